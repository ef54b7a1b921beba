"""Golden vectors for BASELINE configs[0] (an in-core render of a bundled assets/3dmodels scene through the WiVe BVH8 path:
primary rays + one diffuse bounce, 1 spp) from the reference's own bundled meshes. Run in the build container only:

    python tests/golden/make_c1_assets.py

Reads /root/reference/assets/3dmodels/{monkey,cornell_box}.obj (data files the reference ships; parsed as text by the small
reader below), builds a BVH8 over each with this repository's builder, generates the ray sets of SURVEY.md 8d, and records
what oracle/_ref (the traversal restated on the reference's own simd headers, compiled from /root/reference) returns.
Output: tests/golden/c1_assets.npz = inputs (mesh, tree arrays, rays) + expected outputs. The GPU box has no
/root/reference: the tests replay this file through the scalar oracle (CPU suite) and the HIP path (GPU suite).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_binding as ob
import pandora_amd as pa

ASSETS = "/root/reference/assets/3dmodels"


def read_obj(path):
    """Positions and triangles of a Wavefront OBJ: `v x y z` and `f i[/t[/n]] ...` (1-based, negative = relative to the
    end), polygons fanned around their first vertex. Everything else is skipped."""
    pos, tris = [], []
    with open(path) as fh:
        for line in fh:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if tok[0] == "v":
                pos.append([float(x) for x in tok[1:4]])
            elif tok[0] == "f":
                ids = []
                for t in tok[1:]:
                    i = int(t.split("/")[0])
                    ids.append(i - 1 if i > 0 else len(pos) + i)
                for k in range(1, len(ids) - 1):
                    tris.append([ids[0], ids[k], ids[k + 1]])
    return np.asarray(pos, np.float32), np.asarray(tris, np.uint32)


def scene(name, eye, look_at, fov, light_lo, light_hi, res):
    pos, idx = read_obj(os.path.join(ASSETS, name + ".obj"))
    mesh = pa.scenes.Mesh(pos, idx)
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    orc = ob.OracleBvh8.from_host(hb, mesh)
    prim = pa.scenes.primary_rays(eye, look_at, (0, 1, 0), fov, res, res)
    w = orc.trace(prim, use_ref=True)
    prim_id = np.where(w["obj"] == ob.MISS, np.uint32(ob.MISS), hb.prim_ids[np.minimum(w["obj"], hb.prim_ids.size - 1)])
    diff = pa.scenes.diffuse_bounce_rays(mesh, prim, w["t"], prim_id)
    shad = pa.scenes.shadow_rays(mesh, prim, w["t"], prim_id, light_lo, light_hi)
    rays = np.ascontiguousarray(np.concatenate([prim, diff, shad], axis=1))
    out = {"pos": mesh.pos, "idx": mesh.idx, "nodes": hb.nodes, "leaf_idx": hb.leaf_idx, "root": np.uint32(hb.root), "prim_ids": hb.prim_ids,
           "rays": rays, "n_primary": np.uint32(prim.shape[1]), "n_diffuse": np.uint32(diff.shape[1]), "n_shadow": np.uint32(shad.shape[1])}
    for tri in (0, 1):
        c = orc.trace(rays, tri_mode=tri, use_ref=True)
        a = orc.trace(rays, any_hit=True, tri_mode=tri, use_ref=True)
        s = orc.trace(rays, tri_mode=tri)  # the scalar oracle must agree with the simd-backed one before anything is written
        sa = orc.trace(rays, any_hit=True, tri_mode=tri)
        for k in ("t", "obj", "u", "v"):
            assert np.array_equal(c[k].view(np.uint32), s[k].view(np.uint32)), (name, tri, k)
            out[f"tri{tri}_{k}"] = c[k]
        assert np.array_equal(a["hit"], sa["hit"])
        out[f"tri{tri}_any"] = a["hit"]
    print(f"{name}: {mesh.n_tris} triangles, {hb.n_nodes} nodes, {rays.shape[1]} rays, primary hit rate {(w['obj'] != ob.MISS).mean():.2f}, "
          f"shadow occlusion {out['tri0_any'][-shad.shape[1]:].mean():.2f}")
    return {f"{name}_{k}": v for k, v in out.items()}


def main():
    assert ob.have_ref(), "oracle/_ref is not built (needs /root/reference)"
    out = {}
    out.update(scene("monkey", (1.8, 1.2, 3.0), (0, 0, 0), 40, (-0.5, 3.0, -0.5), (0.5, 3.0, 0.5), 80))
    out.update(scene("cornell_box", (278, 273, -800), (278, 273, 0), 39, (213, 548, 227), (343, 548, 332), 80))
    np.savez_compressed(os.path.join(HERE, "c1_assets.npz"), **out)
    print("wrote c1_assets.npz,", os.path.getsize(os.path.join(HERE, "c1_assets.npz")), "bytes")


if __name__ == "__main__":
    main()
