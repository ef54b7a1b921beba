"""Generates the traversal golden vectors from oracle/_ref (the restated loops running on the
reference's own simd headers compiled from /root/reference). Run in the build container:

    python tests/golden/make_golden.py

Outputs (committed, small): bvh8_small.npz, two_level_small.npz, svdag_small.npz.
Each file holds the INPUTS (tree arrays, mesh, rays) and the EXPECTED OUTPUTS; the tests replay the
inputs through the scalar oracle (CPU suite) and through the HIP path (GPU suite).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ctypes as C

import oracle_binding as ob
import pandora_amd as pa


def ray_mix(mesh, hb, res=36):
    orc = ob.OracleBvh8.from_host(hb, mesh)
    prim = pa.scenes.primary_rays((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, res, res)
    w = orc.trace(prim, use_ref=True)
    diff = pa.scenes.diffuse_bounce_rays(mesh, prim, w["t"], w["obj"])
    shad = pa.scenes.shadow_rays(mesh, prim, w["t"], w["obj"], (-0.3, 1.99, -0.3), (0.3, 1.99, 0.3))
    rnd = pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), 1024, seed=3)
    # axis-aligned directions with origins on box planes: exercises the 0 * inf = NaN rule
    ax = pa.scenes.random_rays((-1, 0, -1), (1, 2, 1), 256, seed=5)
    ax[3:6] = 0
    ax[3 + (np.arange(256) % 3), np.arange(256)] = np.where(np.arange(256) % 2 == 0, 1.0, -1.0)
    ax[0, :64] = -1.0
    ax[1, 64:128] = 0.0
    ax[2, 128:192] = -1.0
    return np.ascontiguousarray(np.concatenate([prim, diff, shad, rnd, ax], axis=1))


def main():
    assert ob.have_ref(), "oracle/_ref is not built (needs /root/reference)"
    mesh = pa.scenes.g_small()
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    rays = ray_mix(mesh, hb)
    orc = ob.OracleBvh8.from_host(hb, mesh)
    out = {"nodes": hb.nodes, "leaf_idx": hb.leaf_idx, "root": np.uint32(hb.root), "prim_ids": hb.prim_ids,
           "pos": mesh.pos, "idx": mesh.idx, "rays": rays}
    for tri in (0, 1):
        c = orc.trace(rays, tri_mode=tri, use_ref=True)
        a = orc.trace(rays, any_hit=True, tri_mode=tri, use_ref=True)
        for k in ("t", "obj", "u", "v"):
            out[f"tri{tri}_{k}"] = c[k]
        out[f"tri{tri}_any"] = a["hit"]
    np.savez_compressed(os.path.join(HERE, "bvh8_small.npz"), **out)
    print("bvh8_small:", rays.shape[1], "rays,", hb.n_nodes, "nodes")

    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 48, svdag_res=32)
    sc = ob.scene_from_two_level(tl, mesh)
    far = pa.scenes.random_rays((-300, -300, -300), (300, 300, 300), 512, seed=4, toward=((-1, 0, -1), (1, 2, 1)))
    rays2 = np.ascontiguousarray(np.concatenate([rays, far], axis=1))
    c = sc.trace(rays2, use_ref=True)
    a = sc.trace(rays2, any_hit=True, use_ref=True)
    out = {"pos": mesh.pos, "idx": mesh.idx, "rays": rays2, "top_nodes": tl.top.nodes, "top_root": np.uint32(tl.top.root),
           "leaf_order": tl.top.leaf_order, "dag_words": tl.dags.words, "dag_descs": tl.dags.descs, "n_leaves": np.uint32(tl.n_leaves)}
    for i, b in enumerate(tl.bvhs):
        out[f"leaf{i}_nodes"] = b.nodes
        out[f"leaf{i}_leaf_idx"] = b.leaf_idx
        out[f"leaf{i}_root"] = np.uint32(b.root)
        out[f"leaf{i}_prim_ids"] = b.prim_ids
    for k in ("t", "obj", "leaf", "u", "v", "visits"):
        out[f"closest_{k}"] = c[k]
    out["any_hit"] = a["hit"]
    out["any_visits"] = a["visits"]
    out["closest_counters"] = np.array([c["counters"][k] for k in ("svdag_tests", "svdag_culled", "leaf_visits")], np.uint64)
    out["any_counters"] = np.array([a["counters"][k] for k in ("svdag_tests", "svdag_culled", "leaf_visits")], np.uint64)
    np.savez_compressed(os.path.join(HERE, "two_level_small.npz"), **out)
    print("two_level_small:", rays2.shape[1], "rays,", tl.n_leaves, "leaves, counters", c["counters"])

    # SVDAG: an icosphere, a box and the whole small scene at 32^3 / 64^3, rays near and far
    shapes = {"sphere": pa.scenes.icosphere((0, 0, 0), 1.0, 3), "box": pa.scenes.box((-1, -0.2, -0.5), (1, 0.2, 0.5), 2), "room": mesh}
    out = {}
    for name, m in shapes.items():
        res = 64 if name == "sphere" else 32
        dags = pa.accel.build_svdags(m.pos, m.idx, [np.arange(m.n_tris, dtype=np.uint32)], res)
        lo, hi = m.bounds()
        ext = float((hi - lo).max())
        ctr = (lo + hi) / 2
        sets = [pa.scenes.random_rays(ctr - s * ext, ctr + s * ext, 700, seed=7 + i, toward=(lo - 0.3 * ext, hi + 0.3 * ext)) for i, s in enumerate((1.2, 6.0, 300.0))]
        r = np.ascontiguousarray(np.concatenate(sets, axis=1))
        r[4, ::11] = 0.0  # exact zero direction component (the epsilon clamp, sparse_voxel_dag.cpp:358-364)
        d = dags.descs[0]
        dag = ob.Svdag(dags.words.ctypes.data, int(d["root"]), (C.c_float * 3)(*[float(x) for x in d["bounds_min"]]), float(d["inv_extent"]), 1)
        hit = np.zeros(r.shape[1], np.uint8)
        for i in range(r.shape[1]):
            ray = ob.make_ray(r[0:3, i], r[3:6, i], float(r[6, i]), float(r[7, i]))
            hit[i] = ob.ref().ref_svdag_intersect(C.byref(dag), C.byref(ray))
        out[f"{name}_words"] = dags.words
        out[f"{name}_desc"] = dags.descs
        out[f"{name}_rays"] = r
        out[f"{name}_hit"] = hit
        print("svdag", name, "words", dags.words.size, "hit rate", hit.mean())
    np.savez_compressed(os.path.join(HERE, "svdag_small.npz"), **out)


if __name__ == "__main__":
    main()
