"""One GPU parity test per BASELINE.json config that fits one GPU, at the size the config is quoted on, against the CPU
oracle (bit-exact): configs[0] the reference's bundled assets (golden fixture), configs[1] the Sponza-class atrium at
benchmark size with the three ray sets of SURVEY.md 8d, configs[2] a >= 10 M-triangle two-level scene with 64^3 SVDAGs
(oracle on samples + size-independent properties over the full sets), and the on-disk BVH8 (D1 / N3) on its way to the GPU."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa
from bvh_helpers import with_gaps

pytestmark = pytest.mark.gpu
MISS = np.uint32(0xFFFFFFFF)
THREADS = max(1, min(16, os.cpu_count() or 1))


def _bits(a):
    if not isinstance(a, np.ndarray):
        a = a.cpu().numpy()
    return a.view(np.uint32)


def _need_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch


def _host_bvh(z, prefix=""):
    g = lambda k: np.ascontiguousarray(z[prefix + k])
    return pa.accel.HostBvh8(g("nodes"), g("leaf_idx"), int(z[prefix + "root"]), 0, g("prim_ids"))


# ------------------------------------------------------------------------------------------------ configs[0]
@pytest.mark.parametrize("name", ["monkey", "cornell_box"])
def test_config0_bundled_assets(gpu_ctx, name):
    """The reference's assets/3dmodels meshes, primary + one diffuse bounce + shadow rays, closest and any hit, both triangle
    tests: expected values recorded from oracle/_ref in the build container (tests/golden/make_c1_assets.py)."""
    z = np.load(os.path.join(ob.GOLDEN, "c1_assets.npz"))
    p = name + "_"
    bvh = gpu_ctx.upload_bvh8(_host_bvh(z, p), gpu_ctx.upload_mesh(z[p + "pos"], z[p + "idx"]))
    rays = np.ascontiguousarray(z[p + "rays"])
    assert rays.shape[1] == int(z[p + "n_primary"]) + int(z[p + "n_diffuse"]) + int(z[p + "n_shadow"])
    for lanes in (4, 2):
        gpu_ctx.set_ray_layout(lanes)
        for tri in (0, 1):
            got = gpu_ctx.trace_bvh8(bvh, rays, tri_mode=tri)
            for k in ("t", "obj", "u", "v"):
                bad = np.nonzero(_bits(got[k]) != _bits(z[f"{p}tri{tri}_{k}"]))[0]
                assert bad.size == 0, f"{name} lanes {lanes} tri{tri} {k}: {bad.size} rays differ, first {bad[:8]}"
            assert np.array_equal(_bits(gpu_ctx.trace_bvh8(bvh, rays, kind=pa.capi.ANY, tri_mode=tri)["hit"]), z[f"{p}tri{tri}_any"])
    gpu_ctx.set_ray_layout(4)


# ------------------------------------------------------------------------------------------------ configs[1]
@pytest.fixture(scope="module")
def atrium():
    torch = _need_gpu()
    mesh = pa.scenes.g_atrium(262144)
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    ctx = pa.accel.Context(0)
    bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
    yield torch, ctx, mesh, hb, bvh, ob.OracleBvh8.from_host(hb, mesh)
    ctx.close()


def _ray_sets(ctx, mesh, hb, bvh, side):
    prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, side, side)
    first = ctx.trace_bvh8(bvh, prim)
    obj = _bits(first["obj"])
    prim_id = np.where(obj == MISS, MISS, hb.prim_ids[np.minimum(obj, hb.prim_ids.size - 1)])
    diff = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"], prim_id)
    shad = pa.scenes.shadow_rays(mesh, prim, first["t"], prim_id, (10, 11.9, 6), (20, 11.9, 12))
    return {"primary": prim, "diffuse": diff, "shadow": shad}


@pytest.mark.timeout(400)
def test_config1_atrium_all_ray_sets_all_modes(atrium):
    """R-primary, R-diffuse and R-shadow (4.2 M rays each) x closest / any x Moller-Trumbore / watertight, every ray
    compared with the oracle."""
    torch, ctx, mesh, hb, bvh, orc = atrium
    sets = _ray_sets(ctx, mesh, hb, bvh, 2048)
    assert sets["diffuse"].shape[1] > 4_000_000
    for name, rays in sets.items():
        d = torch.from_numpy(rays).cuda()
        for tri in (pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT):
            got = ctx.trace_bvh8(bvh, d, tri_mode=tri)
            got_any = ctx.trace_bvh8(bvh, d, kind=pa.capi.ANY, tri_mode=tri)
            ctx.sync()
            want = orc.trace(rays, tri_mode=tri, threads=THREADS)
            for k in ("t", "obj", "u", "v"):
                bad = int((_bits(got[k]) != _bits(want[k])).sum())
                assert bad == 0, f"{name} tri {tri} {k}: {bad} of {rays.shape[1]} rays differ"
            want_any = orc.trace(rays, any_hit=True, tri_mode=tri, threads=THREADS)
            assert np.array_equal(_bits(got_any["hit"]), want_any["hit"]), f"{name} tri {tri} any"
    occluded = want_any["hit"].mean()
    assert 0.05 < occluded < 0.95  # the shadow set is neither trivially lit nor trivially dark


@pytest.mark.timeout(400)
def test_config1_atrium_headline_batch_at_bench_size(atrium):
    """The benchmark's own batch: 2^24 R-diffuse rays (every primary ray hits inside the atrium), closest and any hit, both
    kernel layouts -- all rays against the oracle."""
    torch, ctx, mesh, hb, bvh, orc = atrium
    rays = _ray_sets(ctx, mesh, hb, bvh, 4096)["diffuse"]
    assert rays.shape[1] >= (1 << 24) - 4096
    d = torch.from_numpy(rays).cuda()
    want = orc.trace(rays, threads=THREADS)
    want_any = orc.trace(rays, any_hit=True, threads=THREADS)
    for lanes in (4, 2):
        ctx.set_ray_layout(lanes)
        got = ctx.trace_bvh8(bvh, d)
        got_any = ctx.trace_bvh8(bvh, d, kind=pa.capi.ANY)
        ctx.sync()
        for k in ("t", "obj", "u", "v"):
            bad = int((_bits(got[k]) != _bits(want[k])).sum())
            assert bad == 0, f"lanes {lanes} {k}: {bad} of {rays.shape[1]} rays differ"
        assert np.array_equal(_bits(got_any["hit"]), want_any["hit"]), f"lanes {lanes} any"
    ctx.set_ray_layout(4)


# ------------------------------------------------------------------------------------------------ configs[2]
def _prim_of_two_level(tl, leaf, obj):
    out = np.full(obj.shape, MISS, np.uint32)
    hit = obj != MISS
    for l in np.unique(leaf[hit]):
        m = hit & (leaf == l)
        out[m] = tl.bvhs[int(tl.top.leaf_order[int(l)])].prim_ids[obj[m]]
    return out


@pytest.mark.timeout(600)
def test_config2_two_level_11M_triangles_svdag64():
    """San-Miguel-class stand-in: 11.4 M triangles, 256 batching points with 64^3 SVDAGs. 4.2 M primary + their diffuse
    bounce rays: a 1-in-8 sample of every set goes through the oracle (hits, per-ray visit counts, SVDAG counters of the
    sample traced on its own); over the full sets the size-independent properties hold."""
    torch = _need_gpu()
    mesh = pa.scenes.g_city(24, 18750)
    assert mesh.n_tris >= 10_000_000
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 65536, svdag_res=64)
    assert tl.n_leaves >= 128
    scene = ob.scene_from_two_level(tl, mesh)
    ctx = pa.accel.Context(0)
    try:
        tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
        size = 20.0 * 24
        prim = pa.scenes.primary_rays((0.1 * size, 14.0, 0.1 * size), (0.6 * size, 0.0, 0.6 * size), (0, 1, 0), 60, 2048, 2048)
        d_prim = torch.from_numpy(prim).cuda()
        full, cnt = ctx.trace(d_prim)
        full_any, cnt_any = ctx.trace(d_prim, kind=pa.capi.ANY)
        obj, leaf, t = _bits(full["obj"]), _bits(full["leaf"]), full["t"].cpu().numpy()
        hit = obj != MISS
        assert 0.3 < hit.mean() < 0.9
        # properties over the full set: any-hit agrees with closest-hit, the SVDAG culls and never more than it tests, every
        # parked ray was counted, a ray that hit was parked at least once
        assert np.array_equal(_bits(full_any["hit"]) != 0, hit)
        assert cnt["svdag_tests"] > cnt["svdag_culled"] > 0 and cnt["leaf_visits"] == int(_bits(full["visits"]).sum())
        assert np.all(_bits(full["visits"])[hit] >= 1) and cnt["rounds"] == int(_bits(full["visits"]).max())
        prim_id = _prim_of_two_level(tl, leaf, obj)
        diff = pa.scenes.diffuse_bounce_rays(mesh, prim, t, prim_id)
        d_diff = torch.from_numpy(diff).cuda()
        dfull, dcnt = ctx.trace(d_diff)
        assert dcnt["leaf_visits"] == int(_bits(dfull["visits"]).sum())
        # with tfar cut just short of the reported hit, the ray finds nothing at or beyond it (a bounce ray that starts on a
        # surface can report t = 0, which cannot be cut shorter)
        dhit = (_bits(dfull["obj"]) != MISS) & (dfull["t"].cpu().numpy() > 0)
        t_was = dfull["t"].cpu().numpy()[dhit]
        short = diff[:, dhit].copy()
        short[7] = np.nextafter(t_was, np.float32(0))
        cut, _ = ctx.trace(torch.from_numpy(short).cuda())
        assert not np.any((_bits(cut["obj"]) != MISS) & (cut["t"].cpu().numpy() >= t_was))
        # the oracle on a sample of each set (own trace of the sample, so the counters are comparable)
        for name, rays in (("primary", prim), ("diffuse", diff)):
            sub = np.ascontiguousarray(rays[:, 3::8])
            d_sub = torch.from_numpy(sub).cuda()
            for kind in (pa.capi.CLOSEST, pa.capi.ANY):
                got, c = ctx.trace(d_sub, kind=kind)
                want = scene.trace(sub, any_hit=(kind == pa.capi.ANY), lanes=3, threads=THREADS)
                for k in (("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")):
                    bad = int((_bits(got[k]) != _bits(want[k])).sum())
                    assert bad == 0, f"{name} kind {kind} {k}: {bad} of {sub.shape[1]} rays differ"
                for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                    assert c[k] == want["counters"][k], f"{name} kind {kind} {k}"
            # the sample's results are what the full trace returned for those rays
            ref = full if name == "primary" else dfull
            got, _ = ctx.trace(d_sub)
            for k in ("t", "obj", "leaf", "u", "v", "visits"):
                assert np.array_equal(_bits(got[k]), _bits(ref[k])[3::8]), f"{name}: sample vs full trace, {k}"
    finally:
        ctx.close()


@pytest.mark.timeout(600)
def test_config4_instanced_landscape_50k_instances_resident_and_streamed():
    """One GPU's part of configs[4] at a size where the oracle takes seconds: 50 000 instances of 32 tree prototypes (1.3 x 10^8
    instanced triangles) on a 131 K-triangle terrain, ~100 batching points cut over the scene's objects. Primary rays and their
    diffuse bounces (normals of instanced hits through the instance's matrix): resident (rounds queued ahead, ray records) and
    streamed through a cache of a fifth of the images with waiting rays -- hits, the instance of every hit, visit counts
    and counters equal the oracle's, bit for bit."""
    torch = _need_gpu()
    n_inst, size = 50_000, float((50_000 * 64.0) ** 0.5)
    terrain, protos, proto, mats = pa.scenes.g_landscape(n_prototypes=32, n_instances=n_inst, terrain_n=256, size=size, detail=2)
    tl = pa.accel.build_instanced_two_level(terrain.pos, terrain.idx, [(p.pos, p.idx) for p in protos], proto, mats, 16384, svdag_res=0)
    assert tl.n_leaves >= 64
    scene = ob.scene_from_instanced(tl)
    prim = pa.scenes.primary_rays((0.35 * size, 0.06 * size, 0.35 * size), (0.65 * size, 0.0, 0.65 * size), (0, 1, 0), 60, 512, 512)
    first = scene.trace(prim, threads=THREADS)
    assert 0.02 < (first["inst"] != MISS).mean() < 0.9
    normals = tl.hit_normals(first["leaf"], first["obj"], first["inst"])
    diff = pa.scenes.diffuse_bounce_rays(None, prim, first["t"], first["obj"].view(np.uint32), normals=normals)
    want = {name: {kind: scene.trace(rays, any_hit=(kind == pa.capi.ANY), threads=THREADS) for kind in (pa.capi.CLOSEST, pa.capi.ANY)}
            for name, rays in (("primary", prim), ("diffuse", diff))}
    image = [b.n_nodes * 256 + (b.leaf_idx.size + 1) * 48 + 64 * gi.size + 1024 for b, gi in zip(tl.bvhs, tl.group_insts)]
    # (streamed twice: every flushed queue loaded, and the default -- sparse queues traced where their image lies, in host memory)
    for streamed, in_place in ((False, None), (True, 0), (True, None)):
        ctx = pa.accel.Context(0)
        try:
            tl.upload(ctx, streamed=streamed)
            if streamed:
                ctx.set_leaf_cache(max(2 * max(image), sum(image) // 5))
                if in_place is not None:
                    ctx.set_leaf_in_place(in_place)
            for name, rays in (("primary", prim), ("diffuse", diff)):
                d_rays = torch.from_numpy(rays).cuda()
                for kind in (pa.capi.CLOSEST, pa.capi.ANY):
                    got, cnt = ctx.trace(d_rays, kind=kind)
                    exp = want[name][kind]
                    for k in (("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits", "inst")):
                        bad = int((_bits(got[k]) != _bits(exp[k])).sum())
                        assert bad == 0, f"streamed {streamed} (in place {in_place}) {name} kind {kind} {k}: {bad} of {rays.shape[1]} rays differ"
                    assert cnt["leaf_visits"] == exp["counters"]["leaf_visits"]
                    if streamed:
                        assert (cnt["leaf_loads"] > 0 or in_place is None) and cnt["rounds"] >= int(exp["visits"].max())
                    else:
                        assert cnt["rounds"] == int(exp["visits"].max())
        finally:
            ctx.close()


# ------------------------------------------------------------------------------------------------ D1 / N3 on the GPU
def test_bvh8_file_to_gpu_equals_golden(gpu_ctx):
    """bvh8_to_bytes -> bvh8_from_bytes -> upload -> trace reproduces the golden vectors, for the dense array our writer
    emits and for the same tree laid out with allocator-block gaps full of garbage."""
    z, mesh, hb, _ = ob.load_golden_bvh8()
    dense = pa.accel.HostBvh8(np.ascontiguousarray(z["nodes"]), np.ascontiguousarray(z["leaf_idx"]), int(z["root"]), 0, np.ascontiguousarray(z["prim_ids"]))
    rays = np.ascontiguousarray(z["rays"])
    mid = gpu_ctx.upload_mesh(z["pos"], z["idx"])
    for label, host in (("dense", dense), ("gaps", with_gaps(dense))):
        back = pa.accel.bvh8_from_bytes(pa.accel.bvh8_to_bytes(host), host.prim_ids)
        assert back.nodes.tobytes() == np.ascontiguousarray(host.nodes).tobytes() and np.array_equal(back.leaf_idx, host.leaf_idx)
        bvh = gpu_ctx.upload_bvh8(back, mid)
        for tri in (0, 1):
            got = gpu_ctx.trace_bvh8(bvh, rays, tri_mode=tri)
            for k in ("t", "obj", "u", "v"):
                assert np.array_equal(_bits(got[k]), _bits(z[f"tri{tri}_{k}"])), f"{label} tri{tri} {k}"
            assert np.array_equal(_bits(gpu_ctx.trace_bvh8(bvh, rays, kind=pa.capi.ANY, tri_mode=tri)["hit"]), z[f"tri{tri}_any"]), label
