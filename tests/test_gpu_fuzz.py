"""GPU parity on hostile input: rays with zero / negative-zero / infinite / NaN components, inverted and negative
intervals, unnormalised directions, far-away origins (NaN hit distances included: the leaf loop then replays the
reference's sequential test order); meshes with degenerate, duplicated and huge triangles; every leaf
size. The kernel takes its x86-exact slab path for these (NaN rule of minps/maxps). Everything must equal the oracle
bit for bit."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu
FLT_MAX = np.float32(3.4028234663852886e38)


def _bits(a):
    """Raw bit patterns, with every NaN mapped to one pattern: x86 and CDNA generate / propagate different NaN
    payloads and signs (0xFFC00000 vs 0x7FC00000), which is not part of the path's contract."""
    a = np.asarray(a)
    b = a.view(np.uint32).copy()
    if a.dtype == np.float32:
        b[np.isnan(a)] = 0x7FC00000
    return b


def _nasty_mesh(rng, n_tris):
    pos = rng.uniform(-2, 2, (n_tris * 3, 3)).astype(np.float32)
    idx = np.arange(n_tris * 3, dtype=np.uint32).reshape(-1, 3)
    k = n_tris // 16
    pos[idx[:k, 1]] = pos[idx[:k, 0]]                        # zero-area (two equal vertices)
    pos[idx[k:2 * k, 2]] = (pos[idx[k:2 * k, 0]] + pos[idx[k:2 * k, 1]]) / 2  # collinear
    pos[idx[2 * k:3 * k].ravel(), 1] = np.float32(0.5)        # axis-aligned, flat boxes
    pos[idx[3 * k:4 * k].ravel()] *= np.float32(1e4)          # huge
    idx[4 * k:5 * k] = idx[5 * k:6 * k]                       # duplicates (equal t ties)
    pos[idx[6 * k:7 * k].ravel()] *= np.float32(1e-4)         # tiny, near the origin
    return pos, idx


def _nasty_rays(rng, n):
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    tnear = np.zeros(n, np.float32)
    tfar = np.full(n, FLT_MAX, np.float32)
    sel = lambda frac: rng.random(n) < frac
    for ax in range(3):
        d[sel(0.08), ax] = 0.0
        d[sel(0.04), ax] = -0.0
        d[sel(0.01), ax] = np.inf
        d[sel(0.01), ax] = np.nan
        o[sel(0.01), ax] = np.nan
        o[sel(0.01), ax] = np.float32(1e30)
        o[sel(0.02), ax] = np.float32(0.5)   # on the plane of the flat triangles
    d[sel(0.01)] = 0.0                        # no direction at all
    d[sel(0.1)] *= np.float32(1e-3)           # unnormalised
    d[sel(0.1)] *= np.float32(1e3)
    tnear[sel(0.05)] = np.float32(-1.0)
    tnear[sel(0.05)] = np.float32(2.0)
    tfar[sel(0.1)] = np.float32(1.5)
    tfar[sel(0.02)] = np.float32(-1.0)        # inverted interval
    tfar[sel(0.02)] = np.inf
    tfar[sel(0.01)] = np.nan
    tnear[sel(0.01)] = np.nan
    return pa.scenes.pack_rays(o, d, tnear, tfar)


@pytest.mark.parametrize("max_leaf", [1, 2, 3, 4])
def test_hostile_rays_and_meshes(gpu_ctx, max_leaf):
    rng = np.random.default_rng(100 + max_leaf)
    pos, idx = _nasty_mesh(rng, 1600)
    hb = pa.accel.build_bvh8(pos, idx, max_leaf=max_leaf)
    bvh = gpu_ctx.upload_bvh8(hb, gpu_ctx.upload_mesh(pos, idx))

    class M:
        pass
    m = M()
    m.pos, m.idx = pos, idx
    orc = ob.OracleBvh8.from_host(hb, m)
    rays = _nasty_rays(rng, 40000)
    with np.errstate(all="ignore"):
        for tri in (pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT):
            want = orc.trace(rays, tri_mode=tri, threads=8)
            wa = orc.trace(rays, any_hit=True, tri_mode=tri, threads=8)["hit"]
            for lanes in (4, 2):  # both bottom-level kernel layouts
                gpu_ctx.set_ray_layout(lanes)
                got = gpu_ctx.trace_bvh8(bvh, rays, tri_mode=tri)
                for k in ("t", "obj", "u", "v"):
                    g, w = _bits(got[k]), _bits(want[k])
                    bad = np.nonzero(g != w)[0]
                    assert bad.size == 0, f"tri{tri} lanes{lanes} {k}: {bad.size} rays differ, first {bad[:6]} rays {rays[:, bad[:3]].T}"
                ga = gpu_ctx.trace_bvh8(bvh, rays, kind=pa.capi.ANY, tri_mode=tri)["hit"]
                assert np.array_equal(ga, wa), f"tri{tri} lanes{lanes} any-hit: {(ga != wa).sum()} rays differ"
    gpu_ctx.set_ray_layout(4)


@pytest.mark.parametrize("svdag", [0, 16])
def test_hostile_rays_two_level(svdag):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    rng = np.random.default_rng(7 + svdag)
    pos, idx = _nasty_mesh(rng, 1200)
    keep = np.abs(pos[idx.ravel()]).reshape(-1, 9).max(axis=1) < 100.0  # (the voxeliser wants a bounded scene)
    idx = np.ascontiguousarray(idx[keep])
    tl = pa.accel.build_two_level(pos, idx, 48, svdag_res=svdag)
    ctx = pa.accel.Context(0)
    tl.upload(ctx, ctx.upload_mesh(pos, idx))

    class M:
        pass
    m = M()
    m.pos, m.idx = pos, idx
    scene = ob.scene_from_two_level(tl, m)
    rays = _nasty_rays(rng, 30000)
    with np.errstate(all="ignore"):
        for kind in (pa.capi.CLOSEST, pa.capi.ANY):
            got, cnt = ctx.trace(rays, kind=kind)
            want = scene.trace(rays, any_hit=(kind == pa.capi.ANY), lanes=3, threads=8)
            keys = ("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")
            for k in keys:
                g, w = _bits(got[k]), _bits(want[k])
                bad = np.nonzero(g != w)[0]
                assert bad.size == 0, f"kind {kind} {k}: {bad.size} rays differ, first {bad[:6]} rays {rays[:, bad[:3]].T} got {np.asarray(got[k])[bad[:3]]} want {np.asarray(want[k])[bad[:3]]}"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt[k] == want["counters"][k], k
    ctx.close()
