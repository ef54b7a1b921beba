"""GPU parity: single-level WiVe BVH8 traversal through the C ABI vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.asarray(a).view(np.uint32)


def _check_closest(got, want, label):
    for k in ("t", "obj", "u", "v"):
        g, w = _bits(got[k]), _bits(want[k])
        bad = np.nonzero(g != w)[0]
        assert bad.size == 0, f"{label}: {k} differs on {bad.size} rays, first {bad[:5]}: got {got[k][bad[:5]]} want {want[k][bad[:5]]}"


@pytest.fixture(scope="module")
def small(gpu_ctx):
    mesh = pa.scenes.g_small()
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    bvh = gpu_ctx.upload_bvh8(hb, gpu_ctx.upload_mesh(mesh.pos, mesh.idx))
    return mesh, hb, bvh, ob.OracleBvh8.from_host(hb, mesh)


def _ray_sets(mesh, orc, res=96):
    prim = pa.scenes.primary_rays((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, res, res)
    w = orc.trace(prim)
    diff = pa.scenes.diffuse_bounce_rays(mesh, prim, w["t"], w["obj"])
    shad = pa.scenes.shadow_rays(mesh, prim, w["t"], w["obj"], (-0.3, 1.99, -0.3), (0.3, 1.99, 0.3))
    rnd = pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), 5000, seed=3)
    # Rays that pass exactly through (0, 0, 0): the unused child slots of a node are zero boxes with handle 0
    # (wive_bvh8_build8_impl.h:142-144), such a ray "hits" them (tmin == tmax) and the reference then treats handle 0 as a
    # one-triangle leaf at leaf index 0 (wive_bvh8_impl.h:84-137) -- a real triangle, tested once more; the kernels must too.
    rng = np.random.default_rng(17)
    sgn = rng.choice(np.float32([-1.0, 1.0]), (3, 2000))
    s = (rng.random(2000, np.float32) * 4 + np.float32(0.5)).astype(np.float32)
    through = pa.scenes.pack_rays((-sgn * s).T, sgn.T.copy(), np.zeros(2000, np.float32), np.full(2000, 3.4028234663852886e38, np.float32))
    return {"primary": prim, "diffuse": diff, "shadow": shad, "random": rnd, "through_origin": through}


# (lanes per ray, refill threshold, LDS stack window): window 0 = automatic, a small window forces most of
# the stack through the global spill buffer
# leaf_min: how many rays of a wave wait at a leaf before the triangle code runs (1 = every step)
@pytest.mark.parametrize("refill_min,window,leaf_min", [(3, 0, 0), (1, 0, 1), (5, 0, 3), (2, 4, 16), (2, 1, 7), (16, 8, 0)])
@pytest.mark.parametrize("tri_mode", [pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT])
@pytest.mark.parametrize("lanes", [4, 2], ids=["quad", "pair"])
def test_closest_and_any_match_oracle(gpu_ctx, small, lanes, tri_mode, refill_min, window, leaf_min):
    mesh, hb, bvh, orc = small
    gpu_ctx.set_ray_layout(lanes)
    gpu_ctx.set_kernel_variant(refill_min, window, leaf_min)
    for name, rays in _ray_sets(mesh, orc).items():
        got = gpu_ctx.trace_bvh8(bvh, rays, tri_mode=tri_mode)
        want = orc.trace(rays, tri_mode=tri_mode)
        _check_closest(got, want, f"{name}/tri{tri_mode}")
        got_any = gpu_ctx.trace_bvh8(bvh, rays, kind=pa.capi.ANY, tri_mode=tri_mode)["hit"]
        want_any = orc.trace(rays, any_hit=True, tri_mode=tri_mode)["hit"]
        assert np.array_equal(got_any, want_any), f"{name}: any-hit differs"
    gpu_ctx.set_kernel_variant(0, 0, 0)
    gpu_ctx.set_ray_layout(4)


@pytest.mark.parametrize("lanes", [4, 2], ids=["quad", "pair"])
def test_traversal_work_equals_the_reference_algorithm(gpu_ctx, small, lanes):
    """The kernels cull lazily at pop time where the reference compacts its stack after every hit: the two are
    claimed equivalent, so the GPU must visit exactly as many inner nodes, pop as many leaves and test as many triangles as
    the CPU restatement of the reference's loop does -- whatever the scheduling knobs."""
    mesh, hb, bvh, orc = small
    gpu_ctx.set_ray_layout(lanes)
    gpu_ctx.set_stats(True)
    try:
        for (refill_min, window, leaf_min) in ((0, 0, 0), (1, 2, 1), (9, 5, 11)):
            gpu_ctx.set_kernel_variant(refill_min, window, leaf_min)
            for name, rays in _ray_sets(mesh, orc, res=64).items():
                for any_hit in (False, True):
                    gpu_ctx.trace_bvh8(bvh, rays, kind=pa.capi.ANY if any_hit else pa.capi.CLOSEST)
                    got = gpu_ctx.counters()
                    want = orc.trace(rays, any_hit=any_hit)["counters"]
                    # (any-hit: the reference leaves a leaf at its first hit, wive_bvh8_impl.h:282-290; the kernels test the leaf's
                    # <= 4 triangles side by side, so they count the ones after it too)
                    for k in ("inner_visits", "leaf_pops") + (() if any_hit else ("tri_tests",)):
                        assert got[k] == want[k], f"{name} any={any_hit} variant {(refill_min, window, leaf_min)}: {k} {got[k]} != {want[k]}"
                    assert got["tri_tests"] >= want["tri_tests"]
    finally:
        gpu_ctx.set_stats(False)
        gpu_ctx.set_kernel_variant(0, 0, 0)
        gpu_ctx.set_ray_layout(4)
