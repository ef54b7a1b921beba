"""GPU parity: two-level path (PauseableBVH4 + SVDAG early-out + per-leaf BVH8) vs the CPU oracle."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.asarray(a).view(np.uint32)


@pytest.fixture(scope="module", params=[0, 32], ids=["no_svdag", "svdag32"])
def two_level(request):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    ctx = pa.accel.Context(0)
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=request.param)
    tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
    yield ctx, mesh, tl, ob.scene_from_two_level(tl, mesh)
    ctx.close()


def _ray_sets(mesh):
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    orc = ob.OracleBvh8.from_host(hb, mesh)
    prim = pa.scenes.primary_rays((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 96, 96)
    w = orc.trace(prim)
    diff = pa.scenes.diffuse_bounce_rays(mesh, prim, w["t"], w["obj"])
    shad = pa.scenes.shadow_rays(mesh, prim, w["t"], w["obj"], (-0.3, 1.99, -0.3), (0.3, 1.99, 0.3))
    rnd = pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), 5000, seed=3)
    far = pa.scenes.random_rays((-300, -300, -300), (300, 300, 300), 5000, seed=4, toward=((-1, 0, -1), (1, 2, 1)))
    return {"primary": prim, "diffuse": diff, "shadow": shad, "random": rnd, "far": far}


@pytest.mark.parametrize("lanes,window,leaf_min", [(4, 0, 0), (4, 2, 9), (2, 0, 0), (2, 3, 5)])
def test_two_level_matches_oracle(two_level, lanes, window, leaf_min):
    ctx, mesh, tl, orc = two_level
    ctx.set_ray_layout(lanes)  # the bottom-level kernel's layout: quad (16 rays per wave) or pair (32)
    ctx.set_kernel_variant(0, window, leaf_min)
    for name, rays in _ray_sets(mesh).items():
        got, cnt = ctx.trace(rays)
        want = orc.trace(rays, lanes=3)
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            bad = np.nonzero(_bits(got[k]) != _bits(want[k]))[0]
            assert bad.size == 0, f"{name}: {k} differs on {bad.size} rays, first {bad[:5]}: got {got[k][bad[:5]]} want {want[k][bad[:5]]}"
        for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
            assert cnt[k] == want["counters"][k], f"{name}: counter {k}: {cnt[k]} != {want['counters'][k]}"
        got_any, cnt_any = ctx.trace(rays, kind=pa.capi.ANY)
        want_any = orc.trace(rays, any_hit=True, lanes=3)
        assert np.array_equal(_bits(got_any["hit"]), want_any["hit"]), f"{name}: any-hit differs"
        assert np.array_equal(_bits(got_any["visits"]), want_any["visits"]), f"{name}: any-hit visits differ"
        for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
            assert cnt_any[k] == want_any["counters"][k], f"{name}: any counter {k}"


@pytest.mark.parametrize("passes", [1, 2, 3])
def test_deferred_svdag_casts_match_oracle(two_level, passes):
    """pgpu_set_svdag_defer: the casts as a kernel of their own (requests sorted by batching point, DAG staged in LDS) give the
    same hits, visit counts and SVDAG counters as the casts inside the episode -- both are checked against the same oracle."""
    ctx, mesh, tl, orc = two_level
    ctx.set_ray_layout(4)
    ctx.set_kernel_variant(0, 0, 0)
    ctx.set_svdag_defer(1, passes)  # every pass, however small
    try:
        for name, rays in _ray_sets(mesh).items():
            got, cnt = ctx.trace(rays)
            want = orc.trace(rays, lanes=3)
            for k in ("t", "obj", "leaf", "u", "v", "visits"):
                bad = np.nonzero(_bits(got[k]) != _bits(want[k]))[0]
                assert bad.size == 0, f"{name}: {k} differs on {bad.size} rays, first {bad[:5]}"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt[k] == want["counters"][k], f"{name}: counter {k}: {cnt[k]} != {want['counters'][k]}"
            got_any, cnt_any = ctx.trace(rays, kind=pa.capi.ANY)
            want_any = orc.trace(rays, any_hit=True, lanes=3)
            assert np.array_equal(_bits(got_any["hit"]), want_any["hit"]), f"{name}: any-hit differs"
            assert np.array_equal(_bits(got_any["visits"]), want_any["visits"]), f"{name}: any-hit visits differ"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt_any[k] == want_any["counters"][k], f"{name}: any counter {k}"
    finally:
        ctx.set_svdag_defer(0)


@pytest.mark.parametrize("variant", [2, 3, 4])
@pytest.mark.parametrize("refill_min,cast_trips", [(0, 0), (1, 1), (64, 8), (7, 3)])
def test_top_level_lanes_kernel_matches_oracle(two_level, refill_min, cast_trips, variant):
    """pgpu_set_top_variant(2): the top-level kernel whose lanes take the next ray as soon as theirs has parked or left, with
    BVH4 steps and SVDAG cast trips as separate wave phases -- same hits, visit counts and SVDAG counters as the oracle, for
    any refill threshold and phase length, with and without the rounds queued ahead."""
    ctx, mesh, tl, orc = two_level
    ctx.set_ray_layout(0)
    ctx.set_kernel_variant(0, 0, 0)
    ctx.set_top_variant(variant, refill_min, cast_trips)  # (3: lockstep first episode, independent lanes afterwards)
    ctx.set_round_lookahead(refill_min != 7)
    try:
        for name, rays in _ray_sets(mesh).items():
            got, cnt = ctx.trace(rays)
            want = orc.trace(rays, lanes=3)
            for k in ("t", "obj", "leaf", "u", "v", "visits"):
                bad = np.nonzero(_bits(got[k]) != _bits(want[k]))[0]
                assert bad.size == 0, f"{name}: {k} differs on {bad.size} rays, first {bad[:5]}"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt[k] == want["counters"][k], f"{name}: counter {k}: {cnt[k]} != {want['counters'][k]}"
            got_any, cnt_any = ctx.trace(rays, kind=pa.capi.ANY)
            want_any = orc.trace(rays, any_hit=True, lanes=3)
            assert np.array_equal(_bits(got_any["hit"]), want_any["hit"]), f"{name}: any-hit differs"
            assert np.array_equal(_bits(got_any["visits"]), want_any["visits"]), f"{name}: any-hit visits differ"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt_any[k] == want_any["counters"][k], f"{name}: any counter {k}"
    finally:
        ctx.set_top_variant(0)
        ctx.set_round_lookahead(True)


@pytest.mark.parametrize("parts", [2, 3, 4])
def test_a_batch_cut_into_parts_matches_oracle(two_level, parts):
    """pgpu_set_batch_parts: a batch as several batches side by side on their own streams and work buffers -- same hits, visit
    counts and summed counters as the oracle on the whole batch, from host arrays and from device arrays, closest and any hit;
    a length that is no multiple of anything."""
    import torch
    ctx, mesh, tl, orc = two_level
    ctx.set_ray_layout(0)
    ctx.set_kernel_variant(0, 0, 0)
    n = 4 * 65536 + 4097
    rays = np.ascontiguousarray(np.concatenate([
        pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), n - 50000, seed=11),
        pa.scenes.random_rays((-300, -300, -300), (300, 300, 300), 50000, seed=12, toward=((-1, 0, -1), (1, 2, 1)))], axis=1))
    want = orc.trace(rays, lanes=3)
    want_any = orc.trace(rays, any_hit=True, lanes=3)
    ctx.set_batch_parts(parts)
    try:
        for src in (rays, torch.from_numpy(rays).cuda()):
            got, cnt = ctx.trace(src)
            for k in ("t", "obj", "leaf", "u", "v", "visits"):
                g = got[k].cpu().numpy() if torch.is_tensor(got[k]) else got[k]
                bad = np.nonzero(_bits(g) != _bits(want[k]))[0]
                assert bad.size == 0, f"{k} differs on {bad.size} rays, first {bad[:5]}"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt[k] == want["counters"][k], f"counter {k}: {cnt[k]} != {want['counters'][k]}"
            assert cnt["rounds"] >= 1
            got_any, cnt_any = ctx.trace(src, kind=pa.capi.ANY)
            for k in ("hit", "visits"):
                g = got_any[k].cpu().numpy() if torch.is_tensor(got_any[k]) else got_any[k]
                assert np.array_equal(_bits(g), _bits(want_any[k])), f"any-hit {k} differs"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt_any[k] == want_any["counters"][k], f"any counter {k}"
        assert ctx.last_kernel_ms() > 0
    finally:
        ctx.set_batch_parts(0)


@pytest.mark.parametrize("n", [1, 63, 256, 257, 1025])
@pytest.mark.parametrize("variant", [1, 2, 3, 4])
def test_tiny_batches_through_the_round_loop(two_level, n, variant):
    """Batches smaller than a block, a wave, a compaction slice: the rounds queued ahead size their launches from the previous
    counts and must neither drop nor repeat a ray."""
    ctx, mesh, tl, orc = two_level
    ctx.set_ray_layout(0)
    ctx.set_kernel_variant(0, 0, 0)
    ctx.set_top_variant(variant)
    try:
        rays = np.ascontiguousarray(_ray_sets(mesh)["diffuse"][:, 100:100 + n])
        for kind, keys in ((pa.capi.CLOSEST, ("t", "obj", "leaf", "u", "v", "visits")), (pa.capi.ANY, ("hit", "visits"))):
            got, cnt = ctx.trace(rays, kind=kind)
            want = orc.trace(rays, any_hit=(kind == pa.capi.ANY), lanes=3)
            for k in keys:
                assert np.array_equal(_bits(got[k]), _bits(want[k])), f"n={n} {k}"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt[k] == want["counters"][k], f"n={n} counter {k}"
    finally:
        ctx.set_top_variant(0)


def test_regrouped_mesh_upload_matches_whole_mesh(two_level):
    """TwoLevelScene.sub_mesh + prim_remap (what a rank of a sharded scene uploads: only the triangles of its own batching
    points, renumbered) with every point resident: the triangles arrive in another order under other ids, the hits are
    the same. The partial case runs in tests/test_gpu_sharded.py (two ranks, each with half of the mesh)."""
    ctx, mesh, tl, orc = two_level
    rays = _ray_sets(mesh)["diffuse"]
    want, wcnt = ctx.trace(rays)
    part_pos, part_idx, remap = tl.sub_mesh(mesh.pos, mesh.idx, np.ones(tl.n_leaves, bool))
    assert part_idx.shape[0] == mesh.n_tris and not np.array_equal(remap, np.arange(mesh.n_tris, dtype=np.uint32))
    other = pa.accel.Context(0)
    try:
        tl.upload(other, other.upload_mesh(part_pos, part_idx), prim_remap=remap)
        got, cnt = other.trace(rays)
    finally:
        other.close()
    for k in ("t", "obj", "leaf", "u", "v", "visits"):
        assert np.array_equal(_bits(got[k]), _bits(want[k])), k
    assert cnt["leaf_visits"] == wcnt["leaf_visits"]


def test_svdag_policy_skips_the_test_for_resident_points(two_level):
    """pgpu_set_svdag_policy(ctx, 1): batching points whose BVH is in HBM are visited without the proxy test. With every point
    resident that is the scene without SVDAGs -- same hits as with them, the visit counts and counters of the oracle's scene
    without DAGs."""
    import dataclasses
    ctx, mesh, tl, orc = two_level
    bare = ob.scene_from_two_level(dataclasses.replace(tl, dags=None), mesh)
    rays = _ray_sets(mesh)["diffuse"]
    ctx.set_ray_layout(0)
    ctx.set_kernel_variant(0, 0, 0)
    ref, _ = ctx.trace(rays)
    ctx.set_svdag_policy(1)
    try:
        got, cnt = ctx.trace(rays)
        got_any, cnt_any = ctx.trace(rays, kind=pa.capi.ANY)
    finally:
        ctx.set_svdag_policy(0)
    want = bare.trace(rays, lanes=3)
    for k in ("t", "obj", "leaf", "u", "v", "visits"):
        assert np.array_equal(_bits(got[k]), _bits(want[k])), k
    for k in ("t", "obj", "leaf", "u", "v"):
        assert np.array_equal(_bits(got[k]), _bits(ref[k])), f"{k}: the hits must not depend on the policy"
    assert cnt["svdag_tests"] == 0 and cnt["leaf_visits"] == want["counters"]["leaf_visits"]
    want_any = bare.trace(rays, any_hit=True, lanes=3)
    assert np.array_equal(_bits(got_any["hit"]), want_any["hit"]) and np.array_equal(_bits(got_any["visits"]), want_any["visits"])
    again, cnt2 = ctx.trace(rays)  # back to the reference's behaviour
    assert np.array_equal(_bits(again["visits"]), _bits(ref["visits"]))
    if tl.dags is not None:
        assert cnt2["svdag_tests"] > 0


def test_tiled_city_assembled_from_two_ranks_parts_matches_oracle():
    """bench.py's rank-local scene build: the top level from the tiles' exchanged bounds and the SVDAG array laid together from
    two ranks' arrays (pgpu_svdag_rebase), over the geometry and BVH8s of all tiles -- traced on the GPU and by the oracle."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    tiles = pa.scenes.CityTiles(4, 2500)
    n = tiles.n_tiles
    one = pa.accel.tiles_local_part(tiles, np.ones(n, bool), svdag_res=32)
    parts = [pa.accel.tiles_local_part(tiles, np.arange(n) % 2 == r, svdag_res=32) for r in range(2)]
    tl = pa.accel.tiles_assemble(n, [{k: p[k] for k in pa.accel.TILE_SHARED_KEYS} for p in parts], one)
    mesh = pa.scenes.Mesh(one["pos"], one["idx"])
    ctx = pa.accel.Context(0)
    try:
        tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
        orc = ob.scene_from_two_level(tl, mesh)
        lo, hi = mesh.bounds()
        rays = np.ascontiguousarray(np.concatenate([
            pa.scenes.random_rays(lo, hi + (0, 10, 0), 20000, seed=21),
            pa.scenes.random_rays(lo - 50, hi + 50, 20000, seed=22, toward=(lo, hi))], axis=1))
        got, cnt = ctx.trace(rays)
        want = orc.trace(rays, lanes=3)
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            assert np.array_equal(_bits(got[k]), _bits(want[k])), k
        for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
            assert cnt[k] == want["counters"][k], k
        assert cnt["svdag_culled"] > 0 and (got["obj"].view(np.uint32) != 0xFFFFFFFF).mean() > 0.2
    finally:
        ctx.close()
