"""GPU parity of leaves streamed from host memory (BASELINE config 5): whatever the size of the HBM leaf cache, the
two-level loop must return what the CPU oracle returns for the same two-level scene (bit-exact, counters included) --
and therefore what it returns with every leaf resident."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.asarray(a).view(np.uint32)


def _oracle(tl, mesh, rays):
    """{kind: (results, counters)} from the CPU oracle, shaped like Context.trace's return value."""
    scene = ob.scene_from_two_level(tl, mesh)
    out = {}
    for kind in (pa.capi.CLOSEST, pa.capi.ANY):
        res = scene.trace(rays, any_hit=(kind == pa.capi.ANY), threads=8)
        cnt = dict(res["counters"])
        cnt["rounds"] = int(res["visits"].max())  # a ray pauses once per round, so the longest ray sets the round count
        out[kind] = (res, cnt)
    return out


def _rays(mesh):
    prim = pa.scenes.primary_rays((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 80, 80)
    rnd = pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), 5000, seed=21)
    return np.ascontiguousarray(np.concatenate([prim, rnd], axis=1))


@pytest.mark.parametrize("hold,in_place", [(0, 0), (256, 0), (4096, 0), (256, 4096), (1, 1)],
                         ids=["no_waiting", "hold256", "hold4096", "hold256_sparse_queues_in_host_memory", "everything_in_host_memory"])
@pytest.mark.parametrize("copy_mode", [0, 1], ids=["gather_kernel", "memcpy_per_image"])
@pytest.mark.parametrize("svdag", [0, 32])
def test_streamed_leaves_match_resident(svdag, copy_mode, hold, in_place):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=svdag)
    rays = _rays(mesh)
    want = _oracle(tl, mesh, rays)

    ctx = pa.accel.Context(0)
    tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), streamed=True)
    ctx.set_leaf_copy(copy_mode)
    ctx.set_leaf_hold(hold)  # rays wait at a point until it holds a ray per `hold` bytes of image (more rounds, fewer loads, same results)
    # flushed queues with fewer rays than one per `in_place` bytes of image are traced where the image lies, in pinned host memory
    # (last case: every queue waits and every flush is sparse -- no image is ever loaded unless it is in the cache already)
    ctx.set_leaf_in_place(in_place)
    slot = max(((b.n_nodes * 256 + 255) // 256) * 256 + (((b.leaf_idx.size + 1) * 48 + 255) // 256) * 256 for b in tl.bvhs)  # (records end in one zero record of padding)
    total = sum(((b.n_nodes * 256 + 255) // 256) * 256 + (((b.leaf_idx.size + 1) * 48 + 255) // 256) * 256 for b in tl.bvhs)
    for budget, label in ((0, "all fit"), (3 * slot + 17, "3 slots"), (slot, "1 slot"), (total // 2, "half"), (total - 256, "all but one")):
        ctx.set_leaf_cache(budget)
        for kind in (pa.capi.CLOSEST, pa.capi.ANY):
            got, cnt = ctx.trace(rays, kind=kind)
            exp, ecnt = want[kind]
            keys = ("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")
            for k in keys:
                assert np.array_equal(_bits(got[k]), _bits(exp[k])), f"{label}: {k}"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt[k] == ecnt[k], f"{label}: {k}"
            assert cnt["rounds"] == ecnt["rounds"] if hold == 0 else cnt["rounds"] >= ecnt["rounds"], f"{label}: rounds"
            # (a trace that finds everything it needs in the cache loads nothing: possible once the cache holds half of the images)
            assert cnt["leaf_loads"] > 0 or budget == 0 or budget >= total // 2 or in_place
            if in_place == 1:
                    assert cnt["leaf_loads"] <= 2, f"{label}: only an image smaller in bytes than its queue in rays qualifies for a load"
            if budget and budget < 4 * slot and hold == 0:
                assert cnt["leaf_loads"] > budget // slot, f"{label}: more BVHs were needed than fit, so space must have been recycled (loads {cnt['leaf_loads']})"
            assert cnt["streamed_bytes"] >= cnt["leaf_loads"] * 256 or cnt["leaf_loads"] == 0
    # pgpu_set_svdag_policy(ctx, 1) keeps the proxy test for points whose geometry is not in HBM: here that is every point
    ctx.set_leaf_cache(0)
    ctx.set_svdag_policy(1)
    got, cnt = ctx.trace(rays)
    exp, ecnt = want[pa.capi.CLOSEST]
    for k in ("t", "obj", "leaf", "u", "v", "visits"):
        assert np.array_equal(_bits(got[k]), _bits(exp[k])), f"policy 1: {k}"
    assert cnt["svdag_tests"] == ecnt["svdag_tests"] and cnt["svdag_culled"] == ecnt["svdag_culled"]
    with pytest.raises(pa.capi.PandoraError):
        ctx.set_leaf_cache(64)
        ctx.trace(rays)
    ctx.close()


def test_staged_bvh_cannot_be_traced_single_level():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    mesh = pa.scenes.g_small()
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    ctx = pa.accel.Context(0)
    bid = ctx.stage_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
    with pytest.raises(pa.capi.PandoraError) as e:
        ctx.trace_bvh8(bid, _rays(mesh))
    assert e.value.code == pa.capi.ERR_NO_SCENE
    ctx.close()


def test_bvh_staged_after_a_trace_gets_a_cache_that_fits():
    """A BVH staged after the first pgpu_trace can be larger than the cache slots that trace sized: the cache has to be
    rebuilt, or the copy overruns its slot."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    mesh = pa.scenes.g_small()
    rays = _rays(mesh)
    small = pa.accel.build_two_level(mesh.pos, mesh.idx, 48, svdag_res=0)
    large = pa.accel.build_two_level(mesh.pos, mesh.idx, 500, svdag_res=16)
    assert max(b.n_nodes for b in large.bvhs) > max(b.n_nodes for b in small.bvhs)
    ctx = pa.accel.Context(0)
    mid = ctx.upload_mesh(mesh.pos, mesh.idx)
    for tl in (small, large):  # the second upload stages bigger images into the same context and replaces the top level
        tl.upload(ctx, mid, streamed=True)
        want = _oracle(tl, mesh, rays)
        for kind in (pa.capi.CLOSEST, pa.capi.ANY):
            got, cnt = ctx.trace(rays, kind=kind)
            exp, ecnt = want[kind]
            for k in (("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")):
                assert np.array_equal(_bits(got[k]), _bits(exp[k])), k
            assert cnt["leaf_visits"] == ecnt["leaf_visits"] and cnt["svdag_culled"] == ecnt["svdag_culled"]
    ctx.close()


def test_sharded_building_blocks_refuse_staged_contexts():
    """pgpu_trace_leaves / pgpu_trace_leaf_records do not stream: with staged BVHs in the context they must refuse rather
    than walk a cache slot that may hold another BVH's image."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64)
    ctx = pa.accel.Context(0)
    tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), streamed=True)
    dev = torch.device("cuda", 0)
    rays = torch.from_numpy(_rays(mesh)).to(dev)
    n = rays.shape[1]
    hits = {k: torch.zeros(n, dtype=torch.float32 if k in ("t", "u", "v") else torch.int32, device=dev) for k in ("t", "obj", "leaf", "u", "v")}
    work = torch.zeros(n, dtype=torch.int32, device=dev)
    with pytest.raises(pa.capi.PandoraError) as e:
        ctx.trace_leaves(pa.capi.CLOSEST, pa.capi.TRI_MT, rays, work, work, n, hits)
    assert e.value.code == pa.capi.ERR_NO_SCENE
    with pytest.raises(pa.capi.PandoraError) as e:
        ctx.trace_leaf_records(pa.capi.CLOSEST, pa.capi.TRI_MT, torch.zeros(9 * 4, device=dev), 4, torch.zeros(16, device=dev))
    assert e.value.code == pa.capi.ERR_NO_SCENE
    ctx.close()


def test_streamed_city_random_cache_settings():
    """A city of ~50 batching points with images of different sizes, 120 K shuffled diffuse rays: whatever the cache size, the
    waiting threshold and the copy mode, the streamed trace returns the resident trace's arrays bit for bit (which
    tests/test_gpu_scale.py holds against the oracle), and the counters that do not depend on scheduling agree."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    mesh = pa.scenes.g_city(3, 20000)
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 4096, svdag_res=16)
    assert tl.n_leaves >= 32
    res = pa.accel.Context(0)
    try:
        tl.upload(res, res.upload_mesh(mesh.pos, mesh.idx))
        prim = pa.scenes.primary_rays((6.0, 14.0, 6.0), (36.0, 0.0, 36.0), (0, 1, 0), 60, 360, 360)
        first, _ = res.trace(prim)
        rays = pa.scenes.diffuse_bounce_rays(mesh, prim, np.asarray(first["t"]), tl.prim_ids_of(np.asarray(first["leaf"]), np.asarray(first["obj"])))
        want = {kind: res.trace(rays, kind=kind) for kind in (pa.capi.CLOSEST, pa.capi.ANY)}
    finally:
        res.close()
    images = [((b.n_nodes * 256 + 255) // 256) * 256 + (((b.leaf_idx.size + 1) * 48 + 255) // 256) * 256 for b in tl.bvhs]
    assert max(images) > 2 * min(images)
    rng = np.random.default_rng(5)
    ctx = pa.accel.Context(0)
    try:
        tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), streamed=True)
        loads = []
        for trial in range(16):
            budget = int(rng.uniform(max(images), sum(images)))
            hold = int(rng.choice([0, 16, 256, 2048, 1 << 20]))
            in_place = int(rng.choice([0, 1, 512, 4096]))
            ctx.set_leaf_cache(budget)
            ctx.set_leaf_hold(hold)
            ctx.set_leaf_in_place(in_place)
            ctx.set_leaf_copy(trial & 1)
            layout = int(rng.choice([0, 2, 4]))  # (4: the quad kernel, whose nodes go to LDS by direct loads -- from host memory too)
            ctx.set_ray_layout(layout)
            for kind in (pa.capi.CLOSEST, pa.capi.ANY):
                got, cnt = ctx.trace(rays, kind=kind)
                exp, ecnt = want[kind]
                for k in (("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")):
                    assert np.array_equal(_bits(got[k]), _bits(exp[k])), f"budget {budget} hold {hold} in place {in_place} layout {layout}: {k}"
                for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                    assert cnt[k] == ecnt[k], f"budget {budget} hold {hold}: {k}"
                assert cnt["rounds"] >= ecnt["rounds"]
                loads.append(cnt["leaf_loads"])
        assert max(loads) > tl.n_leaves  # some setting had to bring images in more than once
    finally:
        ctx.close()
