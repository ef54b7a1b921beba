"""GPU parity of instanced leaves (OfflineBVHLeaf::Instance; semantics REF-UNPINNED, see tests/test_instances.py) against the
CPU oracle, bit for bit: a single-level instanced BVH, and the two-level path over an instanced landscape (BASELINE
configs[4]-style) with SVDAGs -- resident and with the batching points streamed from host memory through a small leaf cache."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu


def _bits(a):
    if not isinstance(a, np.ndarray):
        a = a.cpu().numpy()
    return a.view(np.uint32)


def _need_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch


def _rays(size, n_side=128, n_rnd=6000):
    prim = pa.scenes.primary_rays((0.05 * size, 0.25 * size, 0.05 * size), (0.6 * size, 0.0, 0.6 * size), (0, 1, 0), 60, n_side, n_side)
    rnd = pa.scenes.random_rays((0, 0, 0), (size, 0.15 * size, size), n_rnd, seed=3)
    far = pa.scenes.random_rays((-5 * size, size, -5 * size), (6 * size, 4 * size, 6 * size), n_rnd // 2, seed=4, toward=((0, 0, 0), (size, 4, size)))
    return np.ascontiguousarray(np.concatenate([prim, rnd, far], axis=1))


def test_single_level_instanced_bvh_matches_oracle():
    torch = _need_gpu()
    terrain, protos, proto, mats = pa.scenes.g_landscape(n_prototypes=5, n_instances=300, terrain_n=48, size=60.0, detail=2)
    tl = pa.accel.build_instanced_two_level(terrain.pos, terrain.idx, [(p.pos, p.idx) for p in protos], proto, mats, 10**9)
    assert tl.n_leaves == 1
    orc = ob.bvhs_from_instanced(tl)[0]
    rays = _rays(60.0)
    ctx = pa.accel.Context(0)
    try:
        bvh = tl.upload(ctx)[0]
        for tri in (pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT):
            want = orc.trace(rays, tri_mode=tri, threads=8)
            assert (want["inst"] != ob.MISS).mean() > 0.05
            for src in (rays, torch.from_numpy(rays).cuda()):
                got = ctx.trace_bvh8(bvh, src, tri_mode=tri)
                ctx.sync()
                for k in ("t", "obj", "u", "v", "inst"):
                    bad = np.nonzero(_bits(got[k]) != _bits(want[k]))[0]
                    assert bad.size == 0, f"tri {tri} {k}: {bad.size} rays differ, first {bad[:8]}"
                any_got = ctx.trace_bvh8(bvh, src, kind=pa.capi.ANY, tri_mode=tri)
                ctx.sync()
                assert np.array_equal(_bits(any_got["hit"]), orc.trace(rays, any_hit=True, tri_mode=tri, threads=8)["hit"])
        # a tiny LDS stack window: the nested traversal's entries go through the spill path
        ctx.set_kernel_variant(3, 3, 0)
        got = ctx.trace_bvh8(bvh, rays)
        want = orc.trace(rays, threads=8)
        for k in ("t", "obj", "u", "v", "inst"):
            assert np.array_equal(_bits(got[k]), _bits(want[k])), k
    finally:
        ctx.close()


@pytest.mark.parametrize("svdag", [0, 32])
def test_two_level_instanced_landscape_resident_and_streamed(svdag):
    _need_gpu()
    terrain, protos, proto, mats = pa.scenes.g_landscape(n_prototypes=6, n_instances=500, terrain_n=64, size=80.0, detail=2)
    tl = pa.accel.build_instanced_two_level(terrain.pos, terrain.idx, [(p.pos, p.idx) for p in protos], proto, mats, 1500, svdag_res=svdag)
    assert tl.n_leaves >= 6
    scene = ob.scene_from_instanced(tl)
    rays = _rays(80.0, 96, 4000)
    want = {k: scene.trace(rays, any_hit=(k == pa.capi.ANY), threads=8) for k in (pa.capi.CLOSEST, pa.capi.ANY)}
    assert (want[pa.capi.CLOSEST]["inst"] != ob.MISS).mean() > 0.03
    slot = max(b.n_nodes * 256 + (b.leaf_idx.size + 1) * 48 + 64 * gi.size + 1024 for b, gi in zip(tl.bvhs, tl.group_insts))
    # (in_place 0: every flushed queue is loaded; None = the default: sparse queues are traced where their image lies, in host memory)
    for streamed, budget, in_place in ((False, 0, None), (True, 0, 0), (True, 2 * slot, 0), (True, 2 * slot, None)):
        ctx = pa.accel.Context(0)
        try:
            tl.upload(ctx, streamed=streamed)
            if streamed:
                ctx.set_leaf_cache(budget)
                if in_place is not None:
                    ctx.set_leaf_in_place(in_place)
            for kind in (pa.capi.CLOSEST, pa.capi.ANY):
                got, cnt = ctx.trace(rays, kind=kind)
                exp = want[kind]
                for k in (("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits", "inst")):
                    bad = np.nonzero(_bits(got[k]) != _bits(exp[k]))[0]
                    assert bad.size == 0, f"streamed {streamed} budget {budget} kind {kind} {k}: {bad.size} rays differ, first {bad[:8]}"
                for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                    assert cnt[k] == exp["counters"][k], k
                if streamed and budget and in_place == 0:
                    assert cnt["leaf_loads"] > 2
        finally:
            ctx.close()


def test_upload_rejects_what_the_kernel_cannot_follow():
    _need_gpu()
    mesh = pa.scenes.g_small()
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    ident = np.eye(4, dtype=np.float32)[:3].reshape(12)
    ctx = pa.accel.Context(0)
    try:
        proto = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
        # two instances sharing one leaf (built without the `alone` flag)
        boxes = np.array([[0, 0, 0, 1, 1, 1], [2, 0, 0, 3, 1, 1]], np.float32)
        shared = pa.accel.build_bvh8_boxes(boxes, None)
        objs = np.array([[pa.capi.INSTANCE, 0], [pa.capi.INSTANCE, 1]], np.uint32)
        with pytest.raises(pa.capi.PandoraError) as e:
            ctx.upload_bvh8_objects(shared, objs, [(proto, ident), (proto, ident)])
        assert e.value.code == pa.capi.ERR_INVALID
        outer = ctx.upload_bvh8_objects(pa.accel.build_bvh8_boxes(boxes, np.ones(2, np.uint8)), objs, [(proto, ident), (proto, ident)])
        # an instance of a BVH that holds instances itself
        with pytest.raises(pa.capi.PandoraError) as e:
            ctx.upload_bvh8_objects(pa.accel.build_bvh8_boxes(boxes[:1], np.ones(1, np.uint8)), objs[:1], [(outer, ident)])
        assert e.value.code == pa.capi.ERR_DEPTH
        # an instance index beyond the table
        with pytest.raises(pa.capi.PandoraError):
            ctx.upload_bvh8_objects(pa.accel.build_bvh8_boxes(boxes, np.ones(2, np.uint8)), objs, [(proto, ident)])
    finally:
        ctx.close()
