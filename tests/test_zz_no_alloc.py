"""pgpu_trace_bvh8_noalloc (the entry point that allocates nothing) against pgpu_trace_bvh8, through the C ABI: two calls in
a row, the second with a smaller batch (the static scratch keeps the first call's contents), closest and any hit, both
triangle tests, and a tree deep enough that the per-ray stacks overflow into the scratch's spill area."""
import ctypes as C

import numpy as np
import pytest

import pandora_amd as pa

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.cpu().numpy().view(np.uint32)


def _alloc(dev, n, kind):
    import torch
    if kind == pa.capi.ANY:
        return {"hit": torch.full((n,), 7, dtype=torch.int32, device=dev)}
    return {k: torch.full((n,), 7, dtype=torch.float32 if k in ("t", "u", "v") else torch.int32, device=dev) for k in ("t", "obj", "leaf", "u", "v")}


def test_no_alloc_entry_point_matches_trace_bvh8():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    dev = torch.device("cuda", 0)
    mesh = pa.scenes.g_atrium(20000)
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    # the LDS stack window holds 12 entries (kernels.h stackWindow); a tree of depth d needs up to 7 d + 8: more than 12 means
    # entries really leave the window and the static spill area is in use
    assert 7 * hb.depth + 8 > 12, "the tree must be deep enough for stack entries to leave the LDS window"
    prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, 192, 192)
    rnd = pa.scenes.random_rays((0, 0, 0), (30, 12, 18), 30000, seed=5)
    big = np.ascontiguousarray(np.concatenate([prim, rnd], axis=1))
    small = np.ascontiguousarray(big[:, 1000:9001:3])  # the second call: fewer rays, different ones first
    ctx = pa.accel.Context(0)
    ref_ctx = pa.accel.Context(0)
    try:
        ids = [c.upload_bvh8(hb, c.upload_mesh(mesh.pos, mesh.idx)) for c in (ctx, ref_ctx)]
        d_big, d_small = torch.from_numpy(big).to(dev), torch.from_numpy(small).to(dev)
        outs = {(kind, tri, which): _alloc(dev, r.shape[1], kind) for kind in (pa.capi.CLOSEST, pa.capi.ANY)
                for tri in (pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT) for which, r in (("big", d_big), ("small", d_small))}
        # `ctx` never calls pgpu_trace_bvh8: it owns no spill buffer, so the kernels can only be using the static scratch.
        # One 64-ray call per kernel variant first: the HIP runtime itself takes memory the first time a kernel is launched
        # (code object, kernel-argument pool), which is not what the check below is about. A 64-ray call needs one block's
        # worth of stack overflow, the calls that follow need hundreds of blocks' worth.
        d_tiny = d_big[:, :64].contiguous()
        for kind in (pa.capi.CLOSEST, pa.capi.ANY):
            for tri in (pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT):
                ctx.trace_bvh8_noalloc(ids[0], d_tiny, _alloc(dev, 64, kind), kind=kind, tri_mode=tri)
        ctx.sync()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(dev)[0]
        for kind in (pa.capi.CLOSEST, pa.capi.ANY):
            for tri in (pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT):
                ctx.trace_bvh8_noalloc(ids[0], d_big, outs[(kind, tri, "big")], kind=kind, tri_mode=tri)
                ctx.trace_bvh8_noalloc(ids[0], d_small, outs[(kind, tri, "small")], kind=kind, tri_mode=tri)
        ctx.sync()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(dev)[0] == free0, "device memory was allocated or freed during pgpu_trace_bvh8_noalloc"
        for kind in (pa.capi.CLOSEST, pa.capi.ANY):
            for tri in (pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT):
                for which, r in (("big", d_big), ("small", d_small)):
                    want = ref_ctx.trace_bvh8(ids[1], r, kind=kind, tri_mode=tri)
                    ref_ctx.sync()
                    got = outs[(kind, tri, which)]
                    for k in (("hit",) if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v")):
                        assert np.array_equal(_bits(got[k]), _bits(want[k])), f"kind {kind} tri {tri} {which}: {k}"
        hit = _bits(outs[(pa.capi.CLOSEST, pa.capi.TRI_MT, "big")]["obj"]) != pa.capi.MISS
        assert 0.2 < hit.mean() <= 1.0
        # sizes beyond the maximum are rejected before anything is launched
        L = pa.capi.lib()
        r = pa.capi.Rays(*[d_big.data_ptr() + i * 4 * d_big.shape[1] for i in range(8)])
        o = outs[(pa.capi.CLOSEST, pa.capi.TRI_MT, "big")]
        h = pa.capi.Hits(*[pa.capi.ptr(o[k]) for k in ("t", "obj", "leaf", "u", "v")], 0)
        rc = L.pgpu_trace_bvh8_noalloc(ctx._h, ids[0], pa.capi.CLOSEST, pa.capi.TRI_MT, C.byref(r), 0x40000000, C.byref(h), None, None)
        assert rc == pa.capi.ERR_INVALID and b"PGPU_NOALLOC_MAX_RAYS" in L.pgpu_last_error(ctx._h)
    finally:
        ctx.close()
        ref_ctx.close()
