"""GPU parity of leaves streamed from host memory (BASELINE config 5): whatever the size of the HBM leaf cache, the
two-level loop must return what it returns with every leaf resident (bit-exact, counters included)."""
import numpy as np
import pytest

import pandora_amd as pa

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.asarray(a).view(np.uint32)


def _rays(mesh):
    prim = pa.scenes.primary_rays((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 80, 80)
    rnd = pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), 5000, seed=21)
    return np.ascontiguousarray(np.concatenate([prim, rnd], axis=1))


@pytest.mark.parametrize("svdag", [0, 32])
def test_streamed_leaves_match_resident(svdag):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=svdag)
    rays = _rays(mesh)
    ref_ctx = pa.accel.Context(0)
    tl.upload(ref_ctx, ref_ctx.upload_mesh(mesh.pos, mesh.idx))
    want = {k: ref_ctx.trace(rays, kind=k) for k in (pa.capi.CLOSEST, pa.capi.ANY)}
    ref_ctx.close()

    ctx = pa.accel.Context(0)
    tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), streamed=True)
    slot = max(((b.n_nodes * 256 + 255) // 256) * 256 + (((b.leaf_idx.size + 1) * 48 + 255) // 256) * 256 for b in tl.bvhs)  # (records end in one zero record of padding)
    for budget, label in ((0, "all fit"), (3 * slot + 17, "3 slots"), (slot, "1 slot")):
        ctx.set_leaf_cache(budget)
        for kind in (pa.capi.CLOSEST, pa.capi.ANY):
            got, cnt = ctx.trace(rays, kind=kind)
            exp, ecnt = want[kind]
            keys = ("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")
            for k in keys:
                assert np.array_equal(_bits(got[k]), _bits(exp[k])), f"{label}: {k}"
            for k in ("svdag_tests", "svdag_culled", "leaf_visits", "rounds"):
                assert cnt[k] == ecnt[k], f"{label}: {k}"
            assert cnt["leaf_loads"] > 0 or budget == 0 and kind == pa.capi.ANY  # (second trace with everything cached loads nothing)
            if budget and budget < 4 * slot:
                assert cnt["leaf_loads"] > budget // slot, f"{label}: more BVHs were needed than fit, so slots must have been recycled (loads {cnt['leaf_loads']})"
            assert cnt["streamed_bytes"] >= cnt["leaf_loads"] * 256 or cnt["leaf_loads"] == 0
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
