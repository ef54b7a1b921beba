"""GPU parity against the committed golden vectors (tests/golden, produced by the restated loops on
the reference's simd headers): the HIP path must reproduce them bit for bit through the C ABI."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.asarray(a).view(np.uint32)


def test_bvh8_golden(gpu_ctx):
    z, mesh, hb, _ = ob.load_golden_bvh8()
    host = pa.accel.HostBvh8(np.ascontiguousarray(z["nodes"]), np.ascontiguousarray(z["leaf_idx"]), int(z["root"]), 0, np.ascontiguousarray(z["prim_ids"]))
    bvh = gpu_ctx.upload_bvh8(host, gpu_ctx.upload_mesh(z["pos"], z["idx"]))
    rays = np.ascontiguousarray(z["rays"])
    for tri, window in ((0, 0), (1, 0), (0, 3), (1, 3)):
        gpu_ctx.set_kernel_variant(3, window, 0)
        got = gpu_ctx.trace_bvh8(bvh, rays, tri_mode=tri)
        for k in ("t", "obj", "u", "v"):
            bad = np.nonzero(_bits(got[k]) != _bits(z[f"tri{tri}_{k}"]))[0]
            assert bad.size == 0, f"tri{tri} {k}: {bad.size} rays differ, first {bad[:8]}"
        assert np.array_equal(_bits(gpu_ctx.trace_bvh8(bvh, rays, kind=pa.capi.ANY, tri_mode=tri)["hit"]), z[f"tri{tri}_any"])


def test_bvh8_golden_from_device_memory(gpu_ctx):
    import torch
    z, mesh, hb, _ = ob.load_golden_bvh8()
    host = pa.accel.HostBvh8(np.ascontiguousarray(z["nodes"]), np.ascontiguousarray(z["leaf_idx"]), int(z["root"]), 0, np.ascontiguousarray(z["prim_ids"]))
    bvh = gpu_ctx.upload_bvh8(host, gpu_ctx.upload_mesh(z["pos"], z["idx"]))
    d = torch.from_numpy(np.ascontiguousarray(z["rays"])).cuda()
    got = gpu_ctx.trace_bvh8(bvh, d, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for k in ("t", "obj", "u", "v"):
        assert np.array_equal(_bits(got[k].cpu().numpy()), _bits(z[f"tri0_{k}"])), k


def test_two_level_golden(gpu_ctx):
    z, mesh, tl, _ = ob.load_golden_two_level()
    ctx = pa.accel.Context(0)  # own context: the top level is per context
    try:
        mesh_id = ctx.upload_mesh(z["pos"], z["idx"])
        ids = [ctx.upload_bvh8(pa.accel.HostBvh8(np.ascontiguousarray(b.nodes), np.ascontiguousarray(b.leaf_idx), b.root, 0, np.ascontiguousarray(b.prim_ids)), mesh_id) for b in tl.bvhs]
        ctx.upload_svdags(pa.accel.HostSvdags(np.ascontiguousarray(z["dag_words"]), np.ascontiguousarray(z["dag_descs"]), 0))
        leaves = np.array([[ids[int(s)], int(s)] for s in z["leaf_order"]], np.uint32)
        ctx.upload_top_bvh4(pa.accel.HostBvh4(np.ascontiguousarray(z["top_nodes"]), z["leaf_order"], int(z["top_root"]), 0), leaves)
        rays = np.ascontiguousarray(z["rays"])
        got, cnt = ctx.trace(rays)
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            bad = np.nonzero(_bits(got[k]) != _bits(z[f"closest_{k}"]))[0]
            assert bad.size == 0, f"{k}: {bad.size} rays differ, first {bad[:8]}"
        assert [cnt[k] for k in ("svdag_tests", "svdag_culled", "leaf_visits")] == list(z["closest_counters"])
        a, cnt = ctx.trace(rays, kind=pa.capi.ANY)
        assert np.array_equal(_bits(a["hit"]), z["any_hit"]) and np.array_equal(_bits(a["visits"]), z["any_visits"])
        assert [cnt[k] for k in ("svdag_tests", "svdag_culled", "leaf_visits")] == list(z["any_counters"])
    finally:
        ctx.close()
