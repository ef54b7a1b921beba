"""Where the time of building the two-level benchmark scene goes (bench.py --config two-level), phase by phase, with the batching
points' BVH8s from the host builder, from the device builder (trees returned) and built and left on the GPU.
Usage: python tools/scene_build_phases.py [city blocks per side, default 24]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import pandora_amd as pa

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rec = {"host_threads": pa.accel.host_threads()}
t = time.perf_counter()
mesh = pa.scenes.g_city(blocks, 18750)
rec["generate_s"] = time.perf_counter() - t
rec["triangles"] = mesh.n_tris
pos, idx = mesh.pos, mesh.idx
t = time.perf_counter()
group_of, ng = pa.accel.partition_prims(pos, idx, 65536)
rec["cut_s"] = time.perf_counter() - t
rec["batching_points"] = int(ng)
t = time.perf_counter()
order = np.argsort(group_of, kind="stable").astype(np.uint32)
counts = np.bincount(group_of, minlength=ng)
starts = np.concatenate([[0], np.cumsum(counts)])
groups = [np.ascontiguousarray(order[starts[g]:starts[g + 1]]) for g in range(ng)]
rec["group_lists_s"] = time.perf_counter() - t
ctx = pa.accel.Context(0)
ctx.build_upload_bvh8_device(pa.scenes.g_small().pos, pa.scenes.g_small().idx)  # (module load)
from concurrent.futures import ThreadPoolExecutor
t = time.perf_counter()
with ThreadPoolExecutor(pa.accel.host_threads()) as pool:
    hosts = list(pool.map(lambda g: pa.accel.build_bvh8(pos, idx, g), groups))
rec["bvh8_host_build_s"] = time.perf_counter() - t
t = time.perf_counter()
mid = ctx.upload_mesh(pos, idx)
for h in hosts:
    ctx.upload_bvh8(h, mid)
rec["bvh8_host_upload_s"] = time.perf_counter() - t
t = time.perf_counter()
devs, _ = pa.accel.build_bvh8_device(ctx, pos, idx, groups)
rec["bvh8_device_build_returned_s"] = time.perf_counter() - t
rec["bvh8_device_ms_on_device"] = ctx.last_kernel_ms()
t = time.perf_counter()
ids, _ = ctx.build_upload_bvh8_device(pos, idx, groups)
rec["bvh8_device_build_resident_s"] = time.perf_counter() - t
rec["bvh8_device_resident_ms_on_device"] = ctx.last_kernel_ms()
t = time.perf_counter()
dags = pa.accel.build_svdags(pos, idx, groups, 64, device_ctx=ctx)
rec["svdags_device_s"] = time.perf_counter() - t
ctx.close()
print(json.dumps(rec))
