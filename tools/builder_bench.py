"""Bottom-level BVH8 construction, host against device (pgpu_build_bvh8 on the box's host threads / pgpu_build_bvh8_device): wall
time of the builds alone, and what a ray pays in the trees (oracle counters on a sample of the bench's R-diffuse batch for the
atrium; node counts for the city's batching points). One JSON line.
Usage: python tools/builder_bench.py [city blocks per side, default 24]"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import oracle_binding as ob
import pandora_amd as pa

MISS = 0xFFFFFFFF
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 24
threads = pa.accel.host_threads()
ctx = pa.accel.Context(0)
rec = {"host_threads": threads}

# ---- configs[1]: the atrium as one tree
mesh = pa.scenes.g_atrium(262144)
t = time.perf_counter()
host = pa.accel.build_bvh8(mesh.pos, mesh.idx)
host_s = time.perf_counter() - t
pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)  # (first call: module load, allocator warm-up)
t = time.perf_counter()
(dev,), fell_back = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)
dev_s = time.perf_counter() - t
dev_ms = ctx.last_kernel_ms()
prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, 512, 512, seed=891379)
first = ob.OracleBvh8.from_host(host, mesh).trace(prim, threads=threads)
obj = first["obj"].view(np.uint32)
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"], np.where(obj == MISS, MISS, host.prim_ids[np.minimum(obj, host.prim_ids.size - 1)]), seed=891379)
visits = {}
for name, hb in (("host", host), ("device", dev)):
    c = ob.OracleBvh8.from_host(hb, mesh).trace(rays, threads=threads)["counters"]
    visits[name] = {k: c[k] / rays.shape[1] for k in ("inner_visits", "leaf_pops", "tri_tests")}
rec["atrium"] = {"triangles": mesh.n_tris, "host_build_s": host_s, "device_build_s_wall": dev_s, "device_build_ms_on_device": dev_ms, "host_built_groups": fell_back,
                 "nodes": {"host": host.n_nodes, "device": dev.n_nodes}, "depth": {"host": host.depth, "device": dev.depth}, "visits_per_diffuse_ray": visits}

# ---- configs[2]: the city's batching points, all trees of the scene
mesh = pa.scenes.g_city(blocks, 18750)
group_of, ng = pa.accel.partition_prims(mesh.pos, mesh.idx, 65536)
order = np.argsort(group_of, kind="stable").astype(np.uint32)
starts = np.concatenate([[0], np.cumsum(np.bincount(group_of, minlength=ng))])
groups = [np.ascontiguousarray(order[starts[g]:starts[g + 1]]) for g in range(ng)]
t = time.perf_counter()
with ThreadPoolExecutor(threads) as pool:
    hosts = list(pool.map(lambda g: pa.accel.build_bvh8(mesh.pos, mesh.idx, g), groups))
host_s = time.perf_counter() - t
t = time.perf_counter()
devs, fell_back = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, groups)
dev_s = time.perf_counter() - t
rec["city"] = {"triangles": mesh.n_tris, "batching_points": ng, "host_build_s": host_s, "device_build_s_wall": dev_s, "device_build_ms_on_device": ctx.last_kernel_ms(),
               "host_built_groups": fell_back, "nodes": {"host": sum(h.n_nodes for h in hosts), "device": sum(d.n_nodes for d in devs)},
               "deepest": {"host": max(h.depth for h in hosts), "device": max(d.depth for d in devs)}}
ctx.close()
print(json.dumps(rec))
