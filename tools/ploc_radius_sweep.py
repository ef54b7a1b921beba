"""The device builder's clustering radius (PANDORA_BVH8_DEVICE_PLOC_RADIUS) against what a ray pays in the tree: oracle counters per
diffuse ray on the benchmark atrium, build time on the device, and the same for the city's 256 batching points as one two-level
trace. Usage: python tools/ploc_radius_sweep.py [radii, default 1 2 3 4 6 8 16 32]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch  # noqa: F401  (first: one HIP runtime per process)

import oracle_binding as ob
import pandora_amd as pa

MISS = 0xFFFFFFFF
radii = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 16, 32]
ctx = pa.accel.Context(0)
threads = pa.accel.host_threads()
mesh = pa.scenes.g_atrium(262144)
host = pa.accel.build_bvh8(mesh.pos, mesh.idx)
prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, 512, 512, seed=891379)
first = ob.OracleBvh8.from_host(host, mesh).trace(prim, threads=threads)
obj = first["obj"].view(np.uint32)
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"], np.where(obj == MISS, MISS, host.prim_ids[np.minimum(obj, host.prim_ids.size - 1)]), seed=891379)
city = pa.scenes.g_city(24, 18750)
group_of, ng = pa.accel.partition_prims(city.pos, city.idx, 65536)
order = np.argsort(group_of, kind="stable").astype(np.uint32)
starts = np.concatenate([[0], np.cumsum(np.bincount(group_of, minlength=ng))])
groups = [np.ascontiguousarray(order[starts[g]:starts[g + 1]]) for g in range(ng)]
out = []
for tree, r in [("lbvh", 0)] + [("ploc", r) for r in radii]:
    os.environ["PANDORA_BVH8_DEVICE_TREE"] = tree
    os.environ["PANDORA_BVH8_DEVICE_PLOC_RADIUS"] = str(max(r, 1))
    pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)
    (dev,), fell_back = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)
    ms = ctx.last_kernel_ms()
    rec = {"tree": tree, "radius": r, "atrium_build_ms": ms, "atrium_nodes": dev.n_nodes, "atrium_depth": dev.depth, "host_built": fell_back}
    for name, rr in (("diffuse", rays), ("primary", prim)):
        c = ob.OracleBvh8.from_host(dev, mesh).trace(rr, threads=threads)["counters"]
        rec[name] = {k: round(c[k] / rr.shape[1], 3) for k in ("inner_visits", "leaf_pops", "tri_tests")}
    devs, fell_back = pa.accel.build_bvh8_device(ctx, city.pos, city.idx, groups)
    rec.update({"city_build_ms": ctx.last_kernel_ms(), "city_nodes": sum(d.n_nodes for d in devs), "city_deepest": max(d.depth for d in devs), "city_host_built": fell_back})
    print(json.dumps(rec), flush=True)
    out.append(rec)
ctx.close()
