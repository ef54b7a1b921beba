"""The two-level loop against a single-level BVH8 over the SAME scene and rays (the 11.4 M-triangle city and the R-diffuse batch of
bench.py --config two-level): what the reference's batching structure -- PauseableBVH4, SVDAG tests, parking and per-point queues --
costs next to one tree that holds everything. The bench's own single-level headline is on a 260 K-triangle atrium and says little
about that. Prints one JSON line.
Usage: python tools/single_vs_two_level.py [city blocks, default 24]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pandora_amd as pa

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
ctx = pa.accel.Context(0)
mesh, tl = bench.city_scene(pa, blocks, svdag=64, ctx=ctx)
mid = ctx.upload_mesh(mesh.pos, mesh.idx)
tl.upload(ctx, mid)
prim = bench.city_camera(pa, blocks, 4096, 891379)
first, _ = ctx.trace(torch.from_numpy(prim).to(dev))
prim_id = tl.prim_ids_of(first["leaf"].cpu().numpy(), first["obj"].cpu().numpy())
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"].cpu().numpy(), prim_id, seed=891379)
d = torch.from_numpy(rays).to(dev)
n = rays.shape[1]


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out2, cnt = ctx.trace(d)
two_ms = timed(lambda: ctx.trace(d, out=out2))
ctx.set_svdag_policy(1)
two_nosvdag_ms = timed(lambda: ctx.trace(d, out=out2))
ctx.set_svdag_policy(0)
rec = {"triangles": mesh.n_tris, "rays": n, "two_level": {"ms": two_ms, "Mrays_s": n / two_ms / 1e3, "batching_point_visits_per_ray": cnt["leaf_visits"] / n, "rounds": cnt["rounds"]},
       "two_level_without_svdag_tests_on_resident_points": {"ms": two_nosvdag_ms, "Mrays_s": n / two_nosvdag_ms / 1e3}}
t0 = time.perf_counter()
hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
rec["single_level_host_tree"] = {"build_s": time.perf_counter() - t0, "nodes": hb.n_nodes, "depth": hb.depth}
b1 = ctx.upload_bvh8(hb, mid)
o1 = ctx.trace_bvh8(b1, d)
ms = timed(lambda: ctx.trace_bvh8(b1, d, out=o1))
rec["single_level_host_tree"].update({"ms": ms, "Mrays_s": n / ms / 1e3})
same_t = int((o1["t"].view(torch.int32) == out2["t"].view(torch.int32)).sum())
rec["rays_with_the_same_nearest_distance_in_both_structures"] = same_t / n
t0 = time.perf_counter()
(b2,), _ = ctx.build_upload_bvh8_device(mesh.pos, mesh.idx)
rec["single_level_device_tree"] = {"build_s": time.perf_counter() - t0}
o2 = ctx.trace_bvh8(int(b2), d)
ms = timed(lambda: ctx.trace_bvh8(int(b2), d, out=o2))
rec["single_level_device_tree"].update({"ms": ms, "Mrays_s": n / ms / 1e3})
print(json.dumps(rec))
ctx.close()
