"""Visit statistics of the bottom-level kernel inside the two-level loop (bench.py's configs[2] scene and diffuse batch):
how long is a traversal per batching-point visit, and what does a wave step cost? Usage: python tools/two_level_stats.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import pandora_amd as pa

blocks = int(os.environ.get("CITY_BLOCKS", "24"))
side = int(os.environ.get("SIDE", "4096"))
ctx = pa.accel.Context(0)
dev = torch.device("cuda:0")
mesh, tl = bench.city_scene(pa, blocks, svdag=int(os.environ.get("SVDAG", "64")))
tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
prim = bench.city_camera(pa, blocks, side, 891379)
first, _ = ctx.trace(torch.from_numpy(prim).to(dev))
prim_id = tl.prim_ids_of(first["leaf"].cpu().numpy(), first["obj"].cpu().numpy())
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"].cpu().numpy(), prim_id, seed=891379)
d_rays = torch.from_numpy(rays).to(dev)
ctx.set_stats(True)
out, cnt = ctx.trace(d_rays)
ctx.sync()
cnt = {**cnt, **{k: v for k, v in ctx.counters().items() if v}}
n = rays.shape[1]
lv = max(1, cnt["leaf_visits"])
print(json.dumps({"rays": n, "counters": cnt, "per_leaf_visit": {k: cnt[k] / lv for k in ("inner_visits", "leaf_pops", "tri_tests")},
                  "per_wave_step": {"ray_visits": (cnt["inner_visits"] + cnt["leaf_pops"]) / max(1, cnt["wave_steps"])},
                  "leaf_visits_per_ray": lv / n}, indent=1))
