"""What the per-batching-point work lists are worth to the bottom level -- the measurement behind DESIGN 3.4's verdict on a fused
(no parking) two-level kernel, which would hand rays to the BVH8 traversal in arrival order. bench.py's configs[2] scene and
diffuse batch; first top-level episode, then the bottom level (pgpu_trace_leaves) on the SAME parked rays as
  sorted   -- by batching point (what pgpu_trace's binning produces),
  arrival  -- in ray-id order (a shuffled batch: neighbours in the list are at unrelated batching points),
  random   -- a random permutation.
Usage: python tools/leaf_coherence.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pandora_amd as pa

blocks = int(os.environ.get("CITY_BLOCKS", "24"))
side = int(os.environ.get("SIDE", "4096"))
ctx = pa.accel.Context(0)
dev = torch.device("cuda:0")
mesh, tl = bench.city_scene(pa, blocks, svdag=64, ctx=ctx)
tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
prim = bench.city_camera(pa, blocks, side, 891379)
first, _ = ctx.trace(torch.from_numpy(prim).to(dev))
prim_id = tl.prim_ids_of(first["leaf"].cpu().numpy(), first["obj"].cpu().numpy())
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"].cpu().numpy(), prim_id, seed=891379)
del first
n = rays.shape[1]
d_rays = torch.from_numpy(rays).to(dev)
i32 = dict(dtype=torch.int32, device=dev)
node, stack = torch.empty(n, **i32), torch.empty(n, dtype=torch.int64, device=dev)
hits = {k: torch.empty(n, dtype=torch.float32 if k in ("t", "u", "v") else torch.int32, device=dev) for k in ("t", "obj", "leaf", "u", "v", "visits")}
parked = torch.empty(n, **i32)
ctx.ray_state_init(d_rays[7], node, stack, hits)
ctx.top_episode(pa.capi.CLOSEST, d_rays, None, n, node, stack, hits["t"], hits["visits"], None, parked)
torch.cuda.synchronize()
pl = parked.cpu().numpy().view(np.uint32)
ids = np.nonzero(pl != 0xFFFFFFFF)[0].astype(np.uint32)
leaf = pl[ids]
print("rays %d, parked after the first episode %d, batching points %d" % (n, ids.size, tl.n_leaves), flush=True)
orders = {"sorted by batching point": np.argsort(leaf, kind="stable"), "arrival (ray-id) order": np.arange(ids.size),
          "random permutation": np.random.default_rng(1).permutation(ids.size)}
t0_state = hits["t"].clone()
res = {}
for lanes in (0, 4):
    ctx.set_ray_layout(lanes)
    for name, o in orders.items():
        wr, wl = torch.from_numpy(ids[o].view(np.int32).copy()).to(dev), torch.from_numpy(leaf[o].view(np.int32).copy()).to(dev)
        best = 1e9
        for _ in range(4):
            hits["t"].copy_(t0_state)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.trace_leaves(pa.capi.CLOSEST, pa.capi.TRI_MT, d_rays, wr, wl, ids.size, hits)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        res["%s, %s" % ("pair layout" if lanes == 0 else "quad layout", name)] = best
        print("%-60s %.3f ms  (%.2f ns per leaf visit)" % ("%s, %s" % ("pair layout" if lanes == 0 else "quad layout", name), best, best * 1e6 / ids.size), flush=True)
import json
print(json.dumps({"rays": int(n), "leaf_visits": int(ids.size), "ms": res}))
