"""Times pgpu_trace_bvh8 on bench.py's configs[1] scene and R-diffuse batch under the kernel's scheduling knobs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pandora_amd as pa

side = int(os.environ.get("SIDE", "4096"))
ctx = pa.accel.Context(0)
dev = torch.device("cuda:0")
mesh = pa.scenes.g_atrium(262144)
hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, side, side, seed=891379)
first = ctx.trace_bvh8(bvh, torch.from_numpy(prim).to(dev))
obj = first["obj"].cpu().numpy().view(np.uint32)
prim_id = np.where(obj == 0xFFFFFFFF, 0xFFFFFFFF, hb.prim_ids[np.minimum(obj, hb.prim_ids.size - 1)])
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"].cpu().numpy(), prim_id, seed=891379)
d_rays = torch.from_numpy(rays).to(dev)
out = ctx.trace_bvh8(bvh, d_rays)


def timed(reps=8):
    ctx.trace_bvh8(bvh, d_rays, out=out)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        ctx.trace_bvh8(bvh, d_rays, out=out)
        ctx.sync()
        ms.append(ctx.last_kernel_ms())
    return sum(ms) / len(ms)


settings = [dict()] + [dict(refill=r, leaf=l) for r in (2, 3, 4, 5) for l in (3, 4, 5, 6)] + [dict(chunk=c) for c in (64, 128, 256)] + \
           [dict(window=w) for w in (8, 10, 12, 16)]
print("rays", rays.shape[1])
for s in settings:
    ctx.set_kernel_variant(s.get("refill", 0), s.get("window", 0), s.get("leaf", 0))
    ctx.set_tuning(0, s.get("chunk", 0))
    print(s, "%.3f ms" % timed(), flush=True)
