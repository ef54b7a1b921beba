"""Times pgpu_trace on bench.py's configs[2] scene and diffuse batch under the bottom-level kernel's scheduling knobs.
Usage: python tools/two_level_sweep.py  (prints one line per setting)"""
import os
import sys
import time

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
# SHUFFLE=0: the bounce rays in the order of the pixels that spawned them (what a wavefront integrator hands over) instead of shuffled
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"].cpu().numpy(), prim_id, seed=891379,
                                     shuffle_seed=94892380 if os.environ.get("SHUFFLE", "1") == "1" else None)
grid = int(os.environ.get("SORT_GRID", "0"))  # upper bound for ray reordering: the batch sorted on the host by origin cell (+ direction octant), sort not timed
if grid:
    import numpy as np
    lo, hi = rays[0:3].min(axis=1, keepdims=True), rays[0:3].max(axis=1, keepdims=True)
    cell = np.minimum(grid - 1, ((rays[0:3] - lo) / (hi - lo) * grid).astype(np.int64))
    key = (cell[0] * grid + cell[2]) * grid + cell[1]
    if os.environ.get("SORT_OCT", "1") == "1":
        key = key * 8 + ((rays[3] > 0) * 1 + (rays[4] > 0) * 2 + (rays[5] > 0) * 4)
    cell_bits = int(os.environ.get("SORT_BITS", "0"))
    if cell_bits:  # `cell_bits` bits of cell, handed out to the axes by extent (halve the longest cell side), + 3 bits of octant
        ext = (hi - lo).ravel().astype(np.float64)
        bits = [0, 0, 0]
        for _ in range(cell_bits):
            a = int(np.argmax(ext / (2.0 ** np.array(bits))))
            bits[a] += 1
        key = np.zeros(rays.shape[1], np.int64)
        for a in range(3):
            n = 1 << bits[a]
            c = np.minimum(n - 1, ((rays[a] - lo[a]) / (hi[a] - lo[a]) * n).astype(np.int64))
            key = key * n + c
        key = key * 8 + ((rays[3] > 0) * 1 + (rays[4] > 0) * 2 + (rays[5] > 0) * 4)
        print("cell bits per axis", bits, flush=True)
    rays = np.ascontiguousarray(rays[:, np.argsort(key, kind="stable")])
d_rays = torch.from_numpy(rays).to(dev)
out, _ = ctx.trace(d_rays)


d_prim = torch.from_numpy(prim).to(dev)
out_any, _ = ctx.trace(d_rays, kind=pa.capi.ANY)
out_prim, _ = ctx.trace(d_prim)


def timed_one(r, o, kind, reps=6):
    ctx.trace(r, out=o, kind=kind)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.trace(r, out=o, kind=kind)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def timed():
    return timed_one(d_rays, out, pa.capi.CLOSEST)


def timed_more():
    return "any %.3f ms, primary %.3f ms" % (timed_one(d_rays, out_any, pa.capi.ANY), timed_one(d_prim, out_prim, pa.capi.CLOSEST))


print("checksum", {k: float(v.double().sum()) for k, v in out.items() if torch.is_tensor(v)}, flush=True)
settings = [dict(), dict(), dict(pol=1)]
if os.environ.get("SWEEP", "0") == "1":
    settings = [dict()] + [dict(chunk=c) for c in (32, 64, 256)] + [dict(refill=r) for r in (3, 4, 8, 12)] + [dict(leaf=l) for l in (4, 6, 12, 16)] + [dict(bpc=b) for b in (3, 5)] + [dict()]
if os.environ.get("SWEEP", "0") == "top":  # the top-level kernel's variants (pgpu_set_top_variant)
    settings = [dict(), dict(top=4), dict(top=4, parts=1), dict(parts=1), dict(top=4, pol=1), dict(top=2), dict(top=3)] + [dict(top=3, trefill=r, trips=t) for r in (8, 24, 32) for t in (2, 8)] + [dict(top=3, pol=1), dict(pol=1), dict()]
if os.environ.get("SWEEP", "0") == "parts":  # a batch as parts side by side (pgpu_set_batch_parts)
    settings = [dict(parts=1), dict(parts=2), dict(parts=3), dict(parts=4), dict(parts=2, top=3), dict(parts=3, top=3), dict(parts=2, pol=1), dict(parts=1)]
for s in settings:
    ctx.set_batch_parts(s.get("parts", 0))
    ctx.set_svdag_policy(s.get("pol", 0))  # 1: SVDAG early-out only for batching points that are not resident
    ctx.set_ray_layout(s.get("lanes", 0))
    ctx.set_round_lookahead(bool(s.get("look", 1)))
    ctx.set_top_variant(s.get("top", 1), s.get("trefill", 0), s.get("trips", 0))
    ctx.set_kernel_variant(s.get("refill", 0), 0, s.get("leaf", 0))
    ctx.set_tuning(s.get("bpc", 0), s.get("chunk", 0))
    print(s, "%.3f ms" % timed(), timed_more(), flush=True)
