"""Does the two-level loop gain from running two half-batches side by side (the top-level episode of one half next to the
bottom level of the other)? Two contexts, each with the scene, each tracing half of bench.py's configs[2] batch on a stream of
its own from a host thread of its own -- against one context tracing the whole batch. Usage: python tools/overlap_halves.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pandora_amd as pa

blocks = int(os.environ.get("CITY_BLOCKS", "24"))
side = int(os.environ.get("SIDE", "4096"))
parts = int(os.environ.get("PARTS", "2"))
dev = torch.device("cuda:0")
ctxs = [pa.accel.Context(0) for _ in range(parts)]
mesh, tl = bench.city_scene(pa, blocks, svdag=64, ctx=ctxs[0])
for c in ctxs:
    tl.upload(c, c.upload_mesh(mesh.pos, mesh.idx))
prim = bench.city_camera(pa, blocks, side, 891379)
first, _ = ctxs[0].trace(torch.from_numpy(prim).to(dev))
prim_id = tl.prim_ids_of(first["leaf"].cpu().numpy(), first["obj"].cpu().numpy())
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"].cpu().numpy(), prim_id, seed=891379)
del first
n = rays.shape[1]
d_all = torch.from_numpy(rays).to(dev)
cuts = np.linspace(0, n, parts + 1).astype(np.int64)
d_part = [torch.from_numpy(np.ascontiguousarray(rays[:, a:b])).to(dev) for a, b in zip(cuts[:-1], cuts[1:])]
out_all, _ = ctxs[0].trace(d_all)
outs = [c.trace(r)[0] for c, r in zip(ctxs, d_part)]
torch.cuda.synchronize()
for k in ("t", "obj"):
    assert torch.equal(torch.cat([o[k] for o in outs]), out_all[k]), k


def whole(reps=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctxs[0].trace(d_all, out=out_all)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def side_by_side(reps=8):
    def run(i):
        for _ in range(reps):
            ctxs[i].trace(d_part[i], out=outs[i])
    torch.cuda.synchronize()
    th = [threading.Thread(target=run, args=(i,)) for i in range(parts)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def one_after_the_other(reps=8):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(parts):
            ctxs[i].trace(d_part[i], out=outs[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    print("whole batch, one context            %.3f ms" % whole(), flush=True)
    print("%d parts, one after the other        %.3f ms" % (parts, one_after_the_other()), flush=True)
    print("%d parts side by side (own streams)  %.3f ms" % (parts, side_by_side()), flush=True)
