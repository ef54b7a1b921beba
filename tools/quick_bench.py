"""Ad-hoc timing of the single-level kernel (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pandora_amd as pa

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
mesh = pa.scenes.g_atrium(int(sys.argv[2]) if len(sys.argv) > 2 else 262144)
t0 = time.time(); hb = pa.accel.build_bvh8(mesh.pos, mesh.idx); print("build %.2fs nodes %d depth %d tris %d" % (time.time() - t0, hb.n_nodes, hb.depth, mesh.n_tris), flush=True)
ctx = pa.accel.Context(0)
lanes = int(os.environ.get("LANES", "4"))
ctx.set_ray_layout(lanes)
bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, n_side, n_side)
dprim = torch.from_numpy(prim).cuda()
ctx.set_stats(True)
out = ctx.trace_bvh8(bvh, dprim); ctx.sync()
print("primary: %.3f ms, %.1f Mrays/s" % (ctx.last_kernel_ms(), prim.shape[1] / ctx.last_kernel_ms() / 1e3), ctx.counters(), flush=True)
ctx.set_stats(False)
ms = []
for it in range(5):
    ctx.trace_bvh8(bvh, dprim, out=out); ctx.sync(); ms.append(ctx.last_kernel_ms())
print("primary (stats off): best %.3f ms -> %.1f Mrays/s" % (min(ms), prim.shape[1] / min(ms) / 1e3), flush=True)
ctx.set_stats(True)
t = out["t"].cpu().numpy(); obj = out["obj"].cpu().numpy().view(np.uint32)
print("primary hit frac", (obj != 0xFFFFFFFF).mean())
primid = np.where(obj == 0xFFFFFFFF, np.uint32(0xFFFFFFFF), hb.prim_ids[np.minimum(obj, hb.prim_ids.size - 1)])
diff = pa.scenes.diffuse_bounce_rays(mesh, prim, t, primid)
ddiff = torch.from_numpy(diff).cuda()
n = diff.shape[1]
o2 = ctx.trace_bvh8(bvh, ddiff); ctx.sync()
print("diffuse (stats on): %.3f ms, %.1f Mrays/s" % (ctx.last_kernel_ms(), n / ctx.last_kernel_ms() / 1e3), ctx.counters(), flush=True)
lpr = lanes
if lanes == 4:
    variants = ((3, 0, 64, 0, 0), (2, 0, 64, 0, 0), (4, 0, 64, 0, 0), (6, 0, 64, 0, 0), (3, 0, 64, 0, 2), (3, 0, 64, 0, 3), (3, 0, 64, 0, 5), (3, 0, 64, 0, 6),
                (3, 0, 64, 8, 0), (3, 0, 64, 11, 0), (3, 0, 64, 12, 0), (3, 0, 64, 16, 0), (3, 0, 64, 21, 0), (3, 0, 64, 28, 0), (3, 0, 64, 64, 0), (3, 4, 64, 0, 0), (3, 0, 32, 0, 0), (3, 0, 128, 0, 0))
else:
    variants = ((0, 0, 64, 0, 0), (4, 0, 64, 0, 0), (8, 0, 64, 0, 0), (12, 0, 64, 0, 0), (6, 0, 64, 0, 4), (6, 0, 64, 0, 6), (6, 0, 64, 0, 10), (6, 0, 64, 0, 12), (6, 0, 64, 0, 16),
                (6, 0, 64, 8, 0), (6, 0, 64, 14, 0), (6, 0, 64, 20, 0), (6, 0, 64, 64, 0), (6, 3, 64, 0, 0), (6, 2, 64, 0, 0), (6, 0, 128, 0, 0), (6, 0, 256, 0, 0))
for rmin, bpc, chunk, win, lmin in variants:
    ctx.set_kernel_variant(rmin, win, lmin)
    ctx.set_tuning(bpc, chunk)
    ctx.set_stats(True)
    ctx.trace_bvh8(bvh, ddiff, out=o2); ctx.sync()
    cn = ctx.counters()
    ctx.set_stats(False)
    ms = []
    for it in range(5):
        ctx.trace_bvh8(bvh, ddiff, out=o2); ctx.sync(); ms.append(ctx.last_kernel_ms())
    g = 64 // lpr
    print("diffuse lpr=%d refill_min=%d bpc=%d chunk=%d win=%d leaf_min=%d: best %.3f ms median %.3f ms -> %.1f Mrays/s | steps %.1fM leaf passes %.1fM refills %.1fM util %.2f" % (
        lpr, rmin, bpc, chunk, win, lmin, min(ms), sorted(ms)[2], n / min(ms) / 1e3, cn["wave_steps"] / 1e6, cn["leaf_passes"] / 1e6, cn["refill_passes"] / 1e6,
        (cn["inner_visits"] + cn["leaf_pops"]) / max(1, cn["wave_steps"] * g)), flush=True)
ctx.set_kernel_variant(0, 0, 0)
ctx.set_tuning(0, 0)
ms = []
for it in range(5):
    ctx.trace_bvh8(bvh, ddiff, kind=pa.capi.ANY); ctx.sync(); ms.append(ctx.last_kernel_ms())
print("any-hit diffuse: best %.3f ms -> %.1f Mrays/s" % (min(ms), n / min(ms) / 1e3))
print("hit frac diffuse", (o2["obj"].cpu().numpy().view(np.uint32) != 0xFFFFFFFF).mean())
