"""Ad-hoc timing of the two-level wavefront loop (development aid). Usage: quick_bench_two_level.py [scene] [n_side] [prims_per_point] [svdag_res]
scene: atrium | city<N>x<T> (N blocks per side, T triangles per block)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pandora_amd as pa

scene = sys.argv[1] if len(sys.argv) > 1 else "atrium"
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
ppp = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
svdag = int(sys.argv[4]) if len(sys.argv) > 4 else 0
cache_mb = float(sys.argv[5]) if len(sys.argv) > 5 else -1.0  # >= 0: leaves streamed from host memory through a cache of that size (0 = all fit)
t0 = time.time()
if scene == "atrium":
    mesh = pa.scenes.g_atrium()
    cam = ((3, 3, 9), (28, 5, 9), (0, 1, 0), 70)
else:
    nb, tpb = scene[4:].split("x")
    mesh = pa.scenes.g_city(int(nb), int(tpb))
    size = 20.0 * int(nb)
    cam = ((0.1 * size, 14.0, 0.1 * size), (0.6 * size, 0.0, 0.6 * size), (0, 1, 0), 60)
print("mesh %d tris (%.1fs)" % (mesh.n_tris, time.time() - t0), flush=True)
t0 = time.time()
tl = pa.accel.build_two_level(mesh.pos, mesh.idx, ppp, svdag_res=svdag)
print("two-level build %.1fs: %d batching points, top depth %d" % (time.time() - t0, len(tl.bvhs), tl.top.depth), flush=True)
ctx = pa.accel.Context(0)
tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), streamed=cache_mb >= 0)
if cache_mb >= 0:
    ctx.set_leaf_cache(int(cache_mb * (1 << 20)))
    if "LEAF_HOLD" in os.environ:
        ctx.set_leaf_hold(int(os.environ["LEAF_HOLD"]))
    ctx.set_leaf_copy(int(os.environ.get("LEAF_COPY", "0")))  # 0: one gather kernel per launch group, 1: one hipMemcpyAsync per image
    print("streamed leaves: %.1f MB of BVH images in host memory, cache %s" % (
        sum(b.n_nodes * 256 + b.leaf_idx.size * 48 for b in tl.bvhs) / 1e6, "holds all" if cache_mb == 0 else "%.0f MB" % cache_mb), flush=True)
prim = pa.scenes.primary_rays(cam[0], cam[1], cam[2], cam[3], n_side, n_side)
d = torch.from_numpy(prim).cuda()
def run(rays, kind, label, reps=4):
    best, cnt = 1e30, None
    for it in range(reps):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        out, cnt = ctx.trace(rays, kind=kind)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t1)
    n = rays.shape[1]
    print("%s: wall %.3f ms (%.1f Mrays/s), kernels %.3f ms, rounds %d, leaf visits/ray %.2f, svdag culled %d of %d, leaf loads %d (%.1f MB)" % (
        label, best * 1e3, n / best / 1e6, ctx.last_kernel_ms(), cnt["rounds"], cnt["leaf_visits"] / n, cnt["svdag_culled"], cnt["svdag_tests"],
        cnt["leaf_loads"], cnt["streamed_bytes"] / 1e6), flush=True)
    return out
out = run(d, pa.capi.CLOSEST, "primary closest")
t = out["t"].cpu().numpy(); obj = out["obj"].cpu().numpy().view(np.uint32); leaf = out["leaf"].cpu().numpy().view(np.uint32)
hit = obj != 0xFFFFFFFF
print("hit frac %.3f" % hit.mean())
primid = np.full(obj.shape, 0xFFFFFFFF, np.uint32)
for l in np.unique(leaf[hit]):
    m = hit & (leaf == l)
    primid[m] = tl.bvhs[int(tl.top.leaf_order[int(l)])].prim_ids[obj[m]]
diff = pa.scenes.diffuse_bounce_rays(mesh, prim, t, primid)
dd = torch.from_numpy(diff).cuda()
run(dd, pa.capi.CLOSEST, "diffuse closest")
run(dd, pa.capi.ANY, "diffuse any")
# the same loop through the sharded-loop building blocks (one rank: no exchange, torch sorts/gathers per round)
from pandora_amd import sharded
if cache_mb >= 0:
    sys.exit(0)
eng = sharded.GpuEngine(ctx, torch.device("cuda", 0))
owner = np.zeros(tl.n_leaves, np.int64)
for label, rays_d, kind in (("sharded-loop primary closest", d, pa.capi.CLOSEST), ("sharded-loop diffuse closest", dd, pa.capi.CLOSEST)):
    best = 1e30
    for it in range(3):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st, info = sharded.trace_sharded(eng, rays_d, owner, kind)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t1)
    print("%s: wall %.3f ms (%.1f Mrays/s), rounds %d" % (label, best * 1e3, rays_d.shape[1] / best / 1e6, info["rounds"]), flush=True)
