import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pandora_amd as pa
mesh = pa.scenes.g_atrium(262144)
hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
ctx = pa.accel.Context(0)
bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, 4096, 4096)
dprim = torch.from_numpy(prim).cuda()
out = ctx.trace_bvh8(bvh, dprim); ctx.sync()
t = out["t"].cpu().numpy(); obj = out["obj"].cpu().numpy().view(np.uint32)
primid = np.where(obj == 0xFFFFFFFF, np.uint32(0xFFFFFFFF), hb.prim_ids[np.minimum(obj, hb.prim_ids.size - 1)])
diff = pa.scenes.diffuse_bounce_rays(mesh, prim, t, primid)
dd = torch.from_numpy(diff).cuda()
o2 = ctx.trace_bvh8(bvh, dd); ctx.sync()
for label, rays, o, kind in (("primary", dprim, out, pa.capi.CLOSEST), ("diffuse", dd, o2, pa.capi.CLOSEST), ("diffuse any", dd, None, pa.capi.ANY)):
    ms = []
    for it in range(8):
        ctx.trace_bvh8(bvh, rays, kind=kind) if o is None else ctx.trace_bvh8(bvh, rays, out=o); ctx.sync(); ms.append(ctx.last_kernel_ms())
    print("%s: best %.3f ms median %.3f ms" % (label, min(ms), sorted(ms)[4]), flush=True)
# how much would reordering the incoherent batch buy? (host-side sort, not timed: an upper bound for a device binning pass)
def morton_key(o, d, bits):
    lo, hi = o.min(axis=1, keepdims=True), o.max(axis=1, keepdims=True)
    q = np.minimum(((o - lo) / (hi - lo) * (1 << bits)).astype(np.uint64), (1 << bits) - 1)
    key = np.zeros(o.shape[1], np.uint64)
    for b in range(bits):
        for a in range(3):
            key |= ((q[a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    octant = (d[0] > 0).astype(np.uint64) | ((d[1] > 0).astype(np.uint64) << np.uint64(1)) | ((d[2] > 0).astype(np.uint64) << np.uint64(2))
    return key, octant
for bits, use_oct in ((3, False), (5, False), (5, True), (8, True), (0, True)):
    key, octant = morton_key(diff[0:3], diff[3:6], max(bits, 1))
    k = (key if bits else np.zeros_like(key)) * np.uint64(8) + (octant if use_oct else 0)
    order = np.argsort(k, kind="stable")
    ds = torch.from_numpy(np.ascontiguousarray(diff[:, order])).cuda()
    ms = []
    for it in range(6):
        ctx.trace_bvh8(bvh, ds, out=o2); ctx.sync(); ms.append(ctx.last_kernel_ms())
    print("diffuse sorted by %d-bit/axis origin cell%s (%d bins): best %.3f ms" % (bits, " + direction octant" if use_oct else "", len(np.unique(k)), min(ms)), flush=True)
