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
