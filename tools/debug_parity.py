"""Development aid: prints the first mismatching rays (all fields) of the single-level kernel vs the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_binding as ob
import pandora_amd as pa

mesh = pa.scenes.g_small()
hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
ctx = pa.accel.Context(0)
bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
orc = ob.OracleBvh8.from_host(hb, mesh)
res = int(sys.argv[1]) if len(sys.argv) > 1 else 96
prim = pa.scenes.primary_rays((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, res, res)
ctx.set_stats(True)
got = ctx.trace_bvh8(bvh, prim)
print("gpu counters", ctx.counters())
want = orc.trace(prim)
print("orc counters", want["counters"])
for k in ("t", "obj", "u", "v"):
    bad = np.nonzero(np.asarray(got[k]).view(np.uint32) != np.asarray(want[k]).view(np.uint32))[0]
    print(k, "mismatches", bad.size, "first", bad[:6], "got", np.asarray(got[k])[bad[:6]], "want", np.asarray(want[k])[bad[:6]])
ga = ctx.trace_bvh8(bvh, prim, kind=pa.capi.ANY)["hit"]
wa = orc.trace(prim, any_hit=True)["hit"]
print("any mismatches", int((ga != wa).sum()), "gpu hits", int(ga.sum()), "orc hits", int(wa.sum()))
