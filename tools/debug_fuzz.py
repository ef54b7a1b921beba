"""Development aid: classify GPU-vs-oracle mismatches of the hostile-input fuzz."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_binding as ob
import pandora_amd as pa
import test_gpu_fuzz as tf
max_leaf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(100 + max_leaf)
pos, idx = tf._nasty_mesh(rng, 1600)
hb = pa.accel.build_bvh8(pos, idx, max_leaf=max_leaf)
ctx = pa.accel.Context(0)
bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(pos, idx))
class M: pass
m = M(); m.pos, m.idx = pos, idx
orc = ob.OracleBvh8.from_host(hb, m)
rays = tf._nasty_rays(rng, 40000)
with np.errstate(all="ignore"):
    got = ctx.trace_bvh8(bvh, rays); want = orc.trace(rays, threads=8)
bits = lambda a: np.asarray(a).view(np.uint32)
bad = np.zeros(rays.shape[1], bool)
for k in ("t", "obj", "u", "v"):
    b = bits(got[k]) != bits(want[k]); print(k, int(b.sum())); bad |= b
r = rays[:, bad]
nan_o = np.isnan(r[0:3]).any(0); nan_d = np.isnan(r[3:6]).any(0); inf_d = np.isinf(r[3:6]).any(0); zero_d = (r[3:6] == 0).any(0)
nan_t = np.isnan(r[6]) | np.isnan(r[7]); big_o = (np.abs(r[0:3]) > 1e20).any(0)
print("bad", int(bad.sum()), "nan_o", int(nan_o.sum()), "nan_d", int(nan_d.sum()), "inf_d", int(inf_d.sum()), "zero_d", int(zero_d.sum()), "nan_t", int(nan_t.sum()), "big_o", int(big_o.sum()))
plain = ~(nan_o | nan_d | inf_d | zero_d | nan_t | big_o)
print("plain", int(plain.sum()))
ids = np.nonzero(bad)[0]
for i in list(ids[plain][:5]) + list(ids[~plain][:8]):
    print(i, rays[:, i], "got", got["t"][i], bits(got["obj"])[i], got["u"][i], got["v"][i], "want", want["t"][i], want["obj"][i], want["u"][i], want["v"][i])
