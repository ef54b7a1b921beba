"""Build time of the SVDAGs of bench.py's configs[2] scene (11.4 M triangles, 256 batching points): host builder vs device builder."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pandora_amd as pa

blocks = int(os.environ.get("CITY_BLOCKS", "24"))
mesh = pa.scenes.g_city(blocks, int(os.environ.get("TRIS_PER_BLOCK", "18750")))
group_of, ng = pa.accel.partition_prims(mesh.pos, mesh.idx, 65536)
order = np.argsort(group_of, kind="stable").astype(np.uint32)
counts = np.bincount(group_of, minlength=ng)
starts = np.concatenate([[0], np.cumsum(counts)])
groups = [np.ascontiguousarray(order[starts[g]:starts[g + 1]]) for g in range(ng)]
ctx = pa.accel.Context(0)
out = {"triangles": int(mesh.idx.shape[0]), "batching_points": int(ng)}
for res in (64, 128):
    pa.accel.build_svdags(mesh.pos, mesh.idx, groups[:2], res, device_ctx=ctx)  # warm-up (module load)
    t0 = time.perf_counter(); dev = pa.accel.build_svdags(mesh.pos, mesh.idx, groups, res, device_ctx=ctx); t1 = time.perf_counter()
    host = pa.accel.build_svdags(mesh.pos, mesh.idx, groups, res); t2 = time.perf_counter()
    out["res%d" % res] = {"device_s": t1 - t0, "host_s": t2 - t1, "words": int(dev.words.size), "words_host": int(host.words.size),
                          "words_before": dev.words_before, "words_before_host": host.words_before}
print(json.dumps(out, indent=1))
