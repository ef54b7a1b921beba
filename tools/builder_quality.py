"""Builder quality on the CPU: visits per ray of bench.py's configs[1] scene and R-diffuse batch (reduced image) through the oracle,
for the tree the host builder emits under its current settings (environment knobs read by bvh_builder.cpp, if any).
Usage: python tools/builder_quality.py [side]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np

import oracle_binding as ob
import pandora_amd as pa

side = int(sys.argv[1]) if len(sys.argv) > 1 else 384
MISS = 0xFFFFFFFF
mesh = pa.scenes.g_atrium(int(os.environ.get("TRIS", "262144")))
t0 = time.perf_counter()
hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
build_s = time.perf_counter() - t0
orc = ob.OracleBvh8.from_host(hb, mesh)
prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, side, side, seed=891379)
first = orc.trace(prim, threads=8)
obj = first["obj"].view(np.uint32)
prim_id = np.where(obj == MISS, MISS, hb.prim_ids[np.minimum(obj, hb.prim_ids.size - 1)])
rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"], prim_id, seed=891379)
n = rays.shape[1]
for name, r, any_hit in (("diffuse closest", rays, False), ("diffuse any", rays, True), ("primary closest", prim, False)):
    c = orc.trace(r, any_hit=any_hit, threads=8)["counters"]
    m = r.shape[1]
    print("%-16s inner %.3f  leaf pops %.3f  tri tests %.3f   (nodes %d, depth %d, build %.2f s)" % (
        name, c["inner_visits"] / m, c["leaf_pops"] / m, c["tri_tests"] / m, hb.n_nodes, hb.depth, build_s), flush=True)
