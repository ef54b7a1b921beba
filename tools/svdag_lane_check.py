"""the GPU casts SVDAGs with 3 lanes; the reference's vectors carry a fourth (x, y, z, 1) lane whose mask constant
(sparse_voxel_dag.cpp:475) only burns iterations. Counts, on >= 10^7 rays, the rays on which the boolean differs between
(a) the C restatement with 4 lanes, (b) the C restatement with 3 lanes, (c) the cast written on the reference's own simd
types (oracle/_ref, needs /root/reference at build time). CPU only. Writes one JSON object."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pandora_amd as pa
import oracle_binding as ob

n_total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 12_000_000
# a shell with displaced patches voxelised at 128^3: thin features, empty interior, many partially filled nodes
mesh = pa.scenes.merge([pa.scenes.icosphere((0, 0, 0), 1.0, 4), pa.scenes.box((0.4, -0.3, -0.25), (1.9, 0.3, 0.25), 6),
                        pa.scenes.cylinder((-1.2, -1.0, 0.3), 0.15, 2.2, 24, 8)])
dags = pa.accel.build_svdags(mesh.pos, mesh.idx, [np.arange(mesh.n_tris, dtype=np.uint32)], 128)
desc = dags.descs[0]
words = np.ascontiguousarray(dags.words, np.uint32)
dag = ob.Svdag(words.ctypes.data, int(desc["root"]), (C.c_float * 3)(*[float(x) for x in desc["bounds_min"]]), float(desc["inv_extent"]), 1)
lo, hi = mesh.bounds(); ext = float((hi - lo).max()); ctr = (lo + hi) / 2
orc = ob.oracle()
orc.orc_svdag_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
orc.orc_svdag_trace.restype = None
ref = ob.ref() if ob.have_ref() else None
if ref is not None:
    ref.ref_svdag_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    ref.ref_svdag_trace.restype = None
report = {"rays": 0, "grid": 128, "dag_words": int(words.size), "hits": 0, "mismatch_3_vs_4_lanes": 0, "mismatch_3_lanes_vs_reference_simd": 0 if ref else None, "sets": []}
per = n_total // 6
t0 = time.time()
for i, (scale, label) in enumerate(((1.2, "origins within 1.2 extents"), (6.0, "6 extents"), (300.0, "300 extents"))):
    for zero_axis in (False, True):
        rays = pa.scenes.random_rays(ctr - scale * ext, ctr + scale * ext, per, seed=100 + 2 * i + int(zero_axis), toward=(lo - 0.1 * ext, hi + 0.1 * ext))
        if zero_axis:
            rays[3 + (i % 3), ::11] = 0.0  # one in eleven rays has an exactly zero direction component
        rays = np.ascontiguousarray(rays, np.float32)
        out3, out4, outr = (np.zeros(per, np.uint8) for _ in range(3))
        orc.orc_svdag_trace(C.byref(dag), 3, rays.ctypes.data, per, out3.ctypes.data)
        orc.orc_svdag_trace(C.byref(dag), 4, rays.ctypes.data, per, out4.ctypes.data)
        m34 = int((out3 != out4).sum())
        mr = None
        if ref is not None:
            ref.ref_svdag_trace(C.byref(dag), rays.ctypes.data, per, outr.ctypes.data)
            mr = int((out3 != outr).sum())
            report["mismatch_3_lanes_vs_reference_simd"] += mr
        report["rays"] += per
        report["hits"] += int(out3.sum())
        report["mismatch_3_vs_4_lanes"] += m34
        report["sets"].append({"origins": label, "zero_direction_components": zero_axis, "rays": per, "hit_fraction": round(float(out3.mean()), 4),
                               "mismatch_3_vs_4_lanes": m34, "mismatch_3_lanes_vs_reference_simd": mr})
        print(report["sets"][-1], flush=True)
report["seconds"] = round(time.time() - t0, 1)
print(json.dumps(report))
