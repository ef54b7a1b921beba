"""Bit-exact comparison of the GPU path with the CPU oracle on the benchmark's FULL ray sets (16.7 M primary + the R-diffuse
batch), both kernel layouts, closest and any hit, both triangle tests. Writes one JSON object (development aid; the test
suite checks the same property at sizes that finish in seconds)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pandora_amd as pa
import oracle_binding as ob

side = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
builder = sys.argv[2] if len(sys.argv) > 2 else "host"  # "resident": the tree built on the GPU and left there (pgpu_build_upload_bvh8_device)
mesh = pa.scenes.g_atrium(262144)
ctx = pa.accel.Context(0)
if builder == "resident":
    (bvh,), _ = ctx.build_upload_bvh8_device(mesh.pos, mesh.idx)
    hb = ctx.download_bvh8(bvh, np.arange(mesh.n_tris, dtype=np.uint32))
else:
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
orc = ob.OracleBvh8.from_host(hb, mesh)
prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, side, side)
w = ctx.trace_bvh8(bvh, prim)
obj = w["obj"].view(np.uint32)
primid = np.where(obj == 0xFFFFFFFF, np.uint32(0xFFFFFFFF), hb.prim_ids[np.minimum(obj, hb.prim_ids.size - 1)])
diff = pa.scenes.diffuse_bounce_rays(mesh, prim, w["t"], primid)
threads = max(1, min(16, os.cpu_count() or 1))
report = {"rays": {"primary": int(prim.shape[1]), "diffuse": int(diff.shape[1])}, "triangles": int(mesh.n_tris), "builder": builder, "bvh8_nodes": hb.n_nodes, "cases": []}
for name, rays in (("primary", prim), ("diffuse", diff)):
    d = torch.from_numpy(rays).cuda()
    for tri, tri_name in ((pa.capi.TRI_MT, "moller-trumbore"), (pa.capi.TRI_WATERTIGHT, "watertight")):
        t0 = time.time()
        want = orc.trace(rays, tri_mode=tri, threads=threads)
        want_any = orc.trace(rays, any_hit=True, tri_mode=tri, threads=threads)["hit"]
        cpu_s = time.time() - t0
        for lanes in (4, 2):
            ctx.set_ray_layout(lanes)
            got = ctx.trace_bvh8(bvh, d, tri_mode=tri)
            got_any = ctx.trace_bvh8(bvh, d, kind=pa.capi.ANY, tri_mode=tri)["hit"]
            torch.cuda.synchronize()
            bad = {k: int((got[k].cpu().numpy().view(np.uint32) != want[k].view(np.uint32)).sum()) for k in ("t", "obj", "u", "v")}
            bad["any"] = int((got_any.cpu().numpy().view(np.uint32) != want_any.view(np.uint32)).sum())
            case = {"rays": name, "triangle_test": tri_name, "lanes_per_ray": lanes, "mismatching_rays": bad, "oracle_seconds": round(cpu_s, 1)}
            report["cases"].append(case)
            print(case, flush=True)
ctx.set_ray_layout(4)
report["all_bit_exact"] = all(v == 0 for c in report["cases"] for v in c["mismatching_rays"].values())
print(json.dumps(report))
