"""Bit-exact comparison of the two-level path (PauseableBVH4 + SVDAG early-out + per-point BVH8) with the CPU oracle on a
San-Miguel-class synthetic scene at benchmark size, resident and streamed through a small leaf cache. Writes one JSON
object (development aid; the test suite checks the same property at sizes that finish in seconds)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import pandora_amd as pa
import oracle_binding as ob

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 24
side = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mesh = pa.scenes.g_city(nb, 18750)
tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 65536, svdag_res=64)
scene = ob.scene_from_two_level(tl, mesh)
size = 20.0 * nb
prim = pa.scenes.primary_rays((0.1 * size, 14.0, 0.1 * size), (0.6 * size, 0.0, 0.6 * size), (0, 1, 0), 60, side, side)
threads = max(1, min(16, os.cpu_count() or 1))
report = {"triangles": int(mesh.n_tris), "batching_points": len(tl.bvhs), "svdag_res": 64, "cases": []}

def compare(ctx, label, rays, want, want_any):
    d = torch.from_numpy(rays).cuda()
    got, cnt = ctx.trace(d)
    got_any, cnt_any = ctx.trace(d, kind=pa.capi.ANY)
    torch.cuda.synchronize()
    bad = {k: int((got[k].cpu().numpy().view(np.uint32) != want[k].view(np.uint32)).sum()) for k in ("t", "obj", "leaf", "u", "v", "visits")}
    bad["any"] = int((got_any["hit"].cpu().numpy().view(np.uint32) != want_any["hit"].view(np.uint32)).sum())
    bad["any_visits"] = int((got_any["visits"].cpu().numpy().view(np.uint32) != want_any["visits"].view(np.uint32)).sum())
    counters_equal = all(cnt[k] == want["counters"][k] and cnt_any[k] == want_any["counters"][k] for k in ("svdag_tests", "svdag_culled", "leaf_visits"))
    case = {"case": label, "rays": int(rays.shape[1]), "mismatching_rays": bad, "counters_equal": bool(counters_equal),
            "leaf_loads": int(cnt["leaf_loads"]), "streamed_MB": round(cnt["streamed_bytes"] / 1e6, 1)}
    report["cases"].append(case)
    print(case, flush=True)
    return got

ctx = pa.accel.Context(0)
tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
t0 = time.time()
want = scene.trace(prim, lanes=3, threads=threads)
want_any = scene.trace(prim, any_hit=True, lanes=3, threads=threads)
print("oracle primary %.1fs" % (time.time() - t0), flush=True)
got = compare(ctx, "resident / primary", prim, want, want_any)
obj = want["obj"]; leaf = want["leaf"]; hit = obj != 0xFFFFFFFF
primid = np.full(obj.shape, 0xFFFFFFFF, np.uint32)
for l in np.unique(leaf[hit]):
    m = hit & (leaf == l)
    primid[m] = tl.bvhs[int(tl.top.leaf_order[int(l)])].prim_ids[obj[m]]
diff = pa.scenes.diffuse_bounce_rays(mesh, prim, want["t"], primid)
t0 = time.time()
want_d = scene.trace(diff, lanes=3, threads=threads)
want_d_any = scene.trace(diff, any_hit=True, lanes=3, threads=threads)
print("oracle diffuse %.1fs" % (time.time() - t0), flush=True)
compare(ctx, "resident / diffuse", diff, want_d, want_d_any)
ctx.close()
# the same scene with every leaf in pinned host memory and a cache that holds a fraction of them
ctx = pa.accel.Context(0)
tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), streamed=True)
ctx.set_leaf_cache(256 << 20)
sub = np.ascontiguousarray(diff[:, : diff.shape[1] // 4])
want_s = scene.trace(sub, lanes=3, threads=threads)
want_s_any = scene.trace(sub, any_hit=True, lanes=3, threads=threads)
compare(ctx, "streamed, 256 MB cache / diffuse (first quarter)", sub, want_s, want_s_any)
ctx.close()
report["all_bit_exact"] = all(v == 0 for c in report["cases"] for v in c["mismatching_rays"].values()) and all(c["counters_equal"] for c in report["cases"])
print(json.dumps(report))
