#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s of incoherent diffuse-bounce rays (BASELINE.json `metric`).

Default (`--config single`, the configuration the metric is quoted on): a step = one closest-hit pass of the
single-level WiVe BVH8 kernel over one batch of R-diffuse rays (one cosine-weighted bounce per primary hit, order
shuffled) of the Sponza-class synthetic atrium (BASELINE.json configs[1]; the real asset is not available offline,
SURVEY.md 8d). `--config two-level`: the same ray set through the two-level path (PauseableBVH4 + 64^3 SVDAGs + per-point
BVH8) of a San-Miguel-class synthetic city (configs[2]). Rays and scene are resident in HBM when the timed region starts.

`--gpus N`: one process per GPU. Started by the driver under torch.distributed.run the ranks are already there; started
plainly (`python bench.py --gpus N`) this process starts the N ranks itself, before anything touches a GPU. Every rank
holds the whole in-core scene and traces its own batch (the path shards by rays: weak scaling, no collective in the data
path) -- and, for N > 1, the line also carries a `sharded` record: the two-level loop with the batching points of the city
sharded over the ranks' HBM and ray records routed by RCCL all-to-all (pgpu_trace_sharded, configs[3]).

Prints ONE JSON line (rank 0). `roofline` prices the traversal against HBM with the algorithmic bytes of SURVEY.md 8d;
`cpu_baseline` times the CPU port of the reference's traversal (oracle/) on a bounded sample of the same rays on the
host cores and checks the GPU result against it bit for bit.
"""
from __future__ import annotations

import argparse
import json
import os
import resource
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MISS = np.uint32(0xFFFFFFFF)


def builder_detail(builder: str) -> str:
    """What built the BVH8s, with the settings the library reads from the environment."""
    if builder == "host":
        if os.environ.get("PANDORA_BVH8_BUILDER") == "greedy":
            return "host: greedy top-down binned SAH"
        return "host: binary SAH tree, %s reinsertion passes, collapsed by dynamic programming" % os.environ.get("PANDORA_BVH8_REINSERT", "10")
    if os.environ.get("PANDORA_BVH8_DEVICE_TREE") == "lbvh":
        return "device: radix tree over Morton keys, collapsed by cost tables"
    return "device: locally-ordered clustering (radius %s) over Morton order, collapsed by cost tables" % os.environ.get("PANDORA_BVH8_DEVICE_PLOC_RADIUS", "3")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["single", "two-level", "landscape"], default="single",
                    help="configs[1] (default, the headline), configs[2], or one GPU's part of configs[4] (instanced landscape, batching points streamed from pinned host memory)")
    ap.add_argument("--landscape-instances", type=int, default=1_000_000)
    ap.add_argument("--landscape-prototypes", type=int, default=64)
    ap.add_argument("--landscape-terrain", type=int, default=1024, help="terrain grid side (2 n^2 triangles)")
    ap.add_argument("--leaf-hold", type=int, default=-1, help="landscape: pgpu_set_leaf_hold (image bytes per waiting ray before a point is loaded; 0 = never wait, -1 = the library's default)")
    ap.add_argument("--leaf-copy", type=int, default=0, help="landscape: 0 = one gather kernel per launch group reads the pinned images, 1 = one hipMemcpyAsync per image")
    ap.add_argument("--leaf-cache-frac", type=float, default=0.25, help="landscape: HBM leaf cache as a fraction of the staged batching points' bytes (0 = everything resident, nothing staged)")
    ap.add_argument("--side", type=int, default=4096, help="primary-ray image side; batch ~ side^2 diffuse rays")
    ap.add_argument("--tris", type=int, default=262144)
    ap.add_argument("--kind", choices=["closest", "any"], default="closest")
    ap.add_argument("--rays", choices=["diffuse", "shadow", "primary"], default="diffuse",
                    help="single-level config: the ray set of SURVEY.md 8d -- R-diffuse (shuffled cosine bounces, the headline), R-shadow (hit point -> ceiling light, tfar = distance; use with --kind any) or R-primary (camera rays)")
    ap.add_argument("--tri", choices=["moller-trumbore", "watertight"], default="moller-trumbore",
                    help="leaf test: the branch the reference compiles (triangle_intersect.cpp:425-462) or its watertight variant (:340-423)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="rays of the batch the CPU port traces (and the GPU result is compared on); 0 = automatic: the whole batch for the single-level config (~20 s of CPU work), 2^21 rays for two-level")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--city-blocks", type=int, default=24, help="two-level / sharded scene: city blocks per side (24 -> 11.4 M triangles)")
    ap.add_argument("--svdag-defer", type=int, default=0, help="two-level: SVDAG casts of passes with at least this many rays run as a kernel of their own (pgpu_set_svdag_defer; 0 = inside the episode, the default)")
    ap.add_argument("--svdag-policy", choices=["always", "nonresident"], default="always",
                    help="two-level / sharded: which batching points get the SVDAG early-out -- every one that has an SVDAG (the reference's behaviour) or only those whose geometry is not in this GPU's HBM (pgpu_set_svdag_policy 1; same hits)")
    ap.add_argument("--svdag", type=int, default=64, help="two-level / sharded scene: SVDAG resolution (0 = no proxy geometry, svdagRes 0 in the reference)")
    ap.add_argument("--sharded-side", type=int, default=4096, help="N > 1: primary image side per rank of the sharded record (batch ~ side^2 diffuse rays per rank)")
    ap.add_argument("--sharded-blocks", type=int, default=64, help="N > 1: city blocks per side of the sharded record's scene (64 -> 80.8 M triangles, 2048 batching points, ~10 GB of host memory per rank while building; 89 -> 156 M triangles, Moana's unique-triangle count, ~20 GB per rank)")
    ap.add_argument("--sharded-owners", choices=["lpt", "contiguous"], default="lpt",
                    help="N > 1: owner rank per batching point -- spread by size over all ranks (every rank gets a sample of the whole scene: balanced ray load, "
                         "(N-1)/N of the visits travel) or contiguous runs of the top level's leaf order (neighbouring space on one rank: little travels when the "
                         "rays are at home where they start, tests/test_gpu_sharded.py, but a camera that looks at one region loads one rank)")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: skip the sharded record")
    ap.add_argument("--sharded-timeout", type=int, default=420, help="seconds after which the sharded record is given up and the line printed without it")
    ap.add_argument("--force-sharded", action="store_true", help="N = 1: run the sharded record too (one rank owns every batching point; rehearsal of the N > 1 path)")
    ap.add_argument("--no-two-level", action="store_true", help="N = 1, --config single: skip the `two_level` sub-record (configs[2], timed after the headline)")
    ap.add_argument("--two-level-steps", type=int, default=10, help="timed steps of the `two_level` sub-record")
    ap.add_argument("--top-variant", type=int, default=0, help="two-level: pgpu_set_top_variant (0 automatic, 1 lockstep waves, 2 independent lanes, 3 lockstep first episode then lanes)")
    ap.add_argument("--batch-parts", type=int, default=0, help="two-level: pgpu_set_batch_parts (0 = the library's default: two parts from 2^21 rays on; 1 = the batch as one)")
    ap.add_argument("--builder", choices=["host", "device", "resident"], default="host",
                    help="bottom-level BVH8 construction: host = binary SAH tree collapsed by dynamic programming (the default, the best trees), device = pgpu_build_bvh8_device (locally-ordered clustering along the Morton order on the GPU, every batching point in one pass, trees returned to the host and uploaded like the host builder's), resident = pgpu_build_upload_bvh8_device (the same trees, never leaving the GPU)")
    ap.add_argument("--cut", choices=["host", "device"], default="host", help="two-level: the cut into batching points: host = binned-SAH BVH2 cut (pgpu_partition_prims, the default), device = the same rule on an LBVH built on the GPU (pgpu_partition_prims_device)")
    ap.add_argument("--dry", action="store_true", help="no GPU: the ranks only meet (gloo) and reduce made-up figures; checks the launch path")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------- starting the ranks
def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N children (one per GPU) with the torch.distributed
    environment and wait for them. This process never touches a GPU; rank 0's stdout (the JSON line) is inherited."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PANDORA_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # wait for all of them; a rank that fails leaves the others waiting at a barrier or inside a collective: give them a moment
    # (they may be printing the line), then end the ranks this process started rather than hang until a communicator times out
    failed_at = None
    while any(p.poll() is None for p in procs):
        if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
            failed_at = time.monotonic()
        if failed_at is not None and time.monotonic() - failed_at > 30.0:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    return max(abs(c) for c in codes)


# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_trace_single(orc, rays, any_hit, tri_mode, threads):
    """The CPU port on `threads` host threads: the AVX2 form (the loops restated on the reference's own simd headers,
    oracle/_ref, built in the container) when it is there, else the scalar C form. Same results either way."""
    import oracle_binding as ob
    if not ob.have_ref():
        return orc.trace(rays, any_hit=any_hit, tri_mode=tri_mode, threads=threads), "scalar C (oracle/pandora_oracle.c)"
    from concurrent.futures import ThreadPoolExecutor
    n = rays.shape[1]
    cuts = np.linspace(0, n, 4 * threads + 1).astype(np.int64)
    parts = [np.ascontiguousarray(rays[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    with ThreadPoolExecutor(threads) as pool:  # (ctypes releases the GIL)
        res = list(pool.map(lambda r: orc.trace(r, any_hit=any_hit, tri_mode=tri_mode, use_ref=True), parts))
    out = {k: np.concatenate([r[k] for r in res]) for k in res[0] if k != "counters"}
    return out, "AVX2 (loops restated on the reference's simd8/simd4 headers, oracle/_ref)"


def avx2_ratio():
    try:
        with open(os.path.join(ROOT, "profiles", "r2_cpu_baseline_ratio.json")) as fh:
            return json.load(fh)["avx2_over_scalar"]
    except Exception:
        return None


# ---------------------------------------------------------------------------------------------- the configurations
def run_single(args, pa, torch, ctx, dev, rank):
    """configs[1]: in-core single-level BVH8. Returns (step fn, n rays, finish fn -> (config, roofline, cpu_baseline))."""
    mesh = pa.scenes.g_atrium(args.tris)
    t_build = time.perf_counter()
    build_device_ms = None
    if args.builder == "device":
        (hb,), _ = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)
        build_device_ms = ctx.last_kernel_ms()
    elif args.builder == "resident":
        (bvh,), _ = ctx.build_upload_bvh8_device(mesh.pos, mesh.idx)
        build_device_ms = ctx.last_kernel_ms()
    else:
        hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    build_s = time.perf_counter() - t_build
    if args.builder == "resident":
        hb = ctx.download_bvh8(bvh, np.arange(mesh.n_tris, dtype=np.uint32))  # (for the CPU baseline and the hit -> triangle table; not part of the build)
    else:
        bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
    ready_s = time.perf_counter() - t_build
    stream = torch.cuda.current_stream().cuda_stream
    prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, args.side, args.side, seed=891379 + 7919 * rank)
    d_prim = torch.from_numpy(prim).to(dev)
    first = ctx.trace_bvh8(bvh, d_prim, stream=stream)
    torch.cuda.synchronize()
    t_hit = first["t"].cpu().numpy()
    obj = first["obj"].cpu().numpy().view(np.uint32)
    prim_id = np.where(obj == MISS, MISS, hb.prim_ids[np.minimum(obj, hb.prim_ids.size - 1)])
    if args.rays == "shadow":
        rays = pa.scenes.shadow_rays(mesh, prim, t_hit, prim_id, (10.0, 11.9, 6.0), (20.0, 11.9, 12.0), seed=891379 + 7919 * rank)
    elif args.rays == "primary":
        rays = prim
    else:
        rays = pa.scenes.diffuse_bounce_rays(mesh, prim, t_hit, prim_id, seed=891379 + 7919 * rank)
    del d_prim, first
    n = int(rays.shape[1])
    d_rays = torch.from_numpy(rays).to(dev)
    kind = pa.capi.CLOSEST if args.kind == "closest" else pa.capi.ANY
    tri_mode = pa.capi.TRI_MT if args.tri == "moller-trumbore" else pa.capi.TRI_WATERTIGHT
    out = ctx.trace_bvh8(bvh, d_rays, kind=kind, tri_mode=tri_mode, stream=stream)  # allocates the result arrays once

    def step():
        ctx.trace_bvh8(bvh, d_rays, kind=kind, tri_mode=tri_mode, out=out, stream=stream)

    def finish(k_ms, world):
        scene_bytes = hb.scene_bytes(mesh.n_verts)
        per_ray = 52 if args.kind == "closest" else 36
        alg_bytes = per_ray * n + scene_bytes
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        hit_frac = float((out["obj"].cpu().numpy().view(np.uint32) != MISS).mean()) if args.kind == "closest" else float(out["hit"].cpu().numpy().mean())
        cpu = None
        if not args.no_cpu:
            import oracle_binding as ob
            threads = pa.accel.host_threads()
            sample = min(n, args.cpu_sample or n)
            orc = ob.OracleBvh8.from_host(hb, mesh)
            sub = np.ascontiguousarray(rays[:, :sample])
            c0 = time.perf_counter()
            want, form = cpu_trace_single(orc, sub, args.kind == "any", tri_mode, threads)
            c1 = time.perf_counter()
            if args.kind == "closest":
                same = all(np.array_equal(out[k][:sample].cpu().numpy().view(np.uint32), want[k].view(np.uint32)) for k in ("t", "obj", "u", "v"))
            else:
                same = bool(np.array_equal(out["hit"][:sample].cpu().numpy().view(np.uint32), want["hit"]))
            # visit counters (secondary figure) from the scalar form on a small slice: bytes a ray pulls through the caches
            cnt = orc.trace(np.ascontiguousarray(sub[:, :min(sample, 1 << 18)]), any_hit=(args.kind == "any"), tri_mode=tri_mode, threads=threads)["counters"]
            m = min(sample, 1 << 18)
            vpr = {k: cnt[k] / m for k in ("inner_visits", "leaf_pops", "tri_tests")}
            cpu = {"value": sample / (c1 - c0) / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": f"first {sample} rays of the same R-{args.rays} batch; WiVeBVH8::intersect restated, {form}, {threads} threads",
                   "avx2_over_scalar_in_container": avx2_ratio(), "gpu_matches_bit_exact": bool(same), "visits_per_ray": vpr,
                   "traversal_model_bytes_per_ray": 256.0 * vpr["inner_visits"] + 4.0 * vpr["tri_tests"] + 48.0 * vpr["tri_tests"]}
        # HBM-side bytes per launch of the dominant kernel come from rocprofv3 PMC passes of this same command
        # (tools/pmc_passes.sh), which cannot run inside the timed process; the committed summary is quoted.
        traffic, traffic_src, issue = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if args.rays == "diffuse" and args.kind == "closest" and args.tri == "moller-trumbore" and os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("rays_per_launch") == n:
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/pmc_traffic.json (" + tj["how"] + ")"
                issue = tj.get("issue")  # what actually bounds the kernel: VALU issue slots used, lanes active, waves waiting (same PMC passes)
        config = {"workload": "configs[1]: Sponza-class synthetic atrium (%d triangles), in-core single-level WiVe BVH8, R-%s %s-hit" % (mesh.n_tris, args.rays, args.kind),
                  "rays_per_gpu": n, "triangles": mesh.n_tris, "bvh8_nodes": hb.n_nodes, "bvh8_depth": hb.depth,
                  "bvh8_builder": args.builder, "bvh8_builder_detail": builder_detail(args.builder), "bvh8_build_s": build_s, "bvh8_build_device_ms": build_device_ms, "bvh8_built_and_uploaded_s": ready_s,
                  "triangle_test": args.tri, "hit_fraction": hit_frac, "parallelism": "rays sharded over %d GPU(s), scene replicated" % world}
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": traffic_src, "issue": issue, "kernel": "bvh8TraceKernel", "kernel_ms": k_ms,
                    "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_ray": per_ray, "scene_bytes": scene_bytes}
        if cpu is not None:
            roofline["traversal_model_GBps"] = cpu["traversal_model_bytes_per_ray"] * n / (k_ms * 1e-3) / 1e9
        return config, roofline, cpu

    return step, n, finish


def city_scene(pa, blocks, build_group=None, svdag=64, ctx=None, builder="host", cut="host"):
    """ctx: build the SVDAGs on that context's GPU (pgpu_build_svdags_device: 33 ms instead of 0.9 s for this scene);
    builder: "device" / "resident" = the batching points' BVH8s too (pgpu_build_bvh8_device / pgpu_build_upload_bvh8_device)."""
    mesh = pa.scenes.g_city(blocks, 18750)
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 65536, svdag_res=svdag, build_group=build_group, svdag_device_ctx=ctx,
                                  bvh_device_ctx=ctx if builder == "device" else None, resident_ctx=ctx if builder == "resident" else None,
                                  cut_ctx=ctx if cut == "device" else None)
    return mesh, tl


def city_camera(pa, blocks, side, seed):
    size = 20.0 * blocks
    return pa.scenes.primary_rays((0.1 * size, 14.0, 0.1 * size), (0.6 * size, 0.0, 0.6 * size), (0, 1, 0), 60, side, side, seed=seed)


def run_two_level(args, pa, torch, ctx, dev, rank):
    """configs[2]: PauseableBVH4 + 64^3 SVDAGs + per-point BVH8 on one GPU (the whole round loop is the step)."""
    t_build = time.perf_counter()
    mesh, tl = city_scene(pa, args.city_blocks, svdag=args.svdag, ctx=ctx, builder=args.builder, cut=args.cut)
    build_s = time.perf_counter() - t_build
    tl.upload(ctx, 0 if args.builder == "resident" else ctx.upload_mesh(mesh.pos, mesh.idx))
    ready_s = time.perf_counter() - t_build
    if args.svdag_defer:
        ctx.set_svdag_defer(args.svdag_defer, 2)
    ctx.set_svdag_policy(1 if args.svdag_policy == "nonresident" else 0)
    ctx.set_top_variant(args.top_variant)
    ctx.set_batch_parts(args.batch_parts)
    stream = torch.cuda.current_stream().cuda_stream
    prim = city_camera(pa, args.city_blocks, args.side, 891379 + 7919 * rank)
    first, _ = ctx.trace(torch.from_numpy(prim).to(dev), stream=stream)
    prim_id = tl.prim_ids_of(first["leaf"].cpu().numpy(), first["obj"].cpu().numpy())
    rays = pa.scenes.diffuse_bounce_rays(mesh, prim, first["t"].cpu().numpy(), prim_id, seed=891379 + 7919 * rank)
    del first
    n = int(rays.shape[1])
    d_rays = torch.from_numpy(rays).to(dev)
    kind = pa.capi.CLOSEST if args.kind == "closest" else pa.capi.ANY
    tri_mode = pa.capi.TRI_MT if args.tri == "moller-trumbore" else pa.capi.TRI_WATERTIGHT
    out, cnt = ctx.trace(d_rays, kind=kind, tri_mode=tri_mode, stream=stream)
    state = {"cnt": cnt}

    def step():
        _, state["cnt"] = ctx.trace(d_rays, kind=kind, tri_mode=tri_mode, out=out, stream=stream)

    def finish(k_ms, world):
        cnt = state["cnt"]
        if tl.bvh_ids is not None:  # trees that never left the GPU: the figures below and the CPU baseline want them on the host
            tl.fetch_bvhs(ctx)
        scene_bytes = tl.device_bytes(mesh.n_verts)
        per_ray = 52 if args.kind == "closest" else 36
        # per (ray, batching-point visit) a park + a resume record of 44 bytes each (ray 32 + token 12)
        alg_bytes = per_ray * n + 88 * cnt["leaf_visits"] + scene_bytes
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        cpu = None
        if not args.no_cpu:
            import oracle_binding as ob
            threads = pa.accel.host_threads()
            sample = min(n, args.cpu_sample or (1 << 21))
            sub = np.ascontiguousarray(rays[:, :sample])
            import dataclasses
            # (policy "nonresident" with every point resident: the oracle's scene without SVDAGs has the same visits and counters)
            scene = ob.scene_from_two_level(tl if args.svdag_policy == "always" else dataclasses.replace(tl, dags=None), mesh)
            c0 = time.perf_counter()
            want = scene.trace(sub, any_hit=(args.kind == "any"), tri_mode=tri_mode, lanes=3, threads=threads)
            c1 = time.perf_counter()
            keys = ("hit", "visits") if args.kind == "any" else ("t", "obj", "leaf", "u", "v", "visits")
            same = all(np.array_equal(out[k][:sample].cpu().numpy().view(np.uint32), want[k].view(np.uint32)) for k in keys)
            cpu = {"value": sample / (c1 - c0) / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": f"first {sample} rays of the same R-diffuse batch; PauseableBVH4::intersectT + SparseVoxelDAG::intersectScalar + WiVeBVH8::intersect restated in scalar C (oracle/pandora_oracle.c), {threads} pthreads",
                   "gpu_matches_bit_exact": bool(same)}
        config = {"workload": "configs[2]: San-Miguel-class synthetic city (%d triangles), two-level: PauseableBVH4 over %d batching points with %d^3 SVDAG early-out%s + per-point WiVe BVH8, R-diffuse %s-hit" % (mesh.n_tris, tl.n_leaves, args.svdag, "" if args.svdag_policy == "always" else " for non-resident points only (here: none)", args.kind),
                  "rays_per_gpu": n, "triangles": mesh.n_tris, "batching_points": tl.n_leaves, "triangle_test": args.tri,
                  "bvh8_builder": args.builder, "bvh8_builder_detail": builder_detail(args.builder), "cut": args.cut, "scene_build_s": build_s, "scene_built_and_uploaded_s": ready_s,
                  "rounds": cnt["rounds"], "leaf_visits_per_ray": cnt["leaf_visits"] / n, "svdag_tests": cnt["svdag_tests"], "svdag_culled": cnt["svdag_culled"],
                  "parallelism": "rays sharded over %d GPU(s), scene replicated" % world}
        # memory-side bytes per step from rocprofv3 PMC passes of this same command (tools/pmc_two_level_traffic.py), quoted as for
        # the single-level config: counters cannot be read from inside the timed process
        traffic, traffic_src, issue = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_two_level.json")
        if args.kind == "closest" and args.tri == "moller-trumbore" and args.svdag == 64 and args.svdag_policy == "always" and os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("rays_per_step") == n:
                traffic, traffic_src = tj["traffic_bytes_per_step"], "profiles/pmc_traffic_two_level.json (" + tj["how"] + ")"
                issue = tj.get("issue")
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": traffic_src, "issue": issue,
                    "kernel": "the round loop (topTraverseKernel + leafScan / binByLeaf + bvh8TracePairKernel, all rounds), device time between the loop's first and last event",
                    "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_ray": per_ray, "bytes_per_leaf_visit": 88, "scene_bytes": scene_bytes}
        return config, roofline, cpu

    return step, n, finish


def run_landscape(args, pa, torch, ctx, dev, rank):
    """One GPU's part of configs[4]: terrain + instanced tree prototypes (OfflineBVHLeaf::Instance leaves), cut into batching
    points over the scene's objects; the prototypes are resident, the batching points' own BVH8s (terrain pieces + instance
    leaves) are staged in pinned host memory and streamed through an HBM cache of --leaf-cache-frac of their size."""
    n_inst = args.landscape_instances
    size = float((n_inst * 64.0) ** 0.5)  # one tree per 64 square units
    t0 = time.perf_counter()
    terrain, protos, proto, mats = pa.scenes.g_landscape(n_prototypes=args.landscape_prototypes, n_instances=n_inst, terrain_n=args.landscape_terrain,
                                                         size=size, detail=3)
    tl = pa.accel.build_instanced_two_level(terrain.pos, terrain.idx, [(p.pos, p.idx) for p in protos], proto, mats, 65536, svdag_res=0)
    build_s = time.perf_counter() - t0
    streamed = args.leaf_cache_frac > 0
    tl.upload(ctx, streamed=streamed)
    image = [b.n_nodes * 256 + (b.leaf_idx.size + 1) * 48 + 64 * gi.size + 1024 for b, gi in zip(tl.bvhs, tl.group_insts)]
    if streamed:
        ctx.set_leaf_cache(max(2 * max(image), int(args.leaf_cache_frac * sum(image))))
        ctx.set_leaf_copy(args.leaf_copy)
        if args.leaf_hold >= 0:
            ctx.set_leaf_hold(args.leaf_hold)
    stream = torch.cuda.current_stream().cuda_stream
    prim = pa.scenes.primary_rays((0.35 * size, 0.06 * size, 0.35 * size), (0.65 * size, 0.0, 0.65 * size), (0, 1, 0), 60, args.side, args.side, seed=891379 + 7919 * rank)
    first, _ = ctx.trace(torch.from_numpy(prim).to(dev), stream=stream)
    f_obj = first["obj"].cpu().numpy()
    normals = tl.hit_normals(first["leaf"].cpu().numpy(), f_obj, first["inst"].cpu().numpy())
    inst_hits = float((first["inst"].cpu().numpy().view(np.uint32) != MISS).mean())
    rays = pa.scenes.diffuse_bounce_rays(None, prim, first["t"].cpu().numpy(), f_obj.view(np.uint32), seed=891379 + 7919 * rank, normals=normals)
    del first
    n = int(rays.shape[1])
    d_rays = torch.from_numpy(rays).to(dev)
    kind = pa.capi.CLOSEST if args.kind == "closest" else pa.capi.ANY
    tri_mode = pa.capi.TRI_MT if args.tri == "moller-trumbore" else pa.capi.TRI_WATERTIGHT
    out, cnt = ctx.trace(d_rays, kind=kind, tri_mode=tri_mode, stream=stream)
    state = {"cnt": cnt}

    def step():
        _, state["cnt"] = ctx.trace(d_rays, kind=kind, tri_mode=tri_mode, out=out, stream=stream)

    def finish(k_ms, world):
        cnt = state["cnt"]
        proto_tris = [int(i.reshape(-1, 3).shape[0]) for _, i in tl.protos]
        unique_tris = terrain.n_tris + sum(proto_tris)
        instanced_tris = terrain.n_tris + int(np.asarray(proto_tris, np.int64)[np.asarray(proto, np.int64)].sum())
        scene_bytes = sum(image) + sum(256 * b.n_nodes + 4 * b.leaf_idx.size + 48 * t for b, t in zip(tl.proto_bvhs, proto_tris)) + 112 * int(tl.top.nodes.shape[0])
        per_ray = 52 if args.kind == "closest" else 36
        alg_bytes = per_ray * n + 88 * cnt["leaf_visits"] + scene_bytes
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        cpu = None
        if not args.no_cpu:
            import oracle_binding as ob
            threads = pa.accel.host_threads()
            sample = min(n, args.cpu_sample or (1 << 20))
            sub = np.ascontiguousarray(rays[:, :sample])
            scene = ob.scene_from_instanced(tl)
            c0 = time.perf_counter()
            want = scene.trace(sub, any_hit=(args.kind == "any"), tri_mode=tri_mode, threads=threads)
            c1 = time.perf_counter()
            keys = ("hit", "visits") if args.kind == "any" else ("t", "obj", "leaf", "u", "v", "visits", "inst")
            same = all(np.array_equal(out[k][:sample].cpu().numpy().view(np.uint32), want[k].view(np.uint32)) for k in keys)
            cpu = {"value": sample / (c1 - c0) / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": f"first {sample} rays of the same R-diffuse batch; PauseableBVH4::intersectT + WiVeBVH8::intersect with instance leaves (ray to instance space, nested traversal) restated in scalar C (oracle/pandora_oracle.c), {threads} pthreads",
                   "gpu_matches_bit_exact": bool(same)}
        config = {"workload": "configs[4], one GPU's part: synthetic instanced landscape (%d instances of %d tree prototypes on a %d-triangle terrain = %.3g instanced triangles from %d unique), %d batching points cut over the scene's objects, %s, R-diffuse %s-hit"
                              % (n_inst, len(protos), terrain.n_tris, instanced_tris, unique_tris, tl.n_leaves,
                                 ("batching points streamed from pinned host memory through an HBM cache of %.0f %% of their %d bytes" % (100 * args.leaf_cache_frac, sum(image))) if streamed else "everything resident", args.kind),
                  "rays_per_gpu": n, "instances": n_inst, "instanced_triangles": instanced_tris, "unique_triangles": unique_tris, "batching_points": tl.n_leaves,
                  "flattened_bytes_if_not_instanced": float(instanced_tris) * (48 + 256 / 3.0), "primary_rays_hitting_an_instance": inst_hits,
                  "triangle_test": args.tri, "rounds": cnt["rounds"], "leaf_visits_per_ray": cnt["leaf_visits"] / n,
                  "leaf_loads_per_step": cnt.get("leaf_loads"), "leaf_bytes_streamed_per_step": cnt.get("streamed_bytes"), "scene_build_s": build_s,
                  "parallelism": "rays sharded over %d GPU(s), scene replicated" % world}
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "the round loop (topTraverseKernel + leafScan / binByLeaf + bvh8TraceInstKernel per group of cached batching points, all rounds), device time between the loop's first and last event",
                    "kernel_ms": k_ms, "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_ray": per_ray, "bytes_per_leaf_visit": 88, "scene_bytes": scene_bytes}
        return config, roofline, cpu

    return step, n, finish


def sharded_scene(args, pa, dist, rank, world, blocks, svdag, svdag_ctx=None):
    """The sharded record's scene, built rank-locally: the city as tiles (scenes.CityTiles: a tile = a batching point), owners
    decided from the tiles' triangle counts -- known without making them --, every rank makes geometry, BVH8s and SVDAGs of its
    own tiles only, the ranks exchange the tiles' bounds and DAGs and each assembles the top level (accel.tiles_assemble).
    -> (tiles, two-level scene, owner per tile, this rank's part, seconds spent)"""
    from pandora_amd import sharded
    t0 = time.perf_counter()
    tiles = pa.scenes.CityTiles(blocks, 18750)
    w = tiles.weights()
    if args.sharded_owners == "contiguous":  # runs of a Z-order curve over the tile grid, balanced by triangle count
        cells = np.array([tiles.tile_cell(t) for t in range(tiles.n_tiles)], np.int64)
        key = np.zeros(tiles.n_tiles, np.int64)
        for b in range(16):
            key |= ((cells[:, 0] >> b) & 1) << (2 * b + 1) | ((cells[:, 1] >> b) & 1) << (2 * b)
        order = np.argsort(key, kind="stable")
        owner_by_tile = np.empty(tiles.n_tiles, np.int64)
        owner_by_tile[order] = sharded.assign_leaves_contiguous(w[order], world)
    else:
        owner_by_tile = np.asarray(sharded.assign_leaves(w, world), np.int64)  # balanced by triangle count
    part = pa.accel.tiles_local_part(tiles, owner_by_tile == rank, svdag_res=svdag, svdag_device_ctx=svdag_ctx)
    mine = {k: part[k] for k in pa.accel.TILE_SHARED_KEYS}
    if world > 1:
        shared = [None] * world
        dist.all_gather_object(shared, mine)
    else:
        shared = [mine]
    tl = pa.accel.tiles_assemble(tiles.n_tiles, shared, part)
    return tiles, tl, owner_by_tile, part, time.perf_counter() - t0


def run_sharded(args, pa, torch, dist, dev, rank, world):
    """configs[3]: batching points sharded over the ranks' HBM, records routed by RCCL (pgpu_trace_sharded). Every rank
    makes only the geometry and the BVHs of its own batching points. Returns the `sharded` record (rank 0) or None."""
    from pandora_amd import sharded
    t0 = time.perf_counter()
    ctx = pa.accel.Context(dev.index)
    tiles, tl, owner_by_tile, part, scene_s = sharded_scene(args, pa, dist, rank, world, args.sharded_blocks, args.svdag, svdag_ctx=ctx)
    print("rank %d: scene_build_s %.1f (%d of %d tiles, %d triangles)" % (rank, scene_s, len(part["tiles"]), tiles.n_tiles, part["idx"].shape[0]), file=sys.stderr, flush=True)
    leaf_tile = np.asarray(tl.top.leaf_order, np.int64)  # tile (= group) per leaf slot
    owner = owner_by_tile[leaf_tile]
    part_pos, part_idx = part["pos"], part["idx"]
    n_tris_total = int(tiles.weights().sum())
    if part_idx.shape[0] == 0:  # (a rank without batching points still needs a mesh to name)
        part_pos, part_idx = np.zeros((3, 3), np.float32), np.array([[0, 1, 2]], np.uint32)
    tl.upload(ctx, ctx.upload_mesh(part_pos, part_idx), resident=(owner == rank))
    mesh_bytes = float(part_pos.nbytes + part_idx.nbytes)
    part["pos"] = part["idx"] = None
    del part_pos, part_idx
    sharded.comm_init(ctx, dev)  # RCCL communicator of the library's own; the unique id travels through torch.distributed
    ctx.comm_selftest()
    ctx.set_leaf_owners(owner)
    ctx.set_svdag_policy(1 if args.svdag_policy == "nonresident" else 0)
    build_s = time.perf_counter() - t0
    prim = city_camera(pa, args.sharded_blocks, args.sharded_side, 891379 + 7919 * rank)
    first, _ = ctx.trace_sharded(torch.from_numpy(prim).to(dev))
    # the bounce needs the normal of whatever was hit, on any rank's tiles: the tiles this rank's camera sees are made once more here
    f_obj, f_leaf = first["obj"].cpu().numpy().view(np.uint32), first["leaf"].cpu().numpy().view(np.uint32)
    hit = np.nonzero(f_obj != MISS)[0]
    normals = np.zeros((f_obj.size, 3), np.float64)
    normals[hit] = tiles.normals_of(leaf_tile[f_leaf[hit].astype(np.int64)], f_obj[hit])
    rays = pa.scenes.diffuse_bounce_rays(None, prim, first["t"].cpu().numpy(), f_obj, seed=891379 + 7919 * rank, normals=normals)
    del normals, first
    d_rays = torch.from_numpy(rays).to(dev)
    n = int(rays.shape[1])
    out, cnt = ctx.trace_sharded(d_rays)  # warm-up, allocates the result arrays and the record buffers
    steps = max(1, min(args.steps, 10))
    multi = world > 1
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        _, cnt = ctx.trace_sharded(d_rays, out=out)
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mx = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    sums = torch.tensor([n, cnt["leaf_visits"], cnt["records_routed"], cnt["exchange_bytes"], 1,
                         sum(256 * b.n_nodes + 48 * b.leaf_idx.size for b in tl.bvhs if b is not None), mesh_bytes], dtype=torch.float64, device=dev)
    if multi:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    rec = None
    if rank == 0:
        total_rays, visits, routed, xbytes, ranks, resident, mesh_total = (float(x) for x in sums.tolist())
        el = float(mx.item())
        rec = {"workload": "configs[3]-class: synthetic city (%d triangles, %d batching points, %d^3 SVDAGs), batching points sharded over %d GPUs' HBM, R-diffuse closest-hit, records routed by all-to-all" % (n_tris_total, tl.n_leaves, args.svdag, world),
               "engine": "pgpu_trace_sharded (C ABI, RCCL grouped ncclSend/ncclRecv)", "owners": args.sharded_owners, "svdag_policy": args.svdag_policy, "torch_backend": dist.get_backend() if multi else None, "ranks_seen": int(ranks),
               "comm_selftest": "ok", "value": total_rays * steps / el / 1e6, "unit": "Mrays/s", "steps": steps, "ms_per_step": 1e3 * el / steps,
               "rays_per_gpu": n, "rounds": cnt["rounds"], "leaf_visits_per_ray": visits / total_rays, "records_routed_per_step": routed,
               "records_routed_fraction": routed / max(1.0, visits), "exchange_bytes_per_step_all_ranks": xbytes,
               "bytes_per_all_to_all_per_rank": xbytes / max(1.0, ranks * 2 * max(1, cnt["rounds"])),
               "resident_bvh_bytes_total": resident, "resident_mesh_bytes_total": mesh_total, "scene_build_s_rank0": build_s,
               "scene": "built rank-locally: every rank makes geometry, BVH8s and SVDAGs of its own tiles only (scenes.CityTiles, accel.tiles_local_part / tiles_assemble)",
               "host_peak_rss_gb_rank0": resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2**20}
    ctx.close()
    return rec


# ---------------------------------------------------------------------------------------------- main
def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)

    # Libraries below us write to stdout (RCCL prints a version banner there): the one JSON line must be the only thing on
    # it, so file descriptor 1 points at stderr until the line is written.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.dry:
        # launch-path check without a GPU: the ranks meet on gloo, reduce made-up figures the way the real run does
        from pandora_amd.sharded import reduce_job
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        seen = torch.zeros(max(world, 1), dtype=torch.int64)
        seen[rank] = 1
        if world > 1:
            dist.all_reduce(seen)
        elapsed, total = reduce_job(0.001 * (rank + 1), 1000, None)
        # the sharded record's scene on a small city: own tiles only, bounds and DAGs exchanged, top level assembled on every rank
        import pandora_amd as pa
        tiles, tl, owner_by_tile, part, scene_s = sharded_scene(args, pa, dist, rank, world, 4, 16)
        print("rank %d: scene_build_s %.2f (%d of %d tiles)" % (rank, scene_s, len(part["tiles"]), tiles.n_tiles), file=sys.stderr, flush=True)
        tops = torch.tensor([float(np.asarray(tl.top.nodes, np.float64).sum()), float(len(part["tiles"])), float(tl.dags.words.size)], dtype=torch.float64)
        all_tops = [torch.zeros_like(tops) for _ in range(world)]
        if world > 1:
            dist.all_gather(all_tops, tops)
        else:
            all_tops = [tops]
        if rank == 0:
            emit({"metric": "Mrays/s incoherent diffuse rays", "value": total / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "dry": True,
                  "ranks_seen": [int(i) for i in torch.nonzero(seen).flatten().tolist()], "steps": args.steps, "warmup": args.warmup,
                  "scene_tiles": tiles.n_tiles, "scene_tiles_built_per_rank": [int(t[1]) for t in all_tops],
                  "scene_same_top_level_on_every_rank": bool(all(float(t[0]) == float(all_tops[0][0]) and float(t[2]) == float(all_tops[0][2]) for t in all_tops))})
        if world > 1:
            dist.destroy_process_group()
        # (test hook: a rank that fails after the line is out -- the launcher must hand its status on, tests/test_bench_launch.py)
        return 3 if os.environ.get("PANDORA_BENCH_DRY_FAIL_RANK") == str(rank) else 0

    import pandora_amd as pa
    from pandora_amd.sharded import reduce_job

    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    def timed(ctx, step, steps, warmup):
        """W untimed steps, then exactly `steps` between barrier + synchronize on both sides; max over ranks."""
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        kernel_ms = []
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            # per-launch kernel time from HIP events recorded by the library on this same stream; reading
            # them needs the step to have finished, which also keeps one batch in flight at a time
            ctx.sync(torch.cuda.current_stream().cuda_stream)
            kernel_ms.append(ctx.last_kernel_ms())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, float(np.mean(kernel_ms))

    if world > 1:
        args.no_cpu = True  # (contract: the CPU baseline is timed on rank 0 at N = 1 only; the other ranks would wait for it)
    ctx = pa.accel.Context(dev.index)
    step, n, finish = {"single": run_single, "two-level": run_two_level, "landscape": run_landscape}[args.config](args, pa, torch, ctx, dev, rank)
    elapsed, k_ms = timed(ctx, step, args.steps, args.warmup)
    elapsed, total_rays = reduce_job(elapsed, n, dev)  # max over ranks, rays summed over ranks

    line = None
    failed = False
    if rank == 0:
        config, roofline, cpu = finish(k_ms, world)
        line = {"metric": "Mrays/s incoherent diffuse rays", "value": total_rays * args.steps / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config, "roofline": roofline, "cpu_baseline": cpu}
        if cpu is not None and not cpu.get("gpu_matches_bit_exact", True):
            failed = True
    ctx.close()
    del step, finish

    if world == 1 and args.config == "single" and not args.no_two_level:
        # configs[2] in the same line: the reference's actual hot path (offline_batching...h:406-437, pauseable_bvh4.h:225-355) --
        # PauseableBVH4 + SVDAG early-out + per-point BVH8, the whole round loop per step -- timed after the headline on a
        # context of its own, with its own roofline and CPU baseline. The headline above is not touched by it.
        t_sub = time.perf_counter()
        try:
            import copy
            a2 = copy.copy(args)
            a2.kind, a2.tri, a2.rays, a2.cpu_sample = "closest", "moller-trumbore", "diffuse", 0
            ctx2 = pa.accel.Context(dev.index)
            step2, n2, finish2 = run_two_level(a2, pa, torch, ctx2, dev, rank)
            steps2 = max(1, args.two_level_steps)
            el2, k2 = timed(ctx2, step2, steps2, 2)
            config2, roofline2, cpu2 = finish2(k2, world)
            ctx2.close()
            line["two_level"] = {"value": n2 * steps2 / el2 / 1e6, "unit": "Mrays/s", "steps": steps2, "warmup": 2, "ms_per_step": 1e3 * el2 / steps2,
                                 "config": config2, "roofline": roofline2, "cpu_baseline": cpu2,
                                 "gpu_matches_bit_exact": None if cpu2 is None else cpu2["gpu_matches_bit_exact"],
                                 "wall_s_including_scene_build": None}
            line["two_level"]["wall_s_including_scene_build"] = time.perf_counter() - t_sub
            if cpu2 is not None and not cpu2["gpu_matches_bit_exact"]:
                failed = True
        except Exception as e:  # the headline still goes out, the exit status says that the sub-record failed
            line["two_level"] = {"error": f"{type(e).__name__}: {e}"}
            failed = True

    if (world > 1 or args.force_sharded) and not args.no_sharded:
        # The headline figure stands on its own: whatever happens to the sharded record -- an exception, or a rank that died and
        # left the others waiting inside a collective -- rank 0 still prints its line, with the failure in it AND in the exit
        # status: a run whose multi-GPU path hung or raised must not read as green.
        def give_up():
            if rank == 0:
                line["sharded"] = {"error": "no result after %d s (a rank failed inside a collective?)" % args.sharded_timeout}
                line["sharded_error"] = line["sharded"]["error"]
                emit(line)
            os._exit(3)

        watchdog = threading.Timer(args.sharded_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()
        try:
            rec = run_sharded(args, pa, torch, dist, dev, rank, world)
        except Exception as e:
            rec = {"error": f"{type(e).__name__}: {e}"}
        watchdog.cancel()
        if rec is not None and "error" in rec:
            failed = True
        if rank == 0:
            line["sharded"] = rec
            # the figures a reader of a SCALE record wants, at the top level: the headline `value` is replicas of configs[1] and
            # scales whatever the sharded loop does
            if rec is not None and "error" not in rec:
                line["sharded_mrays"] = rec["value"]
                line["sharded_ms_per_step"] = rec["ms_per_step"]
                line["sharded_ranks_seen"] = rec["ranks_seen"]
                line["sharded_records_routed_fraction"] = rec["records_routed_fraction"]
                if rec["ranks_seen"] != world:
                    line["sharded_error"] = "%d of %d ranks took part" % (rec["ranks_seen"], world)
                    failed = True
            else:
                line["sharded_error"] = (rec or {}).get("error", "no record")
                failed = True
    if rank == 0:
        emit(line)
    if world > 1:
        if failed:
            os._exit(3)  # (ranks that did not fail may be waiting in a collective for one that did: leave without the group's goodbye)
        dist.destroy_process_group()
    return 3 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
