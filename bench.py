#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s of incoherent diffuse-bounce rays (BASELINE.json `metric`).

A step = one closest-hit pass of the single-level WiVe BVH8 kernel over one batch of R-diffuse rays
(one cosine-weighted bounce per primary hit, order shuffled) of the Sponza-class synthetic atrium
(BASELINE.json configs[1]; the real asset is not available offline). Rays and the scene
are resident in HBM when the timed region starts. With --gpus N every rank holds the whole (in-core)
scene and traces its own batch: the path shards by rays, no collective in the data path (weak scaling).

Prints ONE JSON line (rank 0). `roofline` prices the traversal kernel against HBM with the algorithmic
bytes of SURVEY.md 8d (52 B/ray + the scene read once per launch); `cpu_baseline` times the CPU oracle
(a port of the reference's traversal, oracle/) on a bounded sample of the same rays on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--side", type=int, default=4096, help="primary-ray image side; batch ~ side^2 diffuse rays")
    ap.add_argument("--tris", type=int, default=262144)
    ap.add_argument("--kind", choices=["closest", "any"], default="closest")
    ap.add_argument("--tri", choices=["moller-trumbore", "watertight"], default="moller-trumbore",
                    help="leaf test: the branch the reference compiles (triangle_intersect.cpp:425-462) or its watertight variant (:340-423)")
    ap.add_argument("--cpu-sample", type=int, default=1 << 24, help="rays of the batch the CPU oracle traces (and the GPU result is compared on): the whole batch by default, ~20 s of CPU work")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import pandora_amd as pa

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    # ---- scene (replicated per rank) and ray sets
    mesh = pa.scenes.g_atrium(args.tris)
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    ctx = pa.accel.Context(dev.index)
    bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
    stream = torch.cuda.current_stream().cuda_stream

    prim = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, args.side, args.side, seed=891379 + 7919 * rank)
    d_prim = torch.from_numpy(prim).to(dev)
    first = ctx.trace_bvh8(bvh, d_prim, stream=stream)
    torch.cuda.synchronize()
    t_hit = first["t"].cpu().numpy()
    obj = first["obj"].cpu().numpy().view(np.uint32)
    prim_id = np.where(obj == 0xFFFFFFFF, np.uint32(0xFFFFFFFF), hb.prim_ids[np.minimum(obj, hb.prim_ids.size - 1)])
    rays = pa.scenes.diffuse_bounce_rays(mesh, prim, t_hit, prim_id, seed=891379 + 7919 * rank)
    del d_prim, first
    n = int(rays.shape[1])
    d_rays = torch.from_numpy(rays).to(dev)
    kind = pa.capi.CLOSEST if args.kind == "closest" else pa.capi.ANY
    tri_mode = pa.capi.TRI_MT if args.tri == "moller-trumbore" else pa.capi.TRI_WATERTIGHT
    out = ctx.trace_bvh8(bvh, d_rays, kind=kind, tri_mode=tri_mode, stream=stream)  # allocates the result arrays once

    def step():
        ctx.trace_bvh8(bvh, d_rays, kind=kind, tri_mode=tri_mode, out=out, stream=stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        # per-launch kernel time from HIP events recorded by the library on this same stream; reading
        # them needs the step to have finished, which also keeps one batch in flight at a time
        ctx.sync(stream)
        kernel_ms.append(ctx.last_kernel_ms())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    from pandora_amd.sharded import reduce_job
    elapsed, total_rays = reduce_job(elapsed, n, dev)  # max over ranks, rays summed over ranks

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_rays * args.steps / elapsed / 1e6
        k_ms = float(np.mean(kernel_ms))
        scene_bytes = hb.scene_bytes(mesh.n_verts)
        per_ray = 52 if args.kind == "closest" else 36
        alg_bytes = per_ray * n + scene_bytes
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        hit_frac = float((out["obj"].cpu().numpy().view(np.uint32) != 0xFFFFFFFF).mean()) if args.kind == "closest" else float(out["hit"].cpu().numpy().mean())
        cpu = None
        if not args.no_cpu:
            import oracle_binding as ob
            threads = max(1, min(16, os.cpu_count() or 1))
            sample = min(n, args.cpu_sample)
            orc = ob.OracleBvh8.from_host(hb, mesh)
            sub = np.ascontiguousarray(rays[:, :sample])
            c0 = time.perf_counter()
            want = orc.trace(sub, any_hit=(args.kind == "any"), tri_mode=tri_mode, threads=threads)
            c1 = time.perf_counter()
            if args.kind == "closest":
                same = all(np.array_equal(out[k][:sample].cpu().numpy().view(np.uint32), want[k].view(np.uint32)) for k in ("t", "obj", "u", "v"))
            else:
                same = bool(np.array_equal(out["hit"][:sample].cpu().numpy().view(np.uint32), want["hit"]))
            cpu = {"value": sample / (c1 - c0) / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                   "sample": f"first {sample} rays of the same R-diffuse batch, oracle/pandora_oracle.c (scalar restatement of WiVeBVH8::intersect), {threads} pthreads",
                   "gpu_matches_bit_exact": bool(same),
                   "visits_per_ray": {k: want["counters"][k] / sample for k in ("inner_visits", "leaf_pops", "tri_tests")}}
            # SURVEY.md 8d, secondary figure: bytes a ray pulls through the caches (256 B per inner visit, 4 B per leaf index,
            # 48 B per triangle record), from the oracle's visit counters -- what L2 / Infinity Cache must serve
            vpr = cpu["visits_per_ray"]
            cpu["traversal_model_bytes_per_ray"] = 256.0 * vpr["inner_visits"] + 4.0 * vpr["tri_tests"] + 48.0 * vpr["tri_tests"]
        # HBM-side bytes per launch of the dominant kernel come from rocprofv3 PMC passes of this same command
        # (tools/pmc_passes.sh), which cannot run inside the timed process; the committed summary is quoted.
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if args.kind == "closest" and args.tri == "moller-trumbore" and os.path.exists(tpath):
            with open(tpath) as fh:
                tj = json.load(fh)
            if tj.get("rays_per_launch") == n:
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/pmc_traffic.json (" + tj["how"] + ")"
        line = {
            "metric": "Mrays/s incoherent diffuse rays", "value": value, "unit": "Mrays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: Sponza-class synthetic atrium (%d triangles), in-core single-level WiVe BVH8, R-diffuse %s-hit" % (mesh.n_tris, args.kind),
                       "rays_per_gpu": n, "triangles": mesh.n_tris, "bvh8_nodes": hb.n_nodes, "bvh8_depth": hb.depth,
                       "triangle_test": args.tri, "hit_fraction": hit_frac, "parallelism": "rays sharded over %d GPU(s), scene replicated" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel": "bvh8TraceKernel", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": alg_bytes, "bytes_per_ray": per_ray, "scene_bytes": scene_bytes},
            "cpu_baseline": cpu,
        }
        if cpu is not None:
            line["roofline"]["traversal_model_GBps"] = cpu["traversal_model_bytes_per_ray"] * n / (k_ms * 1e-3) / 1e9
        print(json.dumps(line), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
