#!/usr/bin/env python3
"""Timing of the leaf-sharded two-level loop (BASELINE config 4): batching points split over the ranks' HBM, ray
records routed with all_to_all (RCCL). Launch like bench.py:
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/bench_sharded.py --steps K
With one process it runs the same loop without an exchange. Prints one JSON line on rank 0 (weak scaling: every rank
brings its own rays; the scene is fixed)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=8, help="city blocks per side")
    ap.add_argument("--tris-per-block", type=int, default=40000)
    ap.add_argument("--prims-per-point", type=int, default=32768)
    ap.add_argument("--svdag", type=int, default=64)
    ap.add_argument("--side", type=int, default=2048)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import pandora_amd as pa
    from pandora_amd import sharded
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local if world > 1 else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    mesh = pa.scenes.g_city(args.blocks, args.tris_per_block)
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, args.prims_per_point, svdag_res=args.svdag)
    owner_in = sharded.assign_leaves([b.n_nodes * 256 + b.leaf_idx.size * 48 for b in tl.bvhs], world)
    owner = owner_in[np.asarray(tl.top.leaf_order, np.int64)]
    ctx = pa.accel.Context(dev.index)
    tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), resident=(owner == rank))
    size = 20.0 * args.blocks
    prim = pa.scenes.primary_rays((0.1 * size, 14.0, 0.1 * size), (0.6 * size, 0.0, 0.6 * size), (0, 1, 0), 60, args.side, args.side, seed=891379 + 7919 * rank)
    rays = torch.from_numpy(prim).to(dev)
    eng = sharded.GpuEngine(ctx, dev)
    info = None
    for _ in range(args.warmup):
        _, info = sharded.trace_sharded(eng, rays, owner, pa.capi.CLOSEST)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st, info = sharded.trace_sharded(eng, rays, owner, pa.capi.CLOSEST)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    n = rays.shape[1]
    vals = torch.tensor([el, float(n), float(info["routed"]), float(info["leaf_visits"])], dtype=torch.float64, device=dev)
    if world > 1:
        mx = vals.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        el = float(mx[0])
    if rank == 0:
        total = float(vals[1])
        print(json.dumps({"metric": "Mrays/s primary rays, two-level, batching points sharded over ranks", "value": total * args.steps / el / 1e6, "unit": "Mrays/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True,
                          "scaling": "weak", "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "configs[3]-style: synthetic city, %d triangles, %d batching points, SVDAG %d^3" % (mesh.n_tris, tl.n_leaves, args.svdag),
                                     "rays_per_gpu": n, "rounds": info["rounds"], "leaf_visits_per_ray": float(vals[3]) / total,
                                     "records_routed_fraction": float(vals[2]) / max(1.0, float(vals[3])),
                                     "resident_bytes_rank0": int(sum(b.n_nodes * 256 + b.leaf_idx.size * 48 for i, b in enumerate(tl.bvhs) if owner_in[i] == 0))}}), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
