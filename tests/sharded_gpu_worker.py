"""One rank of the 2-process GPU test of the leaf-sharded loop (both ranks share GPU 0; records travel over gloo
through host memory). Started by tests/test_gpu_sharded.py with RANK / WORLD_SIZE / MASTER_* set."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import pandora_amd as pa  # noqa: E402
from pandora_amd import sharded  # noqa: E402


def main():
    # mode 0 / 1: Python loop (torch / device records), 2: pgpu_trace_sharded, 3: pgpu_trace_sharded on the instanced landscape,
    # 4: pgpu_trace_sharded on a city whose BVHs every rank builds for its own batching points only, 5: the same with owners
    # contiguous in the top level and the rays at home where they start
    out_dir, kind, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    device_records = bool(mode)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        if mode in (3, 6):  # 6: every rank's batching points staged in host memory, streamed through a cache of a fifth of their bytes
            instanced_landscape(out_dir, kind, rank, world, dev, streamed=(mode == 6))
            return
        homed = False
        if mode in (4, 5):  # a city of ~0.35 M triangles in ~60 batching points, every rank building only its own (as bench.py does)
            # 5: owners contiguous in the top level's leaf order, and every ray at home on the rank that owns the point it starts in
            homed = mode == 5
            mesh, tl, owner_by_input = city_scene(world, rank, contiguous=homed)
            mode = 2
        else:
            mesh = pa.scenes.g_small()
            tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=32)
            owner_by_input = sharded.assign_leaves([b.n_nodes for b in tl.bvhs], world)
        owner = owner_by_input[np.asarray(tl.top.leaf_order, np.int64)]  # per leaf slot
        ctx = pa.accel.Context(0)
        if mode == 2:  # (as bench.py does: only the triangles of this rank's batching points are uploaded)
            part_pos, part_idx, remap = tl.sub_mesh(mesh.pos, mesh.idx, owner == rank)
            assert 0 < part_idx.shape[0] < mesh.n_tris
            tl.upload(ctx, ctx.upload_mesh(part_pos, part_idx), resident=(owner == rank), prim_remap=remap)
        else:
            tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), resident=(owner == rank))
        rays = np.load(os.path.join(out_dir, "rays.npy"))
        if homed:
            slot_bounds = tl.leaf_bounds[np.asarray(tl.top.leaf_order, np.int64)]
            mine = np.nonzero(sharded.home_ranks_by_origin(slot_bounds, owner, rays) == rank)[0]
            np.save(os.path.join(out_dir, f"ids{rank}.npy"), mine)
            rays = rays[:, mine]
        else:
            rays = rays[:, rank::world]
        d_rays = torch.from_numpy(np.ascontiguousarray(rays)).to(dev)
        if mode == 2:
            # the whole loop inside the library; the collectives go through host memory + gloo (two ranks share one GPU, which RCCL refuses)
            sharded.comm_init(ctx, dev, host_staged=True)
            ctx.comm_selftest()
            ctx.set_leaf_owners(owner)
            ctx.set_svdag_policy(int(os.environ.get("PANDORA_TEST_SVDAG_POLICY", "0")))
            fail_spec = os.environ.get("PANDORA_TEST_FAIL", "")  # "<rank>:<completed rounds>": that rank's loop fails there
            if fail_spec:
                bad_rank, after = (int(x) for x in fail_spec.split(":"))
                if rank == bad_rank:
                    ctx.debug_fail_sharded(after)
                try:
                    ctx.trace_sharded(d_rays, kind=kind)
                    code = 0
                except pa.capi.PandoraError as e:
                    code = e.code
                np.savez(os.path.join(out_dir, f"r{rank}.npz"), code=code)
                # the communicator is still usable: the ranks left the loop in the same round, nobody is inside a collective
                ctx.debug_fail_sharded(-1)
                st, cnt = ctx.trace_sharded(d_rays, kind=kind)
                np.savez(os.path.join(out_dir, f"again{rank}.npz"), **{k: v.cpu().numpy() for k, v in st.items()})
                ctx.close()
                return
            st, cnt = ctx.trace_sharded(d_rays, kind=kind)
            info = {"routed": cnt["records_routed"], "rounds": cnt["rounds"], "leaf_visits": cnt["leaf_visits"]}
            film = torch.full((5, 3), rank + 1, dtype=torch.int64, device=dev)
            ctx.film_allreduce(film)
            assert bool((film == world * (world + 1) // 2).all()), "pgpu_film_allreduce over the host-staged transport"
        else:
            eng = sharded.GpuEngine(ctx, dev, device_records=device_records)
            st, info = sharded.trace_sharded(eng, d_rays, owner, kind)
            cnt = ctx.counters()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), routed=info["routed"], rounds=info["rounds"], leaf_visits=info["leaf_visits"],
                 svdag_tests=cnt["svdag_tests"], svdag_culled=cnt["svdag_culled"],
                 **{k: v.cpu().numpy() for k, v in st.items() if k not in ("node", "stack")})
        ctx.close()
    finally:
        dist.destroy_process_group()


def city_mesh():
    return pa.scenes.g_city(4, 20000)


def city_scene(world, rank, contiguous=False):
    """(mesh, two-level scene with the BVHs of `rank`'s batching points only -- all of them when rank is None --, owner per input group)"""
    mesh = city_mesh()
    box = {}

    def mine(sizes, top):
        if contiguous:  # runs of the top level's leaf order (neighbouring space), balanced by triangle count
            order = np.asarray(top.leaf_order, np.int64)
            box["owner"] = np.empty(len(sizes), np.int64)
            box["owner"][order] = sharded.assign_leaves_contiguous(np.asarray(sizes)[order], world)
        else:
            box["owner"] = sharded.assign_leaves(sizes, world)
        return np.ones(len(sizes), bool) if rank is None else box["owner"] == rank

    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 8192, svdag_res=32, build_group=mine)
    return mesh, tl, box["owner"]


def landscape_scene():
    terrain, protos, proto, mats = pa.scenes.g_landscape(n_prototypes=6, n_instances=500, terrain_n=64, size=80.0, detail=2)
    return pa.accel.build_instanced_two_level(terrain.pos, terrain.idx, [(p.pos, p.idx) for p in protos], proto, mats, 1500, svdag_res=32)


def instanced_landscape(out_dir, kind, rank, world, dev, streamed=False):
    """Batching points (terrain triangles + tree instances) alternate between the ranks; every rank holds all prototypes. A ray
    that parks at the other rank's point travels there, may hit a tree through its instance, and the instance id comes back."""
    tl = landscape_scene()
    owner = (np.arange(tl.n_leaves) % world).astype(np.uint32)  # per leaf slot
    ctx = pa.accel.Context(0)
    tl.upload(ctx, streamed=streamed, resident=(owner == rank))
    if streamed:
        order = np.asarray(tl.top.leaf_order, np.int64)
        image = [tl.bvhs[g].n_nodes * 256 + (tl.bvhs[g].leaf_idx.size + 1) * 48 + 64 * tl.group_insts[g].size + 1024 for s, g in enumerate(order) if owner[s] == rank]
        ctx.set_leaf_cache(max(2 * max(image), sum(image) // 5))
        ctx.set_leaf_in_place(0)  # first with every queue's image loaded; the second trace below with sparse queues traced in host memory
    rays = np.load(os.path.join(out_dir, "rays.npy"))[:, rank::world]
    d_rays = torch.from_numpy(np.ascontiguousarray(rays)).to(dev)
    sharded.comm_init(ctx, dev, host_staged=True)
    ctx.set_leaf_owners(owner)
    st, cnt = ctx.trace_sharded(d_rays, kind=kind)
    if streamed:
        assert cnt["leaf_loads"] > len(image) and cnt["streamed_bytes"] > sum(image), "the cache holds a fifth of this rank's images: they must be loaded more than once"
        ctx.set_leaf_in_place(128)  # (the default)
        st2, cnt2 = ctx.trace_sharded(d_rays, kind=kind)  # again, from a warm cache and with sparse queues traced where their image lies: same results
        for k in st:
            assert torch.equal(st[k], st2[k]), k
        assert cnt2["leaf_loads"] < cnt["leaf_loads"]
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), routed=cnt["records_routed"], rounds=cnt["rounds"], leaf_visits=cnt["leaf_visits"],
             svdag_tests=cnt["svdag_tests"], svdag_culled=cnt["svdag_culled"], **{k: v.cpu().numpy() for k, v in st.items()})
    ctx.close()


if __name__ == "__main__":
    main()
