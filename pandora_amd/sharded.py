"""Two-level traversal with the batching points sharded over several GPUs (BASELINE config 4).

Every rank holds the top-level BVH4 and the SVDAGs (KBs..MBs) and the bottom-level BVH8s of the batching points
it owns. Rays live on their home rank. One round:

  home:   top-level episode for the active rays  -> (ray, batching point) for every ray that parked
  home:   records {ray 8 x f32 with tfar = current t, home slot} sorted by owner(batching point)
  all-to-all (RCCL over xGMI; counts first, then the records)
  owner:  bottom-level traversal of the received rays through the named batching points
  all-to-all back: {t, obj, u, v}
  home:   merge closer hits, resume the parked rays in the next round

until no rank has a parked ray. This replaces the reference's per-leaf MPMC queues + disk cache
(offline_batching_acceleration_structure.h:158-260, task_graph.cpp:19-56): a leaf's "queue" is whatever the
exchange delivers to its owner in a round. Visit order per ray is the top-level traversal's, exactly as on
one GPU, so results are identical to pgpu_trace.

The loop is written against a small engine interface so that the routing logic can be exercised on CPU ranks
(gloo) with a CPU engine in tests; the product engine is GpuEngine (the C ABI).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import capi

MISS = 0xFFFFFFFF


def assign_leaves(weights, world: int) -> np.ndarray:
    """Owner rank per batching point: greedy longest-processing-time balance of `weights` (bytes or expected
    ray load). Deterministic, so every rank computes the same table."""
    w = np.asarray(weights, np.float64)
    owner = np.zeros(w.size, np.int64)
    load = np.zeros(world, np.float64)
    for i in np.argsort(-w, kind="stable"):
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += w[i]
    return owner


def assign_leaves_contiguous(weights_by_slot, world: int) -> np.ndarray:
    """Owner rank per batching point by position in the top level: the leaf slots of the PauseableBVH4 are in depth-first
    order (pauseable_bvh4.h:461-482), so neighbouring slots are neighbouring subtrees, i.e. neighbouring space. The slot
    sequence is cut into `world` runs of about equal weight: a ray that leaves one batching point mostly enters one a few slots
    away -- with this table that is usually the same rank's. (assign_leaves balances better and scatters neighbours over all
    ranks.) weights_by_slot: one weight per leaf SLOT. Returns the owner per slot. Deterministic."""
    w = np.asarray(weights_by_slot, np.float64)
    if w.size == 0:
        return np.zeros(0, np.int64)
    total = float(w.sum())
    if total <= 0.0:
        return (np.arange(w.size) * world // w.size).astype(np.int64)
    mid = np.cumsum(w) - 0.5 * w  # a point belongs to the run its centre of weight falls into
    return np.minimum(world - 1, (mid * world / total).astype(np.int64))


def home_ranks_by_origin(leaf_bounds_by_slot, owner_by_slot, rays) -> np.ndarray:
    """Home rank per ray = the owner of the batching point its origin lies in (the smallest such box; the nearest box centre for
    an origin outside all of them). What a distributed integrator does when it spawns a bounce ray at a hit point: the ray
    starts where its geometry is. With owners that are contiguous in the top level (assign_leaves_contiguous) most of a ray's
    batching-point visits then stay on its home rank. leaf_bounds_by_slot: [n_leaves, 6] (min, max) in slot order; rays [8, n]."""
    b = np.asarray(leaf_bounds_by_slot, np.float32)
    o = np.asarray(rays, np.float32)[0:3].T
    owner = np.asarray(owner_by_slot, np.int64)
    vol = np.prod(np.maximum(b[:, 3:] - b[:, :3], 1e-12).astype(np.float64), axis=1)
    best = np.full(o.shape[0], -1, np.int64)
    best_vol = np.full(o.shape[0], np.inf)
    for s in np.argsort(vol):  # (n_leaves passes over the rays: fine for the sizes this is used at)
        inside = np.all((o >= b[s, :3]) & (o <= b[s, 3:]), axis=1) & (vol[s] < best_vol)
        best[inside] = s
        best_vol[inside] = vol[s]
    out = best < 0
    if out.any():
        centre = 0.5 * (b[:, :3] + b[:, 3:])
        for i in np.nonzero(out)[0]:
            best[i] = int(np.argmin(((centre - o[i]) ** 2).sum(axis=1)))
    return owner[best]


def reduce_job(elapsed_s: float, n_units: int, device=None):
    """Whole-job figures of a run in which every rank processed its own units (rays) between two barriers:
    (slowest rank's time, units summed over ranks). Works on any backend (nccl needs `device`)."""
    rank, world = _world()
    if world == 1:
        return float(elapsed_s), int(n_units)
    dev = device if device is not None else "cpu"
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([n_units], dtype=torch.int64, device=dev)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def film_allreduce(film: torch.Tensor) -> torch.Tensor:
    """Sums the per-rank films (int64, 40.24 fixed point, see Context.film_add) in place. Integer addition is associative,
    so the result is the film one process would have accumulated from all contributions, bit for bit, whatever the
    number of ranks (the property the reference's fixed-point sensor exists for, sensor.h:29-30)."""
    if film.dtype != torch.int64:
        raise ValueError("film_allreduce: the film is an int64 tensor")
    if _world()[1] > 1:
        dist.all_reduce(film, op=dist.ReduceOp.SUM)
    return film


def _world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def _all_to_all_rows(rows: torch.Tensor, send_counts: torch.Tensor, staging_cpu: bool):
    """rows [k, c] grouped by destination rank, send_counts [world] -> (received rows, recv_counts)."""
    rank, world = _world()
    if world == 1:
        return rows, send_counts.clone()
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    sc, rc = [int(x) for x in send_counts.tolist()], [int(x) for x in recv_counts.tolist()]
    src = rows.cpu() if staging_cpu else rows
    out = torch.empty((sum(rc), rows.shape[1]), dtype=rows.dtype, device=src.device)
    dist.all_to_all_single(out, src.contiguous(), output_split_sizes=rc, input_split_sizes=sc)
    return (out.to(rows.device) if staging_cpu else out), recv_counts


class HostStagedTransport:
    """A pgpu_transport (include/pandora_hip.h) over torch.distributed with host staging: device buffers are copied to the
    host, exchanged on the process group's backend (gloo), and copied back. For set-ups RCCL cannot serve -- several
    ranks sharing one GPU, as in the two-process test -- and as the template for a caller-supplied transport (MPI, ...).
    The product path on an 8-GPU node is pgpu_comm_init (RCCL)."""

    def __init__(self, group=None):
        import ctypes as C
        self._C = C
        self.group = group
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.world = dist.get_world_size(group)
        self._keep = (capi.ALLGATHER_FN(self._allgather), capi.EXCHANGE_FN(self._exchange), capi.ALLREDUCE_FN(self._allreduce))

    def struct(self):
        return capi.Transport(None, *self._keep)

    def _d2h(self, ptr, nbytes, stream):
        host = torch.empty(max(nbytes, 1), dtype=torch.uint8)
        if self.hip.hipStreamSynchronize(stream) != 0 or (nbytes and self.hip.hipMemcpy(host.data_ptr(), ptr, nbytes, 2) != 0):
            raise RuntimeError("device -> host copy failed")
        return host[:nbytes]

    def _h2d(self, ptr, host):
        if host.numel() and self.hip.hipMemcpy(ptr, host.data_ptr(), host.numel(), 1) != 0:
            raise RuntimeError("host -> device copy failed")

    def _allgather(self, user, send, recv, n, stream):
        try:
            mine = self._d2h(send, 4 * n, stream)
            out = torch.empty(4 * n * self.world, dtype=torch.uint8)
            dist.all_gather_into_tensor(out, mine.contiguous(), group=self.group)
            self._h2d(recv, out)
            return 0
        except Exception:
            return 1

    def _exchange(self, user, send, send_counts, recv, recv_counts, row_bytes, stream):
        try:
            sc = [int(send_counts[i]) * row_bytes for i in range(self.world)]
            rc = [int(recv_counts[i]) * row_bytes for i in range(self.world)]
            src = self._d2h(send, sum(sc), stream)
            out = torch.empty(sum(rc), dtype=torch.uint8)
            dist.all_to_all_single(out, src.contiguous(), output_split_sizes=rc, input_split_sizes=sc, group=self.group)
            self._h2d(recv, out)
            return 0
        except Exception:
            return 1

    def _allreduce(self, user, buf, n, stream):
        try:
            t = self._d2h(buf, 8 * n, stream).view(torch.int64).clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self._h2d(buf, t.view(torch.uint8))
            return 0
        except Exception:
            return 1


def comm_init(ctx, device: torch.device, host_staged: bool = False):
    """Gives `ctx` its communicator over the ranks of the default process group: RCCL (the unique id travels through a
    broadcast on the process group) or, with host_staged, the HostStagedTransport above."""
    rank, world = _world()
    if host_staged:
        ctx.comm_init_transport(rank, world, HostStagedTransport())
        return
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid = torch.frombuffer(bytearray(ctx.comm_unique_id()), dtype=torch.uint8).clone()
    if world > 1:
        on = uid.to(device) if dist.get_backend() == "nccl" else uid
        dist.broadcast(on, src=0)
        uid = on.cpu()
    ctx.comm_init(rank, world, bytes(uid.numpy().tobytes()))


class GpuEngine:
    """The product engine: pandora_hip through the C ABI, device tensors."""

    def __init__(self, ctx, device: torch.device, tri_mode: int = capi.TRI_MT, device_records: bool = True):
        self.ctx, self.device, self.tri_mode = ctx, device, tri_mode
        # device_records: pack / unpack / merge the exchanged records with the library's kernels (pgpu_route_pack,
        # pgpu_trace_leaf_records, pgpu_route_merge) instead of torch ops
        self.device_records = device_records
        # The library's kernels and torch's own (sorting, gathering the records) must run on ONE stream so that they
        # are ordered: a side stream of torch's, handed to every library call. (Stream 0 would make the library fall
        # back to its private stream, which torch knows nothing about.)
        self.stream = torch.cuda.Stream(device)

    def stream_context(self):
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        return torch.cuda.stream(self.stream)

    def init_state(self, rays, kind):
        n = rays.shape[1]
        z = lambda dt: torch.empty(n, dtype=dt, device=self.device)
        st = {"node": z(torch.int32), "stack": z(torch.int64), "visits": torch.zeros(n, dtype=torch.int32, device=self.device)}
        if kind == capi.CLOSEST:
            st.update(t=z(torch.float32), obj=z(torch.int32), leaf=z(torch.int32), u=z(torch.float32), v=z(torch.float32))
            self.ctx.ray_state_init(rays[7], st["node"], st["stack"], st, stream=self.stream.cuda_stream)
        else:
            st["hit"] = z(torch.int32)
            self.ctx.ray_state_init(None, st["node"], st["stack"], None, st["hit"], stream=self.stream.cuda_stream)
        return st

    def top_episode(self, kind, rays, active, st, may_park=True):
        n_active = int(active.shape[0])
        parked = torch.empty(n_active, dtype=torch.int32, device=self.device)
        self.ctx.top_episode(kind, rays, active, n_active, st["node"], st["stack"], st.get("t"), st["visits"], st.get("hit"), parked,
                             may_park=may_park, stream=self.stream.cuda_stream)
        return parked

    def trace_leaves(self, kind, rays, work_leaf):
        """rays [8, k] (tfar = current t), work_leaf [k] -> closest: (t, obj, u, v) with obj = MISS where nothing closer; any: hit flags."""
        k = rays.shape[1]
        work_ray = torch.arange(k, dtype=torch.int32, device=self.device)
        order = torch.argsort(work_leaf.to(torch.int64), stable=True).to(torch.int32)  # rays of one leaf adjacent
        wl = work_leaf[order.long()].contiguous()
        wr = work_ray[order.long()].contiguous()
        if kind == capi.CLOSEST:
            hits = {"t": rays[7].clone(), "obj": torch.full((k,), -1, dtype=torch.int32, device=self.device),
                    "leaf": torch.zeros(k, dtype=torch.int32, device=self.device),
                    "u": torch.zeros(k, dtype=torch.float32, device=self.device), "v": torch.zeros(k, dtype=torch.float32, device=self.device)}
            self.ctx.trace_leaves(kind, self.tri_mode, rays, wr, wl, k, hits, stream=self.stream.cuda_stream)
            return hits["t"], hits["obj"], hits["u"], hits["v"]
        hit = torch.zeros(k, dtype=torch.int32, device=self.device)
        self.ctx.trace_leaves(kind, self.tri_mode, rays, wr, wl, k, None, hit, stream=self.stream.cuda_stream)
        return hit

    # -- device-side record path
    def pack(self, kind, rays, active, parked, st, owner_dev, world):
        n_active = int(active.shape[0])
        rec = torch.empty((n_active, 9), dtype=torch.float32, device=self.device)
        slot = torch.empty(n_active, dtype=torch.int32, device=self.device)
        leaf = torch.empty(n_active, dtype=torch.int32, device=self.device)
        counts = self.ctx.route_pack(rays, st.get("t") if kind == capi.CLOSEST else None, active, parked, n_active, owner_dev, world, rec, slot, leaf,
                                     stream=self.stream.cuda_stream)
        k = int(counts.sum())
        return counts, rec[:k], slot[:k], leaf[:k]

    def trace_records(self, kind, records):
        k = int(records.shape[0])
        res = torch.empty((k, 4 if kind == capi.CLOSEST else 1), dtype=torch.float32, device=self.device)
        self.ctx.trace_leaf_records(kind, self.tri_mode, records.contiguous(), k, res, stream=self.stream.cuda_stream)
        return res

    def merge(self, kind, results, slot, leaf, st):
        self.ctx.route_merge(kind, results.contiguous(), slot, leaf, int(slot.shape[0]), st if kind == capi.CLOSEST else None, st.get("hit"),
                             stream=self.stream.cuda_stream)

    def finish(self):
        self.ctx.sync(self.stream.cuda_stream)
        torch.cuda.current_stream(self.device).wait_stream(self.stream)


def trace_sharded(engine, rays: torch.Tensor, owner_of_leaf, kind: int = capi.CLOSEST, staging_cpu: bool | None = None):
    """Traces this rank's rays [8, n] through the sharded scene. owner_of_leaf: int array over leaf slots.
    Returns (state dict: t/obj/leaf/u/v/visits or hit/visits, info dict: rounds, leaf_visits, routed)."""
    if hasattr(engine, "stream_context"):
        with engine.stream_context():
            return _trace_sharded(engine, rays, owner_of_leaf, kind, staging_cpu)
    return _trace_sharded(engine, rays, owner_of_leaf, kind, staging_cpu)


def _trace_sharded(engine, rays: torch.Tensor, owner_of_leaf, kind: int = capi.CLOSEST, staging_cpu: bool | None = None):
    rank, world = _world()
    dev = rays.device
    if staging_cpu is None:
        staging_cpu = world > 1 and dist.get_backend() != "nccl" and dev.type != "cpu"
    owner = torch.as_tensor(np.asarray(owner_of_leaf, np.int64), device=dev)
    n = rays.shape[1]
    st = engine.init_state(rays, kind)
    active = torch.arange(n, dtype=torch.int32, device=dev)
    rounds = leaf_visits = routed = 0
    fast = bool(getattr(engine, "device_records", False))
    owner_dev = owner.to(torch.int32).contiguous() if fast else None
    cdev = "cpu" if (staging_cpu or dev.type == "cpu") else dev
    while True:
        # (a ray pauses at most once per batching point: past that many rounds it may only leave, see pgpu_top_episode)
        parked = engine.top_episode(kind, rays, active, st, rounds < owner.numel()) if active.numel() else torch.empty(0, dtype=torch.int32, device=dev)
        if fast:
            # everything between the all-to-alls stays in the library's kernels; the host only sees the per-rank counts
            counts, rec, slots, leaves = engine.pack(kind, rays, active, parked, st, owner_dev, world)
            total = torch.tensor([int(counts.sum())], dtype=torch.int64, device=cdev)
            if world > 1:
                dist.all_reduce(total)
            if int(total.item()) == 0:
                break
            rounds += 1
            leaf_visits += int(counts.sum())
            routed += int(counts.sum() - counts[rank])
            send_counts = torch.from_numpy(counts.astype(np.int64)).to(cdev)
            got, recv_counts = _all_to_all_rows(rec, send_counts, staging_cpu)
            res = engine.trace_records(kind, got)
            back, _ = _all_to_all_rows(res, recv_counts, staging_cpu)
            engine.merge(kind, back, slots, leaves, st)
            active = slots
            continue
        keep = parked != -1  # int32 view of MISS
        slots = active[keep].long()      # home ray ids that parked
        leaves = parked[keep].long().clamp_(0, owner.numel() - 1)  # (a corrupt id must not become a wild gather)
        total = torch.tensor([slots.numel()], dtype=torch.int64, device=("cpu" if staging_cpu or dev.type == "cpu" else dev))
        if world > 1:
            dist.all_reduce(total)
        if int(total.item()) == 0:
            break
        rounds += 1
        leaf_visits += int(slots.numel())
        dest = owner[leaves]
        order = torch.argsort(dest, stable=True)
        slots, leaves, dest = slots[order], leaves[order], dest[order]
        send_counts = torch.bincount(dest, minlength=world).to(torch.int64)
        routed += int(slots.numel() - int(send_counts[rank]))
        # record = the ray with tfar replaced by its current t (closest hit), then the leaf (bit pattern in a float lane)
        rec = rays[:, slots].t().contiguous()                      # [k, 8]
        if kind == capi.CLOSEST:
            rec[:, 7] = st["t"][slots]
        rec = torch.cat([rec, leaves.to(torch.int32).view(-1, 1).view(torch.float32)], dim=1)  # [k, 9]
        cc = send_counts.cpu() if (staging_cpu or dev.type == "cpu") else send_counts
        got, recv_counts = _all_to_all_rows(rec, cc, staging_cpu)
        k = got.shape[0]
        if k:
            r_in = got[:, :8].t().contiguous()
            wl = got[:, 8].contiguous().view(torch.int32)
            if kind == capi.CLOSEST:
                t, obj, u, v = engine.trace_leaves(kind, r_in, wl)
                res = torch.stack([t, obj.view(torch.float32), u, v], dim=1)
            else:
                res = engine.trace_leaves(kind, r_in, wl).view(torch.float32).view(-1, 1)
        else:
            res = torch.empty((0, 4 if kind == capi.CLOSEST else 1), dtype=torch.float32, device=dev)
        back, _ = _all_to_all_rows(res, recv_counts, staging_cpu)  # rows return in the order they were sent
        if kind == capi.CLOSEST:
            hit = back[:, 1].contiguous().view(torch.int32) != -1
            hs = slots[hit]
            st["t"][hs] = back[hit, 0]
            st["obj"][hs] = back[hit, 1].contiguous().view(torch.int32)
            st["leaf"][hs] = leaves[hit].to(torch.int32)
            st["u"][hs] = back[hit, 2]
            st["v"][hs] = back[hit, 3]
        else:
            hit = back[:, 0].contiguous().view(torch.int32) != 0
            st["hit"][slots[hit]] = 1
        active = slots.to(torch.int32).contiguous()
    engine.finish()
    return st, {"rounds": rounds, "leaf_visits": leaf_visits, "routed": routed}
