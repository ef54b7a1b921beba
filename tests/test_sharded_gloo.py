"""N > 1 path on CPU ranks: the leaf-sharded two-level loop (pandora_amd/sharded.py) over gloo, world_size 2, with the
oracle as the per-rank engine, against the oracle's own single-process two-level traversal (bit-exact)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
import pandora_amd as pa
from pandora_amd import sharded


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, kind, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_engine import OracleEngine
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z, mesh, tl, scene = ob.load_golden_two_level()
        weights = [b.nodes.shape[0] for b in scene.leaf_bvhs]
        owner = sharded.assign_leaves(weights, world)
        rays = np.ascontiguousarray(z["rays"])[:, rank::world]  # every rank brings its own rays
        eng = OracleEngine(scene, owner == rank)
        st, info = sharded.trace_sharded(eng, torch.from_numpy(np.ascontiguousarray(rays)), owner, kind)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), routed=info["routed"], rounds=info["rounds"], leaf_visits=info["leaf_visits"],
                 svdag_tests=eng.counters.svdag_tests, svdag_culled=eng.counters.svdag_culled,
                 **{k: v.numpy() for k, v in st.items() if k not in ("node", "stack")})
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", [0, 1], ids=["closest", "any"])
def test_world2_matches_single_process(kind):
    world = 2
    z, mesh, tl, scene = ob.load_golden_two_level()
    rays = np.ascontiguousarray(z["rays"])
    want = scene.trace(rays, any_hit=bool(kind), lanes=3)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), kind, d), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"r{r}.npz")) for r in range(world)]
    assert sum(int(p["routed"]) for p in parts) > 0, "no ray crossed ranks: the exchange was not exercised"
    keys = ("hit", "visits") if kind else ("t", "obj", "leaf", "u", "v", "visits")
    for r, p in enumerate(parts):
        for k in keys:
            got = np.asarray(p[k]).view(np.uint32)
            exp = np.asarray(want[k])[r::world].view(np.uint32)
            assert np.array_equal(got, exp), f"rank {r}: {k} differs on {(got != exp).sum()} rays"
    assert sum(int(p["leaf_visits"]) for p in parts) == want["counters"]["leaf_visits"]
    assert sum(int(p["svdag_tests"]) for p in parts) == want["counters"]["svdag_tests"]
    assert sum(int(p["svdag_culled"]) for p in parts) == want["counters"]["svdag_culled"]


def test_assign_leaves_balances_and_is_deterministic():
    w = np.array([9, 1, 1, 1, 8, 2, 2, 7, 3], np.float64)
    a, b = sharded.assign_leaves(w, 3), sharded.assign_leaves(w, 3)
    assert np.array_equal(a, b) and set(a.tolist()) == {0, 1, 2}
    loads = np.bincount(a, weights=w, minlength=3)
    assert loads.max() - loads.min() <= w.max()
    assert np.array_equal(sharded.assign_leaves(w, 1), np.zeros(w.size, np.int64))


def _reduce_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        el, n = sharded.reduce_job(1.0 + rank, 1000 * (rank + 1))
        np.save(os.path.join(out_dir, f"red{rank}.npy"), np.array([el, n]))
    finally:
        dist.destroy_process_group()


def test_bench_aggregation_world2():
    """bench.py's whole-job figures: slowest rank's time, rays summed over ranks (every rank must see the same)."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_reduce_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        for r in range(2):
            el, n = np.load(os.path.join(d, f"red{r}.npy"))
            assert el == 2.0 and n == 3000
    assert sharded.reduce_job(0.5, 7) == (0.5, 7)  # no process group: identity
