"""GPU parity of the sharded-loop building blocks (pgpu_ray_state_init / pgpu_top_episode / pgpu_trace_leaves):
driven by pandora_amd.sharded they must reproduce pgpu_trace bit for bit -- in one process, and with the batching
points split over two processes that exchange ray records (gloo through host memory; both share the one GPU)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import pandora_amd as pa
from pandora_amd import sharded

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _bits(a):
    return np.asarray(a).view(np.uint32)


def _rays(mesh):
    prim = pa.scenes.primary_rays((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 64, 64)
    rnd = pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), 4000, seed=5)
    far = pa.scenes.random_rays((-300, -300, -300), (300, 300, 300), 3000, seed=6, toward=((-1, 0, -1), (1, 2, 1)))
    return np.ascontiguousarray(np.concatenate([prim, rnd, far], axis=1))


@pytest.fixture(scope="module")
def scene():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=32)
    ctx = pa.accel.Context(0)
    tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
    yield ctx, mesh, tl
    ctx.close()


@pytest.mark.parametrize("device_records", [True, False], ids=["device_records", "torch_records"])
@pytest.mark.parametrize("kind", [pa.capi.CLOSEST, pa.capi.ANY], ids=["closest", "any"])
def test_single_process_blocks_equal_pgpu_trace(scene, kind, device_records):
    import torch
    ctx, mesh, tl = scene
    rays = _rays(mesh)
    want, wcnt = ctx.trace(rays, kind=kind)
    dev = torch.device("cuda", 0)
    before = ctx.counters()
    st, info = sharded.trace_sharded(sharded.GpuEngine(ctx, dev, device_records=device_records), torch.from_numpy(rays).to(dev), np.zeros(tl.n_leaves, np.int64), kind)
    after = ctx.counters()
    keys = ("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")
    for k in keys:
        assert np.array_equal(_bits(st[k].cpu().numpy()), _bits(want[k])), k
    assert info["rounds"] == wcnt["rounds"] and info["leaf_visits"] == wcnt["leaf_visits"]
    assert after["svdag_tests"] - before["svdag_tests"] == wcnt["svdag_tests"]
    assert after["svdag_culled"] - before["svdag_culled"] == wcnt["svdag_culled"]


def test_trace_rejects_non_resident_leaves():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64)
    ctx = pa.accel.Context(0)
    resident = np.zeros(tl.n_leaves, bool)
    resident[::2] = True
    tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx), resident=resident)
    with pytest.raises(pa.capi.PandoraError) as e:
        ctx.trace(np.ascontiguousarray(_rays(mesh)[:, :64]))
    assert e.value.code == pa.capi.ERR_NO_SCENE
    ctx.close()


@pytest.mark.parametrize("kind", [pa.capi.CLOSEST, pa.capi.ANY], ids=["closest", "any"])
def test_c_abi_loop_one_rank_over_rccl(scene, kind):
    """pgpu_trace_sharded with a real RCCL communicator of one rank (the only size one GPU allows): the library loads
    librccl, creates the communicator, and the loop -- every batching point local -- equals pgpu_trace bit for bit."""
    import torch
    ctx, mesh, tl = scene
    rays = _rays(mesh)
    want, wcnt = ctx.trace(rays, kind=kind)
    ctx.comm_init(0, 1, pa.accel.Context.comm_unique_id())
    try:
        ctx.comm_selftest()  # ncclAllGather, ncclAllReduce and a grouped ncclSend / ncclRecv (to itself) run on the GPU
        ctx.set_leaf_owners(np.zeros(tl.n_leaves, np.uint32))
        got, cnt = ctx.trace_sharded(torch.from_numpy(rays).cuda(), kind=kind)
        for k in (("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")):
            assert np.array_equal(_bits(got[k].cpu().numpy()), _bits(want[k])), k
        for k in ("rounds", "leaf_visits", "svdag_tests", "svdag_culled"):
            assert cnt[k] == wcnt[k], k
        assert cnt["records_routed"] == 0 and cnt["exchange_bytes"] == 0
        # the film all-reduce of one rank leaves the film alone; a 1-rank ncclAllReduce still has to run: do one directly
        film = torch.arange(30, dtype=torch.int64, device="cuda").reshape(10, 3).contiguous()
        ctx.film_allreduce(film)
        assert bool((film.cpu() == torch.arange(30).reshape(10, 3)).all())
    finally:
        ctx.comm_destroy()


def test_set_leaf_owners_checks_residency(scene):
    ctx, mesh, tl = scene
    ctx.comm_destroy()
    with pytest.raises(pa.capi.PandoraError):
        ctx.set_leaf_owners(np.ones(tl.n_leaves, np.uint32))  # rank 1 of a 1-rank world
    with pytest.raises(pa.capi.PandoraError):
        ctx.set_leaf_owners(np.zeros(tl.n_leaves + 1, np.uint32))


@pytest.mark.parametrize("device_records", [2, 1, 0], ids=["c_abi_loop", "device_records", "torch_records"])
@pytest.mark.parametrize("kind", [pa.capi.CLOSEST, pa.capi.ANY], ids=["closest", "any"])
def test_two_processes_sharded_leaves(scene, kind, device_records):
    ctx, mesh, tl = scene
    rays = _rays(mesh)
    want, wcnt = ctx.trace(rays, kind=kind)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "rays.npy"), rays)
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "sharded_gpu_worker.py"), d, str(kind), str(device_records)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
        parts = [np.load(os.path.join(d, f"r{r}.npz")) for r in range(2)]
    assert sum(int(p["routed"]) for p in parts) > 0
    keys = ("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits")
    for r, p in enumerate(parts):
        for k in keys:
            assert np.array_equal(_bits(p[k]), _bits(want[k])[r::2]), f"rank {r}: {k}"
    assert sum(int(p["leaf_visits"]) for p in parts) == wcnt["leaf_visits"]
    assert sum(int(p["svdag_tests"]) for p in parts) == wcnt["svdag_tests"]
    assert sum(int(p["svdag_culled"]) for p in parts) == wcnt["svdag_culled"]


@pytest.mark.parametrize("after_rounds", [0, 1])
def test_a_failing_rank_takes_its_peers_out_of_the_loop_together(scene, after_rounds):
    """Rank 1 fails inside pgpu_trace_sharded (before the first round / after one round, the way a failed launch or allocation
    would): it must still take part in the round's collectives, rank 0 must come back with PGPU_ERR_PEER instead of waiting
    inside an all-to-all, and the same communicator must serve a complete trace right afterwards."""
    ctx, mesh, tl = scene
    rays = _rays(mesh)
    want, _ = ctx.trace(rays)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "rays.npy"), rays)
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PANDORA_TEST_FAIL=f"1:{after_rounds}")
            procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "sharded_gpu_worker.py"), d, str(pa.capi.CLOSEST), "2"], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=120)[0].decode(errors="replace") for p in procs]  # (a stranded rank would run into this timeout)
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
        codes = [int(np.load(os.path.join(d, f"r{r}.npz"))["code"]) for r in range(2)]
        again = [dict(np.load(os.path.join(d, f"again{r}.npz"))) for r in range(2)]
    assert codes == [pa.capi.ERR_PEER, pa.capi.ERR_HIP], codes
    for r, p in enumerate(again):
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            assert np.array_equal(_bits(p[k]), _bits(want[k])[r::2]), f"after the failure, rank {r}: {k}"


def test_two_processes_sharded_svdag_only_for_remote_points(scene):
    """pgpu_set_svdag_policy(ctx, 1) on two ranks: a rank tests the SVDAG of the other rank's batching points (a miss saves a
    record's round trip) and visits its own without the test. Same hits as one context; fewer SVDAG tests."""
    ctx, mesh, tl = scene
    rays = _rays(mesh)
    want, wcnt = ctx.trace(rays)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "rays.npy"), rays)
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PANDORA_TEST_SVDAG_POLICY="1")
            procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "sharded_gpu_worker.py"), d, str(pa.capi.CLOSEST), "2"], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
        parts = [dict(np.load(os.path.join(d, f"r{r}.npz"))) for r in range(2)]
    for r, p in enumerate(parts):
        for k in ("t", "obj", "leaf", "u", "v"):
            assert np.array_equal(_bits(p[k]), _bits(want[k])[r::2]), f"rank {r}: {k}"
    tests = sum(int(p["svdag_tests"]) for p in parts)
    assert 0 < tests < wcnt["svdag_tests"]
    assert sum(int(p["leaf_visits"]) for p in parts) >= wcnt["leaf_visits"]


@pytest.mark.parametrize("mode", [4, 5], ids=["lpt_owners_interleaved_rays", "contiguous_owners_rays_at_home"])
def test_four_processes_sharded_city(mode):
    """Four ranks (sharing the one GPU, records through host memory) on a city of ~60 batching points, each rank building and
    uploading only its own points' BVHs and triangles, 150 K shuffled diffuse rays: several rounds with thousands of records
    per peer and direction, buffers that grow -- results equal one context's pgpu_trace bit for bit."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import sharded_gpu_worker as worker
    mesh, tl, _ = worker.city_scene(4, None)
    assert tl.n_leaves >= 32
    ctx = pa.accel.Context(0)
    try:
        tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
        prim = pa.scenes.primary_rays((8.0, 14.0, 8.0), (48.0, 0.0, 48.0), (0, 1, 0), 60, 400, 400)
        first, _ = ctx.trace(prim)
        rays = pa.scenes.diffuse_bounce_rays(mesh, prim, np.asarray(first["t"]), tl.prim_ids_of(np.asarray(first["leaf"]), np.asarray(first["obj"])))
        got, wcnt = ctx.trace(rays)  # (numpy in, numpy out)
        want = {k: np.asarray(v) for k, v in got.items()}
    finally:
        ctx.close()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "rays.npy"), rays)
        procs = []
        for r in range(4):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "sharded_gpu_worker.py"), d, str(pa.capi.CLOSEST), str(mode)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
        parts = [dict(np.load(os.path.join(d, f"r{r}.npz"))) for r in range(4)]
        ids = [np.load(os.path.join(d, f"ids{r}.npy")) if mode == 5 else np.arange(r, rays.shape[1], 4) for r in range(4)]
    routed, visits = sum(int(p["routed"]) for p in parts), sum(int(p["leaf_visits"]) for p in parts)
    frac = routed / visits
    print(f"records_routed_fraction, 4 ranks, mode {mode}: {frac:.3f} ({routed} of {visits} batching-point visits)")
    if mode == 4:
        # owners spread by size (LPT), rays dealt round-robin: a visit is local with probability 1 / 4
        assert 0.65 < frac < 0.85
    else:
        # owners = contiguous runs of the top level's leaf order, rays at home where they start: the first visit is local, and most
        # of the later ones (a ray that leaves a batching point enters a neighbour) -- well below (N - 1) / N = 0.75
        assert frac < 0.25 and np.array_equal(np.sort(np.concatenate(ids)), np.arange(rays.shape[1]))
        assert min(i.size for i in ids) > 0
    for r, p in enumerate(parts):
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            assert np.array_equal(_bits(p[k]), _bits(want[k])[ids[r]]), f"rank {r}: {k}"
    assert sum(int(p["leaf_visits"]) for p in parts) == wcnt["leaf_visits"]
    assert sum(int(p["svdag_tests"]) for p in parts) == wcnt["svdag_tests"]
    assert sum(int(p["svdag_culled"]) for p in parts) == wcnt["svdag_culled"]


@pytest.mark.parametrize("mode", [3, 6], ids=["resident", "staged_cache_a_fifth"])
@pytest.mark.parametrize("kind", [pa.capi.CLOSEST, pa.capi.ANY], ids=["closest", "any"])
def test_two_processes_sharded_instanced_landscape(kind, mode):
    """Instanced leaves + batching points sharded over two ranks (pgpu_trace_sharded): the instancing BVHs live on their owners,
    the prototypes on both; results -- the instance a hit went through included, which travels back as a fifth value of the
    result row -- equal one context's pgpu_trace, which tests/test_gpu_instances.py checks against the oracle.
    staged: BASELINE configs[4] as stated -- sharded AND streamed: every rank keeps its batching points in pinned host memory and
    streams them through an HBM cache of a fifth of their bytes, for the rays parked there in place and for the records it receives."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import sharded_gpu_worker as worker
    tl = worker.landscape_scene()
    size = 80.0
    rays = np.ascontiguousarray(np.concatenate([
        pa.scenes.primary_rays((0.1 * size, 25.0, 0.1 * size), (0.6 * size, 0.0, 0.6 * size), (0, 1, 0), 55, 64, 64),
        pa.scenes.random_rays((0, 0, 0), (size, 20, size), 4000, seed=21)], axis=1))
    ctx = pa.accel.Context(0)
    try:
        tl.upload(ctx)
        want, wcnt = ctx.trace(rays, kind=kind)
    finally:
        ctx.close()
    if kind == pa.capi.CLOSEST:
        assert (want["inst"] != 0xFFFFFFFF).mean() > 0.02, "the rays should meet some trees"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "rays.npy"), rays)
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "sharded_gpu_worker.py"), d, str(kind), str(mode)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        outs = [p.communicate(timeout=300)[0].decode(errors="replace") for p in procs]
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-3000:]
        parts = [np.load(os.path.join(d, f"r{r}.npz")) for r in range(2)]
    assert sum(int(p["routed"]) for p in parts) > 0
    keys = ("hit", "visits") if kind == pa.capi.ANY else ("t", "obj", "leaf", "u", "v", "visits", "inst")
    for r, p in enumerate(parts):
        for k in keys:
            bad = np.nonzero(_bits(p[k]) != _bits(want[k])[r::2])[0]
            assert bad.size == 0, f"rank {r}: {k} differs on {bad.size} rays, first {bad[:6]}"
    assert sum(int(p["leaf_visits"]) for p in parts) == wcnt["leaf_visits"]
    assert sum(int(p["svdag_tests"]) for p in parts) == wcnt["svdag_tests"]
