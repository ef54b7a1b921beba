"""bench.py's launch path without a GPU: `--gpus N` started plainly (the way the driver starts it for N = 1 ... and the way a
user would for N > 1) must bring up N ranks that meet and reduce, and under torch.distributed.run it must use the ranks it
is given. `--dry` replaces the GPU work by made-up figures; everything else (spawn, rendezvous, reduce, one JSON line) is real."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_plain_start_spawns_the_ranks():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "3"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["dry"] is True and line["steps"] == 3
    # the sharded record's scene, built rank-locally on a small city: half the tiles each, the same top level and DAG array on both
    assert line["scene_tiles_built_per_rank"] == [line["scene_tiles"] // 2] * 2 and line["scene_same_top_level_on_every_rank"] is True
    assert "rank 0: scene_build_s" in p.stderr and "rank 1: scene_build_s" in p.stderr
    # slowest rank's time (2 ms) over the rays of both ranks
    assert abs(line["value"] - 2000 / 0.002 / 1e6) < 1e-6


def test_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1]


def test_one_gpu_default_starts_no_children():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert _json_line(p.stdout)["n_gpus"] == 1


def test_a_failing_rank_fails_the_run():
    """A rank that gives up (a sharded record that errored or hung: bench.py exits 3 there) must not read as a green run."""
    env = dict(_clean_env(), PANDORA_BENCH_DRY_FAIL_RANK="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 3, (p.returncode, p.stderr[-2000:])
    assert _json_line(p.stdout)["n_gpus"] == 2  # the line still went out
