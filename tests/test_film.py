"""Film accumulation: the CPU restatement of Sensor::addPixelContribution against exact integer
arithmetic, its order independence, and the integer all-reduce over ranks (gloo, world size 2). The fixed-point type
is cnl's (not in the container): parity unpinned, the restatement follows its definition (scale by 2^24, truncate)."""
import os
import socket
import tempfile

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_binding as ob
from pandora_amd import sharded


def contributions(n, n_pixels, seed):
    rng = np.random.default_rng(seed)
    pixel = rng.integers(0, n_pixels, n).astype(np.uint32)
    rgb = (rng.random((3, n), np.float32) * rng.choice(np.float32([0.001, 1.0, 37.5, 1e4]), (3, n))).astype(np.float32)
    hostile = rng.random(n)
    rgb[0, hostile < 0.01] = np.nan
    rgb[1, (hostile > 0.01) & (hostile < 0.02)] = np.inf
    rgb[2, (hostile > 0.02) & (hostile < 0.03)] = -np.inf
    rgb[0, (hostile > 0.03) & (hostile < 0.04)] = 6e11        # beyond the 40.24 range
    rgb[1, (hostile > 0.04) & (hostile < 0.06)] *= -1.0       # negative: wraps, i.e. subtracts
    pixel[(hostile > 0.06) & (hostile < 0.07)] = n_pixels + 5  # outside the film
    return pixel, rgb


def exact_film(pixel, rgb, n_pixels):
    """Python-integer model of the rule: drop non-finite / out-of-range, else film += trunc(c * 2^24) mod 2^64."""
    film = [[0, 0, 0] for _ in range(n_pixels)]
    for i in range(pixel.size):
        c = [float(rgb[k, i]) for k in range(3)]
        if pixel[i] >= n_pixels or not all(np.isfinite(x) and abs(x) < 2.0 ** 39 for x in c):
            continue
        for k in range(3):
            film[pixel[i]][k] = (film[pixel[i]][k] + int(np.float32(c[k]) * np.float32(16777216.0))) % (1 << 64)
    return np.array(film, dtype=np.uint64)


def test_oracle_film_matches_exact_integers_and_ignores_order():
    n_pixels = 97
    pixel, rgb = contributions(6000, n_pixels, 1)
    film = np.zeros((n_pixels, 3), np.uint64)
    ob.film_add(film, pixel, rgb[0], rgb[1], rgb[2])
    assert np.array_equal(film, exact_film(pixel, rgb, n_pixels))
    perm = np.random.default_rng(2).permutation(pixel.size)
    shuffled = np.zeros_like(film)
    ob.film_add(shuffled, pixel[perm], rgb[0][perm], rgb[1][perm], rgb[2][perm])
    assert np.array_equal(film, shuffled)
    out = ob.film_resolve(film)
    assert np.array_equal(out, (film.astype(np.float32) * np.float32(2.0 ** -24)))
    assert out[film < (1 << 62)].max() > 0


def test_known_values():
    film = np.zeros((2, 3), np.uint64)
    ob.film_add(film, np.uint32([0, 0, 1, 1]), np.float32([1.0, 0.5, 2.0, np.nan]), np.float32([0.25, 0.25, 0.0, 1.0]), np.float32([0.0, 2.0 ** -24, 3.0, 1.0]))
    assert film.tolist() == [[(1 << 24) + (1 << 23), 1 << 23, 1], [2 << 24, 0, 3 << 24]]
    assert ob.film_resolve(film).tolist() == [[1.5, 0.5, 2.0 ** -24], [2.0, 0.0, 3.0]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_pixels = 64
        pixel, rgb = contributions(5000, n_pixels, 9)
        mine = slice(rank, None, world)  # every rank accumulates its share of the contributions
        film = np.zeros((n_pixels, 3), np.uint64)
        ob.film_add(film, pixel[mine], rgb[0][mine], rgb[1][mine], rgb[2][mine])
        t = torch.from_numpy(film.view(np.int64).copy())
        sharded.film_allreduce(t)
        np.save(os.path.join(out_dir, f"film{rank}.npy"), t.numpy().view(np.uint64))
    finally:
        dist.destroy_process_group()


def test_film_allreduce_world2_equals_one_process():
    n_pixels = 64
    pixel, rgb = contributions(5000, n_pixels, 9)
    want = np.zeros((n_pixels, 3), np.uint64)
    ob.film_add(want, pixel, rgb[0], rgb[1], rgb[2])
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        for r in range(2):
            assert np.array_equal(np.load(os.path.join(d, f"film{r}.npy")), want)
