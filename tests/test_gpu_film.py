"""GPU parity: pgpu_film_add / pgpu_film_resolve through the C ABI against the CPU restatement (bit-exact)."""
import numpy as np
import pytest
import torch

import oracle_binding as ob
from test_film import contributions

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,n_pixels", [(1, 1), (5000, 7), (300000, 4096), (2000000, 1 << 20)])
def test_film_matches_oracle_and_ignores_order(gpu_ctx, n, n_pixels):
    pixel, rgb = contributions(n, n_pixels, 11 + n)
    want = np.zeros((n_pixels, 3), np.uint64)
    ob.film_add(want, pixel, rgb[0], rgb[1], rgb[2])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    for order in (np.arange(n), np.random.default_rng(5).permutation(n)):
        film = torch.zeros((n_pixels, 3), dtype=torch.int64, device=dev)
        p = torch.from_numpy(pixel[order].view(np.int32).copy()).to(dev)
        r, g, b = (torch.from_numpy(np.ascontiguousarray(rgb[k][order])).to(dev) for k in range(3))
        half = n // 2  # two calls accumulate into the same film
        gpu_ctx.film_add(film, p[:half], r[:half], g[:half], b[:half], stream=stream)
        gpu_ctx.film_add(film, p[half:], r[half:], g[half:], b[half:], stream=stream)
        out = torch.empty((n_pixels, 3), dtype=torch.float32, device=dev)
        gpu_ctx.film_resolve(film, out, stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(film.cpu().numpy().view(np.uint64), want)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), ob.film_resolve(want).view(np.uint32))


def test_film_rejects_bad_arguments(gpu_ctx):
    import pandora_amd as pa
    film = torch.zeros((4, 3), dtype=torch.int64, device="cuda")
    with pytest.raises(pa.capi.PandoraError):
        gpu_ctx._L.pgpu_film_add.restype  # (bound)
        pa.accel._check(gpu_ctx._L.pgpu_film_add(gpu_ctx._h, 0, 4, 0, 0, 0, 0, 3, 0), gpu_ctx._h)  # null film
