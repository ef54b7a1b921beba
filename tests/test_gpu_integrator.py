"""GPU parity of the benchmark integrator's ray sources (pgpu_camera_rays, pgpu_lambert_bounce, pgpu_splat_misses; SURVEY.md
8f N4) against the CPU oracle's restatement. Bit-exact: camera rays, path state (rng streams, pixels), bounce origins, the
film given the same contributions. Tolerance 1e-6 absolute: bounce directions and the weights derived from them -- the disk
mapping calls cos / sin, and the device's differ from glibc's cosf / sinf (which the reference calls) by up to 2 ulp."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu
DIR_TOL = 1e-6


def _np(t):
    return t.cpu().numpy()


def _bits(a):
    return (a if isinstance(a, np.ndarray) else _np(a)).view(np.uint32)


def _check_bounce(got_rays, got_paths, want_rays, want_paths, label):
    g = _np(got_rays)
    # origins, tnear, tfar: no transcendental involved
    for row in (0, 1, 2, 6, 7):
        assert np.array_equal(g[row].view(np.uint32), want_rays[row].view(np.uint32)), f"{label}: ray row {row}"
    assert np.array_equal(np.isnan(g[3:6]), np.isnan(want_rays[3:6])), f"{label}: NaN directions (degenerate BSDF frame) differ"
    # |d - d'| <= 1e-6 -- except near the horizon, where z = sqrt(1 - x^2 - y^2) turns a 1-ulp change of x into a change of
    # x dx / z: bounded by 1e-4 there (z >= ~3e-3 for any x, y that are binary32 numbers)
    err = np.nan_to_num(np.abs(g[3:6] - want_rays[3:6])).max(axis=0)
    assert np.percentile(err, 99.0) <= DIR_TOL and err.max() <= 1e-4, f"{label}: directions differ by up to {err.max():.3g} (99th percentile {np.percentile(err, 99.0):.3g})"
    for k in ("rng_state", "rng_inc"):
        assert np.array_equal(_np(got_paths[k]).view(np.uint64), want_paths[k]), f"{label}: {k}"
    assert np.array_equal(_np(got_paths["pixel"]).view(np.uint32), want_paths["pixel"]), label
    for k in ("wr", "wg", "wb"):
        assert np.allclose(_np(got_paths[k]), want_paths[k], rtol=2e-4, atol=0, equal_nan=True), f"{label}: {k}"


def test_camera_bounce_splat_single_level():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    dev = torch.device("cuda", 0)
    mesh = pa.scenes.g_small()
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    orc = ob.OracleBvh8.from_host(hb, mesh)
    cam = pa.accel.make_camera((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 4 / 3, lens_radius=0.02, focal_distance=3.0)
    w, h, spp, seed = 160, 120, 2, 891379
    n = w * h * spp
    ctx = pa.accel.Context(0)
    try:
        ctx.set_keep_shading(True)
        bvh = ctx.upload_bvh8(hb, ctx.upload_mesh(mesh.pos, mesh.idx))
        # camera rays in two chunks (a renderer spawns them in batches): bit-exact, state included
        want_rays, want_paths = ob.camera_rays(cam, w, h, spp, seed, 0, n)
        cut = 7001
        r1, p1 = ctx.camera_rays(cam, w, h, spp, seed, 0, cut, dev)
        r2, p2 = ctx.camera_rays(cam, w, h, spp, seed, cut, n - cut, dev)
        rays = torch.cat([r1, r2], dim=1).contiguous()
        paths = {k: torch.cat([p1[k], p2[k]]).contiguous() for k in p1}
        assert np.array_equal(_bits(rays), want_rays.view(np.uint32))
        for k in ("rng_state", "rng_inc"):
            assert np.array_equal(_np(paths[k]).view(np.uint64), want_paths[k]), k
        assert np.array_equal(_np(paths["pixel"]).view(np.uint32), want_paths["pixel"])
        # trace, bounce, trace, splat -- two bounces deep, the oracle fed with the oracle's own rays at every step
        film = torch.zeros((w * h, 3), dtype=torch.int64, device=dev)
        want_film = np.zeros((w * h, 3), np.uint64)
        sky, albedo = (0.9, 1.0, 1.3), (0.7, 0.5, 0.3)
        for depth in range(3):
            hits = ctx.trace_bvh8(bvh, rays)
            want_hits = orc.trace(want_rays, threads=8)
            same = _bits(hits["obj"]) == want_hits["obj"]
            assert same.mean() > 0.9995 or depth == 0 and same.all(), f"depth {depth}: hits differ on {(~same).sum()} rays"
            ctx.splat_misses(film, hits["obj"], paths, sky)
            # the film is exact for the contributions it is given: the oracle splats the GPU's own path weights
            gp = {k: _np(v).view(np.uint64 if k.startswith("rng") else (np.uint32 if k == "pixel" else np.float32)).copy() for k, v in paths.items()}
            ob.splat_misses(want_film, _bits(hits["obj"]), gp, sky)
            ctx.sync()
            assert np.array_equal(_np(film).view(np.uint64), want_film), f"depth {depth}: film"
            if depth == 2:
                break
            rays_next = ctx.lambert_bounce(bvh, rays, hits, albedo, paths)
            want_next = ob.lambert_bounce(orc, None, want_rays, want_hits, albedo, want_paths)
            ctx.sync()
            if depth == 0:
                _check_bounce(rays_next, paths, want_next, want_paths, "first bounce")
                hit = want_hits["obj"] != ob.MISS
                first_miss = ~hit
                assert hit.mean() > 0.8 and np.array_equal(_np(rays_next)[6] > _np(rays_next)[7], ~hit)
            # an ended path (miss, failed bounce) keeps its slot but not its weight: it adds nothing at the later depths
            dead = _np(rays_next)[6] > _np(rays_next)[7]
            assert dead.any() and all((_np(paths[k])[dead] == 0.0).all() for k in ("wr", "wg", "wb"))
            rays, want_rays = rays_next, want_next
        assert np.count_nonzero(want_film) > 100
        # a pixel whose samples all missed at depth 0 holds spp x sky exactly once each (path_integrator.cpp:113-127), not once per depth
        both = first_miss.reshape(-1, spp).all(axis=1)
        assert both.any() and np.allclose(ob.film_resolve(_np(film).view(np.uint64))[both], np.float32(spp) * np.array(sky, np.float32), rtol=1e-6)
    finally:
        ctx.close()


def test_bounce_from_two_level_hits_and_missing_shading_data():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    dev = torch.device("cuda", 0)
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=16)
    scene = ob.scene_from_two_level(tl, mesh)
    cam = pa.accel.make_camera((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 1.0)
    n = 128 * 128
    ctx = pa.accel.Context(0)
    try:
        mid = ctx.upload_mesh(mesh.pos, mesh.idx)
        tl.upload(ctx, mid)  # no shading data kept: the bounce must refuse
        rays, paths = ctx.camera_rays(cam, 128, 128, 1, 5, 0, n, dev)
        hits, _ = ctx.trace(rays)
        with pytest.raises(pa.capi.PandoraError) as e:
            ctx.lambert_bounce(None, rays, hits, (0.5, 0.5, 0.5), paths)
        assert e.value.code == pa.capi.ERR_NO_SCENE
    finally:
        ctx.close()
    ctx = pa.accel.Context(0)
    try:
        ctx.set_keep_shading(True)
        tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))
        rays, paths = ctx.camera_rays(cam, 128, 128, 1, 5, 0, n, dev)
        want_rays, want_paths = ob.camera_rays(cam, 128, 128, 1, 5, 0, n)
        hits, _ = ctx.trace(rays)
        want_hits = scene.trace(want_rays, threads=8)
        for k in ("t", "obj", "leaf", "u", "v"):
            assert np.array_equal(_bits(hits[k]), want_hits[k].view(np.uint32)), k
        got = ctx.lambert_bounce(None, rays, hits, (0.8, 0.8, 0.8), paths)
        want = ob.lambert_bounce(None, scene, want_rays, want_hits, (0.8, 0.8, 0.8), want_paths)
        ctx.sync()
        _check_bounce(got, paths, want, want_paths, "two-level bounce")
    finally:
        ctx.close()


def test_compacted_paths_are_the_live_ones_in_order_and_the_film_is_the_same():
    """pgpu_compact_paths (the end of a path, path_integrator.cpp:109-127): after a bounce, the rays with tnear <= tfar and their
    path states at the front of new arrays, in their order; a frame traced with the dead slots dropped after every bounce gives
    the film of a frame that keeps them, integer for integer (a dead slot adds nothing), both layouts of the next trace included."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    dev = torch.device("cuda", 0)
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=16)
    cam = pa.accel.make_camera((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 1.0)
    side, spp = 96, 2
    n = side * side * spp
    sky, albedo = (0.9, 1.0, 1.3), (0.7, 0.5, 0.3)
    ctx = pa.accel.Context(0)
    try:
        ctx.set_keep_shading(True)
        tl.upload(ctx, ctx.upload_mesh(mesh.pos, mesh.idx))

        def frame(compact):
            film = torch.zeros((side * side, 3), dtype=torch.int64, device=dev)
            rays, paths = ctx.camera_rays(cam, side, side, spp, 11, 0, n, dev)
            traced = []
            for depth in range(4):
                hits, _ = ctx.trace(rays)
                ctx.splat_misses(film, hits["obj"], paths, sky)
                traced.append(int(rays.shape[1]))
                if depth == 3 or rays.shape[1] == 0:
                    break
                rays = ctx.lambert_bounce(None, rays, hits, albedo, paths)
                if compact:
                    ctx.sync()
                    before = _np(rays).copy()
                    state = {k: _np(v).copy() for k, v in paths.items()}
                    rays, paths = ctx.compact_paths(rays, paths)
                    live = before[6] <= before[7]
                    assert rays.shape[1] == int(live.sum())
                    assert np.array_equal(_np(rays).view(np.uint32), before[:, live].view(np.uint32)), f"depth {depth}: rays"
                    for k, v in state.items():
                        assert np.array_equal(_np(paths[k]).view(np.uint8), np.ascontiguousarray(v[live]).view(np.uint8)), f"depth {depth}: {k}"  # (bit patterns: a weight may be NaN)
            ctx.sync()
            return _np(film).copy(), traced

        kept, slots = frame(False)
        dropped, live = frame(True)
        assert slots == [n] * 4 and live[0] == n and live[1] < n and live[3] < live[1]
        assert np.array_equal(kept, dropped) and np.count_nonzero(kept) > 100
        with pytest.raises(pa.capi.PandoraError):  # in place is refused
            rays, paths = ctx.camera_rays(cam, side, side, spp, 11, 0, n, dev)
            r, _ = ctx._rays_struct(rays)
            import ctypes as C
            k = C.c_size_t(0)
            pa.accel._check(ctx._L.pgpu_compact_paths(ctx._h, C.byref(r), C.byref(ctx._paths_struct(paths)), n, C.byref(r), C.byref(ctx._paths_struct(paths)), C.byref(k), 0), ctx._h)
    finally:
        ctx.close()
