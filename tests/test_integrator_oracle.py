"""CPU checks of the oracle's restatement of the benchmark integrator's ray sources: its PCG32 against the
vectorised numpy one of pandora_amd.scenes (two independent restatements of samplers/rng/pcg.cpp:23-61), camera rays against
plain geometry, bounce rays against the properties a cosine-weighted Lambert bounce must have."""
import numpy as np

import oracle_binding as ob
import pandora_amd as pa


def _scene():
    mesh = pa.scenes.g_small()
    hb = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    return mesh, hb, ob.OracleBvh8.from_host(hb, mesh)


def test_pcg_streams_match_the_numpy_restatement():
    cam = pa.accel.make_camera((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 1.0)
    w = h = 32
    seed, spp = 891379, 2
    n = w * h * spp
    rays, paths = ob.camera_rays(cam, w, h, spp, seed, 0, n)
    rng = pa.scenes.Pcg32(np.arange(n, dtype=np.uint64) + np.uint64(seed))
    r0, r1 = rng.f32(), rng.f32()
    assert np.array_equal(paths["rng_state"], rng.state) and np.array_equal(paths["rng_inc"], rng.inc)
    pix = np.arange(n) // spp
    assert np.array_equal(paths["pixel"], pix.astype(np.uint32))
    # the jitter really is the first two floats of stream seed + i: undo the camera mapping
    cx = ((pix % w).astype(np.float32) + r0) / np.float32(w) - np.float32(0.5)
    cy = ((pix // w).astype(np.float32) + r1) / np.float32(h) - np.float32(0.5)
    sc = np.array(cam.screen_center)[None] + cx[:, None] * np.array(cam.screen_du)[None] + cy[:, None] * np.array(cam.screen_dv)[None]
    d = sc - np.array(cam.lens_center)[None]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    assert np.allclose(rays[3:6].T, d, atol=2e-7)
    assert np.array_equal(rays[0:3].T, np.broadcast_to(np.array(cam.lens_center, np.float32), (n, 3)))
    assert (rays[6] == 0).all() and (rays[7] == np.float32(3.4028234663852886e38)).all()


def test_bounce_rays_leave_the_surface_cosine_weighted():
    mesh, hb, orc = _scene()
    cam = pa.accel.make_camera((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 1.0)
    rays, paths = ob.camera_rays(cam, 96, 96, 1, 7, 0, 96 * 96)
    hits = orc.trace(rays, threads=4)
    before = {k: v.copy() for k, v in paths.items()}
    # miss splat (before the bounce, as a renderer does: path_integrator.cpp:113-127): exactly the rays that missed contribute weight * sky
    film = np.zeros((96 * 96, 3), np.uint64)
    ob.splat_misses(film, hits["obj"], paths, (1.0, 2.0, 3.0))
    out = ob.lambert_bounce(orc, None, rays, hits, (0.5, 0.6, 0.7), paths)
    hit = hits["obj"] != ob.MISS
    dead = out[6] > out[7]
    assert np.array_equal(dead, ~hit) and hit.mean() > 0.9
    o, d = out[0:3, hit].T, out[3:6, hit].T
    p = rays[0:3, hit].T + hits["t"][hit, None] * rays[3:6, hit].T
    assert np.abs(o - p).max() < 1e-4                      # starts at the hit point (barycentric vs o + t d) ...
    n = pa.scenes.geometric_normals(mesh, hb.prim_ids[hits["obj"][hit]])
    n[np.einsum("ij,ij->i", n, -rays[3:6, hit].T) < 0] *= -1
    # The reference's BSDF frame (material.cpp:5-20) picks (1, 0, 0) as the helper vector unless n.x > 0.5 -- for a surface
    # facing -x the helper is parallel to the normal, the tangent is normalize(0) and the bounce direction is NaN. Restated
    # as it is: such rays miss everything. Everywhere else the bounce is a proper one.
    degenerate = np.isnan(d).any(axis=1)
    assert np.array_equal(degenerate, n[:, 0] < -0.999999) and 0 < degenerate.mean() < 0.3
    ok = ~degenerate
    cos = np.einsum("ij,ij->i", n[ok], d[ok])
    assert (cos > -1e-6).all() and abs(np.linalg.norm(d[ok], axis=1) - 1).max() < 1e-5   # ... into the hemisphere of the normal
    assert abs(cos.mean() - 2.0 / 3.0) < 0.02              # E[cos] of a cosine-weighted hemisphere
    assert np.allclose(np.einsum("ij,ij->i", o - p, n)[ok], 5e-6, atol=2e-6)  # RAY_EPSILON along the normal
    # Lambert: weight = previous * albedo (f |cos| / pdf = albedo), and every path got a fresh stream
    live = np.nonzero(hit)[0][ok]
    assert np.allclose(paths["wr"][live], 0.5, rtol=1e-5) and np.allclose(paths["wb"][live], 0.7, rtol=1e-5)
    # a path that missed has ended: its weight is gone, so the miss of its dead ray adds nothing at the next depth
    assert (paths["rng_state"][hit] != before["rng_state"][hit]).all() and (paths["wr"][~hit] == 0.0).all() and (paths["wb"][~hit] == 0.0).all()
    got = ob.film_resolve(film)
    assert np.array_equal(got[:, 0] != 0, ~hit) and np.allclose(got[~hit], [1.0, 2.0, 3.0])


def test_a_terminated_path_contributes_once():
    """Three bounces against a host loop that DROPS ended paths (path_integrator.cpp:109-127: a miss adds weight * sky once and
    the path is over; a failed bounce adds nothing): the batch form, which keeps a dead slot per ended path, must give the same film."""
    mesh, hb, orc = _scene()
    cam = pa.accel.make_camera((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 1.0)
    w = h = 64
    n = w * h
    sky, albedo = (0.9, 1.0, 1.3), (0.7, 0.5, 0.3)
    rays, paths = ob.camera_rays(cam, w, h, 1, 11, 0, n)
    film = np.zeros((n, 3), np.uint64)
    # host loop over compacted live paths, contributions gathered per pixel in exact integers (40.24 fixed point)
    want = np.zeros((n, 3), np.uint64)
    live = np.arange(n)
    l_rays, l_paths = rays.copy(), {k: v.copy() for k, v in paths.items()}
    once = np.zeros(n, np.int64)
    for depth in range(4):
        hits = orc.trace(rays, threads=4)
        ob.splat_misses(film, hits["obj"], paths, sky)
        l_hits = orc.trace(l_rays, threads=4)
        miss = l_hits["obj"] == ob.MISS
        contrib = np.stack([(l_paths["w" + "rgb"[c]][miss] * np.float32(s)).astype(np.float32) for c, s in enumerate(sky)], axis=1)
        ok = (np.abs(contrib) < 549755813888.0).all(axis=1)  # sensor.cpp:41-42: a contribution with a NaN / inf channel is dropped whole
        for c in range(3):
            np.add.at(want[:, c], l_paths["pixel"][miss][ok], (contrib[ok, c] * np.float32(16777216.0)).astype(np.int64).astype(np.uint64))
        once[l_paths["pixel"][miss]] += 1
        if depth == 3:
            break
        rays = ob.lambert_bounce(orc, None, rays, hits, albedo, paths)
        nxt = ob.lambert_bounce(orc, None, l_rays, l_hits, albedo, l_paths)
        keep = nxt[6] <= nxt[7]  # drop what ended: misses and failed bounces
        l_rays = np.ascontiguousarray(nxt[:, keep])
        l_paths = {k: v[keep].copy() for k, v in l_paths.items()}
        live = live[keep]
    assert once.max() == 1 and once.sum() > n // 20
    assert np.array_equal(film, want)
    # a camera ray that missed at depth 0 contributed exactly sky, not (depth + 1) x sky
    first = orc.trace(ob.camera_rays(cam, w, h, 1, 11, 0, n)[0], threads=4)["obj"] == ob.MISS
    assert first.any() and np.allclose(ob.film_resolve(film)[first], sky, rtol=1e-6)
