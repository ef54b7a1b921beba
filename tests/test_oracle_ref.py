"""Scalar oracle vs the restatement running on the reference's own simd headers (oracle/_ref),
fuzzed. Skipped where /root/reference (hence oracle/_ref) is absent; the golden files carry the
same comparison to such machines."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def _bits(a):
    return np.asarray(a).view(np.uint32)


def test_simd_primitives_fuzzed():
    rng = np.random.default_rng(0)
    o, r = ob.oracle(), ob.ref()
    a, b = np.zeros(8, np.uint32), np.zeros(8, np.uint32)
    for m in range(256):
        o.orc_compress_perm8(m, a.ctypes.data)
        r.ref_compress_perm8(m, b.ctypes.data)
        assert np.array_equal(a, b), m
    specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.4028235e38], np.float32)
    for _ in range(2000):
        v = rng.choice(specials, 4) if rng.random() < 0.3 else rng.normal(size=4).astype(np.float32)
        v = np.ascontiguousarray(v, np.float32)
        assert o.orc_hmin_index4(v.ctypes.data) == r.ref_hmin_index4(v.ctypes.data), v
    for m in list(range(16)):
        assert o.orc_bitscan4(m) == r.ref_bitscan4(m)
    for _ in range(500):
        x = int(rng.integers(1, 2**32))
        assert o.orc_bsr32(x) == r.ref_bsr32(x)


def test_inner_node_slab_test_including_nan_and_inf():
    rng = np.random.default_rng(1)
    o, r = ob.oracle(), ob.ref()
    for it in range(3000):
        node = np.zeros(64, np.float32)
        lo = rng.uniform(-2, 2, (3, 8)).astype(np.float32)
        hi = lo + rng.uniform(0, 1.5, (3, 8)).astype(np.float32)
        n_child = int(rng.integers(1, 9))
        lo[:, n_child:] = 0
        hi[:, n_child:] = 0
        for a in range(3):
            node[16 * a: 16 * a + 8] = lo[a]
            node[16 * a + 8: 16 * a + 16] = hi[a]
        kids = rng.integers(1, 2**31, 8).astype(np.uint32)
        perm = np.zeros(8, np.uint32)
        lo_t, hi_t = np.ascontiguousarray(lo.T), np.ascontiguousarray(hi.T)
        o.orc_bvh8_perm(lo_t.ctypes.data, hi_t.ctypes.data, n_child, perm.ctypes.data)
        node.view(np.uint32)[48:56] = kids
        node.view(np.uint32)[56:64] = perm
        org = rng.uniform(-3, 3, 3).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        if it % 3 == 0:  # exact zeros in the direction and origins exactly on slab planes -> 0 * inf
            d[rng.integers(0, 3)] = 0.0
            ax = int(rng.integers(0, 3))
            org[ax] = lo[ax, int(rng.integers(0, n_child))]
        if it % 7 == 0:
            d[rng.integers(0, 3)] = -0.0
        ray = ob.make_ray(org, d, 0.0, float(rng.choice([3.4028235e38, 2.0, 0.5])))
        ck, cd, rk, rd = (np.zeros(8, t) for t in (np.uint32, np.float32, np.uint32, np.float32))
        n1 = o.orc_bvh8_inner(node.ctypes.data, C.byref(ray), ck.ctypes.data, cd.ctypes.data)
        n2 = r.ref_bvh8_inner(node.ctypes.data, C.byref(ray), rk.ctypes.data, rd.ctypes.data)
        assert n1 == n2
        assert np.array_equal(ck[:n1], rk[:n1]) and np.array_equal(_bits(cd[:n1]), _bits(rd[:n1]))


def test_traversal_on_fresh_trees():
    mesh = pa.scenes.merge([pa.scenes.g_small(), pa.scenes.icosphere((0.2, 1.4, -0.3), 0.4, 3)])
    for max_leaf in (1, 2, 4):
        hb = pa.accel.build_bvh8(mesh.pos, mesh.idx, max_leaf=max_leaf)
        orc = ob.OracleBvh8.from_host(hb, mesh)
        rays = pa.scenes.random_rays((-2, -1, -2), (2, 3, 2), 3000, seed=10 + max_leaf)
        for tri in (0, 1):
            a, b = orc.trace(rays, tri_mode=tri), orc.trace(rays, tri_mode=tri, use_ref=True)
            for k in ("t", "obj", "u", "v"):
                assert np.array_equal(_bits(a[k]), _bits(b[k])), (max_leaf, tri, k)
            assert np.array_equal(orc.trace(rays, any_hit=True, tri_mode=tri)["hit"], orc.trace(rays, any_hit=True, tri_mode=tri, use_ref=True)["hit"])


def test_two_level_tokens_step_by_step():
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 40, svdag_res=16)
    sc = ob.scene_from_two_level(tl, mesh)
    rays = pa.scenes.random_rays((-2, -1, -2), (2, 3, 2), 400, seed=21)
    o, r = ob.oracle(), ob.ref()
    for any_hit in (0, 1):
        for i in range(rays.shape[1]):
            ray = ob.make_ray(rays[0:3, i], rays[3:6, i])
            n1, s1, l1 = C.c_uint32(sc.c.root), C.c_uint64(2**64 - 1), C.c_uint32(0)
            n2, s2, l2 = C.c_uint32(sc.c.root), C.c_uint64(2**64 - 1), C.c_uint32(0)
            for _ in range(64):
                a = o.orc_bvh4_step(C.byref(sc.c), any_hit, 4, C.byref(ray), C.byref(n1), C.byref(s1), C.byref(l1), None)
                b = r.ref_bvh4_step(C.byref(sc.c), any_hit, C.byref(ray), C.byref(n2), C.byref(s2), C.byref(l2), None)
                assert a == b and n1.value == n2.value and s1.value == s2.value
                if not a:
                    break
                assert l1.value == l2.value


def test_svdag_three_lane_cast_equals_reference_four_lane_cast():
    mesh = pa.scenes.merge([pa.scenes.icosphere((0, 0, 0), 1.0, 3), pa.scenes.box((0.5, -0.2, -0.2), (1.8, 0.2, 0.2), 2)])
    dags = pa.accel.build_svdags(mesh.pos, mesh.idx, [np.arange(mesh.n_tris, dtype=np.uint32)], 64)
    lo, hi = mesh.bounds()
    ext = float((hi - lo).max())
    ctr = (lo + hi) / 2
    total = 0
    for i, s in enumerate((1.1, 4.0, 50.0, 1000.0)):
        rays = pa.scenes.random_rays(ctr - s * ext, ctr + s * ext, 4000, seed=30 + i, toward=(lo - 0.2 * ext, hi + 0.2 * ext))
        rays[3 + (i % 3), ::13] = 0.0
        want = ob.svdag_trace(dags.words, dags.descs[0], rays, use_ref=True)
        for lanes in (4, 3):
            got = ob.svdag_trace(dags.words, dags.descs[0], rays, lanes=lanes)
            assert np.array_equal(got, want), (s, lanes, int((got != want).sum()))
        total += int(want.sum())
    assert total > 1000
