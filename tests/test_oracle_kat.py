"""Oracle pinned by the reference's own known-answer material (no GPU, no reference needed):
the simd tests' values (permute / compress / count / horizontalMinIndex) and the triangle test's rays."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_binding as ob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def kat():
    with open(os.path.join(GOLDEN, "simd_kat.json")) as f:
        return json.load(f)


def _mask(bools):
    return sum(1 << i for i, b in enumerate(bools) if b)


def test_compress_permutation_matches_reference_simd_tests(kat):
    k = kat["simd8"]
    m = _mask(k["compress"]["mask"])
    perm = np.zeros(8, np.uint32)
    ob.oracle().orc_compress_perm8(m, perm.ctypes.data)
    v2 = np.array(k["v2"])
    assert list(v2[perm][: k["compress"]["count"]]) == k["compress"]["expect_head"]
    assert bin(m).count("1") == k["compress"]["count"]
    assert bin(m & k["compress"]["valid_mask"]).count("1") == k["compress"]["count_valid"]
    assert _mask(k["mask_bits"]["mask"]) == k["mask_bits"]["bitmask"]
    # unselected tail lanes read index 0 (simd8_avx2.ipp:1-17)
    assert list(perm[k["compress"]["count"]:]) == [0] * (8 - k["compress"]["count"])


def test_horizontal_min_index_matches_reference_simd_tests(kat):
    v = np.array(kat["simd4"]["hmin"]["values"], np.float32)
    assert ob.oracle().orc_hmin_index4(v.ctypes.data) == kat["simd4"]["hmin"]["min_index"]
    # ties resolve to the lowest lane; no lane equal (all NaN) falls through to 3 (intrinsics.h:22-34)
    assert ob.oracle().orc_hmin_index4(np.array([2, 1, 1, 5], np.float32).ctypes.data) == 1
    assert ob.oracle().orc_hmin_index4(np.array([np.nan] * 4, np.float32).ctypes.data) == 3


def test_bitscans():
    o = ob.oracle()
    assert [o.orc_bitscan4(m) for m in (1, 2, 4, 8, 6, 12, 0)] == [0, 1, 2, 3, 1, 2, 3]
    for b in range(32):
        assert o.orc_bsr32(1 << b) == b
        assert o.orc_bsr32((1 << b) | 1) == b


def test_sign_shift_zero_counts_as_negative():
    o = ob.oracle()
    assert o.orc_sign_shift(1.0, 1.0, 1.0) == 21
    assert o.orc_sign_shift(0.0, -0.0, 1.0) == 12
    assert o.orc_sign_shift(-1.0, 2.0, -3.0) == 6


@pytest.mark.skipif(not ob.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_reference_simd_library_gives_the_same_known_answers(kat):
    r = ob.ref()
    k = kat["simd8"]
    v2 = np.array(k["v2"], np.float32)
    out = np.zeros(8, np.float32)
    r.ref_permute8(v2.ctypes.data, np.array(k["permute"]["index"], np.uint32).ctypes.data, out.ctypes.data)
    assert list(out) == k["permute"]["expect"]
    r.ref_compress8(v2.ctypes.data, _mask(k["compress"]["mask"]), out.ctypes.data)
    assert list(out[:4]) == k["compress"]["expect_head"]
    assert r.ref_hmin_index8(np.array(k["hmin"]["values"], np.float32).ctypes.data) == k["hmin"]["min_index"]
    assert r.ref_hmin_index4(np.array(kat["simd4"]["hmin"]["values"], np.float32).ctypes.data) == kat["simd4"]["hmin"]["min_index"]


@pytest.mark.parametrize("mode", [0, 1], ids=["moller_trumbore", "watertight"])
def test_triangle_known_answer_rays(mode):
    with open(os.path.join(GOLDEN, "triangle_kat.json")) as f:
        k = json.load(f)
    o = ob.oracle()
    p = np.array(k["intersect1"]["positions"], np.float32)
    for ray in k["intersect1"]["rays"]:
        org, d = np.array(ray["o"], np.float32), np.array(ray["d"], np.float32)
        tfar, u, v = C.c_float(3.4028234663852886e38), C.c_float(), C.c_float()
        hit = o.orc_tri_intersect(mode, org.ctypes.data, d.ctypes.data, 0.0, C.byref(tfar), p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data, C.byref(u), C.byref(v))
        assert bool(hit) == ray["hit"], ray
    a = k["area_scanline"]
    p = np.array(a["positions"], np.float32)
    hits = total = 0
    x = np.float32(a["x_start"])
    while x < np.float32(a["x_end"]):
        org, d = np.array([x, a["o_y"], a["scanline_z"]], np.float32), np.array(a["d"], np.float32)
        tfar, u, v = C.c_float(3.4028234663852886e38), C.c_float(), C.c_float()
        hits += o.orc_tri_intersect(mode, org.ctypes.data, d.ctypes.data, 0.0, C.byref(tfar), p[0].ctypes.data, p[1].ctypes.data, p[2].ctypes.data, C.byref(u), C.byref(v))
        total += 1
        x = np.float32(x + np.float32(a["x_step"]))
    assert hits == total // 2
