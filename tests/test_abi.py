"""The C-ABI library loads without a GPU and exports every symbol include/pandora_hip.h declares;
host-side entry points report errors by code (never by exception)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

import pandora_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pandora_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgpu_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared_symbols()
    assert len(names) >= 30
    lib = C.CDLL(pa.capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/pandora_hip.h but not exported"
    assert set(names) == set(pa.capi.SYMBOLS), set(names) ^ set(pa.capi.SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", pa.capi.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\b(pgpu_[a-z0-9_]+)\b", out))
    assert set(names) <= exported


def test_library_carries_gfx950_code_objects():
    data = open(pa.capi.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"bvh8TraceKernel" in data and b"topTraverseKernel" in data


def test_host_entry_points_return_codes():
    L = pa.capi.lib()
    assert L.pgpu_version().startswith(b"pandora_hip")
    h = C.c_void_p()
    assert L.pgpu_build_bvh8(None, 0, None, 0, None, 0, 4, C.byref(h)) == pa.capi.ERR_INVALID
    b = np.zeros((1, 6), np.float32)
    assert L.pgpu_build_top_bvh4(b.ctypes.data, 1, C.byref(h)) == pa.capi.ERR_SINGLE_LEAF
    assert L.pgpu_last_error(None) == b"null context"
    assert L.pgpu_set_stats(None, 1) == pa.capi.ERR_INVALID
    pos = np.zeros((3, 3), np.float32)
    idx = np.array([[0, 1, 7]], np.uint32)  # vertex index out of range
    g = np.zeros(1, np.uint32)
    ng = C.c_uint32()
    assert L.pgpu_partition_prims(pos.ctypes.data, 3, idx.ctypes.data, 1, 4, g.ctypes.data, C.byref(ng)) == pa.capi.ERR_INVALID


def test_init_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    assert pa.capi.lib().pgpu_init(0, C.byref(h)) == pa.capi.ERR_NO_DEVICE
    try:
        pa.accel.Context(0)
    except pa.capi.PandoraError as e:
        assert e.code == pa.capi.ERR_NO_DEVICE
    else:
        raise AssertionError("Context() must not succeed without a GPU")


def test_product_never_touches_the_oracle():
    """The shipped package must not import, load or link anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pandora_amd")):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower().replace("the oracle proves", ""), os.path.join(dirpath, f)
    ldd = subprocess.run(["ldd", pa.capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "pandora_ref" not in ldd
