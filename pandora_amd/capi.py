"""ctypes binding of libpandora_hip.so (include/pandora_hip.h). Thin: pointers and sizes only.

The library must exist (run `python -c "import __graft_entry__ as g; g.build()"` or
`make -C pandora_amd`); there is no fallback path of any kind.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PANDORA_HIP_LIB") or os.path.join(_HERE, "lib", "libpandora_hip.so")  # (the override is for A/B runs of two builds)

OK, ERR_INVALID, ERR_HIP, ERR_SINGLE_LEAF, ERR_DEPTH, ERR_NO_SCENE, ERR_NO_DEVICE, ERR_PEER = 0, -1, -2, -3, -4, -5, -6, -7
CLOSEST, ANY = 0, 1
TRI_MT, TRI_WATERTIGHT = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
MISS = 0xFFFFFFFF
NO_SVDAG = 0xFFFFFFFF
NO_BVH = 0xFFFFFFFF
INSTANCE = 0xFFFFFFFF

c_f32p = C.POINTER(C.c_float)
c_u32p = C.POINTER(C.c_uint32)
c_u64p = C.POINTER(C.c_uint64)


class Rays(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "tnear", "tfar")]


class Hits(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("t", "obj", "leaf", "u", "v", "visits", "inst")]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("svdag_tests", "svdag_culled", "leaf_visits", "rounds", "inner_visits", "leaf_pops", "tri_tests", "leaf_loads", "streamed_bytes", "wave_steps", "leaf_passes", "refill_passes", "records_routed", "exchange_bytes")]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32), C.c_size_t, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


class Transport(C.Structure):
    _fields_ = [("user", C.c_void_p), ("allgather_u32", ALLGATHER_FN), ("exchange", EXCHANGE_FN), ("allreduce_u64", ALLREDUCE_FN)]


class Camera(C.Structure):
    """pgpu_camera: PerspectiveCamera's member vectors after updatePosition (perspective_camera.cpp:50-65)."""
    _fields_ = [(n, C.c_float * 3) for n in ("lens_center", "lens_du", "lens_dv", "screen_center", "screen_du", "screen_dv")]


class Paths(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rng_state", "rng_inc", "pixel", "wr", "wg", "wb")]


class LeafObj(C.Structure):
    _fields_ = [("mesh", C.c_uint32), ("prim", C.c_uint32)]


class InstanceDesc(C.Structure):
    _fields_ = [("bvh_id", C.c_uint32), ("world_to_local", C.c_float * 12)]


class Leaf(C.Structure):
    _fields_ = [("bvh_id", C.c_uint32), ("svdag", C.c_uint32)]


class SvdagDesc(C.Structure):
    _fields_ = [("root", C.c_uint32), ("bounds_min", C.c_float * 3), ("inv_extent", C.c_float), ("resolution", C.c_uint32)]


# every symbol include/pandora_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "pgpu_build_bvh8": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]),
    "pgpu_host_bvh8_nodes": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pgpu_host_bvh8_leaf_indices": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pgpu_host_bvh8_root": (C.c_uint32, [C.c_void_p]),
    "pgpu_host_bvh8_depth": (C.c_int, [C.c_void_p]),
    "pgpu_host_bvh8_free": (None, [C.c_void_p]),
    "pgpu_build_top_bvh4": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pgpu_host_bvh4_nodes": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pgpu_host_bvh4_leaf_order": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pgpu_host_bvh4_root": (C.c_uint32, [C.c_void_p]),
    "pgpu_host_bvh4_depth": (C.c_int, [C.c_void_p]),
    "pgpu_host_bvh4_free": (None, [C.c_void_p]),
    "pgpu_partition_prims": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint32)]),
    "pgpu_partition_objects": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32)]),
    "pgpu_build_svdags": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]),
    "pgpu_host_svdags_words": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pgpu_host_svdags_descs": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pgpu_host_svdags_words_before_compression": (C.c_size_t, [C.c_void_p]),
    "pgpu_host_svdags_free": (None, [C.c_void_p]),
    "pgpu_svdag_expand": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(SvdagDesc), C.c_void_p]),
    "pgpu_svdag_rebase": (C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32]),
    "pgpu_voxelize": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pgpu_bvh8_serialize": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "pgpu_bvh8_deserialize": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "pgpu_bytes_free": (None, [C.c_void_p]),
    "pgpu_init": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pgpu_shutdown": (None, [C.c_void_p]),
    "pgpu_last_error": (C.c_char_p, [C.c_void_p]),
    "pgpu_version": (C.c_char_p, []),
    "pgpu_upload_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "pgpu_upload_bvh8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "pgpu_upload_bvh8_instanced": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "pgpu_stage_bvh8_instanced": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "pgpu_build_bvh8_boxes": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "pgpu_stage_bvh8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]),
    "pgpu_set_leaf_cache": (C.c_int, [C.c_void_p, C.c_size_t]),
    "pgpu_set_leaf_copy": (C.c_int, [C.c_void_p, C.c_int]),
    "pgpu_set_leaf_hold": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pgpu_set_leaf_in_place": (C.c_int, [C.c_void_p, C.c_uint32]),
    "pgpu_set_svdag_policy": (C.c_int, [C.c_void_p, C.c_int]),
    "pgpu_upload_svdags": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "pgpu_upload_top_bvh4": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]),
    "pgpu_trace_bvh8": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(Rays), C.c_size_t, C.c_int, C.POINTER(Hits), C.c_void_p, C.c_void_p]),
    "pgpu_trace_bvh8_noalloc": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(Rays), C.c_size_t, C.POINTER(Hits), C.c_void_p, C.c_void_p]),
    "pgpu_trace": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Rays), C.c_size_t, C.c_int, C.POINTER(Hits), C.c_void_p, C.POINTER(Counters), C.c_void_p]),
    "pgpu_ray_state_init": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Hits), C.c_void_p, C.c_void_p]),
    "pgpu_top_episode": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(Rays), C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pgpu_trace_leaves": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Rays), C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Hits), C.c_void_p, C.c_void_p]),
    "pgpu_route_pack": (C.c_int, [C.c_void_p, C.POINTER(Rays), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgpu_trace_leaf_records": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "pgpu_route_merge": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Hits), C.c_void_p, C.c_void_p]),
    "pgpu_last_kernel_ms": (C.c_float, [C.c_void_p]),
    "pgpu_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgpu_host_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pgpu_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgpu_device_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "pgpu_device_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgpu_stream_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "pgpu_stream_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgpu_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pgpu_stream_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pgpu_set_stats": (C.c_int, [C.c_void_p, C.c_int]),
    "pgpu_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "pgpu_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "pgpu_set_kernel_variant": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pgpu_set_ray_layout": (C.c_int, [C.c_void_p, C.c_int]),
    "pgpu_mesh_serialize": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pgpu_mesh_deserialize": (C.c_int, [C.c_void_p, C.c_size_t] + [C.c_void_p] * 9),
    "pgpu_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pgpu_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pgpu_comm_init_transport": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Transport)]),
    "pgpu_comm_destroy": (C.c_int, [C.c_void_p]),
    "pgpu_comm_selftest": (C.c_int, [C.c_void_p]),
    "pgpu_set_leaf_owners": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "pgpu_trace_sharded": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(Rays), C.c_size_t, C.POINTER(Hits), C.c_void_p, C.POINTER(Counters), C.c_void_p]),
    "pgpu_film_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pgpu_set_keep_shading": (C.c_int, [C.c_void_p, C.c_int]),
    "pgpu_build_svdags_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]),
    "pgpu_build_bvh8_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p),
                                         C.POINTER(C.c_uint32)]),
    "pgpu_build_upload_bvh8_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p,
                                                C.POINTER(C.c_uint32)]),
    "pgpu_partition_prims_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint32)]),
    "pgpu_bvh8_info": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "pgpu_download_bvh8": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "pgpu_set_round_lookahead": (C.c_int, [C.c_void_p, C.c_int]),
    "pgpu_set_batch_parts": (C.c_int, [C.c_void_p, C.c_int]),
    "pgpu_set_top_variant": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pgpu_debug_fail_sharded": (C.c_int, [C.c_void_p, C.c_int]),
    "pgpu_set_svdag_defer": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "pgpu_camera_rays": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_size_t, C.POINTER(Rays), C.POINTER(Paths), C.c_void_p]),
    "pgpu_lambert_bounce": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(Rays), C.POINTER(Hits), C.c_size_t, C.POINTER(C.c_float), C.POINTER(Rays), C.POINTER(Paths), C.c_void_p]),
    "pgpu_splat_misses": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(Paths), C.c_size_t, C.POINTER(C.c_float), C.c_void_p]),
    "pgpu_compact_paths": (C.c_int, [C.c_void_p, C.POINTER(Rays), C.POINTER(Paths), C.c_size_t, C.POINTER(Rays), C.POINTER(Paths), C.POINTER(C.c_size_t), C.c_void_p]),
    "pgpu_film_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pgpu_film_resolve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C pandora_amd` (no fallback exists)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def ptr(a) -> int:
    """Address of a numpy array or torch tensor (host or device)."""
    if a is None:
        return 0
    if isinstance(a, np.ndarray):
        return a.ctypes.data
    return int(a.data_ptr())


class PandoraError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pandora_hip error {code}: {msg}")
        self.code = code


def view(addr: int, n: int, dtype) -> np.ndarray:
    """numpy view (no copy) of n elements at a host address owned by the library."""
    dt = np.dtype(dtype)
    if n == 0:
        return np.empty(0, dt)
    buf = (C.c_char * (n * dt.itemsize)).from_address(addr)
    return np.frombuffer(buf, dtype=dt, count=n)
