"""Host-side driver over the C ABI, used by the tests and bench.py.

Mirrors the reference's builder -> acceleration-structure flow
(OfflineBatchingAccelerationStructureBuilder::build and ::intersect / ::intersectAny,
projects/pandora/include/pandora/traversal/offline_batching_acceleration_structure.h:106-126,406-437)
for ray *batches*: the per-ray interface with hit/miss callbacks lives in the C++ adapter
include/pandora_amd/gpu_acceleration_structure.hpp; this module is the thin Python equivalent.
All computation happens in libpandora_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import capi
from .capi import PandoraError


def host_threads(cap: int = 16) -> int:
    """Host worker threads this process may use: PANDORA_HOST_THREADS when set, else the node's CPUs divided by the ranks sharing it
    (LOCAL_WORLD_SIZE under torch.distributed.run / bench.py's own launcher), at most `cap`. Eight ranks building their scenes on
    16 threads each would oversubscribe the host they share."""
    env = os.environ.get("PANDORA_HOST_THREADS")
    if env:
        return max(1, int(env))
    local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return max(1, min(cap, (os.cpu_count() or 1) // local))


def _check(rc: int, ctx=None, what: str = ""):
    if rc != capi.OK:
        msg = capi.lib().pgpu_last_error(ctx).decode() if ctx else what
        raise PandoraError(rc, msg or what)


@dataclass
class HostBvh8:
    """Bottom-level BVH8 in the reference's on-disk layout (wive_bvh8.fbs + contiguous_allocator.fbs)."""
    nodes: np.ndarray      # (n_nodes, 64) float32 view of 256-byte nodes
    leaf_idx: np.ndarray   # uint32
    root: int
    depth: int
    prim_ids: np.ndarray   # leaf object j -> triangle id in the mesh

    @property
    def n_nodes(self) -> int:
        return int(self.nodes.shape[0])

    def scene_bytes(self, n_verts: int) -> int:
        return 256 * self.n_nodes + 4 * self.leaf_idx.size + 12 * self.prim_ids.size + 12 * n_verts


def build_bvh8(pos: np.ndarray, idx: np.ndarray, prim_ids: np.ndarray | None = None, max_leaf: int = 4) -> HostBvh8:
    L = capi.lib()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    if prim_ids is None:
        prim_ids = np.arange(idx.shape[0], dtype=np.uint32)
    prim_ids = np.ascontiguousarray(prim_ids, np.uint32)
    h = C.c_void_p()
    _check(L.pgpu_build_bvh8(pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], prim_ids.ctypes.data, prim_ids.size, max_leaf, C.byref(h)),
           what="pgpu_build_bvh8 failed")
    try:
        n = C.c_size_t()
        p = L.pgpu_host_bvh8_nodes(h, C.byref(n))
        nodes = capi.view(p, n.value * 64, np.float32).reshape(-1, 64).copy() if n.value else np.zeros((0, 64), np.float32)
        p = L.pgpu_host_bvh8_leaf_indices(h, C.byref(n))
        leaf_idx = capi.view(p, n.value, np.uint32).copy()
        return HostBvh8(nodes, leaf_idx, int(L.pgpu_host_bvh8_root(h)), int(L.pgpu_host_bvh8_depth(h)), prim_ids)
    finally:
        L.pgpu_host_bvh8_free(h)


def build_bvh8_device(ctx: "Context", pos: np.ndarray, idx: np.ndarray, groups: "list[np.ndarray] | None" = None, max_leaf: int = 4):
    """The BVH8s of the triangle groups `groups` (None: one group, the whole mesh) built on the GPU of `ctx`
    (pgpu_build_bvh8_device: LBVH collapsed into 8-wide nodes). -> (list of HostBvh8, groups the host builder had to take because
    the device tree nested too deep). ctx.last_kernel_ms() = device time of the build."""
    L = capi.lib()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    if groups is None:
        groups = [np.arange(idx.shape[0], dtype=np.uint32)]
    groups = [np.ascontiguousarray(g, np.uint32) for g in groups]
    offsets = np.zeros(len(groups) + 1, np.uint64)
    offsets[1:] = np.cumsum([g.size for g in groups])
    prims = np.ascontiguousarray(np.concatenate(groups), np.uint32)
    handles = (C.c_void_p * len(groups))()
    host_built = C.c_uint32()
    _check(L.pgpu_build_bvh8_device(ctx._h, pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], prims.ctypes.data, offsets.ctypes.data, len(groups),
                                    max_leaf, handles, C.byref(host_built)), ctx._h, what="pgpu_build_bvh8_device failed")
    out = []
    try:
        for g, h in zip(groups, handles):
            n = C.c_size_t()
            p = L.pgpu_host_bvh8_nodes(h, C.byref(n))
            nodes = capi.view(p, n.value * 64, np.float32).reshape(-1, 64).copy() if n.value else np.zeros((0, 64), np.float32)
            p = L.pgpu_host_bvh8_leaf_indices(h, C.byref(n))
            out.append(HostBvh8(nodes, capi.view(p, n.value, np.uint32).copy(), int(L.pgpu_host_bvh8_root(h)), int(L.pgpu_host_bvh8_depth(h)), g))
    finally:
        for h in handles:
            if h:
                L.pgpu_host_bvh8_free(h)
    return out, int(host_built.value)


def make_camera(eye, look_at, up, fov_x_deg: float, aspect: float, lens_radius: float = 0.0, focal_distance: float = 1.0) -> "capi.Camera":
    """PerspectiveCamera::updatePosition (graphics_core/perspective_camera.cpp:50-65) in binary32 for a camera whose transform has
    the columns (right, up, forward, eye). The result is what pgpu_camera_rays takes."""
    f32 = np.float32
    eye, look_at, up = (np.asarray(a, np.float64) for a in (eye, look_at, up))
    fwd = look_at - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    t0, t1, t2, t3 = (v.astype(f32) for v in (right, upv, fwd, eye))
    lr, fd, ar = f32(lens_radius), f32(focal_distance), f32(aspect)
    dist = f32(0.5) / f32(np.tan(f32(np.radians(f32(fov_x_deg) / f32(2.0)))))
    cam = capi.Camera()
    vals = {"lens_center": t3, "lens_du": lr * t0, "lens_dv": lr * t1 / ar, "screen_center": t3 + fd * t2,
            "screen_du": -fd / dist * t0, "screen_dv": -fd / dist * t1 / ar}
    for k, v in vals.items():
        setattr(cam, k, (C.c_float * 3)(*[float(f32(x)) for x in v]))
    return cam


def build_bvh8_boxes(bounds: np.ndarray, alone: np.ndarray | None = None, max_leaf: int = 4) -> HostBvh8:
    """BVH8 over arbitrary leaf objects given by their bounds [n, 6]; object j is leaf object j. alone[j] != 0 gives
    object j a leaf of its own (instances)."""
    L = capi.lib()
    b = np.ascontiguousarray(bounds, np.float32).reshape(-1, 6)
    al = None if alone is None else np.ascontiguousarray(alone, np.uint8)
    h = C.c_void_p()
    _check(L.pgpu_build_bvh8_boxes(b.ctypes.data, b.shape[0], capi.ptr(al), max_leaf, C.byref(h)), what="pgpu_build_bvh8_boxes failed")
    try:
        n = C.c_size_t()
        p = L.pgpu_host_bvh8_nodes(h, C.byref(n))
        nodes = capi.view(p, n.value * 64, np.float32).reshape(-1, 64).copy() if n.value else np.zeros((0, 64), np.float32)
        p = L.pgpu_host_bvh8_leaf_indices(h, C.byref(n))
        leaf_idx = capi.view(p, n.value, np.uint32).copy()
        return HostBvh8(nodes, leaf_idx, int(L.pgpu_host_bvh8_root(h)), int(L.pgpu_host_bvh8_depth(h)), np.arange(b.shape[0], dtype=np.uint32))
    finally:
        L.pgpu_host_bvh8_free(h)


def partition_objects(bounds: np.ndarray, weights: np.ndarray | None, max_weight: int):
    """Scene cut over objects (detail::createSubScenes, batching.cpp:38-265) -> (group_of_object uint32[n], n_groups)."""
    L = capi.lib()
    b = np.ascontiguousarray(bounds, np.float32).reshape(-1, 6)
    w = None if weights is None else np.ascontiguousarray(weights, np.uint64)
    out = np.empty(b.shape[0], np.uint32)
    ng = C.c_uint32()
    _check(L.pgpu_partition_objects(b.ctypes.data, capi.ptr(w), b.shape[0], int(max_weight), out.ctypes.data, C.byref(ng)), what="pgpu_partition_objects failed")
    return out, int(ng.value)


def transform_bounds_many(lo, hi, local_to_world) -> np.ndarray:
    """Transform::transformToWorld(Bounds), graphics_core/transform.cpp:47-61, for n boxes [n, 3] under n matrices [n, 4, 4]:
    the box around the 8 transformed corners -> [n, 6]. Every product is evaluated as ((c0 m0 + c1 m1) + c2 m2) + m3 in
    float32, so one box and a million boxes give the same bits."""
    lo = np.asarray(lo, np.float32).reshape(-1, 3)
    hi = np.asarray(hi, np.float32).reshape(-1, 3)
    m = np.asarray(local_to_world, np.float32).reshape(-1, 4, 4)
    out_lo = np.full((m.shape[0], 3), np.inf, np.float32)
    out_hi = np.full((m.shape[0], 3), -np.inf, np.float32)
    for corner in range(8):
        c = [(hi if corner >> a & 1 else lo)[:, a] for a in range(3)]
        row = lambda r: ((c[0] * m[:, r, 0] + c[1] * m[:, r, 1]) + c[2] * m[:, r, 2]) + m[:, r, 3]
        w = row(3)
        p = np.stack([row(0) / w, row(1) / w, row(2) / w], axis=1)
        out_lo = np.minimum(out_lo, p)
        out_hi = np.maximum(out_hi, p)
    return np.concatenate([out_lo, out_hi], axis=1)


def transform_bounds(lo, hi, local_to_world) -> np.ndarray:
    return transform_bounds_many(np.asarray(lo)[None], np.asarray(hi)[None], np.asarray(local_to_world)[None])[0]


def world_to_local_rows_many(local_to_world) -> np.ndarray:
    """Rows 0..2 of Transform::m_inverseMatrix (transform.cpp:7-11) of n matrices as [n, 12] floats, row-major."""
    m = np.asarray(local_to_world, np.float64).reshape(-1, 4, 4)
    inv = np.linalg.inv(m) if m.shape[0] else m
    return np.ascontiguousarray(inv[:, :3, :].astype(np.float32)).reshape(-1, 12)


def world_to_local_rows(local_to_world) -> np.ndarray:
    return world_to_local_rows_many(np.asarray(local_to_world)[None])[0]


@dataclass
class HostBvh4:
    nodes: np.ndarray       # (n_nodes, 28) float32 view of 112-byte nodes
    leaf_order: np.ndarray  # final leaf slot -> input leaf
    root: int
    depth: int


def bvh8_to_bytes(bvh: HostBvh8) -> bytes:
    """The reference's on-disk flatbuffer of a bottom-level BVH (wive_bvh8.fbs). The leaf-object table is not part of it
    (the reference dumps raw pointers there, offline_bvh_cache.cpp:125-129); save `prim_ids` next to it."""
    L = capi.lib()
    out, n = C.c_void_p(), C.c_size_t()
    nodes = np.ascontiguousarray(bvh.nodes)
    idx = np.ascontiguousarray(bvh.leaf_idx, np.uint32)
    _check(L.pgpu_bvh8_serialize(nodes.ctypes.data, bvh.n_nodes, idx.ctypes.data, idx.size, bvh.root, bvh.prim_ids.size, C.byref(out), C.byref(n)),
           what="pgpu_bvh8_serialize failed")
    try:
        return C.string_at(out.value, n.value)
    finally:
        L.pgpu_bytes_free(out)


def mesh_to_bytes(pos: np.ndarray, idx: np.ndarray, normals: np.ndarray | None = None, texcoords: np.ndarray | None = None) -> bytes:
    """The reference's triangle mesh chunk (triangle_mesh.fbs) as TriangleShape::serialize writes it (triangle_shape.cpp:520-553)."""
    L = capi.lib()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    nrm = None if normals is None else np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
    tex = None if texcoords is None else np.ascontiguousarray(texcoords, np.float32).reshape(-1, 2)
    if (nrm is not None and nrm.shape[0] != pos.shape[0]) or (tex is not None and tex.shape[0] != pos.shape[0]):
        raise PandoraError(capi.ERR_INVALID, "mesh_to_bytes: one normal / texture coordinate per vertex")
    out, n = C.c_void_p(), C.c_size_t()
    _check(L.pgpu_mesh_serialize(pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], capi.ptr(nrm), capi.ptr(tex), C.byref(out), C.byref(n)),
           what="pgpu_mesh_serialize failed")
    try:
        return C.string_at(out.value, n.value)
    finally:
        L.pgpu_bytes_free(out)


def mesh_from_bytes(data: bytes) -> dict:
    """Reads a triangle mesh chunk: {"pos", "idx", "normals", "texcoords", "bounds"} (copies; empty arrays for absent vectors)."""
    L = capi.lib()
    buf = np.frombuffer(data, np.uint8)
    ptrs = [C.c_void_p() for _ in range(4)]
    cnts = [C.c_size_t() for _ in range(4)]
    bounds = np.zeros(6, np.float32)
    _check(L.pgpu_mesh_deserialize(buf.ctypes.data, buf.size, C.byref(ptrs[0]), C.byref(cnts[0]), C.byref(ptrs[1]), C.byref(cnts[1]), C.byref(ptrs[2]),
                                   C.byref(cnts[2]), C.byref(ptrs[3]), C.byref(cnts[3]), bounds.ctypes.data), what="pgpu_mesh_deserialize: not a TriangleMesh flatbuffer")

    def take(i, dtype, width):
        n = int(cnts[i].value)
        if n == 0:
            return np.zeros((0, width), dtype)
        return np.frombuffer(data, dtype, count=n * width, offset=ptrs[i].value - buf.ctypes.data).reshape(n, width).copy()

    return {"pos": take(0, np.float32, 3), "idx": take(1, np.uint32, 3), "normals": take(2, np.float32, 3), "texcoords": take(3, np.float32, 2),
            "bounds": bounds.reshape(2, 3)}


def bvh8_from_bytes(data: bytes, prim_ids: np.ndarray) -> HostBvh8:
    L = capi.lib()
    buf = np.frombuffer(data, np.uint8)
    nodes, n_nodes, idx, n_idx = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
    root, n_obj = C.c_uint32(), C.c_uint64()
    _check(L.pgpu_bvh8_deserialize(buf.ctypes.data, buf.size, C.byref(nodes), C.byref(n_nodes), C.byref(idx), C.byref(n_idx), C.byref(root), C.byref(n_obj)),
           what="pgpu_bvh8_deserialize: not a WiVeBVH8 flatbuffer")
    prim_ids = np.ascontiguousarray(prim_ids, np.uint32)
    if int(n_obj.value) != prim_ids.size:
        raise PandoraError(capi.ERR_INVALID, "Number of leaf objects does not match that of the serialized BVH")  # wive_bvh8_impl.h:31
    node_arr = np.frombuffer(data, np.uint8, count=n_nodes.value * 256, offset=nodes.value - buf.ctypes.data).view(np.float32).reshape(-1, 64).copy() if n_nodes.value else np.zeros((0, 64), np.float32)
    idx_arr = np.frombuffer(data, np.uint32, count=n_idx.value, offset=idx.value - buf.ctypes.data).copy() if n_idx.value else np.zeros(0, np.uint32)
    return HostBvh8(node_arr, idx_arr, int(root.value), host_bvh8_depth(node_arr, int(root.value)), prim_ids)


def host_bvh8_depth(nodes: np.ndarray, root: int) -> int:
    """Nested inner nodes below `root` (iterative; handles as wive_bvh8_impl.h:333-382)."""
    u = nodes.view(np.uint32).reshape(-1, 64)
    depth, todo = 0, [(root, 1)]
    if (root >> 29) != 2:
        return 0
    while todo:
        h, d = todo.pop()
        if (h & 0x1FFFFFFF) >= u.shape[0] or d > 64:
            raise PandoraError(capi.ERR_INVALID, "BVH8 node handle out of range or tree deeper than 64 (damaged file?)")
        depth = max(depth, d)
        for c in u[h & 0x1FFFFFFF, 48:56]:
            if (int(c) >> 29) == 2:
                todo.append((int(c), d + 1))
    return depth


def build_top_bvh4(leaf_bounds: np.ndarray) -> HostBvh4:
    L = capi.lib()
    b = np.ascontiguousarray(leaf_bounds, np.float32).reshape(-1, 6)
    h = C.c_void_p()
    _check(L.pgpu_build_top_bvh4(b.ctypes.data, b.shape[0], C.byref(h)), what="pgpu_build_top_bvh4 failed")
    try:
        n = C.c_size_t()
        p = L.pgpu_host_bvh4_nodes(h, C.byref(n))
        nodes = capi.view(p, n.value * 28, np.float32).reshape(-1, 28).copy()
        p = L.pgpu_host_bvh4_leaf_order(h, C.byref(n))
        order = capi.view(p, n.value, np.uint32).copy()
        return HostBvh4(nodes, order, int(L.pgpu_host_bvh4_root(h)), int(L.pgpu_host_bvh4_depth(h)))
    finally:
        L.pgpu_host_bvh4_free(h)


def partition_prims(pos: np.ndarray, idx: np.ndarray, max_prims: int):
    """-> (group_of_prim uint32[nt], n_groups)."""
    L = capi.lib()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    out = np.empty(idx.shape[0], np.uint32)
    ng = C.c_uint32()
    _check(L.pgpu_partition_prims(pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], max_prims, out.ctypes.data, C.byref(ng)),
           what="pgpu_partition_prims failed")
    return out, int(ng.value)


def partition_prims_device(ctx: "Context", pos: np.ndarray, idx: np.ndarray, max_prims: int):
    """partition_prims on the GPU of `ctx` (pgpu_partition_prims_device: the cut on an LBVH). -> (group_of_prim uint32[nt], n_groups)."""
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    out = np.empty(idx.shape[0], np.uint32)
    ng = C.c_uint32()
    _check(capi.lib().pgpu_partition_prims_device(ctx._h, pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], max_prims, out.ctypes.data, C.byref(ng)), ctx._h)
    return out, int(ng.value)


@dataclass
class HostSvdags:
    words: np.ndarray       # shared uint32 node array
    descs: np.ndarray       # structured: root, bounds_min[3], inv_extent, resolution
    words_before: int


SVDAG_DESC_DTYPE = np.dtype([("root", np.uint32), ("bounds_min", np.float32, 3), ("inv_extent", np.float32), ("resolution", np.uint32)])


def build_svdags(pos, idx, groups: list[np.ndarray], resolution: int, device_ctx: "Context | None" = None) -> HostSvdags:
    """Voxelisation + SVO + DAG merge of every group of triangles. device_ctx: build on that context's GPU
    (pgpu_build_svdags_device) instead of the host; both give DAGs with the same voxels, bounds and word count."""
    L = capi.lib()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    prims = np.ascontiguousarray(np.concatenate(groups), np.uint32)
    offs = np.zeros(len(groups) + 1, np.uint64)
    offs[1:] = np.cumsum([g.size for g in groups])
    h = C.c_void_p()
    if device_ctx is not None:
        _check(L.pgpu_build_svdags_device(device_ctx._h, pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], prims.ctypes.data, offs.ctypes.data,
                                          len(groups), resolution, C.byref(h)), device_ctx._h)
    else:
        _check(L.pgpu_build_svdags(pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], prims.ctypes.data, offs.ctypes.data, len(groups), resolution, C.byref(h)),
               what="pgpu_build_svdags failed")
    try:
        n = C.c_size_t()
        p = L.pgpu_host_svdags_words(h, C.byref(n))
        words = capi.view(p, n.value, np.uint32).copy()
        p = L.pgpu_host_svdags_descs(h, C.byref(n))
        descs = capi.view(p, n.value, SVDAG_DESC_DTYPE).copy()
        return HostSvdags(words, descs, int(L.pgpu_host_svdags_words_before_compression(h)))
    finally:
        L.pgpu_host_svdags_free(h)


def voxelize(pos, idx, prims, resolution: int):
    L = capi.lib()
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    prims = np.ascontiguousarray(prims, np.uint32)
    bits = np.zeros((resolution**3 + 31) // 32, np.uint32)
    bounds = np.zeros(6, np.float32)
    _check(L.pgpu_voxelize(pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], prims.ctypes.data, prims.size, resolution, bounds.ctypes.data, bits.ctypes.data),
           what="pgpu_voxelize failed")
    return bits, bounds


def svdag_expand(words: np.ndarray, desc) -> np.ndarray:
    L = capi.lib()
    d = capi.SvdagDesc(int(desc["root"]), (C.c_float * 3)(*[float(x) for x in desc["bounds_min"]]), float(desc["inv_extent"]), int(desc["resolution"]))
    res = int(desc["resolution"])
    bits = np.zeros((res**3 + 31) // 32, np.uint32)
    _check(L.pgpu_svdag_expand(words.ctypes.data, words.size, C.byref(d), bits.ctypes.data), what="pgpu_svdag_expand failed")
    return bits


class Context:
    """One GPU context (pgpu_ctx). Ray batches are SoA float32 arrays of shape (8, n): numpy (host
    memory, copied by the library) or torch CUDA tensors (device memory, used in place)."""

    def __init__(self, device: int = 0):
        self._L = capi.lib()
        h = C.c_void_p()
        rc = self._L.pgpu_init(device, C.byref(h))
        if rc != capi.OK:
            raise PandoraError(rc, "pgpu_init failed (no GPU visible?)")
        self._h = h
        self.instanced = False  # an instanced BVH was uploaded: closest-hit results then carry "inst"

    def close(self):
        if getattr(self, "_h", None):
            self._L.pgpu_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene upload
    def upload_mesh(self, pos, idx) -> int:
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
        out = C.c_uint32()
        _check(self._L.pgpu_upload_mesh(self._h, pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], C.byref(out)), self._h)
        return int(out.value)

    def upload_bvh8(self, bvh: HostBvh8, mesh_id: int, prim_remap=None) -> int:
        """prim_remap: optional uint32 array, triangle id of the builder's mesh -> triangle id of the uploaded mesh `mesh_id`
        (a rank that uploaded only the triangles of its own batching points, see sub_mesh)."""
        objs = np.empty((bvh.prim_ids.size, 2), np.uint32)
        objs[:, 0] = mesh_id
        objs[:, 1] = bvh.prim_ids if prim_remap is None else prim_remap[bvh.prim_ids]
        nodes = np.ascontiguousarray(bvh.nodes)
        out = C.c_uint32()
        _check(self._L.pgpu_upload_bvh8(self._h, nodes.ctypes.data, bvh.n_nodes, bvh.leaf_idx.ctypes.data, bvh.leaf_idx.size, bvh.root,
                                        objs.ctypes.data, objs.shape[0], C.byref(out)), self._h)
        return int(out.value)

    def build_upload_bvh8_device(self, pos, idx, groups: "list[np.ndarray] | None" = None, max_leaf: int = 4):
        """pgpu_build_upload_bvh8_device: the BVH8s of the triangle groups built on this GPU and left there, as part of the context's
        BVH table (no upload_mesh / upload_bvh8). -> (bvh ids, groups the host builder had to take)."""
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
        if groups is None:
            groups = [np.arange(idx.shape[0], dtype=np.uint32)]
        offsets = np.zeros(len(groups) + 1, np.uint64)
        offsets[1:] = np.cumsum([np.asarray(g).size for g in groups])
        prims = np.ascontiguousarray(np.concatenate(groups), np.uint32)
        ids = np.zeros(len(groups), np.uint32)
        host_built = C.c_uint32()
        _check(self._L.pgpu_build_upload_bvh8_device(self._h, pos.ctypes.data, pos.shape[0], idx.ctypes.data, idx.shape[0], prims.ctypes.data, offsets.ctypes.data,
                                                     len(groups), max_leaf, ids.ctypes.data, C.byref(host_built)), self._h)
        return ids, int(host_built.value)

    def download_bvh8(self, bvh_id: int, prim_ids: np.ndarray) -> HostBvh8:
        """A BVH of this context back as a HostBvh8 (pgpu_download_bvh8); prim_ids = its leaf objects' triangle ids."""
        nn, ni, root, depth = C.c_size_t(), C.c_size_t(), C.c_uint32(), C.c_int()
        _check(self._L.pgpu_bvh8_info(self._h, int(bvh_id), C.byref(nn), C.byref(ni), C.byref(root), C.byref(depth)), self._h)
        nodes = np.zeros((nn.value, 64), np.float32)
        leaf_idx = np.zeros(ni.value, np.uint32)
        _check(self._L.pgpu_download_bvh8(self._h, int(bvh_id), nodes.ctypes.data if nn.value else None, leaf_idx.ctypes.data), self._h)
        return HostBvh8(nodes, leaf_idx, int(root.value), int(depth.value), np.ascontiguousarray(prim_ids, np.uint32))

    def stage_bvh8(self, bvh: HostBvh8, mesh_id: int) -> int:
        """Like upload_bvh8, but the BVH stays in pinned host memory and is streamed into the HBM leaf cache on demand."""
        objs = np.empty((bvh.prim_ids.size, 2), np.uint32)
        objs[:, 0] = mesh_id
        objs[:, 1] = bvh.prim_ids
        nodes = np.ascontiguousarray(bvh.nodes)
        out = C.c_uint32()
        _check(self._L.pgpu_stage_bvh8(self._h, nodes.ctypes.data, bvh.n_nodes, bvh.leaf_idx.ctypes.data, bvh.leaf_idx.size, bvh.root,
                                       objs.ctypes.data, objs.shape[0], C.byref(out)), self._h)
        return int(out.value)

    def upload_bvh8_objects(self, bvh: HostBvh8, objs: np.ndarray, instances=None, staged: bool = False) -> int:
        """A BVH8 whose leaf objects are given explicitly: objs [n, 2] uint32 = (mesh id, primitive id), or (capi.INSTANCE, k)
        for instance k of `instances` = list of (bvh id, world_to_local 12 floats). pgpu_upload_bvh8_instanced /
        pgpu_stage_bvh8_instanced."""
        objs = np.ascontiguousarray(objs, np.uint32).reshape(-1, 2)
        # pgpu_instance: {uint32 bvh_id; float world_to_local[12]} = 13 words per instance
        if isinstance(instances, tuple):  # (bvh ids uint32[n], rows float32[n, 12])
            ids, rows = instances
            n_inst = int(np.asarray(ids).size)
            inst = np.empty((n_inst, 13), np.uint32)
            inst[:, 0] = ids
            inst[:, 1:] = np.ascontiguousarray(rows, np.float32).reshape(n_inst, 12).view(np.uint32)
        else:
            n_inst = len(instances or [])
            inst = np.empty((n_inst, 13), np.uint32)
            for k, (bid, m) in enumerate(instances or []):
                inst[k, 0] = int(bid)
                inst[k, 1:] = np.asarray(m, np.float32).reshape(-1)[:12].view(np.uint32)
        assert C.sizeof(capi.InstanceDesc) == 52
        nodes = np.ascontiguousarray(bvh.nodes)
        out = C.c_uint32()
        fn = self._L.pgpu_stage_bvh8_instanced if staged else self._L.pgpu_upload_bvh8_instanced
        _check(fn(self._h, nodes.ctypes.data, bvh.n_nodes, bvh.leaf_idx.ctypes.data, bvh.leaf_idx.size, bvh.root, objs.ctypes.data, objs.shape[0],
                  C.c_void_p(inst.ctypes.data) if n_inst else None, n_inst, C.byref(out)), self._h)
        if n_inst:
            self.instanced = True
        return int(out.value)

    def set_leaf_cache(self, n_bytes: int):
        _check(self._L.pgpu_set_leaf_cache(self._h, int(n_bytes)), self._h)

    def set_svdag_policy(self, policy: int):
        """0: every batching point with an SVDAG gets the early-out test (the reference). 1: only points whose geometry is not in
        this GPU's HBM (staged or owned by another rank). Same hits; visits and SVDAG counters differ."""
        _check(self._L.pgpu_set_svdag_policy(self._h, int(policy)), self._h)

    def set_leaf_hold(self, image_bytes_per_ray: int):
        """Streamed leaves: an image that is not in the cache is loaded once its batching point holds one ray per this many
        bytes of image; until then the point's rays stay parked (0 = never wait). Default 256."""
        _check(self._L.pgpu_set_leaf_hold(self._h, int(image_bytes_per_ray)), self._h)

    def set_leaf_in_place(self, image_bytes_per_ray: int):
        """Flushed queues with fewer rays than one per this many bytes of image are traced in pinned host memory instead of being
        loaded (pgpu_set_leaf_in_place; default 128, 0 = never). Same results."""
        _check(self._L.pgpu_set_leaf_in_place(self._h, int(image_bytes_per_ray)), self._h)

    def set_leaf_copy(self, mode: int):
        """How staged images reach the HBM cache: 0 = one gather kernel per launch group (default), 1 = one hipMemcpyAsync per image."""
        _check(self._L.pgpu_set_leaf_copy(self._h, int(mode)), self._h)

    def upload_svdags(self, dags: HostSvdags):
        descs = np.ascontiguousarray(dags.descs)
        _check(self._L.pgpu_upload_svdags(self._h, dags.words.ctypes.data, dags.words.size, descs.ctypes.data, descs.size), self._h)

    def upload_top_bvh4(self, top: HostBvh4, leaves: np.ndarray):
        """leaves: (n_leaves, 2) uint32 [bvh_id, svdag index or NO_SVDAG] in final leaf-slot order."""
        leaves = np.ascontiguousarray(leaves, np.uint32).reshape(-1, 2)
        nodes = np.ascontiguousarray(top.nodes)
        _check(self._L.pgpu_upload_top_bvh4(self._h, nodes.ctypes.data, nodes.shape[0], top.root, leaves.ctypes.data, leaves.shape[0]), self._h)

    # -- tracing
    @staticmethod
    def _is_torch(a) -> bool:
        return not isinstance(a, np.ndarray)

    def _rays_struct(self, rays):
        n = int(rays.shape[1])
        assert rays.shape[0] == 8
        if self._is_torch(rays):
            # (rows may be the front of longer rows: what compact_paths returns)
            assert rays.is_cuda and str(rays.dtype) == "torch.float32"
            if rays.is_contiguous():
                base, stride = rays.data_ptr(), n * 4
            else:
                assert rays.stride(1) == 1 and rays.stride(0) >= n
                base, stride = rays.data_ptr(), int(rays.stride(0)) * 4
        else:
            assert rays.dtype == np.float32 and rays.flags["C_CONTIGUOUS"]
            base, stride = rays.ctypes.data, n * 4
        return capi.Rays(*[base + i * stride for i in range(8)]), n

    def _alloc(self, like, n, kind):
        if self._is_torch(like):
            import torch
            dt = torch.float32 if kind == "f" else torch.int32
            return torch.empty(n, dtype=dt, device=like.device)
        return np.empty(n, np.float32 if kind == "f" else np.uint32)

    def trace_bvh8(self, bvh_id: int, rays, kind: int = capi.CLOSEST, tri_mode: int = capi.TRI_MT, out=None, stream: int = 0):
        """Single-level trace. Returns dict(t, obj, leaf, u, v) for CLOSEST or dict(hit) for ANY."""
        r, n = self._rays_struct(rays)
        mem = capi.MEM_DEVICE if self._is_torch(rays) else capi.MEM_HOST
        if kind == capi.CLOSEST:
            if out is None:
                out = {k: self._alloc(rays, n, "f" if k in ("t", "u", "v") else "u") for k in ("t", "obj", "leaf", "u", "v") + (("inst",) if self.instanced else ())}
            h = capi.Hits(capi.ptr(out["t"]), capi.ptr(out["obj"]), capi.ptr(out["leaf"]), capi.ptr(out["u"]), capi.ptr(out["v"]), 0, capi.ptr(out.get("inst")))
            _check(self._L.pgpu_trace_bvh8(self._h, bvh_id, kind, tri_mode, C.byref(r), n, mem, C.byref(h), None, stream), self._h)
        else:
            if out is None:
                out = {"hit": self._alloc(rays, n, "u")}
            _check(self._L.pgpu_trace_bvh8(self._h, bvh_id, kind, tri_mode, C.byref(r), n, mem, None, capi.ptr(out["hit"]), stream), self._h)
        return out

    def trace_bvh8_noalloc(self, bvh_id: int, rays, out: dict, kind: int = capi.CLOSEST, tri_mode: int = capi.TRI_MT, stream: int = 0):
        """pgpu_trace_bvh8_noalloc: device tensors only, `out` allocated by the caller; the library allocates nothing."""
        r, n = self._rays_struct(rays)
        assert self._is_torch(rays)
        if kind == capi.CLOSEST:
            h = capi.Hits(capi.ptr(out["t"]), capi.ptr(out["obj"]), capi.ptr(out["leaf"]), capi.ptr(out["u"]), capi.ptr(out["v"]), 0)
            _check(self._L.pgpu_trace_bvh8_noalloc(self._h, bvh_id, kind, tri_mode, C.byref(r), n, C.byref(h), None, stream), self._h)
        else:
            _check(self._L.pgpu_trace_bvh8_noalloc(self._h, bvh_id, kind, tri_mode, C.byref(r), n, None, capi.ptr(out["hit"]), stream), self._h)
        return out

    def trace(self, rays, kind: int = capi.CLOSEST, tri_mode: int = capi.TRI_MT, out=None, stream: int = 0):
        """Two-level trace through the uploaded top-level BVH4. Returns (results, counters dict)."""
        r, n = self._rays_struct(rays)
        mem = capi.MEM_DEVICE if self._is_torch(rays) else capi.MEM_HOST
        cnt = capi.Counters()
        if kind == capi.CLOSEST:
            if out is None:
                out = {k: self._alloc(rays, n, "f" if k in ("t", "u", "v") else "u") for k in ("t", "obj", "leaf", "u", "v", "visits") + (("inst",) if self.instanced else ())}
            h = capi.Hits(capi.ptr(out["t"]), capi.ptr(out["obj"]), capi.ptr(out["leaf"]), capi.ptr(out["u"]), capi.ptr(out["v"]), capi.ptr(out.get("visits")), capi.ptr(out.get("inst")))
            _check(self._L.pgpu_trace(self._h, kind, tri_mode, C.byref(r), n, mem, C.byref(h), None, C.byref(cnt), stream), self._h)
        else:
            if out is None:
                out = {"hit": self._alloc(rays, n, "u"), "visits": self._alloc(rays, n, "u")}
            h = capi.Hits(0, 0, 0, 0, 0, capi.ptr(out.get("visits")))
            _check(self._L.pgpu_trace(self._h, kind, tri_mode, C.byref(r), n, mem, C.byref(h), capi.ptr(out["hit"]), C.byref(cnt), stream), self._h)
        return out, {f: int(getattr(cnt, f)) for f, _ in capi.Counters._fields_}

    # -- building blocks of the two-level loop (device tensors only; used by pandora_amd.sharded)
    def ray_state_init(self, tfar, node, stack, hits: dict | None, any_hit=None, stream: int = 0):
        n = int(node.shape[0])
        h = None
        if hits is not None:
            h = capi.Hits(capi.ptr(hits["t"]), capi.ptr(hits["obj"]), capi.ptr(hits["leaf"]), capi.ptr(hits["u"]), capi.ptr(hits["v"]), capi.ptr(hits.get("visits")))
        _check(self._L.pgpu_ray_state_init(self._h, n, capi.ptr(tfar), capi.ptr(node), capi.ptr(stack), C.byref(h) if h is not None else None,
                                           capi.ptr(any_hit), stream), self._h)

    def top_episode(self, kind: int, rays, active, n_active: int, node, stack, t_state, visits, any_hit, parked_leaf, may_park: bool = True, stream: int = 0):
        r, _ = self._rays_struct(rays)
        _check(self._L.pgpu_top_episode(self._h, kind, C.byref(r), capi.ptr(active), n_active, capi.ptr(node), capi.ptr(stack), capi.ptr(t_state),
                                        capi.ptr(visits), capi.ptr(any_hit), capi.ptr(parked_leaf), int(may_park), stream), self._h)

    def trace_leaves(self, kind: int, tri_mode: int, rays, work_ray, work_leaf, n_work: int, hits: dict | None, any_hit=None, stream: int = 0):
        r, _ = self._rays_struct(rays)
        h = None
        if hits is not None:
            h = capi.Hits(capi.ptr(hits["t"]), capi.ptr(hits["obj"]), capi.ptr(hits["leaf"]), capi.ptr(hits["u"]), capi.ptr(hits["v"]), 0)
        _check(self._L.pgpu_trace_leaves(self._h, kind, tri_mode, C.byref(r), capi.ptr(work_ray), capi.ptr(work_leaf), n_work,
                                         C.byref(h) if h is not None else None, capi.ptr(any_hit), stream), self._h)

    def route_pack(self, rays, t_state, active, parked_leaf, n_active: int, owner_dev, world: int, records, sent_slot, sent_leaf, stream: int = 0) -> np.ndarray:
        r, _ = self._rays_struct(rays)
        counts = np.zeros(world, np.uint32)
        _check(self._L.pgpu_route_pack(self._h, C.byref(r), capi.ptr(t_state), capi.ptr(active), capi.ptr(parked_leaf), n_active, capi.ptr(owner_dev), world,
                                       counts.ctypes.data, capi.ptr(records), capi.ptr(sent_slot), capi.ptr(sent_leaf), stream), self._h)
        return counts

    def trace_leaf_records(self, kind: int, tri_mode: int, records, k: int, results, stream: int = 0):
        _check(self._L.pgpu_trace_leaf_records(self._h, kind, tri_mode, capi.ptr(records), k, capi.ptr(results), stream), self._h)

    def route_merge(self, kind: int, results, sent_slot, sent_leaf, k: int, hits: dict | None, any_hit=None, stream: int = 0):
        h = None
        if hits is not None:
            h = capi.Hits(capi.ptr(hits["t"]), capi.ptr(hits["obj"]), capi.ptr(hits["leaf"]), capi.ptr(hits["u"]), capi.ptr(hits["v"]), 0)
        _check(self._L.pgpu_route_merge(self._h, kind, capi.ptr(results), capi.ptr(sent_slot), capi.ptr(sent_leaf), k, C.byref(h) if h is not None else None,
                                        capi.ptr(any_hit), stream), self._h)

    # -- several GPUs: batching points sharded over the ranks (pgpu_trace_sharded; one process and one context per GPU)
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 bytes (an ncclUniqueId) that rank 0 makes and hands to every other rank."""
        buf = C.create_string_buffer(128)
        rc = capi.lib().pgpu_comm_unique_id(buf)
        if rc != capi.OK:
            raise PandoraError(rc, "pgpu_comm_unique_id failed (librccl not loadable?)")
        return buf.raw

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """Joins the RCCL communicator (collective)."""
        assert len(unique_id) == 128
        _check(self._L.pgpu_comm_init(self._h, rank, world, unique_id), self._h)

    def comm_init_transport(self, rank: int, world: int, transport):
        """The collectives come from `transport` (an object with .struct(): capi.Transport, kept alive here)."""
        self._transport = transport
        self._transport_struct = transport.struct()
        _check(self._L.pgpu_comm_init_transport(self._h, rank, world, C.byref(self._transport_struct)), self._h)

    def comm_selftest(self):
        """Every collective of the transport once, results checked (collective)."""
        _check(self._L.pgpu_comm_selftest(self._h), self._h)

    def comm_destroy(self):
        _check(self._L.pgpu_comm_destroy(self._h), self._h)

    def set_leaf_owners(self, owner_of_leaf):
        o = np.ascontiguousarray(owner_of_leaf, np.uint32)
        _check(self._L.pgpu_set_leaf_owners(self._h, o.ctypes.data, o.size), self._h)

    def trace_sharded(self, rays, kind: int = capi.CLOSEST, tri_mode: int = capi.TRI_MT, out=None, stream: int = 0):
        """pgpu_trace_sharded (collective; device tensors). Returns (results, counters dict) like trace()."""
        r, n = self._rays_struct(rays)
        assert self._is_torch(rays)
        cnt = capi.Counters()
        if kind == capi.CLOSEST:
            if out is None:
                out = {k: self._alloc(rays, n, "f" if k in ("t", "u", "v") else "u") for k in ("t", "obj", "leaf", "u", "v", "visits") + (("inst",) if self.instanced else ())}
            h = capi.Hits(capi.ptr(out["t"]), capi.ptr(out["obj"]), capi.ptr(out["leaf"]), capi.ptr(out["u"]), capi.ptr(out["v"]), capi.ptr(out.get("visits")),
                          capi.ptr(out.get("inst")))
            _check(self._L.pgpu_trace_sharded(self._h, kind, tri_mode, C.byref(r), n, C.byref(h), None, C.byref(cnt), stream), self._h)
        else:
            if out is None:
                out = {"hit": self._alloc(rays, n, "u"), "visits": self._alloc(rays, n, "u")}
            h = capi.Hits(0, 0, 0, 0, 0, capi.ptr(out.get("visits")))
            _check(self._L.pgpu_trace_sharded(self._h, kind, tri_mode, C.byref(r), n, C.byref(h), capi.ptr(out["hit"]), C.byref(cnt), stream), self._h)
        return out, {f: int(getattr(cnt, f)) for f, _ in capi.Counters._fields_}

    def debug_fail_sharded(self, after_rounds: int = -1):
        """Test hook (pgpu_debug_fail_sharded): this rank's trace_sharded fails after that many completed rounds; -1 = off."""
        _check(self._L.pgpu_debug_fail_sharded(self._h, after_rounds), self._h)

    def film_allreduce(self, film, stream: int = 0):
        """Sums the int64 fixed-point films of all ranks in place (pgpu_film_allreduce)."""
        _check(self._L.pgpu_film_allreduce(self._h, capi.ptr(film), int(film.numel() // 3), stream), self._h)

    def set_top_variant(self, variant: int = 0, refill_min: int = 0, cast_trips: int = 0):
        """Top-level kernel of trace(): 1 lockstep waves with casts inside the step, 4 the same with a wave's casts gathered into phases,
        2 lanes refilled as their rays finish, 3 = 1 for the first episode then 2, 0 automatic (4). Same results."""
        _check(self._L.pgpu_set_top_variant(self._h, variant, refill_min, cast_trips), self._h)

    def set_round_lookahead(self, on: bool = True):
        """trace() on resident scenes queues rounds ahead of the host's knowledge of the parked counts (default); False =
        one host read per round before the next is queued. Same results."""
        _check(self._L.pgpu_set_round_lookahead(self._h, int(on)), self._h)

    def set_batch_parts(self, parts: int = 0):
        """trace() on resident scenes runs a large batch as `parts` batches side by side on streams of their own (2..4; 1 = never;
        0 = default: two parts from 2^21 rays on). Same results."""
        _check(self._L.pgpu_set_batch_parts(self._h, int(parts)), self._h)

    def set_svdag_defer(self, min_rays: int = 0, passes: int = 2):
        """Where trace() runs the SVDAG casts: 0 = inside the top-level episode (default); > 0 = passes of at least that many
        rays cast in a kernel of their own, `passes` times per round (pgpu_set_svdag_defer). Same results bit for bit."""
        _check(self._L.pgpu_set_svdag_defer(self._h, int(min_rays), int(passes)), self._h)

    # -- the benchmark integrator's ray sources on the device (pgpu_camera_rays / pgpu_lambert_bounce / pgpu_splat_misses)
    def set_keep_shading(self, on: bool = True):
        """Uploads that follow also keep the vertices per leaf object on the device (what lambert_bounce reads)."""
        _check(self._L.pgpu_set_keep_shading(self._h, int(on)), self._h)

    @staticmethod
    def _paths_struct(paths: dict):
        return capi.Paths(*[capi.ptr(paths[k]) for k in ("rng_state", "rng_inc", "pixel", "wr", "wg", "wb")])

    def camera_rays(self, cam, width: int, height: int, spp: int, seed: int, first: int, n: int, device, stream: int = 0):
        """-> (rays [8, n] float32 device tensor, path state dict of device tensors)."""
        import torch
        rays = torch.empty((8, n), dtype=torch.float32, device=device)
        paths = {"rng_state": torch.empty(n, dtype=torch.int64, device=device), "rng_inc": torch.empty(n, dtype=torch.int64, device=device),
                 "pixel": torch.empty(n, dtype=torch.int32, device=device)}
        paths.update({k: torch.empty(n, dtype=torch.float32, device=device) for k in ("wr", "wg", "wb")})
        r, _ = self._rays_struct(rays)
        p = self._paths_struct(paths)
        _check(self._L.pgpu_camera_rays(self._h, C.byref(cam), width, height, spp, seed, first, n, C.byref(r), C.byref(p), stream), self._h)
        return rays, paths

    def lambert_bounce(self, bvh_id, rays, hits: dict, albedo, paths: dict, stream: int = 0):
        """One Lambert bounce per hit (bvh_id None: two-level hits, hits["leaf"] names the batching point). Updates `paths`
        in place and returns the bounce rays [8, n] (dead rays where there is no bounce)."""
        import torch
        n = int(rays.shape[1])
        out = torch.empty_like(rays)
        r_in, _ = self._rays_struct(rays)
        r_out, _ = self._rays_struct(out)
        h = capi.Hits(capi.ptr(hits["t"]), capi.ptr(hits["obj"]), capi.ptr(hits.get("leaf")), capi.ptr(hits["u"]), capi.ptr(hits["v"]), 0)
        p = self._paths_struct(paths)
        alb = (C.c_float * 3)(*[float(x) for x in albedo])
        _check(self._L.pgpu_lambert_bounce(self._h, capi.NO_BVH if bvh_id is None else bvh_id, C.byref(r_in), C.byref(h), n, alb, C.byref(r_out), C.byref(p), stream), self._h)
        return out

    def compact_paths(self, rays, paths: dict, stream: int = 0):
        """The paths that are still alive (tnear <= tfar), in their order: -> (rays [8, n_live], path state of n_live), views of
        the front of new arrays of the old length. Waits for the stream (pgpu_compact_paths)."""
        import torch
        n = int(rays.shape[1])
        out = torch.empty((8, n), dtype=torch.float32, device=rays.device)
        pout = {k: torch.empty(n, dtype=v.dtype, device=v.device) for k, v in paths.items()}
        r_in, _ = self._rays_struct(rays)
        r_out, _ = self._rays_struct(out)
        live = C.c_size_t(0)
        _check(self._L.pgpu_compact_paths(self._h, C.byref(r_in), C.byref(self._paths_struct(paths)), n, C.byref(r_out), C.byref(self._paths_struct(pout)), C.byref(live), stream), self._h)
        k = int(live.value)
        return out[:, :k], {key: v[:k] for key, v in pout.items()}

    def splat_misses(self, film, obj, paths: dict, sky, stream: int = 0):
        p = self._paths_struct(paths)
        s = (C.c_float * 3)(*[float(x) for x in sky])
        _check(self._L.pgpu_splat_misses(self._h, capi.ptr(film), int(film.numel() // 3), capi.ptr(obj), C.byref(p), int(obj.numel()), s, stream), self._h)

    def film_add(self, film, pixel, r, g, b, stream: int = 0):
        """Sensor::addPixelContribution over a batch (sensor.cpp:36-49). film: int64 device tensor [n_pixels, 3] (40.24
        fixed point); pixel: int32 device tensor [n]; r, g, b: float32 device tensors [n]. Enqueues on `stream`."""
        n_pixels = int(film.numel() // 3)
        _check(self._L.pgpu_film_add(self._h, capi.ptr(film), n_pixels, capi.ptr(pixel), capi.ptr(r), capi.ptr(g), capi.ptr(b), int(pixel.numel()), stream), self._h)

    def film_resolve(self, film, out, stream: int = 0):
        """Sensor::Pixel -> vec3 (sensor.cpp:96-99): out (float32 device tensor [n_pixels, 3]) = film * 2^-24."""
        _check(self._L.pgpu_film_resolve(self._h, capi.ptr(film), int(film.numel() // 3), capi.ptr(out), stream), self._h)

    def sync(self, stream: int = 0):
        _check(self._L.pgpu_sync(self._h, stream), self._h)

    def last_kernel_ms(self) -> float:
        return float(self._L.pgpu_last_kernel_ms(self._h))

    def set_stats(self, on: bool):
        _check(self._L.pgpu_set_stats(self._h, int(on)), self._h)

    def set_tuning(self, blocks_per_cu: int, chunk: int):
        _check(self._L.pgpu_set_tuning(self._h, blocks_per_cu, chunk), self._h)

    def set_kernel_variant(self, refill_min: int = 0, stack_window: int = 0, leaf_min: int = 0):
        _check(self._L.pgpu_set_kernel_variant(self._h, refill_min, stack_window, leaf_min), self._h)

    def set_ray_layout(self, lanes_per_ray: int):
        """4: quad kernel (16 rays per wave); 2: pair kernel (32 rays per wave)."""
        _check(self._L.pgpu_set_ray_layout(self._h, lanes_per_ray), self._h)

    def counters(self) -> dict:
        cnt = capi.Counters()
        _check(self._L.pgpu_get_counters(self._h, C.byref(cnt)), self._h)
        return {f: int(getattr(cnt, f)) for f, _ in capi.Counters._fields_}


@dataclass
class TwoLevelScene:
    """Host description of the two-level structure the reference's
    OfflineBatchingAccelerationStructureBuilder produces (offline_batching...h:307-386): batching points
    (triangle groups), one bottom-level BVH8 and optionally one SVDAG per point, a BVH4 over the points."""
    groups: list            # per input leaf: uint32 triangle ids
    bvhs: list              # per input leaf: HostBvh8
    top: HostBvh4
    dags: HostSvdags | None
    leaf_bounds: np.ndarray
    bvh_ids: "np.ndarray | None" = None  # per input leaf: the id of its BVH8 when the trees were built on (and never left) a context's GPU

    @property
    def n_leaves(self) -> int:
        return len(self.groups)

    def fetch_bvhs(self, ctx: "Context"):
        """Trees that were built on the device (build_two_level(..., resident_ctx=ctx)) brought back as HostBvh8s, for checks
        and the on-disk format."""
        self.bvhs = [ctx.download_bvh8(int(i), g) if int(i) != capi.NO_BVH else None for i, g in zip(self.bvh_ids, self.groups)]
        return self

    def prim_ids_of(self, leaf: np.ndarray, obj: np.ndarray) -> np.ndarray:
        """Mesh triangle id of every hit record (leaf slot, leaf-object index); MISS stays MISS."""
        leaf, obj = np.asarray(leaf).view(np.uint32), np.asarray(obj).view(np.uint32)
        out = np.full(obj.shape, capi.MISS, np.uint32)
        hit = np.nonzero(obj != np.uint32(capi.MISS))[0]
        # (leaf object j of a point = its j-th triangle) -- one gather through the groups laid end to end
        sizes = np.array([g.size for g in self.groups], np.int64)
        start = np.concatenate([[0], np.cumsum(sizes)])
        cat = np.concatenate(self.groups) if self.groups else np.zeros(0, np.uint32)
        group = np.asarray(self.top.leaf_order, np.int64)[leaf[hit].astype(np.int64)]
        out[hit] = cat[start[group] + obj[hit].astype(np.int64)]
        return out

    def device_bytes(self, n_verts: int = 0) -> int:
        """Bytes of the structure as the traversal reads it: 256-byte nodes, leaf indices, triangles, top level, SVDAG words."""
        b = sum(256 * h.n_nodes + 4 * h.leaf_idx.size + 12 * h.prim_ids.size for h in self.bvhs if h is not None)
        b += 112 * int(self.top.nodes.shape[0]) + 12 * n_verts
        return b + (4 * int(self.dags.words.size) if self.dags is not None else 0)

    def sub_mesh(self, pos, idx, resident):
        """The part of the mesh the batching points marked in `resident` (bool array over leaf slots) need: (pos, idx, remap)
        with remap[triangle id of the whole mesh] = triangle id in the part (MISS elsewhere). A rank of a sharded scene uploads
        this instead of the whole mesh and hands `remap` to upload()."""
        pos = np.asarray(pos, np.float32).reshape(-1, 3)
        idx = np.asarray(idx, np.uint32).reshape(-1, 3)
        order = np.asarray(self.top.leaf_order, np.int64)
        mine = [self.groups[int(order[s])] for s in np.nonzero(np.asarray(resident, bool))[0]]
        tri = np.concatenate(mine) if mine else np.zeros(0, np.uint32)
        remap = np.full(idx.shape[0], capi.MISS, np.uint32)
        remap[tri] = np.arange(tri.size, dtype=np.uint32)
        part = idx[tri]
        used = np.zeros(pos.shape[0], bool)
        used[part.ravel()] = True
        new_id = (np.cumsum(used, dtype=np.int64) - 1).astype(np.uint32)
        return np.ascontiguousarray(pos[used]), np.ascontiguousarray(new_id[part]), remap

    def upload(self, ctx: "Context", mesh_id: int, resident=None, streamed: bool = False, prim_remap=None):
        """Uploads BVHs, SVDAGs and the top level. Returns bvh ids in final leaf-slot order.

        resident: optional bool array over leaf slots -- only those batching points' bottom-level BVHs are
        uploaded (leaves sharded over several GPUs); the others are marked NO_BVH. The top level and the
        SVDAGs are small and always uploaded whole."""
        slot_of = np.empty(self.n_leaves, np.int64)
        slot_of[np.asarray(self.top.leaf_order, np.int64)] = np.arange(self.n_leaves)
        if prim_remap is not None and streamed:
            raise ValueError("prim_remap is for resident uploads")
        put = ctx.stage_bvh8 if streamed else (lambda b, m: ctx.upload_bvh8(b, m, prim_remap))  # streamed: leaves live in pinned host memory (config 5)
        if self.bvh_ids is not None:  # built on this context's GPU and already part of its BVH table
            if streamed or resident is not None:
                raise ValueError("trees built on the device are resident as they are")
            ids = [int(i) for i in self.bvh_ids]
        else:
            ids = [put(b, mesh_id) if (b is not None and (resident is None or bool(resident[slot_of[i]]))) else capi.NO_BVH for i, b in enumerate(self.bvhs)]
        if self.dags is not None:
            ctx.upload_svdags(self.dags)
        leaves = np.empty((self.n_leaves, 2), np.uint32)
        for slot, src in enumerate(self.top.leaf_order):
            leaves[slot, 0] = ids[int(src)]
            leaves[slot, 1] = int(src) if self.dags is not None else capi.NO_SVDAG
        ctx.upload_top_bvh4(self.top, leaves)
        return leaves[:, 0].copy()


def build_two_level(pos, idx, prims_per_batching_point: int, svdag_res: int = 0, max_leaf: int = 4, build_group=None,
                    svdag_device_ctx: "Context | None" = None, bvh_device_ctx: "Context | None" = None, resident_ctx: "Context | None" = None, cut_ctx: "Context | None" = None) -> TwoLevelScene:
    """build_group: optional callable (group sizes: int array[, top: HostBvh4]) -> bool array; only those batching points get a
    bottom-level BVH8 here (the others stay None: they live on another rank). A callable that takes two arguments also gets the
    top level, which is built first (from the groups' bounds) so that ownership can follow its leaf order. The cut, the top
    level and the SVDAGs are always whole. svdag_device_ctx: build the SVDAGs on that context's GPU instead of the host;
    bvh_device_ctx: likewise the bottom-level BVH8s (LBVH quality, see pgpu_build_bvh8_device); resident_ctx: the BVH8s are built on
    that context's GPU and stay there (pgpu_build_upload_bvh8_device: `bvh_ids` instead of `bvhs`; upload() then only adds the
    SVDAGs and the top level, fetch_bvhs() brings the trees back when the host wants them); cut_ctx: the cut into batching points on
    that context's GPU as well (pgpu_partition_prims_device)."""
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    group_of, ng = partition_prims(pos, idx, prims_per_batching_point) if cut_ctx is None else partition_prims_device(cut_ctx, pos, idx, prims_per_batching_point)
    order = np.argsort(group_of, kind="stable").astype(np.uint32)
    counts = np.bincount(group_of, minlength=ng)
    starts = np.concatenate([[0], np.cumsum(counts)])
    groups = [np.ascontiguousarray(order[starts[g]:starts[g + 1]]) for g in range(ng)]
    # bounds per group: the triangles' own bounds reduced over the groups laid end to end (one pass instead of a gather per group)
    tri = pos[idx[order]]  # (n, 3 vertices, 3)
    bounds = np.empty((ng, 6), np.float32)
    nonempty = counts > 0
    at = starts[:-1][nonempty]
    bounds[nonempty, :3] = np.minimum.reduceat(tri.min(axis=1), at, axis=0)
    bounds[nonempty, 3:] = np.maximum.reduceat(tri.max(axis=1), at, axis=0)
    del tri
    top = build_top_bvh4(bounds)
    # the per-group builds are independent C calls (ctypes releases the GIL): run them on a few threads
    workers = max(1, min(host_threads(), len(groups)))
    if build_group is None:
        want = np.ones(ng, bool)
    else:
        import inspect
        two = len(inspect.signature(build_group).parameters) >= 2
        want = np.asarray(build_group(counts, top) if two else build_group(counts), bool)
    build = lambda gi: build_bvh8(pos, idx, groups[gi], max_leaf) if want[gi] else None
    bvh_ids = None
    if resident_ctx is not None:
        sel = [gi for gi in range(ng) if want[gi]]
        ids, _ = resident_ctx.build_upload_bvh8_device(pos, idx, [groups[gi] for gi in sel], max_leaf) if sel else (np.zeros(0, np.uint32), 0)
        bvh_ids = np.full(ng, capi.NO_BVH, np.uint32)
        bvh_ids[sel] = ids
        bvhs = [None] * ng
    elif bvh_device_ctx is not None:  # every wanted batching point's BVH8 in one pass over the GPU (pgpu_build_bvh8_device)
        sel = [gi for gi in range(ng) if want[gi]]
        built, _ = build_bvh8_device(bvh_device_ctx, pos, idx, [groups[gi] for gi in sel], max_leaf) if sel else ([], 0)
        bvhs = [None] * ng
        for gi, b in zip(sel, built):
            bvhs[gi] = b
    elif workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            bvhs = list(pool.map(build, range(ng)))
    else:
        bvhs = [build(gi) for gi in range(ng)]
    dags = build_svdags(pos, idx, groups, svdag_res, device_ctx=svdag_device_ctx) if svdag_res > 0 else None
    return TwoLevelScene(groups, bvhs, top, dags, bounds, bvh_ids)


def svdags_concat(parts: list) -> "tuple[np.ndarray, list[np.ndarray]]":
    """Word arrays of DAG sets built apart, laid one behind the other (pgpu_svdag_rebase): -> (words, per part its descriptors
    with the roots moved along). parts: [(words uint32[], descs SVDAG_DESC_DTYPE[])]."""
    L = capi.lib()
    words, descs, base = [], [], 0
    for w, d in parts:
        w = np.ascontiguousarray(w, np.uint32).copy()
        d = np.asarray(d, SVDAG_DESC_DTYPE).copy()
        _check(L.pgpu_svdag_rebase(w.ctypes.data, w.size, base), what="pgpu_svdag_rebase failed")
        d["root"] += np.uint32(base)
        words.append(w)
        descs.append(d)
        base += int(w.size)
    return (np.concatenate(words) if words else np.zeros(0, np.uint32)), descs


def tiles_local_part(tiles, mine, svdag_res: int = 0, max_leaf: int = 4, svdag_device_ctx: "Context | None" = None) -> dict:
    """One rank's share of a two-level build over a tiled scene (scenes.CityTiles: a tile is a batching point that can be made on
    its own). Makes the geometry of the tiles marked in `mine` and nothing else, and from it the tiles' bounds, bottom-level BVH8s
    and SVDAGs. -> {"tiles": ids, "pos", "idx": the local mesh (own tiles one behind the other), "groups": per own tile its
    triangle ids in the local mesh, "bvhs": per own tile, "bounds": float32[k, 6], "dag_words", "dag_descs", "words_before"}.
    "tiles", "bounds" and the three dag entries are what the other ranks need (tiles_assemble)."""
    from . import scenes
    ids = np.nonzero(np.asarray(mine, bool))[0]
    meshes = [tiles.tile_mesh(int(t)) for t in ids]
    groups, base = [], 0
    for m in meshes:
        groups.append(np.arange(base, base + m.n_tris, dtype=np.uint32))
        base += m.n_tris
    if meshes:
        local = scenes.merge(meshes)
        pos, idx = local.pos, local.idx
    else:
        pos, idx = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32)
    bounds = np.empty((len(meshes), 6), np.float32)
    for k, m in enumerate(meshes):
        bounds[k, :3], bounds[k, 3:] = m.pos.min(axis=0), m.pos.max(axis=0)
    del meshes
    workers = max(1, min(host_threads(), len(groups)))
    build = lambda g: build_bvh8(pos, idx, g, max_leaf)
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            bvhs = list(pool.map(build, groups))
    else:
        bvhs = [build(g) for g in groups]
    part = {"tiles": ids, "pos": pos, "idx": idx, "groups": groups, "bvhs": bvhs, "bounds": bounds,
            "dag_words": None, "dag_descs": None, "words_before": 0}
    if svdag_res > 0 and groups:
        d = build_svdags(pos, idx, groups, svdag_res, device_ctx=svdag_device_ctx)
        part.update(dag_words=d.words, dag_descs=d.descs, words_before=d.words_before)
    return part


def tiles_assemble(n_tiles: int, shared: list, own: dict) -> TwoLevelScene:
    """The whole two-level structure from every rank's shared entries (tiles_local_part; in rank order, the same list on every rank)
    and this rank's own part: top level over all tiles' bounds, all SVDAGs, BVH8s and triangle groups of the own tiles only."""
    bounds = np.empty((n_tiles, 6), np.float32)
    seen = np.zeros(n_tiles, bool)
    for sh in shared:
        t = np.asarray(sh["tiles"], np.int64)
        if seen[t].any():
            raise ValueError("a tile was built by two ranks")
        seen[t] = True
        bounds[t] = sh["bounds"]
    if not seen.all():
        raise ValueError("tiles nobody built: %s" % np.nonzero(~seen)[0][:8])
    dags = None
    with_dags = [sh for sh in shared if sh["dag_words"] is not None]
    if with_dags:
        if sum(len(sh["tiles"]) for sh in with_dags) != n_tiles:
            raise ValueError("SVDAGs for some tiles only")
        words, descs = svdags_concat([(sh["dag_words"], sh["dag_descs"]) for sh in with_dags])
        all_descs = np.zeros(n_tiles, SVDAG_DESC_DTYPE)
        for sh, d in zip(with_dags, descs):
            all_descs[np.asarray(sh["tiles"], np.int64)] = d
        dags = HostSvdags(words, all_descs, int(sum(sh["words_before"] for sh in with_dags)))
    groups = [np.zeros(0, np.uint32)] * n_tiles
    bvhs = [None] * n_tiles
    for k, t in enumerate(np.asarray(own["tiles"], np.int64)):
        groups[int(t)], bvhs[int(t)] = own["groups"][k], own["bvhs"][k]
    return TwoLevelScene(groups, bvhs, build_top_bvh4(bounds), dags, bounds)


TILE_SHARED_KEYS = ("tiles", "bounds", "dag_words", "dag_descs", "words_before")


@dataclass
class InstancedTwoLevelScene:
    """A scene graph with plain triangles and instanced prototypes, cut into batching points the way the reference's builder
    does it (OfflineBatchingAccelerationStructureBuilder, offline_batching_acceleration_structure.cpp:38-57 and
    batching.cpp:38-265, :358-422): shapes larger than primitivesPerBatchingPoint / 8 are split, the objects (sub-shapes and
    instances) are grouped by an object-level BVH2 cut, and every group gets one bottom-level BVH8 whose leaf objects are its
    triangles and its instances (LRUBVHSceneCache::createBVH, offline_bvh_cache.cpp:273-322)."""
    pos: np.ndarray
    idx: np.ndarray
    protos: list                 # [(pos, idx)] prototype meshes in their own space
    proto_bvhs: list             # HostBvh8 per prototype
    inst_proto: np.ndarray
    inst_matrix: np.ndarray      # [n, 4, 4] local-to-world
    group_tris: list             # per batching point: triangle ids (leaf objects 0 .. nt-1)
    group_insts: list            # per batching point: instance ids (leaf objects nt ..)
    bvhs: list                   # HostBvh8 over the leaf objects of every batching point
    top: HostBvh4 | None         # None: a single batching point (use it as a single-level scene)
    dags: HostSvdags | None
    leaf_bounds: np.ndarray

    @property
    def n_leaves(self) -> int:
        return len(self.bvhs)

    def hit_normals(self, leaf, obj, inst) -> np.ndarray:
        """World-space geometric normal [n, 3] of every hit record (leaf slot, leaf-object index, instance leaf object or MISS)
        of a trace through this scene; rows of misses are zero. A hit inside an instance is a triangle of the prototype: its
        normal goes to world space with the inverse transpose of the instance's matrix."""
        leaf, obj, inst = (np.asarray(a).view(np.uint32).astype(np.int64) for a in (leaf, obj, inst))
        out = np.zeros((obj.size, 3), np.float64)
        miss = int(capi.MISS)
        order = np.asarray(self.top.leaf_order, np.int64) if self.top is not None else np.zeros(1, np.int64)
        group = order[np.minimum(leaf, order.size - 1)]
        n_tris = np.array([t.size for t in self.group_tris], np.int64)
        n_inst = np.array([t.size for t in self.group_insts], np.int64)
        tri_cat = np.concatenate(self.group_tris + [np.zeros(0, np.uint32)]).astype(np.int64)
        inst_cat = np.concatenate(self.group_insts + [np.zeros(0, np.int64)]).astype(np.int64)
        tri_start = np.concatenate([[0], np.cumsum(n_tris)])
        inst_start = np.concatenate([[0], np.cumsum(n_inst)])

        def tri_normals(pos, idx, prim):
            t = idx[prim].astype(np.int64)
            p0, p1, p2 = (pos[t[:, c]].astype(np.float64) for c in range(3))
            return np.cross(p1 - p0, p2 - p0)

        plain = np.nonzero((obj != miss) & (inst == miss))[0]
        out[plain] = tri_normals(self.pos, self.idx, tri_cat[tri_start[group[plain]] + obj[plain]])
        nested = np.nonzero((obj != miss) & (inst != miss))[0]
        if nested.size:
            k = inst_cat[inst_start[group[nested]] + (inst[nested] - n_tris[group[nested]])]
            w2l = world_to_local_rows_many(self.inst_matrix[k]).reshape(-1, 3, 4).astype(np.float64)
            proto = np.asarray(self.inst_proto, np.int64)[k]
            for q in np.unique(proto):
                sel = proto == q
                ppos, pidx = self.protos[int(q)]
                local = tri_normals(ppos.reshape(-1, 3), pidx.reshape(-1, 3), self.proto_bvhs[int(q)].prim_ids[obj[nested[sel]]])
                out[nested[sel]] = np.einsum("nij,ni->nj", w2l[sel][:, :, :3], local)
        return out

    def upload(self, ctx: "Context", streamed: bool = False, resident=None):
        """Prototypes are uploaded resident (they are shared by every batching point); the batching points' own BVHs are
        uploaded, or staged in host memory when `streamed`. Returns bvh ids per batching point (input order).
        resident: optional bool array over leaf slots (batching points sharded over several GPUs): only those points' BVHs
        are uploaded here, the others are NO_BVH; every rank holds all prototypes."""
        mesh_id = ctx.upload_mesh(self.pos, self.idx)
        proto_ids = np.array([ctx.upload_bvh8(b, ctx.upload_mesh(p, i)) for b, (p, i) in zip(self.proto_bvhs, self.protos)], np.uint32)
        w2l = world_to_local_rows_many(self.inst_matrix)
        slot_of = None
        if resident is not None:
            slot_of = np.empty(self.n_leaves, np.int64)
            slot_of[np.asarray(self.top.leaf_order, np.int64)] = np.arange(self.n_leaves)
        ids = []
        for g, bvh in enumerate(self.bvhs):
            if resident is not None and not bool(resident[slot_of[g]]):
                ids.append(capi.NO_BVH)
                continue
            tris, insts = self.group_tris[g], self.group_insts[g]
            objs = np.empty((tris.size + insts.size, 2), np.uint32)
            objs[:tris.size, 0] = mesh_id
            objs[:tris.size, 1] = tris
            objs[tris.size:, 0] = capi.INSTANCE
            objs[tris.size:, 1] = np.arange(insts.size)
            table = (proto_ids[self.inst_proto[insts]], w2l[insts])
            ids.append(ctx.upload_bvh8_objects(bvh, objs, table, staged=streamed))
        if self.top is not None:
            if self.dags is not None:
                ctx.upload_svdags(self.dags)
            leaves = np.empty((self.n_leaves, 2), np.uint32)
            for slot, src in enumerate(self.top.leaf_order):
                leaves[slot, 0] = ids[int(src)]
                leaves[slot, 1] = int(src) if self.dags is not None else capi.NO_SVDAG
            ctx.upload_top_bvh4(self.top, leaves)
        return ids


def build_instanced_two_level(pos, idx, protos, inst_proto, inst_matrix, prims_per_batching_point: int, svdag_res: int = 0) -> InstancedTwoLevelScene:
    pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
    nt = idx.shape[0]
    inst_proto = np.asarray(inst_proto, np.int64)
    inst_matrix = np.asarray(inst_matrix, np.float32).reshape(-1, 4, 4)
    proto_bvhs = [build_bvh8(p, i) for p, i in protos]
    tri_lo = pos[idx].min(axis=1)
    tri_hi = pos[idx].max(axis=1)
    # splitLargeSceneObjects (batching.cpp:358-422, called with primitivesPerBatchingPoint / 8): the plain geometry in pieces of that size
    piece_of, n_pieces = partition_prims(pos, idx, max(1, prims_per_batching_point // 8))
    piece_bounds = np.empty((n_pieces, 6), np.float32)
    order = np.argsort(piece_of, kind="stable")
    counts = np.bincount(piece_of, minlength=n_pieces)
    starts = np.concatenate([[0], np.cumsum(counts)])
    pieces = [order[starts[p]:starts[p + 1]].astype(np.uint32) for p in range(n_pieces)]
    for p, tris in enumerate(pieces):
        piece_bounds[p, :3] = tri_lo[tris].min(axis=0)
        piece_bounds[p, 3:] = tri_hi[tris].max(axis=0)
    proto_lo = np.stack([np.asarray(p, np.float32).reshape(-1, 3).min(axis=0) for p, _ in protos]) if protos else np.zeros((0, 3), np.float32)
    proto_hi = np.stack([np.asarray(p, np.float32).reshape(-1, 3).max(axis=0) for p, _ in protos]) if protos else np.zeros((0, 3), np.float32)
    inst_bounds = transform_bounds_many(proto_lo[inst_proto], proto_hi[inst_proto], inst_matrix) if inst_proto.size else np.zeros((0, 6), np.float32)
    # createSubScenes (batching.cpp:38-265): cut over the objects. An instance weighs a fraction of its prototype: it adds
    # traversal work to a batching point but no geometry (own heuristic; the reference counts the instanced scene's primitives).
    obj_bounds = np.concatenate([piece_bounds, inst_bounds])
    proto_weight = np.array([max(1, np.asarray(i).reshape(-1, 3).shape[0] // 64) for _, i in protos], np.uint64)
    weights = np.concatenate([counts.astype(np.uint64), proto_weight[inst_proto] if inst_proto.size else np.zeros(0, np.uint64)])
    group_of, ng = partition_objects(obj_bounds, weights, prims_per_batching_point)
    # members of every group: its pieces' triangles (pieces ascending, triangles ascending inside a piece), then its instances (ascending)
    tri_group = group_of[:n_pieces][piece_of]
    tri_order = np.lexsort((np.arange(nt), piece_of, tri_group)).astype(np.uint32)
    tri_starts = np.concatenate([[0], np.cumsum(np.bincount(tri_group, minlength=ng))])
    inst_group = group_of[n_pieces:]
    inst_order = np.argsort(inst_group, kind="stable")
    inst_starts = np.concatenate([[0], np.cumsum(np.bincount(inst_group, minlength=ng))])
    group_tris, group_insts = [], []
    bounds = np.empty((ng, 6), np.float32)

    def build(g):
        tris = np.ascontiguousarray(tri_order[tri_starts[g]:tri_starts[g + 1]])
        insts = np.ascontiguousarray(inst_order[inst_starts[g]:inst_starts[g + 1]]).astype(np.int64)
        boxes = np.concatenate([np.concatenate([tri_lo[tris], tri_hi[tris]], axis=1), inst_bounds[insts]]).astype(np.float32)
        alone = np.concatenate([np.zeros(tris.size, np.uint8), np.ones(insts.size, np.uint8)])
        return tris, insts, build_bvh8_boxes(boxes, alone), np.concatenate([boxes[:, :3].min(axis=0), boxes[:, 3:].max(axis=0)])

    workers = max(1, min(host_threads(), ng))
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            built = list(pool.map(build, range(ng)))
    else:
        built = [build(g) for g in range(ng)]
    bvhs = []
    for g, (tris, insts, bvh, box) in enumerate(built):
        group_tris.append(tris)
        group_insts.append(insts)
        bvhs.append(bvh)
        bounds[g] = box
    top = build_top_bvh4(bounds) if ng >= 2 else None
    dags = None
    if svdag_res > 0 and top is not None:
        # the voxeliser sees instanced geometry in world space (TriangleShape::voxelize with the instance's transform,
        # triangle_intersect.cpp:245-249): bake each batching point's instances for this step only
        from . import scenes
        all_pos, all_idx, groups, base_v, base_t = [pos], [idx], [], pos.shape[0], nt
        for g in range(ng):
            ids = [group_tris[g].astype(np.uint32)]
            for k in group_insts[g]:
                baked = scenes.bake_instances([scenes.Mesh(*protos[int(inst_proto[k])])], [0], inst_matrix[k:k + 1])
                all_pos.append(baked.pos)
                all_idx.append(baked.idx + np.uint32(base_v))
                ids.append(np.arange(base_t, base_t + baked.n_tris, dtype=np.uint32))
                base_v += baked.n_verts
                base_t += baked.n_tris
            groups.append(np.concatenate(ids))
        dags = build_svdags(np.concatenate(all_pos), np.concatenate(all_idx), groups, svdag_res)
    return InstancedTwoLevelScene(pos, idx, [(np.asarray(p, np.float32), np.asarray(i, np.uint32)) for p, i in protos], proto_bvhs, inst_proto, inst_matrix,
                                  group_tris, group_insts, bvhs, top, dags, bounds)
