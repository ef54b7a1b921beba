"""ctypes binding of the CPU oracle (oracle/_build/libpandora_oracle.so) and, when present, of the
restatement that runs on the reference's own simd headers (oracle/_ref/libpandora_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libpandora_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libpandora_ref.so")
MISS = 0xFFFFFFFF


class Ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("tnear", C.c_float), ("tfar", C.c_float)]


class Mesh(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("idx", C.c_void_p)]


class Bvh8(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("leaf_idx", C.c_void_p), ("root", C.c_uint32), ("obj_prim", C.c_void_p), ("mesh", Mesh), ("inst", C.c_void_p)]


class Instance(C.Structure):
    """orc_instance: child BVH (NULL = the leaf object is a primitive) + world-to-local 3x4 (row-major)."""
    _fields_ = [("child", C.c_void_p), ("world_to_local", C.c_float * 12)]


class Svdag(C.Structure):
    _fields_ = [("data", C.c_void_p), ("root", C.c_uint32), ("bounds_min", C.c_float * 3), ("inv_extent", C.c_float), ("enabled", C.c_int)]


class Leaf(C.Structure):
    _fields_ = [("bvh", Bvh8), ("dag", Svdag)]


class Scene(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("root", C.c_uint32), ("leaves", C.c_void_p), ("n_leaves", C.c_uint32), ("n_nodes", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("svdag_tests", "svdag_culled", "leaf_visits", "inner_visits", "leaf_pops", "tri_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def ensure_built():
    if not os.path.exists(ORACLE_SO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "_build/libpandora_oracle.so"])


_orc = None
_ref = None


def oracle() -> C.CDLL:
    global _orc
    if _orc is None:
        ensure_built()
        _orc = C.CDLL(ORACLE_SO)
        _orc.orc_svdag_intersect.argtypes = [C.POINTER(Svdag), C.POINTER(Ray), C.c_int]
        _orc.orc_bvh4_step.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.POINTER(Ray), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p]
        _orc.orc_bvh4_step_ex.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.POINTER(Ray), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p, C.c_int]
        _orc.orc_tri_intersect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _orc.orc_bvh8_inner.argtypes = [C.c_void_p, C.POINTER(Ray), C.c_void_p, C.c_void_p]
        _orc.orc_bvh8_inner.restype = C.c_uint
        _orc.orc_hmin_index4.argtypes = [C.c_void_p]
        _orc.orc_hmin_index4.restype = C.c_uint
        _orc.orc_compress_perm8.argtypes = [C.c_uint, C.c_void_p]
        _orc.orc_bvh8_perm.argtypes = [C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p]
        _orc.orc_sign_shift.argtypes = [C.c_float, C.c_float, C.c_float]
        _orc.orc_sign_shift.restype = C.c_uint32
        _orc.orc_bvh8_trace.argtypes = [C.POINTER(Bvh8), C.c_int, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 5 + [C.POINTER(Counters), C.c_int]
        _orc.orc_scene_trace.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 7 + [C.POINTER(Counters), C.c_int]
        _orc.orc_bvh8_trace_inst.argtypes = [C.POINTER(Bvh8), C.c_int, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 6 + [C.POINTER(Counters), C.c_int]
        _orc.orc_scene_trace_inst.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 8 + [C.POINTER(Counters), C.c_int]
        _orc.orc_brute_trace.argtypes = [C.POINTER(Mesh), C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    return _orc


def have_ref() -> bool:
    return os.path.exists(REF_SO)


def ref() -> C.CDLL:
    global _ref
    if _ref is None:
        _ref = C.CDLL(REF_SO)
        _ref.ref_bvh8_inner.argtypes = [C.c_void_p, C.POINTER(Ray), C.c_void_p, C.c_void_p]
        _ref.ref_bvh8_inner.restype = C.c_uint
        _ref.ref_hmin_index4.argtypes = [C.c_void_p]
        _ref.ref_hmin_index4.restype = C.c_uint
        _ref.ref_hmin_index8.argtypes = [C.c_void_p]
        _ref.ref_hmin_index8.restype = C.c_uint
        _ref.ref_compress_perm8.argtypes = [C.c_uint, C.c_void_p]
        _ref.ref_permute8.argtypes = [C.c_void_p] * 3
        _ref.ref_compress8.argtypes = [C.c_void_p, C.c_uint, C.c_void_p]
        _ref.ref_minmax8.argtypes = [C.c_void_p] * 4
        _ref.ref_svdag_intersect.argtypes = [C.POINTER(Svdag), C.POINTER(Ray)]
        _ref.ref_bvh4_step.argtypes = [C.POINTER(Scene), C.c_int, C.POINTER(Ray), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p]
        _ref.ref_bvh8_trace.argtypes = [C.POINTER(Bvh8), C.c_int, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 5
        _ref.ref_scene_trace.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_void_p, C.c_size_t] + [C.c_void_p] * 7 + [C.POINTER(Counters)]
    return _ref


def make_ray(o, d, tnear=0.0, tfar=3.4028234663852886e38) -> Ray:
    return Ray((C.c_float * 3)(*o), (C.c_float * 3)(*d), tnear, tfar)


class OracleBvh8:
    """Keeps the numpy arrays alive behind an orc_bvh8."""

    def __init__(self, nodes, leaf_idx, root, prim_ids, pos, idx, instances=None):
        """instances: optional list over leaf objects of None (primitive) or (OracleBvh8 child, world_to_local 3x4)."""
        self.nodes = np.ascontiguousarray(nodes, np.float32)
        self.leaf_idx = np.ascontiguousarray(leaf_idx, np.uint32)
        self.prim_ids = np.ascontiguousarray(prim_ids, np.uint32)
        self.pos = np.ascontiguousarray(pos, np.float32)
        self.idx = np.ascontiguousarray(idx, np.uint32)
        self.inst = None
        if isinstance(instances, tuple):
            # (children: list of OracleBvh8, child index per leaf object int[n] (-1 = primitive), rows float32[n, 12]): the same
            # table filled through numpy, for scenes with 10^5+ instances
            children, which, rows = instances
            self.children = list(children)
            which = np.asarray(which, np.int64)
            table = np.zeros(which.size, np.dtype([("child", np.uint64), ("w2l", np.float32, (12,))], align=True))
            assert table.itemsize == C.sizeof(Instance)
            addr = np.array([C.addressof(c.c) for c in self.children] + [0], np.uint64)
            table["child"] = addr[np.where(which >= 0, which, len(self.children))]
            table["w2l"] = np.asarray(rows, np.float32).reshape(-1, 12)
            self.inst_np = table
            self.inst = (Instance * which.size).from_buffer(table)
        elif instances is not None:
            self.children = [i[0] for i in instances if i is not None]  # keep them alive
            self.inst = (Instance * len(instances))()
            for j, i in enumerate(instances):
                if i is not None:
                    self.inst[j].child = C.cast(C.pointer(i[0].c), C.c_void_p)
                    self.inst[j].world_to_local = (C.c_float * 12)(*[float(x) for x in np.asarray(i[1], np.float32).reshape(-1)[:12]])
        self.c = Bvh8(self.nodes.ctypes.data, self.leaf_idx.ctypes.data, int(root), self.prim_ids.ctypes.data,
                      Mesh(self.pos.ctypes.data, self.idx.ctypes.data), C.cast(self.inst, C.c_void_p) if self.inst is not None else None)

    @classmethod
    def from_host(cls, hb, mesh):
        return cls(hb.nodes, hb.leaf_idx, hb.root, hb.prim_ids, mesh.pos, mesh.idx)

    def trace(self, rays: np.ndarray, any_hit=False, tri_mode=0, threads=1, use_ref=False):
        n = rays.shape[1]
        rays = np.ascontiguousarray(rays, np.float32)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        obj, hit, inst = np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.full(n, MISS, np.uint32)
        cnt = Counters()
        if use_ref:
            assert self.inst is None, "the simd-backed restatement has no instance leaves"
            ref().ref_bvh8_trace(C.byref(self.c), int(any_hit), tri_mode, rays.ctypes.data, n, t.ctypes.data, obj.ctypes.data, u.ctypes.data, v.ctypes.data, hit.ctypes.data)
        else:
            oracle().orc_bvh8_trace_inst(C.byref(self.c), int(any_hit), tri_mode, rays.ctypes.data, n, t.ctypes.data, obj.ctypes.data, u.ctypes.data, v.ctypes.data,
                                         hit.ctypes.data, inst.ctypes.data, C.byref(cnt), threads)
        if any_hit:
            return {"hit": hit, "counters": cnt.as_dict()}
        return {"t": t, "obj": obj, "u": u, "v": v, "inst": inst, "counters": cnt.as_dict()}


class OracleScene:
    """Two-level scene: top BVH4 + per-leaf BVH8 + optional SVDAGs."""

    def __init__(self, top_nodes, root, leaf_bvhs: list[OracleBvh8], dag_words=None, dag_descs=None, leaf_dag=None):
        self.top_nodes = np.ascontiguousarray(top_nodes, np.float32)
        self.leaf_bvhs = leaf_bvhs
        self.dag_words = None if dag_words is None else np.ascontiguousarray(dag_words, np.uint32)
        self.leaves = (Leaf * len(leaf_bvhs))()
        for i, b in enumerate(leaf_bvhs):
            self.leaves[i].bvh = b.c
            di = None if leaf_dag is None else int(leaf_dag[i])
            if di is not None and di != MISS and self.dag_words is not None:
                d = dag_descs[di]
                self.leaves[i].dag = Svdag(self.dag_words.ctypes.data, int(d["root"]), (C.c_float * 3)(*[float(x) for x in d["bounds_min"]]), float(d["inv_extent"]), 1)
            else:
                self.leaves[i].dag = Svdag(0, 0, (C.c_float * 3)(0, 0, 0), 0.0, 0)
        self.c = Scene(self.top_nodes.ctypes.data, int(root), C.cast(self.leaves, C.c_void_p), len(leaf_bvhs), int(self.top_nodes.shape[0]))

    def trace(self, rays: np.ndarray, any_hit=False, tri_mode=0, lanes=3, threads=1, use_ref=False):
        n = rays.shape[1]
        rays = np.ascontiguousarray(rays, np.float32)
        t, u, v = (np.zeros(n, np.float32) for _ in range(3))
        obj, leaf, hit, visits = (np.zeros(n, np.uint32) for _ in range(4))
        inst = np.full(n, MISS, np.uint32)
        cnt = Counters()
        if use_ref:
            ref().ref_scene_trace(C.byref(self.c), int(any_hit), tri_mode, rays.ctypes.data, n, t.ctypes.data, obj.ctypes.data, leaf.ctypes.data,
                                  u.ctypes.data, v.ctypes.data, hit.ctypes.data, visits.ctypes.data, C.byref(cnt))
        else:
            oracle().orc_scene_trace_inst(C.byref(self.c), int(any_hit), tri_mode, lanes, rays.ctypes.data, n, t.ctypes.data, obj.ctypes.data, leaf.ctypes.data,
                                          u.ctypes.data, v.ctypes.data, hit.ctypes.data, visits.ctypes.data, inst.ctypes.data, C.byref(cnt), threads)
        out = {"visits": visits, "counters": cnt.as_dict()}
        if any_hit:
            out["hit"] = hit
        else:
            out.update({"t": t, "obj": obj, "leaf": leaf, "u": u, "v": v, "inst": inst})
        return out


def brute(mesh, rays: np.ndarray, tri_mode=0, threads=4):
    n = rays.shape[1]
    rays = np.ascontiguousarray(rays, np.float32)
    pos = np.ascontiguousarray(mesh.pos, np.float32)
    idx = np.ascontiguousarray(mesh.idx, np.uint32)
    m = Mesh(pos.ctypes.data, idx.ctypes.data)
    t, prim = np.zeros(n, np.float32), np.zeros(n, np.uint32)
    oracle().orc_brute_trace(C.byref(m), idx.shape[0], tri_mode, rays.ctypes.data, n, t.ctypes.data, prim.ctypes.data, threads)
    return t, prim


def film_add(film: np.ndarray, pixel, r, g, b) -> None:
    """orc_film_add on a uint64 array [n_pixels, 3] (in place)."""
    assert film.dtype == np.uint64 and film.flags.c_contiguous
    pixel = np.ascontiguousarray(pixel, np.uint32)
    r, g, b = (np.ascontiguousarray(x, np.float32) for x in (r, g, b))
    f = oracle().orc_film_add
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    f.restype = None
    f(film.ctypes.data, film.size // 3, pixel.ctypes.data, r.ctypes.data, g.ctypes.data, b.ctypes.data, pixel.size)


def film_resolve(film: np.ndarray) -> np.ndarray:
    out = np.zeros(film.shape, np.float32)
    f = oracle().orc_film_resolve
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    f.restype = None
    f(np.ascontiguousarray(film).ctypes.data, film.size // 3, out.ctypes.data)
    return out


# ---------------------------------------------------------------------------- benchmark integrator pieces (N4)
class OrcCamera(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in ("lens_center", "lens_du", "lens_dv", "screen_center", "screen_du", "screen_dv")]


class OrcPaths(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("rng_state", "rng_inc", "pixel", "wr", "wg", "wb")]


def new_paths(n):
    p = {"rng_state": np.zeros(n, np.uint64), "rng_inc": np.zeros(n, np.uint64), "pixel": np.zeros(n, np.uint32)}
    p.update({k: np.zeros(n, np.float32) for k in ("wr", "wg", "wb")})
    return p


def _paths_struct(p):
    return OrcPaths(*[p[k].ctypes.data for k in ("rng_state", "rng_inc", "pixel", "wr", "wg", "wb")])


def camera_rays(cam, width, height, spp, seed, first, n):
    """orc_camera_rays; `cam` = a pandora_amd.capi.Camera (same layout)."""
    rays = np.zeros((8, n), np.float32)
    paths = new_paths(n)
    oc = OrcCamera.from_buffer_copy(bytes(cam))
    f = oracle().orc_camera_rays
    f.argtypes = [C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p, C.POINTER(OrcPaths)]
    f.restype = None
    ps = _paths_struct(paths)
    f(C.byref(oc), width, height, spp, seed, first, n, rays.ctypes.data, C.byref(ps))
    return rays, paths


def lambert_bounce(bvh: "OracleBvh8 | None", scene: "OracleScene | None", rays, hits: dict, albedo, paths: dict):
    """orc_lambert_bounce: updates `paths` in place, returns the bounce rays."""
    n = rays.shape[1]
    rays = np.ascontiguousarray(rays, np.float32)
    out = np.zeros((8, n), np.float32)
    f = oracle().orc_lambert_bounce
    f.argtypes = [C.c_void_p, C.c_void_p] + [C.c_void_p] * 6 + [C.c_size_t, C.POINTER(C.c_float), C.c_void_p, C.POINTER(OrcPaths)]
    f.restype = None
    g = lambda k, dt: np.ascontiguousarray(hits[k], dt) if k in hits else np.zeros(n, dt)
    t, obj, leaf, u, v = g("t", np.float32), g("obj", np.uint32), g("leaf", np.uint32), g("u", np.float32), g("v", np.float32)
    ps = _paths_struct(paths)
    f(C.byref(bvh.c) if bvh is not None else None, C.byref(scene.c) if scene is not None else None, rays.ctypes.data, t.ctypes.data, obj.ctypes.data,
      leaf.ctypes.data, u.ctypes.data, v.ctypes.data, n, (C.c_float * 3)(*[float(x) for x in albedo]), out.ctypes.data, C.byref(ps))
    return out


def splat_misses(film: np.ndarray, obj, paths: dict, sky):
    f = oracle().orc_splat_misses
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(OrcPaths), C.c_size_t, C.POINTER(C.c_float)]
    f.restype = None
    obj = np.ascontiguousarray(obj, np.uint32)
    ps = _paths_struct(paths)
    f(film.ctypes.data, film.size // 3, obj.ctypes.data, C.byref(ps), obj.size, (C.c_float * 3)(*[float(x) for x in sky]))


def scene_from_two_level(tl, mesh) -> OracleScene:
    """OracleScene for a pandora_amd.accel.TwoLevelScene (leaves in final slot order)."""
    leaf_bvhs = [OracleBvh8.from_host(tl.bvhs[int(src)], mesh) for src in tl.top.leaf_order]
    if tl.dags is not None:
        return OracleScene(tl.top.nodes, tl.top.root, leaf_bvhs, tl.dags.words, tl.dags.descs, [int(s) for s in tl.top.leaf_order])
    return OracleScene(tl.top.nodes, tl.top.root, leaf_bvhs)


def bvhs_from_instanced(tl):
    """One OracleBvh8 per batching point of a pandora_amd.accel.InstancedTwoLevelScene (input order): leaf objects = the
    point's triangles, then its instances."""
    import pandora_amd as pa
    protos = [OracleBvh8(b.nodes, b.leaf_idx, b.root, b.prim_ids, p, i) for b, (p, i) in zip(tl.proto_bvhs, tl.protos)]
    out = []
    w2l = pa.accel.world_to_local_rows_many(tl.inst_matrix)
    for g, bvh in enumerate(tl.bvhs):
        tris, insts = tl.group_tris[g], tl.group_insts[g]
        prim_of_obj = np.concatenate([tris, np.zeros(insts.size, np.uint32)]).astype(np.uint32)
        which = np.concatenate([np.full(tris.size, -1, np.int64), np.asarray(tl.inst_proto, np.int64)[insts]])
        rows = np.concatenate([np.zeros((tris.size, 12), np.float32), w2l[insts]])
        out.append(OracleBvh8(bvh.nodes, bvh.leaf_idx, bvh.root, prim_of_obj, tl.pos, tl.idx, instances=(protos, which, rows)))
    return out


def scene_from_instanced(tl) -> OracleScene:
    bvhs = bvhs_from_instanced(tl)
    leaf_bvhs = [bvhs[int(src)] for src in tl.top.leaf_order]
    if tl.dags is not None:
        return OracleScene(tl.top.nodes, tl.top.root, leaf_bvhs, tl.dags.words, tl.dags.descs, [int(s) for s in tl.top.leaf_order])
    return OracleScene(tl.top.nodes, tl.top.root, leaf_bvhs)


# ---------------------------------------------------------------------------- golden fixtures
GOLDEN = os.path.join(ROOT, "tests", "golden")


class _Mesh:
    def __init__(self, pos, idx):
        self.pos, self.idx = pos, idx


class _HostBvh8:
    def __init__(self, nodes, leaf_idx, root, prim_ids):
        self.nodes, self.leaf_idx, self.root, self.prim_ids = nodes, leaf_idx, int(root), prim_ids
        self.n_nodes = int(nodes.shape[0])


def load_golden_bvh8():
    z = np.load(os.path.join(GOLDEN, "bvh8_small.npz"))
    mesh = _Mesh(z["pos"], z["idx"])
    hb = _HostBvh8(z["nodes"], z["leaf_idx"], z["root"], z["prim_ids"])
    return z, mesh, hb, OracleBvh8.from_host(hb, mesh)


class _Top:
    def __init__(self, nodes, root, leaf_order):
        self.nodes, self.root, self.leaf_order = nodes, int(root), leaf_order


class _Dags:
    def __init__(self, words, descs):
        self.words, self.descs = words, descs


class _TwoLevel:
    def __init__(self, z):
        self.n_leaves = int(z["n_leaves"])
        self.bvhs = [_HostBvh8(z[f"leaf{i}_nodes"], z[f"leaf{i}_leaf_idx"], z[f"leaf{i}_root"], z[f"leaf{i}_prim_ids"]) for i in range(self.n_leaves)]
        self.top = _Top(z["top_nodes"], z["top_root"], z["leaf_order"])
        self.dags = _Dags(z["dag_words"], z["dag_descs"])


def load_golden_two_level():
    z = np.load(os.path.join(GOLDEN, "two_level_small.npz"))
    mesh = _Mesh(z["pos"], z["idx"])
    tl = _TwoLevel(z)
    return z, mesh, tl, scene_from_two_level(tl, mesh)


def svdag_trace(words, desc, rays, lanes=3, use_ref=False):
    words = np.ascontiguousarray(words, np.uint32)
    dag = Svdag(words.ctypes.data, int(desc["root"]), (C.c_float * 3)(*[float(x) for x in desc["bounds_min"]]), float(desc["inv_extent"]), 1)
    out = np.zeros(rays.shape[1], np.uint8)
    for i in range(rays.shape[1]):
        ray = make_ray(rays[0:3, i], rays[3:6, i], float(rays[6, i]), float(rays[7, i]))
        out[i] = ref().ref_svdag_intersect(C.byref(dag), C.byref(ray)) if use_ref else oracle().orc_svdag_intersect(C.byref(dag), C.byref(ray), lanes)
    return out
