"""CPU engine for pandora_amd.sharded.trace_sharded built on the oracle (TEST INFRASTRUCTURE ONLY): lets the
multi-rank routing loop run on gloo/CPU ranks. Mirrors pandora_amd.sharded.GpuEngine's interface."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

import oracle_binding as ob

MISS = 0xFFFFFFFF


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("obj", C.c_uint32), ("leaf", C.c_uint32), ("u", C.c_float), ("v", C.c_float)]


class OracleEngine:
    def __init__(self, scene: ob.OracleScene, resident, lanes: int = 3, tri_mode: int = 0):
        self.scene, self.resident, self.lanes, self.tri_mode = scene, np.asarray(resident, bool), lanes, tri_mode
        self.L = ob.oracle()
        self.L.orc_bvh8_intersect.argtypes = [C.POINTER(ob.Bvh8), C.c_int, C.POINTER(ob.Ray), C.POINTER(Hit), C.POINTER(ob.Counters)]
        self.L.orc_bvh8_intersect_any.argtypes = [C.POINTER(ob.Bvh8), C.c_int, C.POINTER(ob.Ray), C.POINTER(ob.Counters)]
        self.counters = ob.Counters()

    def init_state(self, rays, kind):
        n = rays.shape[1]
        st = {"node": torch.full((n,), int(self.scene.c.root), dtype=torch.int32), "stack": torch.full((n,), -1, dtype=torch.int64),
              "visits": torch.zeros(n, dtype=torch.int32)}
        if kind == 0:
            st.update(t=rays[7].clone(), obj=torch.full((n,), -1, dtype=torch.int32), leaf=torch.zeros(n, dtype=torch.int32),
                      u=torch.zeros(n), v=torch.zeros(n))
        else:
            st["hit"] = torch.zeros(n, dtype=torch.int32)
        return st

    def top_episode(self, kind, rays, active, st, may_park=True):
        r = rays.numpy()
        parked = torch.full((active.numel(),), -1, dtype=torch.int32)
        for i, ray in enumerate(active.tolist()):
            if kind == 1 and int(st["hit"][ray]) != 0:
                continue
            tfar = float(st["t"][ray]) if kind == 0 else float(r[7, ray])
            cr = ob.make_ray(r[0:3, ray], r[3:6, ray], float(r[6, ray]), tfar)
            node = C.c_uint32(int(st["node"][ray]) & 0xFFFFFFFF)
            stack = C.c_uint64(int(st["stack"][ray]) & 0xFFFFFFFFFFFFFFFF)
            leaf = C.c_uint32(0)
            if self.L.orc_bvh4_step_ex(C.byref(self.scene.c), kind, self.lanes, C.byref(cr), C.byref(node), C.byref(stack), C.byref(leaf), C.byref(self.counters), int(may_park)):
                st["node"][ray] = int(np.array(node.value, np.uint32).view(np.int32))
                st["stack"][ray] = int(np.array(stack.value, np.uint64).view(np.int64))
                st["visits"][ray] += 1
                parked[i] = int(leaf.value)
        return parked

    def trace_leaves(self, kind, rays, work_leaf):
        r = rays.numpy()
        k = r.shape[1]
        t = torch.from_numpy(r[7].copy())
        obj = torch.full((k,), -1, dtype=torch.int32)
        u, v, hit = torch.zeros(k), torch.zeros(k), torch.zeros(k, dtype=torch.int32)
        for i in range(k):
            leaf = int(work_leaf[i])
            assert self.resident[leaf], "a ray was routed to a rank that does not own its batching point"
            cr = ob.make_ray(r[0:3, i], r[3:6, i], float(r[6, i]), float(r[7, i]))
            bvh = self.scene.leaf_bvhs[leaf].c
            if kind == 0:
                h = Hit()
                if self.L.orc_bvh8_intersect(C.byref(bvh), self.tri_mode, C.byref(cr), C.byref(h), None):
                    t[i], u[i], v[i] = h.t, h.u, h.v
                    obj[i] = int(np.array(h.obj, np.uint32).view(np.int32))
            else:
                hit[i] = 1 if self.L.orc_bvh8_intersect_any(C.byref(bvh), self.tri_mode, C.byref(cr), None) else 0
        return (t, obj, u, v) if kind == 0 else hit

    def finish(self):
        pass
