"""Synthetic geometry and ray sets for parity tests and the benchmark.

The assets BASELINE.json names (Crytek Sponza, San Miguel, Moana) are not available offline, so the
scenes are procedural stand-ins with matched triangle counts, generated in code so they exist on the
GPU box too. Ray sets follow the reference's own ray sources:
  * primary  : PerspectiveCamera::generateRay, src/graphics_core/perspective_camera.cpp:37-47 (lens radius 0)
  * diffuse  : cosine-weighted bounce (src/graphics_core/sampling.h:8-27,44-49) in the frame of
               BSDF::BSDF (src/graphics_core/material.cpp:5-20), origin offset by RAY_EPSILON along the
               normal (src/graphics_core/interaction.cpp:39-46, graphics_core/ray.h:12)
  * shadow   : hit point -> point on a ceiling light, tfar = distance (interaction.cpp:48-55)
Random numbers are PCG32 (src/samplers/rng/pcg.cpp:23-39), one stream per ray.
"""
from __future__ import annotations

import numpy as np

RAY_EPSILON = np.float32(0.000005)  # graphics_core/ray.h:12
FLT_MAX = np.float32(3.4028234663852886e38)


# ----------------------------------------------------------------------------- PCG32 (vectorised)
class Pcg32:
    """One independent PCG32 stream per element (pcg.cpp:23-39: setSequence + uniformU32)."""

    MULT = np.uint64(0x5851F42D4C957F2D)
    DEFAULT_STATE = np.uint64(0x853C49E6748FEA9B)

    def __init__(self, initseq: np.ndarray):
        initseq = np.asarray(initseq, dtype=np.uint64)
        self.state = np.zeros_like(initseq)
        self.inc = (initseq << np.uint64(1)) | np.uint64(1)
        self.u32()
        with np.errstate(over="ignore"):
            self.state = self.state + self.DEFAULT_STATE
        self.u32()

    def u32(self) -> np.ndarray:
        old = self.state
        with np.errstate(over="ignore"):
            self.state = old * self.MULT + self.inc
        xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xorshifted >> rot) | (xorshifted << ((~rot + np.uint32(1)) & np.uint32(31)))

    def f32(self) -> np.ndarray:
        one_minus_eps = np.float32(1.0) - np.finfo(np.float32).eps
        return np.minimum(one_minus_eps, self.u32().astype(np.float32) * np.float32(2.0**-32))


# ----------------------------------------------------------------------------- mesh helpers
class Mesh:
    def __init__(self, pos: np.ndarray, idx: np.ndarray):
        self.pos = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        self.idx = np.ascontiguousarray(idx, dtype=np.uint32).reshape(-1, 3)

    @property
    def n_tris(self) -> int:
        return int(self.idx.shape[0])

    @property
    def n_verts(self) -> int:
        return int(self.pos.shape[0])

    def bounds(self):
        return self.pos.min(axis=0), self.pos.max(axis=0)


def merge(meshes) -> Mesh:
    pos, idx, base = [], [], 0
    for m in meshes:
        pos.append(m.pos)
        idx.append(m.idx + np.uint32(base))
        base += m.n_verts
    return Mesh(np.concatenate(pos), np.concatenate(idx))


def grid_patch(origin, du, dv, nu: int, nv: int, displace=None) -> Mesh:
    """(nu x nv) quads spanning origin + s*du + t*dv; optional displacement f(s, t) -> (n,3) offsets."""
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1, dtype=np.float32), np.linspace(0, 1, nv + 1, dtype=np.float32), indexing="ij")
    s, t = s.ravel(), t.ravel()
    p = np.asarray(origin, np.float32)[None, :] + s[:, None] * np.asarray(du, np.float32)[None, :] + t[:, None] * np.asarray(dv, np.float32)[None, :]
    if displace is not None:
        p = p + displace(s, t).astype(np.float32)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel()
    b = a + (nv + 1)
    tris = np.concatenate([np.stack([a, b, b + 1], 1), np.stack([a, b + 1, a + 1], 1)])
    return Mesh(p, tris)


def box(lo, hi, n: int = 1) -> Mesh:
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    faces = [
        grid_patch(lo, ex, ey, n, n), grid_patch(lo + ez, ey, ex, n, n),
        grid_patch(lo, ey, ez, n, n), grid_patch(lo + ex, ez, ey, n, n),
        grid_patch(lo, ez, ex, n, n), grid_patch(lo + ey, ex, ez, n, n),
    ]
    return merge(faces)


def cylinder(base, radius: float, height: float, seg: int, stacks: int) -> Mesh:
    def disp(s, t):
        ang = s * np.float32(2 * np.pi)
        return np.stack([radius * np.cos(ang), np.zeros_like(s), radius * np.sin(ang)], 1)

    return grid_patch(base, (0, 0, 0), (0, height, 0), seg, stacks, displace=disp)


_UNIT_ICOSPHERES: dict = {}


def icosphere(center, radius: float, level: int) -> Mesh:
    if level not in _UNIT_ICOSPHERES:  # the subdivision is the same for every sphere of a level: do it once
        _UNIT_ICOSPHERES[level] = _unit_icosphere(level)
    v, f = _UNIT_ICOSPHERES[level]
    return Mesh(np.asarray(center)[None, :] + radius * v, f)


def _unit_icosphere(level: int):
    t = (1.0 + 5.0**0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
                  [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
                  [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    for _ in range(level):
        edges = {}
        verts = list(map(tuple, v))

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in edges:
                verts.append(tuple((np.array(verts[a]) + np.array(verts[b])) * 0.5))
                edges[key] = len(verts) - 1
            return edges[key]

        nf = []
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [[a, ab, ca], [b, bc, ab], [c, ca, bc], [ab, bc, ca]]
        v, f = np.array(verts), np.array(nf)
    v = v / np.linalg.norm(v, axis=1, keepdims=True)
    return v, f


def _noise(s, t, seed: int, octaves: int = 4):
    out = np.zeros_like(s, dtype=np.float64)
    rng = np.random.default_rng(seed)
    for o in range(octaves):
        f = 2.0**o * 3.0
        ph = rng.uniform(0, 2 * np.pi, 4)
        out += (np.sin(f * s * 2 * np.pi + ph[0]) * np.cos(f * t * 2 * np.pi + ph[1]) + np.sin(f * (s + t) * np.pi + ph[2])) / 2.0**o
    return out


# ----------------------------------------------------------------------------- scenes
def g_small() -> Mesh:
    """Cornell-style room (5 walls + 2 boxes) and an icosphere: ~1.3 k triangles (SURVEY G-small)."""
    room = [
        grid_patch((-1, 0, -1), (2, 0, 0), (0, 0, 2), 1, 1),  # floor
        grid_patch((-1, 2, -1), (0, 0, 2), (2, 0, 0), 1, 1),  # ceiling
        grid_patch((-1, 0, -1), (0, 2, 0), (2, 0, 0), 1, 1),  # back
        grid_patch((-1, 0, -1), (0, 0, 2), (0, 2, 0), 1, 1),  # left
        grid_patch((1, 0, -1), (0, 2, 0), (0, 0, 2), 1, 1),  # right
    ]
    return merge(room + [box((-0.7, 0, -0.6), (-0.1, 1.2, 0.0)), box((0.2, 0, 0.1), (0.7, 0.6, 0.6)), icosphere((0.45, 0.95, 0.35), 0.33, 3)])


def g_atrium(target_tris: int = 262_144, seed: int = 7) -> Mesh:
    """Sponza-class stand-in: a 30 x 12 x 18 atrium with two colonnades, arches, drapes and clutter.

    Triangle sizes span three orders of magnitude (coarse walls, fine drapes) like an architectural
    scene. The tessellation of the drapes is scaled so that the total lands near target_tris.
    """
    parts = []
    L, H, W = 30.0, 12.0, 18.0
    wall = 24
    parts += [
        grid_patch((0, 0, 0), (L, 0, 0), (0, 0, W), wall, wall),
        grid_patch((0, H, 0), (0, 0, W), (L, 0, 0), wall, wall),
        grid_patch((0, 0, 0), (0, H, 0), (L, 0, 0), wall, wall),
        grid_patch((0, 0, W), (L, 0, 0), (0, H, 0), wall, wall),
        grid_patch((0, 0, 0), (0, 0, W), (0, H, 0), wall, wall),
        grid_patch((L, 0, 0), (0, H, 0), (0, 0, W), wall, wall),
    ]
    # gallery floors on both long sides
    parts += [box((0, 5.6, 0), (L, 6.0, 4.0), 8), box((0, 5.6, W - 4.0), (L, 6.0, W), 8)]
    for row_z in (4.0, W - 4.0):
        for i in range(10):
            x = 1.5 + i * 3.0
            parts.append(cylinder((x, 0, row_z), 0.35, 5.6, 24, 12))
            parts.append(cylinder((x, 6.0, row_z), 0.25, 4.5, 20, 10))
            parts.append(box((x - 0.5, 0, row_z - 0.5), (x + 0.5, 0.3, row_z + 0.5), 2))
            if i < 9:  # arch between neighbouring columns (half torus as a displaced patch)
                def arch(s, t, x0=x, z0=row_z):
                    ang = s * np.float32(np.pi)
                    tube = t * np.float32(2 * np.pi)
                    R, r = 1.5, 0.22
                    return np.stack([x0 + 1.5 - (R + r * np.cos(tube)) * np.cos(ang), 4.1 + (R + r * np.cos(tube)) * np.sin(ang), z0 + r * np.sin(tube)], 1)

                parts.append(grid_patch((0, 0, 0), (0, 0, 0), (0, 0, 0), 24, 10, displace=arch))
    rng = np.random.default_rng(seed)
    for _ in range(24):  # clutter: vases and crates on the floor
        c = rng.uniform((2, 0, 5.5), (L - 2, 0, W - 5.5))
        if rng.random() < 0.5:
            parts.append(icosphere((c[0], 0.45, c[2]), 0.45, 3))
        else:
            s = rng.uniform(0.3, 0.8)
            parts.append(box((c[0] - s, 0, c[2] - s), (c[0] + s, 2 * s, c[2] + s), 3))
    fixed = sum(p.n_tris for p in parts)
    n_drapes = 8
    per = max(2 * 16 * 16, (target_tris - fixed) // n_drapes)
    nu = max(8, int(round((per / 2 * 2.0) ** 0.5)))
    nv = max(8, int(per // (2 * nu)))
    for d in range(n_drapes):
        x0 = 2.0 + d * 3.4
        z0 = 6.2 + (d % 2) * 5.0

        def wave(s, t, dd=d):
            amp = 0.35 * (0.3 + t)
            return np.stack([np.zeros_like(s), 0.05 * _noise(s, t, 100 + dd), amp * _noise(s, t, dd, 5)], 1)

        parts.append(grid_patch((x0, 10.5, z0), (2.6, 0, 0), (0, -6.5, 0), nu, nv, displace=wave))
    return merge(parts)


def _city_block_params(n_blocks_side: int, seed: int):
    """The random draws of g_city, block by block in its order: per block 5 boxes (centre offset, half size) and 5 spheres
    (centre offset, height, radius). A few numbers per block -- every rank of a sharded build draws all of them and makes
    geometry only for its own blocks."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_blocks_side * n_blocks_side):
        boxes = [(rng.uniform((-6, 0, -6), (6, 0, 6)), rng.uniform(1.0, 3.0, 3)) for _ in range(5)]
        spheres = []
        for _ in range(5):
            c = rng.uniform((-8, 0, -8), (8, 0, 8))
            y = 3.0 + rng.uniform(0, 6)
            spheres.append((c, y, rng.uniform(0.6, 1.6)))
        out.append((boxes, spheres))
    return out


def _city_block_parts(params, bx: int, bz: int, sub: int, lvl: int):
    boxes, spheres = params
    cx, cz = 10.0 + 20.0 * bx, 10.0 + 20.0 * bz
    parts = [box((cx + c[0] - s[0], 1.5, cz + c[2] - s[2]), (cx + c[0] + s[0], 1.5 + 4 * s[1], cz + c[2] + s[2]), sub) for c, s in boxes]
    parts += [icosphere((cx + c[0], y, cz + c[2]), r, lvl) for c, y, r in spheres]
    return parts


def _city_shape(n_blocks_side: int, tris_per_block: int):
    tn = max(16, int((tris_per_block * n_blocks_side * n_blocks_side / 8) ** 0.5))  # terrain cells per side
    sub = max(1, int(round((tris_per_block / 10 / 12) ** 0.5)))
    lvl = 3 if tris_per_block < 20000 else (4 if tris_per_block < 80000 else 5)
    return tn, sub, lvl


def g_city(n_blocks_side: int, tris_per_block: int, seed: int = 11) -> Mesh:
    """Large stand-in (San-Miguel / Moana class): a terrain heightfield plus a grid of detailed blocks.

    Spatially well separated clusters, so a primitives-per-batching-point cut yields meaningful
    batching points.
    """
    size = 20.0 * n_blocks_side
    tn, sub, lvl = _city_shape(n_blocks_side, tris_per_block)

    def terrain(s, t):
        return np.stack([np.zeros_like(s), 1.5 * _noise(s, t, seed, 5), np.zeros_like(s)], 1)

    parts = [grid_patch((0, 0, 0), (size, 0, 0), (0, 0, size), tn, tn, displace=terrain)]
    params = _city_block_params(n_blocks_side, seed)
    for bx in range(n_blocks_side):
        for bz in range(n_blocks_side):
            parts += _city_block_parts(params[bx * n_blocks_side + bz], bx, bz, sub, lvl)
    return merge(parts)


class CityTiles:
    """g_city as tiles that can be made one at a time: a tile is `blocks_per_tile` blocks along x by one block along z, with the
    terrain cells whose centre lies over it. The union of all tiles is g_city's triangles (same vertices bit for bit; vertices on
    tile borders are repeated). For builds in which every rank makes only the geometry of its own batching points: a tile IS a
    batching point, its triangle count (`weights`) is known without making it."""

    def __init__(self, n_blocks_side: int, tris_per_block: int, seed: int = 11, blocks_per_tile: int = 2):
        self.n, self.seed, self.bpt = int(n_blocks_side), int(seed), int(blocks_per_tile)
        self.size = 20.0 * self.n
        self.tn, self.sub, self.lvl = _city_shape(self.n, tris_per_block)
        self.params = _city_block_params(self.n, seed)
        self.tiles_x = (self.n + self.bpt - 1) // self.bpt
        self.n_tiles = self.tiles_x * self.n
        self._lin = np.linspace(0, 1, self.tn + 1, dtype=np.float32)  # (grid_patch's parameter values)

    def _cells(self, b0: int, b1: int):
        """terrain cells i with centre over blocks [b0, b1): (2 i + 1) n < 2 tn b, in integers"""
        first = lambda b: min(self.tn, max(0, -((self.n - 2 * self.tn * b) // (2 * self.n))))
        return first(b0), first(b1)

    def tile_blocks(self, t: int):
        tx, bz = divmod(int(t), self.n)
        return [(bx, bz) for bx in range(tx * self.bpt, min(self.n, (tx + 1) * self.bpt))]

    def tile_cell(self, t: int):
        """(x, z) position of the tile in the tile grid (for space-filling orders)"""
        return divmod(int(t), self.n)

    def weights(self) -> np.ndarray:
        w = np.zeros(self.n_tiles, np.int64)
        per_block = 5 * 12 * self.sub * self.sub + 5 * 20 * 4 ** self.lvl
        for t in range(self.n_tiles):
            blocks = self.tile_blocks(t)
            i0, i1 = self._cells(blocks[0][0], blocks[-1][0] + 1)
            j0, j1 = self._cells(blocks[0][1], blocks[0][1] + 1)
            w[t] = per_block * len(blocks) + 2 * (i1 - i0) * (j1 - j0)
        return w

    def tile_mesh(self, t: int) -> Mesh:
        blocks = self.tile_blocks(t)
        i0, i1 = self._cells(blocks[0][0], blocks[-1][0] + 1)
        j0, j1 = self._cells(blocks[0][1], blocks[0][1] + 1)
        parts = []
        if i1 > i0 and j1 > j0:  # the terrain cells, with grid_patch's arithmetic on the stretch of its parameter grid
            s, u = np.meshgrid(self._lin[i0:i1 + 1], self._lin[j0:j1 + 1], indexing="ij")
            s, u = s.ravel(), u.ravel()
            du, dv = np.asarray((self.size, 0, 0), np.float32), np.asarray((0, 0, self.size), np.float32)
            p = np.zeros(3, np.float32)[None, :] + s[:, None] * du[None, :] + u[:, None] * dv[None, :]
            p = p + np.stack([np.zeros_like(s), 1.5 * _noise(s, u, self.seed, 5), np.zeros_like(s)], 1).astype(np.float32)
            nv = j1 - j0
            i, j = np.meshgrid(np.arange(i1 - i0), np.arange(nv), indexing="ij")
            a = (i * (nv + 1) + j).ravel()
            b = a + (nv + 1)
            parts.append(Mesh(p, np.concatenate([np.stack([a, b, b + 1], 1), np.stack([a, b + 1, a + 1], 1)])))
        for bx, bz in blocks:
            parts += _city_block_parts(self.params[bx * self.n + bz], bx, bz, self.sub, self.lvl)
        return merge(parts)

    def normals_of(self, tile: np.ndarray, obj: np.ndarray) -> np.ndarray:
        """Geometric normals (not normalised) of hits given as (tile, triangle of the tile); makes each tile that was hit, once."""
        tile, obj = np.asarray(tile, np.int64), np.asarray(obj, np.int64)
        out = np.zeros((tile.size, 3), np.float64)
        order = np.argsort(tile, kind="stable")
        cuts = np.nonzero(np.diff(tile[order]))[0] + 1
        for run in np.split(order, cuts):
            if run.size == 0:
                continue
            m = self.tile_mesh(int(tile[run[0]]))
            v = m.pos[m.idx[obj[run]]].astype(np.float64)
            out[run] = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])
        return out


def tree_prototype(seed: int, detail: int = 2) -> Mesh:
    """A small tree in its own space (base at the origin, ~4 units tall): a trunk and a few foliage blobs. `detail` is the
    icosphere level of the blobs (2 -> ~1.2 k triangles, 3 -> ~4.5 k)."""
    rng = np.random.default_rng(seed)
    parts = [cylinder((0.0, 0.0, 0.0), 0.15 + 0.1 * rng.random(), 2.0 + rng.random(), 10, 4)]
    for _ in range(3 + int(rng.integers(0, 3))):
        c = rng.uniform((-0.8, 1.8, -0.8), (0.8, 3.6, 0.8))
        parts.append(icosphere(c, 0.5 + 0.6 * rng.random(), detail))
    return merge(parts)


def g_landscape(n_prototypes: int = 8, n_instances: int = 1500, terrain_n: int = 96, size: float = 120.0, detail: int = 2, seed: int = 5):
    """Instanced-landscape stand-in (BASELINE configs[4], SURVEY G-landscape): a heightfield terrain (plain triangles) and
    `n_instances` placements of `n_prototypes` tree meshes. Returns (terrain Mesh, [prototype Mesh], proto id per instance
    int32[n], local-to-world matrices float32[n, 4, 4]): rotation about y, uniform scale, translation onto the terrain."""
    amp = 0.03 * size

    def height(s, t):
        return amp * _noise(s, t, seed, 4)

    terrain = grid_patch((0, 0, 0), (size, 0, 0), (0, 0, size), terrain_n, terrain_n,
                         displace=lambda s, t: np.stack([np.zeros_like(s), height(s, t), np.zeros_like(s)], 1))
    protos = [tree_prototype(seed * 1000 + i, detail) for i in range(n_prototypes)]
    rng = np.random.default_rng(seed + 1)
    st = rng.uniform(0.02, 0.98, (n_instances, 2))
    ang = rng.uniform(0, 2 * np.pi, n_instances)
    scale = rng.uniform(0.6, 1.6, n_instances)
    proto = rng.integers(0, n_prototypes, n_instances).astype(np.int32)
    y = height(st[:, 0], st[:, 1])
    m = np.zeros((n_instances, 4, 4), np.float64)
    m[:, 0, 0] = scale * np.cos(ang)
    m[:, 0, 2] = scale * np.sin(ang)
    m[:, 1, 1] = scale
    m[:, 2, 0] = -scale * np.sin(ang)
    m[:, 2, 2] = scale * np.cos(ang)
    m[:, 0, 3] = st[:, 0] * size
    m[:, 1, 3] = y - 0.05
    m[:, 2, 3] = st[:, 1] * size
    m[:, 3, 3] = 1.0
    return terrain, protos, proto, m.astype(np.float32)


def bake_instances(protos, proto, matrices) -> Mesh:
    """The instanced geometry as plain world-space triangles (Transform::transformPointToWorld per vertex,
    transform.cpp:18-22): what the voxeliser sees (triangle_intersect.cpp:245-249) and a brute-force cross-check."""
    out = []
    for k in range(len(proto)):
        pm = protos[int(proto[k])]
        m = matrices[k].astype(np.float32)
        p = np.concatenate([pm.pos, np.ones((pm.n_verts, 1), np.float32)], axis=1) @ m.T
        out.append(Mesh(p[:, :3] / p[:, 3:4], pm.idx))
    return merge(out)


# ----------------------------------------------------------------------------- ray sets (SoA [8][n])
def _normalize(v):
    return v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)


def pack_rays(o, d, tnear, tfar) -> np.ndarray:
    n = o.shape[0]
    r = np.empty((8, n), dtype=np.float32)
    r[0:3] = o.T
    r[3:6] = d.T
    r[6] = tnear
    r[7] = tfar
    return r


def primary_rays(eye, look_at, up, fov_deg: float, width: int, height: int, seed: int = 891379) -> np.ndarray:
    """Jittered pinhole camera rays, one per pixel (sampler seed as sampler_integrator.h:45)."""
    eye, look_at, up = (np.asarray(a, np.float64) for a in (eye, look_at, up))
    fwd = look_at - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    n = width * height
    rng = Pcg32(np.arange(n, dtype=np.uint64) + np.uint64(seed))
    px = (np.arange(n) % width + rng.f32()) / width - 0.5
    py = (np.arange(n) // width + rng.f32()) / height - 0.5
    half = np.tan(np.radians(fov_deg) / 2)
    d = fwd[None, :] + (2 * half * px)[:, None] * right[None, :] * (width / height) - (2 * half * py)[:, None] * upv[None, :]
    d = _normalize(d).astype(np.float32)
    o = np.broadcast_to(eye.astype(np.float32), (n, 3)).copy()
    return pack_rays(o, d, np.zeros(n, np.float32), np.full(n, FLT_MAX, np.float32))


def geometric_normals(mesh: Mesh, prim: np.ndarray) -> np.ndarray:
    t = mesh.idx[prim]
    p0, p1, p2 = mesh.pos[t[:, 0]], mesh.pos[t[:, 1]], mesh.pos[t[:, 2]]
    return _normalize(np.cross(p1 - p0, p2 - p0).astype(np.float64)).astype(np.float32)


def diffuse_bounce_rays(mesh: Mesh | None, rays: np.ndarray, t_hit: np.ndarray, prim: np.ndarray, seed: int = 891379,
                        shuffle_seed: int | None = 94892380, normals: np.ndarray | None = None) -> np.ndarray:
    """One cosine-weighted bounce ray per ray that hit something; order shuffled (incoherent set).
    normals: optional [n, 3] geometric normals of the hits (scenes whose hits are not triangles of one mesh: instances);
    `prim` then only says which rays hit (anything but 0xFFFFFFFF)."""
    hit = prim != np.uint32(0xFFFFFFFF)
    o = rays[0:3, hit].T.astype(np.float32)
    d = rays[3:6, hit].T.astype(np.float32)
    p = o + t_hit[hit, None].astype(np.float32) * d
    ng = geometric_normals(mesh, prim[hit]) if normals is None else _normalize(np.asarray(normals, np.float64)[hit]).astype(np.float32)
    flip = np.einsum("ij,ij->i", ng, -d) < 0  # triangle_intersect.cpp:459-460
    ng[flip] = -ng[flip]
    n = p.shape[0]
    rng = Pcg32(np.arange(n, dtype=np.uint64) + np.uint64(seed) + np.uint64(1 << 32))
    u1, u2 = rng.f32(), rng.f32()
    # concentric disk + lift (sampling.h): cosine-weighted hemisphere around +z
    ox, oy = 2 * u1 - 1, 2 * u2 - 1
    r = np.where(np.abs(ox) > np.abs(oy), ox, oy)
    theta = np.where(np.abs(ox) > np.abs(oy), (np.pi / 4) * (oy / np.where(ox == 0, 1, ox)), (np.pi / 2) - (np.pi / 4) * (ox / np.where(oy == 0, 1, oy)))
    dx, dy = r * np.cos(theta), r * np.sin(theta)
    dz = np.sqrt(np.maximum(0.0, 1 - dx * dx - dy * dy))
    helper = np.where((ng[:, 0] > 0.5)[:, None], np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))  # material.cpp:5-11
    ss = _normalize(np.cross(ng, helper))
    ts = _normalize(np.cross(ng, ss))
    wi = _normalize(dx[:, None] * ss + dy[:, None] * ts + dz[:, None] * ng).astype(np.float32)
    org = (p + ng * RAY_EPSILON).astype(np.float32)  # interaction.cpp:39-46
    out = pack_rays(org, wi, np.zeros(n, np.float32), np.full(n, FLT_MAX, np.float32))
    if shuffle_seed is not None:
        perm = np.random.default_rng(shuffle_seed).permutation(n)
        out = np.ascontiguousarray(out[:, perm])
    return out


def shadow_rays(mesh: Mesh, rays: np.ndarray, t_hit: np.ndarray, prim: np.ndarray, light_lo, light_hi, seed: int = 891379) -> np.ndarray:
    hit = prim != np.uint32(0xFFFFFFFF)
    o = rays[0:3, hit].T.astype(np.float32)
    d = rays[3:6, hit].T.astype(np.float32)
    p = o + t_hit[hit, None].astype(np.float32) * d
    ng = geometric_normals(mesh, prim[hit])
    flip = np.einsum("ij,ij->i", ng, -d) < 0
    ng[flip] = -ng[flip]
    n = p.shape[0]
    rng = Pcg32(np.arange(n, dtype=np.uint64) + np.uint64(seed) + np.uint64(2 << 32))
    lo, hi = np.asarray(light_lo, np.float32), np.asarray(light_hi, np.float32)
    q = lo[None, :] + np.stack([rng.f32(), rng.f32(), rng.f32()], 1) * (hi - lo)[None, :]
    dirv = q - p
    dist = np.linalg.norm(dirv, axis=1).astype(np.float32)
    wi = (dirv / np.maximum(dist[:, None], 1e-20)).astype(np.float32)
    side = np.where((np.einsum("ij,ij->i", ng, wi) > 0)[:, None], ng, -ng)
    org = (p + side * RAY_EPSILON).astype(np.float32)
    return pack_rays(org, wi, np.zeros(n, np.float32), (dist * np.float32(0.9999)).astype(np.float32))


def random_rays(lo, hi, n: int, seed: int = 1, toward=None) -> np.ndarray:
    """Uniform origins in a box; directions uniform on the sphere or aimed at points in `toward` box."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    if toward is None:
        d = rng.normal(size=(n, 3))
    else:
        tl, th = np.asarray(toward[0], np.float32), np.asarray(toward[1], np.float32)
        d = rng.uniform(tl, th, (n, 3)) - o
    d = _normalize(d).astype(np.float32)
    return pack_rays(o, d, np.zeros(n, np.float32), np.full(n, FLT_MAX, np.float32))
