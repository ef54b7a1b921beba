/*
 * ref_simd_harness.cpp -- TEST INFRASTRUCTURE ONLY. Built into oracle/_ref/libpandora_ref.so.
 *
 * Second, independently written restatement of the traversal loops that runs ON THE REFERENCE'S OWN
 * SIMD LIBRARY: simd/simd8.h and simd/simd4.h are compiled from where they lie under
 * /root/reference/projects/simd/include (they are std-only and build as they are). Nothing from the
 * reference is copied into this repository; this file only #includes those headers by path, and the
 * resulting .so is git-ignored.
 *
 * What this pins: every AVX2/SSE primitive the reference's hot path relies on (minps/maxps NaN rule,
 * vpermps permute, the 256-entry compress LUT, mask8::count(validMask), horizontalMinIndex,
 * horizontalMinVec, bitScan4, bitScanReverse32, int<->float bit casts and per-lane shifts) is the
 * reference's real code here. What it does not pin: the loop structure around them, which is restated
 * from wive_bvh8_impl.h / pauseable_bvh4.h / sparse_voxel_dag.cpp (those headers need EASTL,
 * flatbuffers, Embree, mio and glm, none of which exist in this image, so they are unbuildable here).
 *
 * The scalar oracle (pandora_oracle.c) must agree with this file bit for bit; tests/test_oracle_ref.py
 * checks that, and the tests/golden fixtures are generated from this library.
 */
#include "pandora_oracle.h"
#include "simd/intrinsics.h"
#include "simd/simd4.h"
#include "simd/simd8.h"
#include <array>
#include <cmath>
#include <cstring>
#include <limits>
#include <span>

namespace {

using simd::mask4;
using simd::mask8;
using simd::vec4_f32;
using simd::vec4_u32;
using simd::vec8_f32;
using simd::vec8_u32;

struct Ray8 { // wive_bvh8.h:97-109
    vec8_f32 ox, oy, oz, ix, iy, iz, tnear, tfar;
    vec8_u32 shift;
    explicit Ray8(const orc_ray& r)
        : ox(r.o[0]), oy(r.o[1]), oz(r.o[2])
        , ix(1.0f / r.d[0]), iy(1.0f / r.d[1]), iz(1.0f / r.d[2])
        , tnear(r.tnear), tfar(r.tfar)
        , shift(orc_sign_shift(r.d[0], r.d[1], r.d[2]))
    {
    }
};

struct Node8View {
    vec8_f32 loX, hiX, loY, hiY, loZ, hiZ;
    vec8_u32 kids, perm;
    explicit Node8View(const orc_bvh8_node& n)
        : loX(std::span<const float, 8>(n.minX)), hiX(std::span<const float, 8>(n.maxX))
        , loY(std::span<const float, 8>(n.minY)), hiY(std::span<const float, 8>(n.maxY))
        , loZ(std::span<const float, 8>(n.minZ)), hiZ(std::span<const float, 8>(n.maxZ))
        , kids(std::span<const uint32_t, 8>(n.children)), perm(std::span<const uint32_t, 8>(n.perm))
    {
    }
};

// wive_bvh8_impl.h:209-238 expressed with the reference's vec8 operators
inline unsigned slab8(const Node8View& n, const Ray8& r, vec8_u32& kidsOut, vec8_f32& distOut)
{
    const vec8_f32 ax = (n.loX - r.ox) * r.ix, bx = (n.hiX - r.ox) * r.ix;
    const vec8_f32 ay = (n.loY - r.oy) * r.iy, by = (n.hiY - r.oy) * r.iy;
    const vec8_f32 az = (n.loZ - r.oz) * r.iz, bz = (n.hiZ - r.oz) * r.iz;
    vec8_f32 enter = simd::max(r.tnear, simd::max(simd::min(ax, bx), simd::max(simd::min(ay, by), simd::min(az, bz))));
    vec8_f32 leave = simd::min(r.tfar, simd::min(simd::max(ax, bx), simd::min(simd::max(ay, by), simd::max(az, bz))));
    const vec8_u32 order = (n.perm >> r.shift) & vec8_u32(7u);
    enter = enter.permute(order);
    leave = leave.permute(order);
    const mask8 overlap = enter <= leave;
    const vec8_u32 pack(overlap.computeCompressPermutation());
    kidsOut = n.kids.permute(order).permute(pack);
    distOut = enter.permute(pack);
    return static_cast<unsigned>(overlap.count());
}

struct Stack96 { // wive_bvh8_impl.h:74-76
    alignas(32) std::array<uint32_t, 104> handle;
    alignas(32) std::array<float, 104> dist;
    size_t top = 0;
    Stack96() { dist.fill(std::numeric_limits<float>::max()); handle.fill(0); }
    // :112-134, eight entries at a time through the compress permutation
    void cull(const vec8_f32& tfar)
    {
        size_t out = 0;
        for (size_t i = 0; i < top; i += 8) {
            vec8_f32 d;
            vec8_u32 h;
            d.loadAligned(std::span<const float>(dist.data() + i, 8));
            h.loadAligned(std::span<const uint32_t>(handle.data() + i, 8));
            const mask8 keep = d < tfar;
            const vec8_u32 pack(keep.computeCompressPermutation());
            d.permute(pack).store(std::span<float>(dist.data() + out, 8));
            h.permute(pack).store(std::span<uint32_t>(handle.data() + out, 8));
            const size_t live = std::min<size_t>(8, top - i);
            out += static_cast<size_t>(keep.count((1u << live) - 1u));
        }
        top = out;
    }
};

constexpr uint32_t kInnerTag = 2u;
inline bool isInner(uint32_t h) { return ((h >> 29) & 7u) == kInnerTag; }
inline uint32_t slot(uint32_t h) { return h & 0x1FFFFFFFu; }
inline uint32_t leafCount(uint32_t h) { return ((h >> 29) & 3u) + 1u; }

inline const float* vtx(const orc_mesh& m, uint32_t prim, unsigned k) { return m.pos + 3u * size_t(m.idx[3u * size_t(prim) + k]); }

bool leafTest(const orc_bvh8& b, int triMode, uint32_t h, orc_ray& r, orc_hit* out, bool stopAtFirst)
{
    bool any = false;
    const uint32_t* ids = b.leaf_idx + slot(h);
    for (uint32_t i = 0, n = leafCount(h); i < n; ++i) {
        const uint32_t obj = ids[i];
        const uint32_t prim = b.obj_prim ? b.obj_prim[obj] : obj;
        float u, v;
        if (orc_tri_intersect(triMode, r.o, r.d, r.tnear, &r.tfar, vtx(b.mesh, prim, 0), vtx(b.mesh, prim, 1), vtx(b.mesh, prim, 2), &u, &v)) {
            any = true;
            if (out) {
                out->t = r.tfar;
                out->obj = obj;
                out->u = u;
                out->v = v;
            }
            if (stopAtFirst)
                return true;
        }
    }
    return any;
}

// wive_bvh8_impl.h:57-140 (closest) and :142-201 (any)
bool walk8(const orc_bvh8& b, int triMode, bool anyHit, orc_ray& r, orc_hit* out)
{
    Ray8 ray(r);
    Stack96 st;
    st.handle[0] = b.root;
    st.dist[0] = 0.0f;
    st.top = 1;
    bool found = false;
    while (st.top > 0) {
        const uint32_t h = st.handle[--st.top];
        if (isInner(h)) {
            vec8_u32 kids;
            vec8_f32 dist;
            const unsigned n = slab8(Node8View(b.nodes[slot(h)]), ray, kids, dist);
            if (n > 0) {
                kids.store(std::span<uint32_t>(st.handle.data() + st.top, 8));
                dist.store(std::span<float>(st.dist.data() + st.top, 8));
                st.top += n;
            }
        } else if (anyHit) {
            if (leafTest(b, triMode, h, r, nullptr, true)) {
                r.tfar = -std::numeric_limits<float>::infinity();
                return true;
            }
        } else if (leafTest(b, triMode, h, r, out, false)) {
            found = true;
            ray.tfar.broadcast(r.tfar);
            st.cull(ray.tfar);
        }
    }
    return found;
}

// sparse_voxel_dag.cpp:350-567 on the reference's vec4 types (4 lanes, w = 1)
bool castDag(const orc_svdag& dag, const orc_ray& in)
{
    constexpr int kDepth = 23;
    constexpr float eps = 1.1920928955078125e-07f;
    float o[3] = { in.o[0], in.o[1], in.o[2] }, d[3] = { in.d[0], in.d[1], in.d[2] };
    for (int i = 0; i < 3; ++i) {
        o[i] = 1.0f + (dag.inv_extent * (o[i] - dag.bounds_min[i]));
        if (std::abs(d[i]) < eps)
            d[i] = std::copysign(eps, d[i]);
    }
    const vec4_f32 org(o[0], o[1], o[2], 1.0f), dir(d[0], d[1], d[2], 1.0f);
    const vec4_f32 coef = vec4_f32(1.0f) / -dir.abs();
    vec4_f32 bias = coef * org;
    const mask4 positive = dir > vec4_f32(0.0f);
    const uint32_t mirror = (7u ^ uint32_t(positive.bitMask())) & 7u;
    bias = simd::blend(bias, vec4_f32(3.0f) * coef - bias, positive);

    auto store4 = [](const vec4_f32& v, float* out) { v.store(std::span<float, 4>(out, 4)); };
    float a[4], c[4];
    store4(vec4_f32(2.0f) * coef - bias, a);
    store4(coef - bias, c);
    const float tEnter = std::max(0.0f, std::max(a[0], std::max(a[1], a[2])));
    const float tLeave = std::min(c[0], std::min(c[1], c[2]));
    if (tEnter >= tLeave)
        return false;

    vec4_f32 tMin(tEnter);
    const mask4 firstHalf = (vec4_f32(1.5f) * coef - bias) > tEnter;
    uint32_t idx = uint32_t(firstHalf.bitMask());
    vec4_f32 pos = simd::blend(vec4_f32(1.0f), vec4_f32(1.5f), firstHalf);

    uint32_t parent = dag.root;
    int scale = kDepth - 1;
    uint32_t saved[kDepth + 1] = {};
    auto valid = [&](uint32_t n) { return (dag.data[n] >> 8) & 0xFFu; };
    auto leafs = [&](uint32_t n) { return dag.data[n] & 0xFFu; };

    while (scale < kDepth) {
        const vec4_f32 corner = pos * coef - bias;
        const uint32_t child = 7u - ((idx & 7u) ^ mirror);
        if (valid(parent) >> child & 1u) {
            const float half = std::ldexp(1.0f, (scale - 1) - kDepth);
            const vec4_f32 center = vec4_f32(half) * coef + corner;
            saved[scale] = parent;
            if ((valid(parent) & leafs(parent)) >> child & 1u)
                return true;
            const uint32_t inner = valid(parent) & ~leafs(parent);
            parent = dag.data[parent + 1u + uint32_t(simd::popcount64(inner & ((1u << child) - 1u)))];
            --scale;
            const mask4 far = center > tMin;
            idx = uint32_t(far.bitMask()) & 7u;
            pos = simd::blend(pos, pos + half, far);
        } else {
            const float size = std::ldexp(1.0f, scale - kDepth);
            const vec4_f32 exitT = simd::intBitsToFloat(simd::floatBitsToInt(corner) | vec4_u32(0x0, 0x0, 0x0, 0x7FFFFFF)).horizontalMinVec();
            const mask4 step = corner <= exitT;
            const uint32_t stepBits = uint32_t(step.bitMask()) & 7u;
            pos = simd::blend(pos, pos - size, step);
            tMin = exitT;
            idx ^= stepBits;
            if ((idx & stepBits) != 0) {
                const vec4_u32 changed = simd::blend(vec4_u32(0u), simd::floatBitsToInt(pos) ^ simd::floatBitsToInt(pos + vec4_f32(size)), step);
                uint32_t ch[4];
                changed.store(std::span<uint32_t, 4>(ch, 4));
                scale = simd::bitScanReverse32(ch[0] | ch[1] | ch[2]);
                if (scale <= kDepth)
                    parent = saved[scale];
            }
            const vec4_u32 sh = simd::floatBitsToInt(pos) >> uint32_t(scale);
            pos = simd::intBitsToFloat(sh << uint32_t(scale));
            const vec4_u32 lo = (sh & vec4_u32(1u)) << vec4_u32(0, 1, 2, 0);
            const vec4_u32 hi = (sh & vec4_u32(2u)) << vec4_u32(2, 3, 4, 0);
            uint32_t part[4];
            (lo | hi).store(std::span<uint32_t, 4>(part, 4));
            idx = part[0] | part[1] | part[2];
        }
    }
    return false;
}

struct Node4View { // pauseable_bvh4.h:74-98
    vec4_f32 loX, loY, loZ, hiX, hiY, hiZ;
    uint32_t valid, leaf, depth;
    explicit Node4View(const orc_bvh4_node& n)
        : loX(std::span<const float, 4>(n.minX)), loY(std::span<const float, 4>(n.minY)), loZ(std::span<const float, 4>(n.minZ))
        , hiX(std::span<const float, 4>(n.maxX)), hiY(std::span<const float, 4>(n.maxY)), hiZ(std::span<const float, 4>(n.maxZ))
        , valid(n.bits & 0xFu), leaf((n.bits >> 4) & 0xFu), depth((n.bits >> 8) & 0xFFu)
    {
    }
};

// pauseable_bvh4.h:225-355, one episode (same contract as orc_bvh4_step)
bool walk4(const orc_scene& s, bool anyHit, const orc_ray& r, uint32_t& node, uint64_t& stack, uint32_t& leafOut, orc_counters* cnt)
{
    const vec4_f32 ox(r.o[0]), oy(r.o[1]), oz(r.o[2]);
    const vec4_f32 ix(1.0f / r.d[0]), iy(1.0f / r.d[1]), iz(1.0f / r.d[2]);
    const vec4_f32 tnear(r.tnear), tfar(r.tfar);
    for (;;) {
        const orc_bvh4_node& raw = s.nodes[node];
        const Node4View n(raw);
        const int bit = 4 * int(n.depth);
        const uint64_t want = (stack >> bit) & 0xFull;
        if (want) {
            const mask4 wantMask(bool(want & 1), bool(want & 2), bool(want & 4), bool(want & 8));
            const mask4 validMask(bool(n.valid & 1), bool(n.valid & 2), bool(n.valid & 4), bool(n.valid & 8));
            const vec4_f32 ax = (n.loX - ox) * ix, bx = (n.hiX - ox) * ix;
            const vec4_f32 ay = (n.loY - oy) * iy, by = (n.hiY - oy) * iy;
            const vec4_f32 az = (n.loZ - oz) * iz, bz = (n.hiZ - oz) * iz;
            const vec4_f32 enter = simd::max(tnear, simd::max(simd::min(ax, bx), simd::max(simd::min(ay, by), simd::min(az, bz))));
            const vec4_f32 leave = simd::min(tfar, simd::min(simd::max(ax, bx), simd::min(simd::max(ay, by), simd::max(az, bz))));
            const mask4 visit = ((enter <= leave) && wantMask) && validMask;
            if (visit.any()) {
                unsigned child;
                if (anyHit)
                    child = unsigned(simd::bitScan4(uint32_t(visit.bitMask())));
                else
                    child = simd::blend(vec4_f32(std::numeric_limits<float>::max()), enter, visit).horizontalMinIndex();
                uint64_t rest = uint64_t(visit.bitMask());
                rest ^= (1ull << child);
                stack ^= (want << bit);
                stack |= (rest << bit);
                const uint32_t innerMask = n.valid & ~n.leaf, leafMask = n.valid & n.leaf;
                if (innerMask >> child & 1u) {
                    node = raw.firstChild + uint32_t(simd::popcount32(innerMask & ((1u << child) - 1u)));
                } else {
                    const uint32_t leaf = raw.firstLeaf + uint32_t(simd::popcount32(leafMask & ((1u << child) - 1u)));
                    bool culled = false;
                    if (s.leaves[leaf].dag.enabled) {
                        if (cnt)
                            cnt->svdag_tests++;
                        if (!castDag(s.leaves[leaf].dag, r)) {
                            culled = true;
                            if (cnt)
                                cnt->svdag_culled++;
                        }
                    }
                    if (!culled) {
                        leafOut = leaf;
                        return true;
                    }
                }
                continue;
            }
        }
        stack |= (0xFFFFFFFFFFFFFFFFull << bit);
        if (stack == 0xFFFFFFFFFFFFFFFFull)
            return false;
        node = raw.parent;
    }
}

} // namespace

extern "C" {

void ref_compress_perm8(unsigned mask, uint32_t out[8])
{
    const mask8 m(bool(mask & 1), bool(mask & 2), bool(mask & 4), bool(mask & 8), bool(mask & 16), bool(mask & 32), bool(mask & 64), bool(mask & 128));
    vec8_u32(m.computeCompressPermutation()).store(std::span<uint32_t>(out, 8));
}
unsigned ref_hmin_index4(const float v[4]) { return vec4_f32(v[0], v[1], v[2], v[3]).horizontalMinIndex(); }
unsigned ref_hmin_index8(const float v[8]) { return vec8_f32(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]).horizontalMinIndex(); }
int ref_bitscan4(unsigned mask) { return simd::bitScan4(mask); }
int ref_bsr32(uint32_t mask) { return simd::bitScanReverse32(mask); }
void ref_permute8(const float v[8], const uint32_t idx[8], float out[8])
{
    vec8_f32(std::span<const float, 8>(v, 8)).permute(vec8_u32(std::span<const uint32_t, 8>(idx, 8))).store(std::span<float>(out, 8));
}
void ref_compress8(const float v[8], unsigned mask, float out[8])
{
    const mask8 m(bool(mask & 1), bool(mask & 2), bool(mask & 4), bool(mask & 8), bool(mask & 16), bool(mask & 32), bool(mask & 64), bool(mask & 128));
    vec8_f32(std::span<const float, 8>(v, 8)).compress(m).store(std::span<float>(out, 8));
}
void ref_minmax8(const float a[8], const float b[8], float mn[8], float mx[8])
{
    const vec8_f32 va(std::span<const float, 8>(a, 8)), vb(std::span<const float, 8>(b, 8));
    simd::min(va, vb).store(std::span<float>(mn, 8));
    simd::max(va, vb).store(std::span<float>(mx, 8));
}

unsigned ref_bvh8_inner(const orc_bvh8_node* n, const orc_ray* r, uint32_t kids[8], float dist[8])
{
    vec8_u32 k;
    vec8_f32 d;
    const unsigned c = slab8(Node8View(*n), Ray8(*r), k, d);
    k.store(std::span<uint32_t>(kids, 8));
    d.store(std::span<float>(dist, 8));
    return c;
}

int ref_svdag_intersect(const orc_svdag* d, const orc_ray* r) { return castDag(*d, *r) ? 1 : 0; }
void ref_svdag_trace(const orc_svdag* d, const float* rays, size_t n, unsigned char* out)
{
    for (size_t i = 0; i < n; ++i) {
        orc_ray r;
        for (int a = 0; a < 3; ++a) {
            r.o[a] = rays[size_t(a) * n + i];
            r.d[a] = rays[size_t(3 + a) * n + i];
        }
        r.tnear = rays[6 * n + i];
        r.tfar = rays[7 * n + i];
        out[i] = castDag(*d, r) ? 1 : 0;
    }
}

void ref_bvh8_trace(const orc_bvh8* b, int any_hit, int tri_mode, const float* rays, size_t n,
    float* t, uint32_t* obj, float* u, float* v, uint32_t* anyhit)
{
    for (size_t i = 0; i < n; ++i) {
        orc_ray r { { rays[i], rays[n + i], rays[2 * n + i] }, { rays[3 * n + i], rays[4 * n + i], rays[5 * n + i] }, rays[6 * n + i], rays[7 * n + i] };
        if (any_hit) {
            anyhit[i] = walk8(*b, tri_mode, true, r, nullptr) ? 1u : 0u;
        } else {
            orc_hit h { 0.0f, ORC_MISS, 0u, 0.0f, 0.0f, ORC_MISS };
            walk8(*b, tri_mode, false, r, &h);
            t[i] = r.tfar;
            obj[i] = h.obj;
            u[i] = h.u;
            v[i] = h.v;
        }
    }
}

int ref_bvh4_step(const orc_scene* s, int any_hit, const orc_ray* r, uint32_t* node, uint64_t* stack, uint32_t* leaf, orc_counters* c)
{
    return walk4(*s, any_hit != 0, *r, *node, *stack, *leaf, c) ? 1 : 0;
}

void ref_scene_trace(const orc_scene* s, int any_hit, int tri_mode, const float* rays, size_t n,
    float* t, uint32_t* obj, uint32_t* leaf, float* u, float* v, uint32_t* anyhit, uint32_t* visits, orc_counters* c)
{
    if (c)
        std::memset(c, 0, sizeof(*c));
    for (size_t i = 0; i < n; ++i) {
        orc_ray r { { rays[i], rays[n + i], rays[2 * n + i] }, { rays[3 * n + i], rays[4 * n + i], rays[5 * n + i] }, rays[6 * n + i], rays[7 * n + i] };
        uint32_t node = s->root, at = 0, nvis = 0;
        uint64_t stack = 0xFFFFFFFFFFFFFFFFull;
        orc_hit h { 0.0f, ORC_MISS, 0u, 0.0f, 0.0f, ORC_MISS };
        bool shadowed = false;
        while (walk4(*s, any_hit != 0, r, node, stack, at, c)) {
            ++nvis;
            if (any_hit) {
                if (walk8(s->leaves[at].bvh, tri_mode, true, r, nullptr)) {
                    shadowed = true;
                    break;
                }
            } else if (walk8(s->leaves[at].bvh, tri_mode, false, r, &h)) {
                h.leaf = at;
            }
        }
        if (c)
            c->leaf_visits += nvis;
        if (any_hit) {
            anyhit[i] = shadowed ? 1u : 0u;
        } else {
            t[i] = r.tfar;
            obj[i] = h.obj;
            leaf[i] = h.leaf;
            u[i] = h.u;
            v[i] = h.v;
        }
        if (visits)
            visits[i] = nvis;
    }
}
}
