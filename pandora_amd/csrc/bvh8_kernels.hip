// Bottom-level WiVe BVH8 traversal for MI355X (gfx950, wave64).
//
// Reference semantics: WiVeBVH8::intersect / intersectAny / intersectInnerNode / intersectLeaf
//   projects/pandora/include/pandora/traversal/bvh/wive_bvh8_impl.h:57-140, :142-201, :209-238, :272-290
// and the leaf test TriangleShape::intersectPrimitive
//   projects/pandora/src/shapes/triangle_intersect.cpp:425-462 (Moller-Trumbore, the compiled branch)
//   and :340-423 (watertight variant, selectable).
//
// Mapping. The reference runs one ray at a time with the 8 children of a node in the 8 lanes of a
// ymm register. Here a wave64 carries 8 rays; the 8 lanes of a "ray group" play the role of the 8
// AVX2 lanes. Per step every group pops one stack entry and the wave issues two LDS-DMA loads
// (global_load_lds_dwordx4) that bring the 8 x 256-byte nodes -- or the <=4 x 48-byte leaf triangles
// -- of its 8 rays into LDS, 16 cache lines per wave step instead of 64 for per-lane gathers.
// Child ordering is the node's precomputed octant permutation applied with ds_bpermute, hits are
// compacted with a wave ballot + popcount prefix and pushed to a per-ray 96-entry stack in LDS.
// Waves are persistent: they draw chunks of rays from a global cursor until it runs out.
//
// The reference compacts its whole stack after every leaf hit (dist < tfar). Order is preserved by
// that compaction, so culling lazily at pop time is equivalent: entries pushed before the last hit
// (index < hitMark) must satisfy dist < tfar, entries pushed after it are kept unconditionally.
//
// Float semantics: contraction off; min/max written as compare+select so that NaN behaves like
// x86 minps/maxps (second operand wins), see SURVEY.md F10.
#include "kernels.h"
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace pandora_hip {

namespace {

constexpr uint32_t kWavesPerBlock = 4;
// Per-ray stack: the reference reserves 96 entries (wive_bvh8_impl.h:74-75); a tree with d nested inner
// nodes never holds more than 7 d + 8, so the launch sizes the LDS stack from the deepest uploaded tree
// (more blocks per CU for shallow trees). +64 B per ray: odd ray groups start 16 banks later.
__host__ __device__ constexpr uint32_t stackStrideBytes(uint32_t entries)
{
    return (entries * 8u + 64u) % 128u == 0u ? entries * 8u + 32u : entries * 8u + 64u;
}

__device__ __forceinline__ float min86(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float max86(float a, float b) { return a > b ? a : b; }

__device__ __forceinline__ void ldsDma16(const unsigned char* gsrc, unsigned char* ldsWaveBase)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)ldsWaveBase, 16, 0, 0);
}

__device__ __forceinline__ float ldsF(const unsigned char* p) { return *reinterpret_cast<const float*>(p); }
__device__ __forceinline__ uint32_t ldsU(const unsigned char* p) { return *reinterpret_cast<const uint32_t*>(p); }

struct TriHit {
    float t, u, v;
    bool hit;
};

// triangle_intersect.cpp:425-462, operation order kept (glm::cross / glm::dot expansion)
__device__ __forceinline__ TriHit triMT(float ox, float oy, float oz, float dx, float dy, float dz, float tnear, float tfar,
    float4 a, float4 b, float4 c)
{
    TriHit r;
    r.hit = false;
    r.t = r.u = r.v = 0.0f;
    const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
    const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
    const float hx = dy * e2z - e2y * dz;
    const float hy = dz * e2x - e2z * dx;
    const float hz = dx * e2y - e2x * dy;
    const float det = (e1x * hx + e1y * hy) + e1z * hz;
    if (det > -0.000001f && det < 0.000001f)
        return r;
    const float f = 1.0f / det;
    const float sx = ox - a.x, sy = oy - a.y, sz = oz - a.z;
    const float u = f * ((sx * hx + sy * hy) + sz * hz);
    if (u < 0.0f || u > 1.0f)
        return r;
    const float qx = sy * e1z - e1y * sz;
    const float qy = sz * e1x - e1z * sx;
    const float qz = sx * e1y - e1x * sy;
    const float v = f * ((dx * qx + dy * qy) + dz * qz);
    if (v < 0.0f || u + v > 1.0f)
        return r;
    const float t = f * ((e2x * qx + e2y * qy) + e2z * qz);
    if (t < tnear || t > tfar)
        return r;
    r.hit = true;
    r.t = t;
    r.u = u;
    r.v = v;
    return r;
}

// triangle_intersect.cpp:340-423 (PBRT-v3 watertight test with the double-precision edge fallback)
__device__ __forceinline__ TriHit triWatertight(float ox, float oy, float oz, float dx, float dy, float dz, float tnear, float tfar,
    float4 a, float4 b, float4 c)
{
    TriHit r;
    r.hit = false;
    r.t = r.u = r.v = 0.0f;
    const float ax = fabsf(dx), ay = fabsf(dy), az = fabsf(dz);
    const int kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);
    float p0[3] = { a.x - ox, a.y - oy, a.z - oz };
    float p1[3] = { b.x - ox, b.y - oy, b.z - oz };
    float p2[3] = { c.x - ox, c.y - oy, c.z - oz };
    float d[3] = { dx, dy, dz };
    // permute(v, kx, ky, kz) with kx = kz+1, ky = kx+1 (mod 3), written as selects to stay in registers
    auto sel = [](const float* v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); };
    const int kx = kz == 2 ? 0 : kz + 1;
    const int ky = kx == 2 ? 0 : kx + 1;
    const float ddx = sel(d, kx), ddy = sel(d, ky), ddz = sel(d, kz);
    float p0x = sel(p0, kx), p0y = sel(p0, ky), p0z = sel(p0, kz);
    float p1x = sel(p1, kx), p1y = sel(p1, ky), p1z = sel(p1, kz);
    float p2x = sel(p2, kx), p2y = sel(p2, ky), p2z = sel(p2, kz);
    const float Sx = -ddx / ddz, Sy = -ddy / ddz, Sz = 1.0f / ddz;
    p0x += Sx * p0z;
    p0y += Sy * p0z;
    p1x += Sx * p1z;
    p1y += Sy * p1z;
    p2x += Sx * p2z;
    p2y += Sy * p2z;
    float e0 = p1x * p2y - p1y * p2x;
    float e1 = p2x * p0y - p2y * p0x;
    float e2 = p0x * p1y - p0y * p1x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        const double p2txp1ty = (double)p2x * (double)p1y;
        const double p2typ1tx = (double)p2y * (double)p1x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        const double p0txp2ty = (double)p0x * (double)p2y;
        const double p0typ2tx = (double)p0y * (double)p2x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        const double p1txp0ty = (double)p1x * (double)p0y;
        const double p1typ0tx = (double)p1y * (double)p0x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f))
        return r;
    const float det = e0 + e1 + e2;
    if (det == 0.0f)
        return r;
    p0z *= Sz;
    p1z *= Sz;
    p2z *= Sz;
    const float tScaled = e0 * p0z + e1 * p1z + e2 * p2z;
    if (det < 0.0f && (tScaled >= tnear * det || tScaled < tfar * det))
        return r;
    else if (det > 0.0f && (tScaled <= tnear * det || tScaled > tfar * det))
        return r;
    const float invDet = 1.0f / det;
    r.hit = true;
    r.u = e0 * invDet;
    r.v = e1 * invDet;
    r.t = tScaled * invDet;
    return r;
}

// LPR = lanes per ray: 8 (one child per lane, the reference's ymm mapping) or 4 (two children per lane,
// 16 rays per wave: the per-ray bookkeeping and the leaf phase are amortised over twice as many rays).
template <int LPR>
struct Geo {
    static constexpr uint32_t G = 64 / LPR;      // ray groups per wave
    static constexpr uint32_t CPL = 8 / LPR;     // children per lane
    static constexpr uint32_t NDMA = 16 / LPR;   // LDS-DMA instructions per 256-byte node
    static constexpr uint32_t kStage = G * 256;  // staging bytes per wave
    static constexpr uint32_t kOrder = CPL > 1 ? G * 64 : 0; // child-ordering scratch per wave
    static constexpr uint64_t kOneBitPerGroup = LPR == 8 ? 0x0101010101010101ull : 0x1111111111111111ull;
    static constexpr uint32_t kGroupMask = (1u << LPR) - 1u;
    // swizzle of the 16-byte piece position inside one DMA instruction's region, chosen so that the ray groups
    // of one LDS lane group read disjoint banks (LPR 8: ds_read_b32, 32 banks; LPR 4: ds_read_b64, 64 banks)
    __device__ static uint32_t swz(uint32_t g) { return LPR == 8 ? ((g & 3u) << 1) : (((g >> 2) & 1u) << 1); }
    // LDS byte offset (inside the wave's staging area) of 16-byte piece P of group g's node / triangle block
    __device__ static uint32_t pieceAt(uint32_t g, uint32_t P) { return (P / LPR) * 1024u + (g * LPR + ((P % LPR) ^ swz(g))) * 16u; }
};

template <int LPR, bool ANY, int TRI, bool STATS>
__global__ __launch_bounds__(256) void bvh8TraceKernel(Bvh8TraceArgs a)
{
    using GE = Geo<LPR>;
    constexpr uint32_t G = GE::G, CPL = GE::CPL;
    static_assert(LPR == 8 || LPR == 4, "lanes per ray");
    extern __shared__ __align__(16) unsigned char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t g = lane / LPR, q = lane % LPR;
    unsigned char* const stage = lds + wave * GE::kStage;
    uint2* const order = reinterpret_cast<uint2*>(lds + kWavesPerBlock * GE::kStage + wave * GE::kOrder + g * 64u);
    uint2* const stack = reinterpret_cast<uint2*>(lds + kWavesPerBlock * (GE::kStage + GE::kOrder) + (wave * G + g) * stackStrideBytes(a.stack_entries));
    const uint32_t groupShift = g * LPR;
    // Only the first stack_entries entries of a ray's stack live in LDS; deeper ones (rare: the window is
    // sized for the common case so that more blocks fit a CU) go to a per-group slice of a global spill
    // buffer, written and read with agent-scope accesses so that the lanes of the group see each other's
    // entries.
    const uint32_t W = a.stack_entries;
    unsigned long long* const spill = a.spill ? a.spill + (size_t(blockIdx.x * kWavesPerBlock + wave) * G + g) * a.spill_entries : nullptr;
    auto stackStore = [&](int idx, uint2 v) {
        if (uint32_t(idx) < W) {
            stack[idx] = v;
        } else {
            __hip_atomic_store(&spill[uint32_t(idx) - W], (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto stackLoad = [&](int idx) {
        if (uint32_t(idx) < W)
            return stack[idx];
        const unsigned long long v = __hip_atomic_load(&spill[uint32_t(idx) - W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2(uint32_t(v), uint32_t(v >> 32));
    };

    // LDS read offsets of this lane: children q*CPL .. q*CPL+CPL-1 of node array j
    // (minX,maxX,minY,maxY,minZ,maxZ,children,perm; array j = bytes [32 j, 32 j + 32) of the node)
    const unsigned char* nodeRd[8];
#pragma unroll
    for (uint32_t j = 0; j < 8; ++j)
        nodeRd[j] = stage + GE::pieceAt(g, 2u * j + ((q * CPL * 4u) >> 4)) + ((q * CPL * 4u) & 15u);
    // triangle (q & 3) of a leaf: 16-byte pieces 3t, 3t+1, 3t+2
    const unsigned char* triRd[3];
#pragma unroll
    for (uint32_t j = 0; j < 3; ++j)
        triRd[j] = stage + GE::pieceAt(g, 3u * (q & 3u) + j);
    const uint32_t myPiece = q ^ GE::swz(g); // piece (within each DMA instruction's LPR pieces) this lane fetches

    // per-ray state, replicated over the lanes of the group
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, tnear = 0, tfar = 0;
    float hitU = 0, hitV = 0;
    uint32_t hitObj = PGPU_MISS_ID, shift = 0, rayId = 0, bvhId = 0, tag = 0;
    const unsigned char* nodes = nullptr;
    const unsigned char* tris = nullptr;
    int sp = 0, hitMark = 0;
    bool active = false, exact = false;
    bool pend = false;   // a leaf has been popped (handle) but not intersected yet
    uint32_t handle = 0; // last popped stack entry
    const unsigned char *rdNearX = nodeRd[0], *rdFarX = nodeRd[1], *rdNearY = nodeRd[2], *rdFarY = nodeRd[3], *rdNearZ = nodeRd[4], *rdFarZ = nodeRd[5];
    uint32_t chunkNext = 0, chunkEnd = 0; // wave-uniform
    bool exhausted = false;
    uint32_t statInner = 0, statLeaf = 0, statTri = 0, statIter = 0, statLeafPass = 0, statRefill = 0;

    for (;;) {
        // ------------------------------------------------------------ refill idle groups
        const uint64_t idle = __ballot(!active) & GE::kOneBitPerGroup;
        const uint32_t nIdle = __popcll(idle);
        // a refill pass costs a full trip through the ray-setup code for the whole wave: wait until a few
        // groups are idle (or nothing else is left to do) so that one pass serves several rays
        if (!exhausted && (nIdle >= a.refill_min || nIdle == G)) {
            if (chunkNext >= chunkEnd) {
                uint32_t c = 0;
                if (lane == 0)
                    c = atomicAdd(a.cursor, a.chunk);
                c = __builtin_amdgcn_readfirstlane(c);
                chunkNext = c;
                chunkEnd = min(c + a.chunk, a.n_work);
                if (c >= a.n_work) {
                    exhausted = true;
                    chunkEnd = chunkNext;
                }
            }
            const uint32_t rank = __popcll(idle & ((1ull << groupShift) - 1ull));
            const uint32_t w = chunkNext + rank;
            const bool take = !active && w < chunkEnd;
            chunkNext = min(chunkNext + nIdle, chunkEnd);
            if (take) {
                rayId = a.work_ray ? a.work_ray[w] : w;
                bvhId = a.work_bvh ? a.work_bvh[w] : a.fixed_bvh;
                tag = a.work_tag ? a.work_tag[w] : 0u;
                const Bvh8Dev bd = a.bvhs[bvhId];
                nodes = bd.nodes;
                tris = bd.tris;
                ox = a.ox[rayId];
                oy = a.oy[rayId];
                oz = a.oz[rayId];
                dx = a.dx[rayId];
                dy = a.dy[rayId];
                dz = a.dz[rayId];
                tnear = a.tnear[rayId];
                tfar = a.tfar_in[rayId];
                ix = 1.0f / dx; // wive_bvh8_impl.h:66-68: no zero guard
                iy = 1.0f / dy;
                iz = 1.0f / dz;
                shift = ((dx > 0 ? 1u : 0u) | (dy > 0 ? 2u : 0u) | (dz > 0 ? 4u : 0u)) * 3u; // :384-388
                // rays that can meet 0 * inf (or carry inf/NaN themselves) take the x86-exact slab test
                const float big = 3.4028234663852886e38f;
                const bool tame = fabsf(ix) <= big && fabsf(iy) <= big && fabsf(iz) <= big && ix != 0.0f && iy != 0.0f && iz != 0.0f
                    && fabsf(ox) <= big && fabsf(oy) <= big && fabsf(oz) <= big && tnear == tnear && tfar == tfar;
                exact = !tame || bd.needs_exact != 0u;
                rdNearX = ix < 0.0f ? nodeRd[1] : nodeRd[0];
                rdFarX = ix < 0.0f ? nodeRd[0] : nodeRd[1];
                rdNearY = iy < 0.0f ? nodeRd[3] : nodeRd[2];
                rdFarY = iy < 0.0f ? nodeRd[2] : nodeRd[3];
                rdNearZ = iz < 0.0f ? nodeRd[5] : nodeRd[4];
                rdFarZ = iz < 0.0f ? nodeRd[4] : nodeRd[5];
                hitObj = PGPU_MISS_ID;
                hitU = hitV = 0.0f;
                if (q == 0)
                    stack[0] = make_uint2(bd.root, __float_as_uint(0.0f)); // :80-82
                sp = 1;
                hitMark = 0;
                active = true;
            }
            if (STATS)
                statRefill += (lane == 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (__ballot(active) == 0) {
            if (exhausted)
                break;
            continue;
        }

        // ------------------------------------------------------------ pop: the top LPR entries at once
        // A ray that popped a leaf keeps the handle and sits out (pend) until the wave runs a leaf pass: the
        // triangle code is long, so it runs once several rays of the wave wait for it (or nothing else is
        // left to do) rather than in every step for the one or two rays that happen to be at a leaf.
        const bool popNow = active && !pend;
        bool isInner = false;
        {
            const int e = sp - 1 - int(q);
            uint2 ent = make_uint2(0u, 0u);
            if (popNow && e >= 0)
                ent = stackLoad(e);
            const bool ok = popNow && e >= 0 && (e >= hitMark || __uint_as_float(ent.y) < tfar);
            const uint32_t gb = uint32_t(__ballot(ok) >> groupShift) & GE::kGroupMask;
            const uint32_t first = gb != 0 ? uint32_t(__builtin_ctz(gb)) : 0u;
            const uint32_t popped = uint32_t(__builtin_amdgcn_ds_bpermute(int((lane - q + first) << 2), int(ent.x)));
            if (popNow) {
                if (gb != 0) {
                    sp = sp - 1 - int(first);
                    handle = popped;
                    isInner = (popped >> 29) == 2u;
                    pend = !isInner;
                } else {
                    sp = max(sp - LPR, 0); // every candidate was culled
                }
                hitMark = min(hitMark, sp);
            }
        }
        const uint64_t pendMask = __ballot(pend);
        const bool leafPass = pendMask != 0 && (uint32_t(__popcll(pendMask & GE::kOneBitPerGroup)) >= a.leaf_min || __ballot(isInner) == 0);
        const bool isLeaf = pend && leafPass;
        const bool have = isInner || isLeaf;
        const uint32_t index = handle & 0x1FFFFFFFu;
        const uint32_t count = ((handle >> 29) & 3u) + 1u; // :378-382
        if (STATS) {
            statIter += (lane == 0);
            statLeafPass += (lane == 0 && leafPass);
        }

        // ------------------------------------------------------------ fetch node / leaf triangles into LDS
        {
            const unsigned char* base = isInner ? nodes + size_t(index) * 256u : tris + size_t(index) * 48u;
            const uint32_t pieces = isInner ? 16u : 3u * count;
#pragma unroll
            for (uint32_t h = 0; h < GE::NDMA; ++h)
                if (have && h * LPR + myPiece < pieces)
                    ldsDma16(base + (h * LPR + myPiece) * 16u, stage + h * 1024u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }

        // ------------------------------------------------------------ inner node: slab test of this lane's children
        if (isInner) {
            float tmin[CPL], tmax[CPL];
            uint32_t child[CPL], perm[CPL];
            float nX[CPL], fX[CPL], nY[CPL], fY[CPL], nZ[CPL], fZ[CPL];
            const bool slow = __ballot(exact) != 0;
#pragma unroll
            for (uint32_t i = 0; i < CPL; ++i) {
                child[i] = ldsU(nodeRd[6] + 4u * i);
                perm[i] = ldsU(nodeRd[7] + 4u * i);
            }
            if (!slow) {
                // No ray of this wave can produce a NaN (finite non-zero 1/d, finite origin, ordered boxes), so
                // min(t1,t2) is the plane on the ray's near side: picked through the LDS address (rdNear/rdFar,
                // set per ray from the sign of 1/d), and the max/min chains are plain IEEE max3/min3.
#pragma unroll
                for (uint32_t i = 0; i < CPL; ++i) {
                    nX[i] = ldsF(rdNearX + 4u * i);
                    fX[i] = ldsF(rdFarX + 4u * i);
                    nY[i] = ldsF(rdNearY + 4u * i);
                    fY[i] = ldsF(rdFarY + 4u * i);
                    nZ[i] = ldsF(rdNearZ + 4u * i);
                    fZ[i] = ldsF(rdFarZ + 4u * i);
                }
#pragma unroll
                for (uint32_t i = 0; i < CPL; ++i) {
                    const float tnx = (nX[i] - ox) * ix, tfx = (fX[i] - ox) * ix;
                    const float tny = (nY[i] - oy) * iy, tfy = (fY[i] - oy) * iy;
                    const float tnz = (nZ[i] - oz) * iz, tfz = (fZ[i] - oz) * iz;
                    tmin[i] = fmaxf(tnear, fmaxf(fmaxf(tnx, tny), tnz));
                    tmax[i] = fminf(tfar, fminf(fminf(tfx, tfy), tfz));
                }
            } else {
#pragma unroll
                for (uint32_t i = 0; i < CPL; ++i) {
                    const float lox = ldsF(nodeRd[0] + 4u * i), hix = ldsF(nodeRd[1] + 4u * i);
                    const float loy = ldsF(nodeRd[2] + 4u * i), hiy = ldsF(nodeRd[3] + 4u * i);
                    const float loz = ldsF(nodeRd[4] + 4u * i), hiz = ldsF(nodeRd[5] + 4u * i);
                    const float tx1 = (lox - ox) * ix, tx2 = (hix - ox) * ix; // :212-217
                    const float ty1 = (loy - oy) * iy, ty2 = (hiy - oy) * iy;
                    const float tz1 = (loz - oz) * iz, tz2 = (hiz - oz) * iz;
                    const float txMin = min86(tx1, tx2), tyMin = min86(ty1, ty2), tzMin = min86(tz1, tz2);
                    const float txMax = max86(tx1, tx2), tyMax = max86(ty1, ty2), tzMax = max86(tz1, tz2);
                    tmin[i] = max86(tnear, max86(txMin, max86(tyMin, tzMin))); // :224
                    tmax[i] = min86(tfar, min86(txMax, min86(tyMax, tzMax)));  // :225
                }
            }
            // slot j (far -> near) takes the child named by the octant permutation (:229-232,235); a child that
            // fails tmin <= tmax (:233) travels as 0xFFFFFFFF
            uint32_t pChild[CPL];
            float pTmin[CPL];
            if (CPL == 1) {
                const uint32_t src = (lane - q) + ((perm[0] >> shift) & 7u);
                pChild[0] = uint32_t(__builtin_amdgcn_ds_bpermute(int(src << 2), int(tmin[0] <= tmax[0] ? child[0] : 0xFFFFFFFFu)));
                pTmin[0] = __int_as_float(__builtin_amdgcn_ds_bpermute(int(src << 2), __float_as_int(tmin[0])));
            } else {
#pragma unroll
                for (uint32_t i = 0; i < CPL; ++i)
                    order[q * CPL + i] = make_uint2(tmin[i] <= tmax[i] ? child[i] : 0xFFFFFFFFu, __float_as_uint(tmin[i]));
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (uint32_t i = 0; i < CPL; ++i) {
                    const uint2 e = order[(perm[i] >> shift) & 7u];
                    pChild[i] = e.x;
                    pTmin[i] = __uint_as_float(e.y);
                }
            }
            // compress (:234-237): rank of every kept slot among the kept slots of this ray, slot order
            uint32_t rank = 0, total = 0;
            bool keep[CPL];
#pragma unroll
            for (uint32_t i = 0; i < CPL; ++i) {
                keep[i] = pChild[i] != 0xFFFFFFFFu;
                const uint32_t gb = uint32_t(__ballot(keep[i]) >> groupShift) & GE::kGroupMask;
                rank += __popc(gb & ((1u << q) - 1u));
                total += __popc(gb);
            }
#pragma unroll
            for (uint32_t i = 0; i < CPL; ++i) {
                if (keep[i])
                    stackStore(sp + int(rank), make_uint2(pChild[i], __float_as_uint(pTmin[i])));
                rank += keep[i] ? 1u : 0u;
            }
            sp += int(total);
            if (STATS)
                statInner += (q == 0);
        }

        // ------------------------------------------------------------ leaf: <= 4 triangles, one per lane
        if (isLeaf) {
            TriHit th;
            th.hit = false;
            th.t = th.u = th.v = 0.0f;
            uint32_t obj = 0;
            if (q < count) {
                const float4 p0 = *reinterpret_cast<const float4*>(triRd[0]);
                const float4 p1 = *reinterpret_cast<const float4*>(triRd[1]);
                const float4 p2 = *reinterpret_cast<const float4*>(triRd[2]);
                obj = __float_as_uint(p0.w);
                th = TRI == PGPU_TRI_WATERTIGHT_ID ? triWatertight(ox, oy, oz, dx, dy, dz, tnear, tfar, p0, p1, p2)
                                                   : triMT(ox, oy, oz, dx, dy, dz, tnear, tfar, p0, p1, p2);
                if (STATS)
                    statTri += 1;
            }
            if (STATS)
                statLeaf += (q == 0);
            // The reference tests the leaf's objects in order, each seeing the tfar left by the previous
            // one and accepting t <= tfar (:272-280). The survivor of that sequence is the LAST object
            // whose t equals the minimum of the valid t's.
            float best = th.hit ? th.t : __builtin_inff();
            best = fminf(best, __shfl_xor(best, 1));
            best = fminf(best, __shfl_xor(best, 2));
            const bool winner = th.hit && th.t == best;
            const uint32_t gb = uint32_t(__ballot(winner) >> groupShift) & 0xFu;
            if (gb != 0) {
                const uint32_t w = 31u - __builtin_clz(gb);
                const int srcAddr = int((lane - q + w) << 2);
                if (ANY) {
                    // :193-196: first hit ends the ray
                    if (q == 0)
                        a.any_out[rayId] = 1u;
                    active = false;
                    sp = 0;
                } else {
                    tfar = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(th.t)));
                    hitU = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(th.u)));
                    hitV = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(th.v)));
                    hitObj = uint32_t(__builtin_amdgcn_ds_bpermute(srcAddr, int(obj)));
                    hitMark = sp; // everything still on the stack predates this hit
                }
            }
            pend = false;
        }

        // ------------------------------------------------------------ retire rays whose stack ran dry
        if (active && sp == 0 && !pend) {
            if (q == 0) {
                if (ANY) {
                    if (a.init_miss)
                        a.any_out[rayId] = 0u;
                } else {
                    a.t_out[rayId] = tfar;
                    if (hitObj != PGPU_MISS_ID) {
                        a.obj_out[rayId] = hitObj;
                        a.leaf_out[rayId] = tag;
                        a.u_out[rayId] = hitU;
                        a.v_out[rayId] = hitV;
                    } else if (a.init_miss) {
                        a.obj_out[rayId] = PGPU_MISS_ID;
                        a.leaf_out[rayId] = 0u;
                        a.u_out[rayId] = 0.0f;
                        a.v_out[rayId] = 0.0f;
                    }
                }
            }
            active = false;
        }
        if (!active)
            exact = false;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    if (STATS && a.stats) {
        atomicAdd(&a.stats[0], (unsigned long long)statInner);
        atomicAdd(&a.stats[1], (unsigned long long)statLeaf);
        atomicAdd(&a.stats[2], (unsigned long long)statTri);
        if (lane == 0) {
            atomicAdd(&a.stats[3], (unsigned long long)statIter);
            atomicAdd(&a.stats[4], (unsigned long long)statLeafPass);
            atomicAdd(&a.stats[5], (unsigned long long)statRefill);
        }
    }
}

} // namespace

size_t bvh8TraceLdsBytes(uint32_t stackEntries, int lanesPerRay)
{
    const uint32_t groups = 64u / uint32_t(lanesPerRay);
    const uint32_t order = lanesPerRay == 8 ? 0u : groups * 64u;
    return kWavesPerBlock * (groups * 256u + order) + kWavesPerBlock * groups * stackStrideBytes(stackEntries);
}

template <int LPR>
static hipError_t launchLpr(const Bvh8TraceArgs& args, int kind, int triMode, bool stats, uint32_t blocks, hipStream_t stream)
{
    const dim3 grid(blocks), block(256);
    const size_t ldsBytes = bvh8TraceLdsBytes(args.stack_entries, LPR);
#define PANDORA_LAUNCH(ANYV, TRIV, STATV) \
    hipLaunchKernelGGL((bvh8TraceKernel<LPR, ANYV, TRIV, STATV>), grid, block, ldsBytes, stream, args)
    if (kind == PGPU_ANY_ID) {
        if (triMode == PGPU_TRI_WATERTIGHT_ID) {
            if (stats) PANDORA_LAUNCH(true, 1, true); else PANDORA_LAUNCH(true, 1, false);
        } else {
            if (stats) PANDORA_LAUNCH(true, 0, true); else PANDORA_LAUNCH(true, 0, false);
        }
    } else {
        if (triMode == PGPU_TRI_WATERTIGHT_ID) {
            if (stats) PANDORA_LAUNCH(false, 1, true); else PANDORA_LAUNCH(false, 1, false);
        } else {
            if (stats) PANDORA_LAUNCH(false, 0, true); else PANDORA_LAUNCH(false, 0, false);
        }
    }
#undef PANDORA_LAUNCH
    return hipGetLastError();
}

hipError_t launchBvh8Trace(const Bvh8TraceArgs& args, int kind, int triMode, bool stats, int lanesPerRay, uint32_t blocks, hipStream_t stream)
{
    return lanesPerRay == 4 ? launchLpr<4>(args, kind, triMode, stats, blocks, stream) : launchLpr<8>(args, kind, triMode, stats, blocks, stream);
}

} // namespace pandora_hip
