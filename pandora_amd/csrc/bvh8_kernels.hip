// Bottom-level WiVe BVH8 traversal for MI355X (gfx950, wave64).
//
// Reference semantics: WiVeBVH8::intersect / intersectAny / intersectInnerNode / intersectLeaf
//   projects/pandora/include/pandora/traversal/bvh/wive_bvh8_impl.h:57-140, :142-201, :209-238, :272-290
// and the leaf test TriangleShape::intersectPrimitive
//   projects/pandora/src/shapes/triangle_intersect.cpp:425-462 (Moller-Trumbore, the compiled branch)
//   and :340-423 (watertight variant, selectable).
//
// Mapping. The reference runs one ray at a time with the 8 children of a node in the 8 lanes of a ymm register.
// Here a wave64 carries 16 rays, one per quad of lanes: lane q owns children 2q and 2q+1 of the node its ray
// visits and triangle q of its leaf, and values move inside a quad with DPP quad_perm modifiers. Per step every
// ray pops one stack entry and the wave issues four LDS-DMA loads (global_load_lds_dwordx4) that bring the
// 16 x 256-byte nodes of its rays into LDS. Child ordering is the node's precomputed octant permutation applied
// through an LDS slot table, kept children are ranked with a byte-packed quad sum and pushed to a per-ray stack in
// LDS (a window; deeper entries spill to HBM). Leaf triangles are loaded straight into registers and tested in
// batches (rays wait at a leaf until several of the wave do). Waves are persistent: they draw chunks of rays from
// per-XCD cursors until the work list runs out. DESIGN.md 3.1 has the details and the measurements.
//
// The reference compacts its whole stack after every leaf hit (dist < tfar). Order is preserved by
// that compaction, so culling lazily at pop time is equivalent: entries pushed before the last hit
// (index < hitMark) must satisfy dist < tfar, entries pushed after it are kept unconditionally.
//
// Float semantics: contraction off; min/max written as compare+select so that NaN behaves like
// x86 minps/maxps (second operand wins), see SURVEY.md F10.
#include "kernels.h"
#include "wave_ops.h"
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace pandora_hip {

namespace {

constexpr uint32_t kWavesPerBlock = 4;
// Per-ray stack: the reference reserves 96 entries (wive_bvh8_impl.h:74-75); a tree with d nested inner
// nodes never holds more than 7 d + 8, so the launch sizes the LDS window from the deepest uploaded tree and
// from what lets 6 blocks share a CU. The stride is an odd number of 8-byte entries, which spreads the 16
// stacks of a wave over the LDS banks.
__host__ __device__ constexpr uint32_t stackStrideBytes(uint32_t entries) { return (entries * 8u) | 8u; }
// Quad layout: the four lanes of a quad pop by reading the ray's top four entries, 32 contiguous bytes, with ds_read_b64 --
// banked over 64 dwords per half-wave, i.e. per 8 rays. A stride of an odd number of 32-byte units puts the 8 rays' reads on
// 8 different eighths of the bank row whenever their stack depths agree modulo 4 (88 bytes, the old stride for 11 entries, made
// rays 0 / 3 / 6 and 1 / 4 / 7 overlap whatever the depths).
__host__ __device__ constexpr uint32_t quadStackStrideBytes(uint32_t entries) { return (((entries * 8u + 31u) / 32u) | 1u) * 32u; }

__device__ __forceinline__ float min86(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float max86(float a, float b) { return a > b ? a : b; }

// One triangle test, split the way the leaf loop needs it: `pass` = every rejection that does not involve the ray
// interval; (ka, kb) = what the interval test compares (Moller-Trumbore: t itself; watertight: tScaled and det);
// hit = pass and inside [tnear, tfar] for the tfar the ray had when the leaf was entered.
struct TriHit {
    float t, u, v;
    float ka, kb;
    bool pass, hit;
};

// the interval rejection of the two tests, for an arbitrary far end (triangle_intersect.cpp:450 and :412-415)
template <int TRI>
__device__ __forceinline__ bool intervalReject(float ka, float kb, float tnear, float tfar)
{
    if (TRI == PGPU_TRI_WATERTIGHT_ID)
        return (kb < 0.0f && (ka >= tnear * kb || ka < tfar * kb)) || (kb > 0.0f && (ka <= tnear * kb || ka > tfar * kb));
    return ka < tnear || ka > tfar;
}

// triangle_intersect.cpp:425-462, operation order kept (glm::cross / glm::dot expansion). The reference leaves at the first
// failed test; a wave leaves only when all its lanes do, so every lane computes everything and the four rejections (the same
// comparisons, so NaNs fall through them exactly as in the reference) are combined at the end.
__device__ __forceinline__ TriHit triMT(float ox, float oy, float oz, float dx, float dy, float dz, float tnear, float tfar,
    float4 a, float4 b, float4 c)
{
    const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
    const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
    const float hx = dy * e2z - e2y * dz;
    const float hy = dz * e2x - e2z * dx;
    const float hz = dx * e2y - e2x * dy;
    const float det = (e1x * hx + e1y * hy) + e1z * hz;
    const bool rejectDet = det > -0.000001f && det < 0.000001f;
    const float f = 1.0f / det;
    const float sx = ox - a.x, sy = oy - a.y, sz = oz - a.z;
    const float u = f * ((sx * hx + sy * hy) + sz * hz);
    const bool rejectU = u < 0.0f || u > 1.0f;
    const float qx = sy * e1z - e1y * sz;
    const float qy = sz * e1x - e1z * sx;
    const float qz = sx * e1y - e1x * sy;
    const float v = f * ((dx * qx + dy * qy) + dz * qz);
    const bool rejectV = v < 0.0f || u + v > 1.0f;
    const float t = f * ((e2x * qx + e2y * qy) + e2z * qz);
    TriHit r;
    r.pass = !(rejectDet || rejectU || rejectV);
    r.ka = t;
    r.kb = 0.0f;
    r.hit = r.pass && !intervalReject<0>(t, 0.0f, tnear, tfar);
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
    const bool rejectSign = (e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f);
    const float det = e0 + e1 + e2;
    const bool rejectDet = det == 0.0f;
    p0z *= Sz;
    p1z *= Sz;
    p2z *= Sz;
    const float tScaled = e0 * p0z + e1 * p1z + e2 * p2z;
    const float invDet = 1.0f / det;
    r.pass = !(rejectSign || rejectDet);
    r.ka = tScaled;
    r.kb = det;
    r.hit = r.pass && !intervalReject<1>(tScaled, det, tnear, tfar);
    r.u = e0 * invDet;
    r.v = e1 * invDet;
    r.t = tScaled * invDet;
    return r;
}

// ---------------------------------------------------------------------------------------------------
// Wave layout: a wave64 carries 16 rays, one per quad of lanes (DPP quad_perm moves data inside a quad for
// free). Lane q of a quad owns children 2q and 2q+1 of the node its ray visits and triangle q of its leaf.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kRaysPerWave = 16;
constexpr uint32_t kStagePerWave = kRaysPerWave * 256u; // one 256-byte node per ray

typedef float f2 __attribute__((ext_vector_type(2)));

// element `i` of a wave-uniform array through a 32-bit byte offset (batches are limited to 2^30 rays): one shift serves
// all output arrays and the stores use the SGPR-base addressing form
template <typename T>
__device__ __forceinline__ T& at32(T* base, uint32_t byteOffset)
{
    return *reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(base) + byteOffset);
}

// The LDS side of a global_load_lds is M0 + immediate offset + lane * 16 (measured: tools/dma_probe.hip): the
// immediate offset moves the global AND the LDS address. Piece group h of a node (64 bytes further in memory)
// must land 1024 bytes further in LDS, so the LDS base handed over is 960 h below the wanted one.
template <int H>
__device__ __forceinline__ void nodeDma(const unsigned char* gsrc, __attribute__((address_space(3))) unsigned char* stageUniform)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)(stageUniform + H * 960), 16, H * 64, 0);
}

typedef __attribute__((address_space(3))) unsigned char lds_u8;
// the other array of a min/max pair (they sit 32 bytes apart and the staging area is 64-byte aligned)
__device__ __forceinline__ const lds_u8* ldsFlip32(const lds_u8* p) { return (const lds_u8*)(uintptr_t)(uint32_t(uintptr_t(p)) ^ 32u); }
__device__ __forceinline__ f2 ldsF2(const lds_u8* p) { return *reinterpret_cast<const __attribute__((address_space(3))) f2*>(p); }
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 ldsU2(const lds_u8* p)
{
    const u2v v = *reinterpret_cast<const __attribute__((address_space(3))) u2v*>(p);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void ldsStoreU2(lds_u8* p, uint2 v)
{
    u2v w;
    w.x = v.x;
    w.y = v.y;
    *reinterpret_cast<__attribute__((address_space(3))) u2v*>(p) = w;
}
__device__ __forceinline__ void ldsStoreU4(lds_u8* p, uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    u4v r;
    r.x = x;
    r.y = y;
    r.z = z;
    r.w = w;
    *reinterpret_cast<__attribute__((address_space(3))) u4v*>(p) = r;
}

// INST = the BVHs may hold instance leaves (OfflineBVHLeaf::Instance, offline_bvh_cache.cpp:54-61): a leaf whose single object
// is another BVH under a transform. The ray goes to the instance's space (Transform::transformToLocal, transform.cpp:40-45; t
// keeps its meaning) and the nested traversal runs on the same per-ray stack above a floor (spBase); when the stack is back at
// the floor the ray is reloaded and the outer traversal goes on with whatever tfar the instance left. One level of nesting.
template <bool ANY, int TRI, bool STATS, bool INST>
__device__ __forceinline__ void bvh8TraceBody(const Bvh8TraceArgs a) // (by value: through a reference the compiler keeps the arguments in SGPRs it then has to spill)
{
    extern __shared__ __align__(16) unsigned char ldsGeneric[];
    lds_u8* const lds = (lds_u8*)ldsGeneric;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    const uint32_t g = lane >> 2, q = lane & 3u;
    const uint32_t W = a.stack_entries;

    // LDS: [4 waves x 16 rays x 256 B node staging][4 waves x 16 rays x stack window]
    lds_u8* const stageU = lds + wave * kStagePerWave; // wave-uniform
    // 16-byte piece P of ray g's node sits at (P / 4) * 1024 + (g * 4 + ((P % 4) ^ swz)) * 16 with swz = 2 for the
    // quads 4..7 and 12..15: a lane reads its children pair (8 bytes of each 32-byte array) with ds_read_b64 and
    // the two halves of the wave then cover all 64 banks exactly once.
    const uint32_t swz = ((g >> 2) & 1u) << 1;
    lds_u8* const pairEven = stageU + g * 64u + q * 8u + (swz ? 32u : 0u); // arrays minX, minY, minZ, children (+1024 each)
    lds_u8* const pairOdd = stageU + g * 64u + q * 8u + (swz ? 0u : 32u);  // arrays maxX, maxY, maxZ, permutation
    // after the bounds have been read, the 64 bytes of the ray's minX/maxX arrays hold the 8 (child, tmin) slots
    lds_u8* const orderBase = stageU + g * 64u;
    lds_u8* const stackBase = lds + kWavesPerBlock * kStagePerWave + (wave * kRaysPerWave + g) * quadStackStrideBytes(W);
    unsigned long long* const spill = a.spill ? a.spill + (size_t(blockIdx.x * kWavesPerBlock + wave) * kRaysPerWave + g) * a.spill_entries : nullptr;
    const uint32_t myPiece16 = (q ^ swz) * 16u; // the 16-byte piece (of every 64) this lane fetches
    const uint32_t qbit = 1u << q;
    const uint32_t quadLane0x4 = (lane & ~3u) << 2;

    auto stackStore = [&](uint32_t idx, uint2 v) {
        if (idx < W) {
            ldsStoreU2(stackBase + idx * 8u, v);
        } else {
            __hip_atomic_store(&spill[idx - W], (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto stackLoad = [&](uint32_t idx) {
        if (idx < W)
            return ldsU2(stackBase + idx * 8u);
        const unsigned long long v = __hip_atomic_load(&spill[idx - W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2(uint32_t(v), uint32_t(v >> 32));
    };


    // per-ray state, replicated over the 4 lanes of the quad
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, tnear = 0, tfar = 0;
    float hitU = 0, hitV = 0;
    uint32_t hitObj = PGPU_MISS_ID, rayId = 0, tag = 0, handle = 0;
    // bits 4:0 = the octant's bit offset into a permutation word (:384-388); bit 8 = this ray needs the x86-exact
    // slab test (v_bfe_u32 reads only the low 5 bits of its offset operand)
    uint32_t shift = 0;
    const unsigned char* nodePiece = nullptr; // nodes + this lane's piece offset
    const unsigned char* triMine = nullptr;   // tris + q * 48: triangle q of leaf slot 0
    int sp = 0, hitMark = 0;
    // INST: floor of the nested traversal's part of the stack, the instance being traversed / of the current hit (leaf-object
    // ids of the outer BVH; PGPU_MISS_ID = none) and the outer BVH
    int spBase = 0;
    uint32_t curInst = PGPU_MISS_ID, hitInst = PGPU_MISS_ID, outerBvh = 0;
    // 0 idle, 1 traversing, 2 waiting at a popped leaf (handle) for the wave's next leaf pass, 3 (inside one step
    // only) popped an inner node (handle)
    int st = 0;
    // LDS offsets of the near-side plane pair per axis (min array if 1/d >= 0 else max array); the far side is 32 bytes
    // away in the direction given by bit 5
    const lds_u8 *rdNearX = pairEven, *rdNearY = pairEven, *rdNearZ = pairEven;
    uint32_t chunkNext = 0, chunkEnd = 0; // wave-uniform
    uint32_t range = blockIdx.x & 7u;     // blocks are dealt round-robin over the 8 XCDs
    // (a launch sized before the list's length was known: blocks the list cannot feed leave without probing the cursors. Only the
    // instanced variant reads the list's length from device memory: in the plain kernel, which is bound by instruction issue, the
    // extra scalar state cost 3 % on the single-level benchmark)
    bool exhausted = INST && a.n_work_dev != nullptr && blockIdx.x >= max(8u, (min(*a.n_work_dev, a.n_work) + 2u * kRaysPerWave * 4u - 1u) / (2u * kRaysPerWave * 4u));
    uint32_t statInner = 0, statLeaf = 0, statTri = 0, statIter = 0, statLeafPass = 0, statRefill = 0;

    // Ray set-up for the BVH `bd`: lane q of the quad brings component q of the origin (va) and of the direction (vb); lane 3
    // brings tnear and tfar. One correctly rounded division per lane; the quad shares the 11 values with DPP.
    auto setRay = [&](float va, float vb, const Bvh8Dev& bd) {
        nodePiece = bd.nodes + myPiece16;
        triMine = bd.tris + q * 48u;
        const float inv = 1.0f / vb; // wive_bvh8_impl.h:66-68: no zero guard (lane 3: unused)
        // rays that can meet 0 * inf (or carry inf/NaN themselves) take the x86-exact slab test
        const float big = 3.4028234663852886e38f;
        const bool tame = q < 3u ? (fabsf(inv) <= big && inv != 0.0f && fabsf(va) <= big) : (va == va && vb == vb);
        shift = quadOr(q < 3u && vb > 0.0f ? qbit : 0u) * 3u; // :384-388
        shift |= (quadOr(tame ? 0u : 1u) | (bd.needs_exact != 0u ? 1u : 0u)) << 8;
        ox = quadBcast<0>(va);
        oy = quadBcast<1>(va);
        oz = quadBcast<2>(va);
        tnear = quadBcast<3>(va);
        dx = quadBcast<0>(vb);
        dy = quadBcast<1>(vb);
        dz = quadBcast<2>(vb);
        tfar = quadBcast<3>(vb);
        ix = quadBcast<0>(inv);
        iy = quadBcast<1>(inv);
        iz = quadBcast<2>(inv);
        rdNearX = ix < 0.0f ? pairOdd : pairEven;
        rdNearY = (iy < 0.0f ? pairOdd : pairEven) + 1024;
        rdNearZ = (iz < 0.0f ? pairOdd : pairEven) + 2048;
    };

    for (bool done = false; !done;) {
        // ------------------------------------------------------------ refill idle quads
        const uint64_t idleMask = __ballot(st == 0);
        const uint32_t nIdle = uint32_t(__popcll(idleMask & kQuadLane0));
        uint64_t liveMask = ~idleMask;
        // a refill pass costs a trip through the ray-setup code for the whole wave: wait until a few quads are
        // idle (or nothing else is left to do) so that one pass serves several rays
        if (!exhausted && (nIdle >= a.refill_min || nIdle == kRaysPerWave)) {
            // The work list is cut into 8 contiguous ranges, one per group of blocks that share an XCD (and its
            // L2): neighbouring rays -- and, in the two-level loop, the rays of one leaf -- stay on one XCD. A wave
            // whose home range has run dry helps with the next one.
            for (uint32_t tried = 0; chunkNext >= chunkEnd && !exhausted;) {
                uint32_t c = 0;
                if (lane == 0)
                    c = atomicAdd(a.cursor + range * 8u, a.chunk);
                c = uint32_t(__builtin_amdgcn_readfirstlane(int(c)));
                uint32_t nWork = a.n_work, rangeSize = a.range_size;
                if (INST && a.n_work_dev) {
                    nWork = min(*a.n_work_dev, a.n_work);
                    rangeSize = (((nWork + 7u) / 8u + a.chunk - 1u) / a.chunk) * a.chunk;
                }
                const uint32_t lo = range * rangeSize, hi = min(lo + rangeSize, nWork);
                if (lo < hi && c < hi - lo) {
                    chunkNext = lo + c;
                    chunkEnd = min(chunkNext + a.chunk, hi);
                } else if (++tried == 8u) {
                    exhausted = true;
                } else {
                    range = (range + 1u) & 7u;
                }
            }
            // idle quads below this one (the marker bit sits in lane 3 so that a quad does not count itself)
            const uint64_t marks = idleMask & kQuadLane3;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(marks >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(marks), 0u));
            const uint32_t w = chunkNext + rank;
            const bool take = st == 0 && w < chunkEnd;
            chunkNext = uint32_t(__builtin_amdgcn_readfirstlane(int(min(chunkNext + nIdle, chunkEnd))));
            liveMask |= __ballot(take);
            if (take) {
                rayId = a.work_ray ? a.work_ray[w] : w;
                const uint32_t bvhId = a.work_bvh ? a.work_bvh[w] : a.fixed_bvh;
                tag = a.work_tag ? a.work_tag[w] : 0u;
                const Bvh8Dev bd = a.bvhs[min(bvhId, a.n_bvhs - 1u)];
                // lane q of the quad loads component q of the origin and of the direction (lane 3: tnear, tfar)
                const float* const loadA = q == 0 ? a.ox : (q == 1 ? a.oy : (q == 2 ? a.oz : a.tnear));
                const float* const loadB = q == 0 ? a.dx : (q == 1 ? a.dy : (q == 2 ? a.dz : a.tfar_in));
                if (INST && a.ray_aos) {
                    // rays as 32-byte records {ox, oy, oz, tnear}{dx, dy, dz, .} (written by the first top-level episode): the quad
                    // reads one sector instead of eight; the current far end still comes from the per-ray state
                    const float* const rec = reinterpret_cast<const float*>(a.ray_aos + 2u * size_t(rayId));
                    setRay(rec[q], q == 3u ? a.tfar_in[rayId] : rec[4u + q], bd);
                } else {
                    setRay(loadA[rayId], loadB[rayId], bd);
                }
                if (INST) {
                    spBase = 0;
                    curInst = hitInst = PGPU_MISS_ID;
                    outerBvh = min(bvhId, a.n_bvhs - 1u);
                }
                hitObj = PGPU_MISS_ID;
                hitU = hitV = 0.0f;
                if (q == 0)
                    ldsStoreU2(stackBase, make_uint2(bd.root, __float_as_uint(0.0f))); // :80-82
                sp = 1;
                hitMark = 0;
                st = (bvhId < a.n_bvhs && bd.nodes != nullptr) ? 1 : 0; // (a work item whose geometry is not resident here is dropped)
            }
            if (STATS)
                statRefill += (lane == 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (liveMask == 0) {
            done = exhausted;
        } else {
        // ------------------------------------------------------------ pop: the top 4 entries at once
        // A ray that popped a leaf keeps the handle and sits out (st 2) until the wave runs a leaf pass: the
        // triangle code is long, so it runs once several rays of the wave wait for it (or nothing else is
        // left to do) rather than in every step for the one or two rays that happen to be at a leaf.
        {
            const bool run = st == 1;
            const int top = sp - 1;
            const int e = top - int(q);
            const bool valid = run && e >= (INST ? spBase : 0);
            const bool deep = __ballot(uint32_t(sp) > W) != 0; // some stack reaches beyond the LDS window
            uint2 ent = make_uint2(0u, 0u);
            if (valid) {
                if (!deep)
                    ent = ldsU2(stackBase + uint32_t(e) * 8u);
                else
                    ent = stackLoad(uint32_t(e));
            }
            const bool ok = valid && (e >= hitMark || __uint_as_float(ent.y) < tfar);
            const uint32_t m = quadOr(ok ? qbit : 0u);
            const uint32_t first = uint32_t(__builtin_ctz(m | 16u));
            const uint32_t popped = uint32_t(__builtin_amdgcn_ds_bpermute(int(quadLane0x4 + (first << 2)), int(ent.x)));
            const bool got = run && m != 0u;
            sp = run ? (m != 0u ? top - int(first) : max(sp - 4, INST ? spBase : 0)) : sp; // (no survivor: all 4 candidates were culled)
            hitMark = min(hitMark, sp);
            handle = got ? popped : handle;
            st = got ? ((popped >> 29) == 2u ? 3 : 2) : st;
        }
        const uint64_t pendMask = __ballot(st == 2);
        const uint64_t innerMask = __ballot(st == 3);
        const uint32_t leafPassU = uint32_t(pendMask != 0) & (uint32_t(uint32_t(__popcll(pendMask & kQuadLane0)) >= a.leaf_min) | uint32_t(innerMask == 0));
        const bool leafPass = leafPassU != 0u;
        const uint32_t index = handle & 0x1FFFFFFFu;
        if (STATS) {
            statIter += (lane == 0);
            statLeafPass += (lane == 0 && leafPass);
        }

        // ------------------------------------------------------------ fetch: node -> LDS (4 x 1 KiB LDS-DMA per wave),
        // leaf triangles -> registers (record q of the leaf's <= 4 consecutive 48-byte records)
        if (st == 3) {
            const unsigned char* src = nodePiece + size_t(index) * 256u;
            nodeDma<0>(src, stageU);
            nodeDma<1>(src, stageU);
            nodeDma<2>(src, stageU);
            nodeDma<3>(src, stageU);
        }

        // ------------------------------------------------------------ leaf pass: <= 4 triangles, one per lane
        if (leafPass) {
            const bool isLeaf = st == 2;
            const bool triLane = isLeaf && q <= ((handle >> 29) & 3u); // count - 1 = low two flag bits (:378-382)
            // (t = +inf marks "no hit"; u, v and obj of such lanes are never selected)
            float tHit = __builtin_inff(), uHit = 0.0f, vHit = 0.0f, tRaw = 0.0f, ka = 0.0f, kb = 0.0f;
            uint32_t obj = 0, pass = 0, instWord = 0;
            if (triLane) {
                typedef float f4v __attribute__((ext_vector_type(4)));
                typedef const __attribute__((address_space(1))) f4v* gf4;
                gf4 t = (gf4)(triMine + size_t(index) * 48u);
                const f4v r0 = t[0], r1 = t[1], r2 = t[2];
                const float4 p0 = make_float4(r0.x, r0.y, r0.z, r0.w), p1 = make_float4(r1.x, r1.y, r1.z, r1.w), p2 = make_float4(r2.x, r2.y, r2.z, r2.w);
                obj = __float_as_uint(r0.w);
                if (INST)
                    instWord = __float_as_uint(r1.w); // bit 31: this slot is an instance (its record is a degenerate triangle: no hit)
                const TriHit th = TRI == PGPU_TRI_WATERTIGHT_ID ? triWatertight(ox, oy, oz, dx, dy, dz, tnear, tfar, p0, p1, p2)
                                                                : triMT(ox, oy, oz, dx, dy, dz, tnear, tfar, p0, p1, p2);
                uHit = th.u;
                vHit = th.v;
                tRaw = th.t;
                ka = th.ka;
                kb = th.kb;
                pass = th.pass ? 1u : 0u;
                if (th.hit)
                    tHit = th.t;
                if (STATS)
                    statTri += 1;
            }
            if (STATS)
                statLeaf += (isLeaf && q == 0);
            // The reference tests the leaf's objects in order, each seeing the tfar left by the previous one
            // (:272-280). With Moller-Trumbore and ordinary numbers a test accepts t <= tfar, so the survivor of
            // that sequence is the LAST object whose t equals the minimum of the accepted t's: one quad reduction.
            // A NaN t breaks that argument (every comparison with it is false: it is accepted, and so is whatever
            // comes after it), and the watertight test compares tScaled against tfar * det, which is not the same
            // as t <= tfar to the last bit: those cases replay the sequence over the quad's 4 lanes.
            uint32_t winners;
            float tWin = tHit;
            if (TRI == PGPU_TRI_WATERTIGHT_ID || __ballot(pass != 0u && tRaw != tRaw) != 0) {
                float cur = tfar;
                uint32_t win = 0;
#define PANDORA_LEAF_STEP(I)                                                                                         \
    {                                                                                                                \
        const float kaI = quadBcast<I>(ka), kbI = quadBcast<I>(kb), tI = quadBcast<I>(tRaw);                         \
        const bool acc = quadPerm<I, I, I, I>(int(pass)) != 0 && !intervalReject<TRI>(kaI, kbI, tnear, cur);         \
        cur = acc ? tI : cur;                                                                                        \
        win = acc ? (1u << I) : win;                                                                                 \
    }
                PANDORA_LEAF_STEP(0)
                PANDORA_LEAF_STEP(1)
                PANDORA_LEAF_STEP(2)
                PANDORA_LEAF_STEP(3)
#undef PANDORA_LEAF_STEP
                winners = win;
                tWin = tRaw;
            } else {
                const float best = quadMin(tHit);
                winners = quadOr(tHit == best && tHit != __builtin_inff() ? qbit : 0u);
            }
            const int srcAddr = int(quadLane0x4 + ((31u - uint32_t(__builtin_clz(winners | 1u))) << 2)); // (no winner: unused)
            const float wt = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(tWin)));
            const float wu = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(uHit)));
            const float wv = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(vHit)));
            const uint32_t wobj = uint32_t(__builtin_amdgcn_ds_bpermute(srcAddr, int(obj)));
            if (isLeaf) {
                st = 1;
                if ((winners & 0xFu) != 0u) {
                    if (ANY) {
                        // :193-196: first hit ends the ray
                        if (q == 0)
                            at32(a.any_out, rayId << 2) = 1u;
                        st = 0;
                        sp = 0;
                        shift = 0;
                    } else {
                        tfar = wt;
                        hitU = wu;
                        hitV = wv;
                        hitObj = wobj;
                        if (INST)
                            hitInst = curInst;
                        hitMark = sp; // everything still on the stack predates this hit
                    }
                }
            }
            if (INST) {
                // An instance leaf holds exactly one object (checked at upload): enter its BVH. Only from the outer level
                // (an instance inside an instance is rejected at upload).
                const uint32_t iw = uint32_t(quadPerm<0, 0, 0, 0>(int(instWord)));
                const uint32_t iobj = uint32_t(quadPerm<0, 0, 0, 0>(int(obj)));
                if (isLeaf && (iw & 0x80000000u) != 0u && curInst == PGPU_MISS_ID) {
                    const InstDev* const I = reinterpret_cast<const InstDev*>(a.bvhs[outerBvh].inst) + (iw & 0x7FFFFFFFu);
                    const uint32_t childId = I->child;
                    const Bvh8Dev cd = a.bvhs[min(childId, a.n_bvhs - 1u)];
                    if (childId < a.n_bvhs && cd.nodes != nullptr) {
                        // Transform::transformToLocal with glm's association (c0 x + c1 y) + (c2 z + c3 w): row q of the
                        // world-to-local matrix in lane q (lane 3 carries tnear / the current tfar through)
                        const float* const row = I->m + 4u * min(q, 2u);
                        const float m0 = row[0], m1 = row[1], m2 = row[2], m3 = row[3];
                        const float lo = (m0 * ox + m1 * oy) + (m2 * oz + m3 * 1.0f);
                        const float ld = (m0 * dx + m1 * dy) + (m2 * dz + m3 * 0.0f);
                        const float keepNear = tnear, keepFar = tfar;
                        setRay(q == 3u ? keepNear : lo, q == 3u ? keepFar : ld, cd);
                        spBase = sp;
                        if (q == 0)
                            stackStore(uint32_t(sp), make_uint2(cd.root, __float_as_uint(0.0f)));
                        sp += 1;
                        curInst = iobj;
                    }
                }
            }
        }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // ------------------------------------------------------------ inner node: slab test of this lane's two children
        if (st == 3) {
            const uint2 child = ldsU2(pairEven + 3072);
            const uint2 perm = ldsU2(pairOdd + 3072);
            f2 tmin, tmax;
            if (__ballot(shift >= 256u) == 0) {
                // No ray of this wave can produce a NaN (finite non-zero 1/d, finite origin, ordered boxes), so
                // min(t1,t2) is the plane on the ray's near side: picked through the LDS address (rdNear/rdFar,
                // set per ray from the sign of 1/d), and the max/min chains are plain IEEE max3/min3.
                const f2 nX = ldsF2(rdNearX), fX = ldsF2(ldsFlip32(rdNearX)), nY = ldsF2(rdNearY), fY = ldsF2(ldsFlip32(rdNearY));
                const f2 nZ = ldsF2(rdNearZ), fZ = ldsF2(ldsFlip32(rdNearZ));
                const f2 tnx = (nX - ox) * ix, tfx = (fX - ox) * ix;
                const f2 tny = (nY - oy) * iy, tfy = (fY - oy) * iy;
                const f2 tnz = (nZ - oz) * iz, tfz = (fZ - oz) * iz;
                tmin.x = fmaxf(tnear, fmaxf(fmaxf(tnx.x, tny.x), tnz.x));
                tmin.y = fmaxf(tnear, fmaxf(fmaxf(tnx.y, tny.y), tnz.y));
                tmax.x = fminf(tfar, fminf(fminf(tfx.x, tfy.x), tfz.x));
                tmax.y = fminf(tfar, fminf(fminf(tfx.y, tfy.y), tfz.y));
            } else {
                const f2 loX = ldsF2(pairEven), hiX = ldsF2(pairOdd), loY = ldsF2(pairEven + 1024), hiY = ldsF2(pairOdd + 1024);
                const f2 loZ = ldsF2(pairEven + 2048), hiZ = ldsF2(pairOdd + 2048);
                const float lo[3][2] = { { loX.x, loX.y }, { loY.x, loY.y }, { loZ.x, loZ.y } };
                const float hi[3][2] = { { hiX.x, hiX.y }, { hiY.x, hiY.y }, { hiZ.x, hiZ.y } };
                float tm[2], tM[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const float tx1 = (lo[0][i] - ox) * ix, tx2 = (hi[0][i] - ox) * ix; // :212-217
                    const float ty1 = (lo[1][i] - oy) * iy, ty2 = (hi[1][i] - oy) * iy;
                    const float tz1 = (lo[2][i] - oz) * iz, tz2 = (hi[2][i] - oz) * iz;
                    const float txMin = min86(tx1, tx2), tyMin = min86(ty1, ty2), tzMin = min86(tz1, tz2);
                    const float txMax = max86(tx1, tx2), tyMax = max86(ty1, ty2), tzMax = max86(tz1, tz2);
                    tm[i] = max86(tnear, max86(txMin, max86(tyMin, tzMin))); // :224
                    tM[i] = min86(tfar, min86(txMax, min86(tyMax, tzMax)));  // :225
                }
                tmin.x = tm[0];
                tmin.y = tm[1];
                tmax.x = tM[0];
                tmax.y = tM[1];
            }
            // slot j (far -> near) takes the child named by the octant permutation (:229-232,235); a child that
            // fails tmin <= tmax (:233) travels as 0xFFFFFFFF. The 8 (child, tmin) pairs of the ray go through the
            // (already consumed) first 64 bytes of its staged node.
            {
                ldsStoreU4(orderBase + q * 16u, tmin.x <= tmax.x ? child.x : 0xFFFFFFFFu, __float_as_uint(tmin.x),
                    tmin.y <= tmax.y ? child.y : 0xFFFFFFFFu, __float_as_uint(tmin.y));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint2 s0 = ldsU2(orderBase + (__builtin_amdgcn_ubfe(perm.x, shift, 3u) << 3));
            const uint2 s1 = ldsU2(orderBase + (__builtin_amdgcn_ubfe(perm.y, shift, 3u) << 3));
            // compress (:234-237): slots are pushed in slot order; this lane's slots are 2q and 2q+1
            const uint32_t k0 = s0.x != 0xFFFFFFFFu ? 1u : 0u, k1 = s1.x != 0xFFFFFFFFu ? 1u : 0u;
            const uint32_t counts = quadAdd((k0 + k1) << (q * 8u)); // byte l = kept slots of lane l
            const uint32_t at0 = __builtin_amdgcn_sad_u8(counts & ((1u << (q * 8u)) - 1u), 0u, uint32_t(sp));
            const uint32_t at1 = at0 + k0;
            const int spNew = int(__builtin_amdgcn_sad_u8(counts, 0u, uint32_t(sp)));
            if (__ballot(uint32_t(spNew) > W) == 0) {
                if (k0)
                    ldsStoreU2(stackBase + at0 * 8u, s0);
                if (k1)
                    ldsStoreU2(stackBase + at1 * 8u, s1);
            } else {
                if (k0)
                    stackStore(at0, s0);
                if (k1)
                    stackStore(at1, s1);
            }
            sp = spNew;
            st = 1;
            if (STATS)
                statInner += (q == 0);
        }

        // ------------------------------------------------------------ retire rays whose stack ran dry
        if (INST && st == 1 && sp == spBase && curInst != PGPU_MISS_ID) {
            // the instance's BVH is done: back to the outer traversal with the ray as it was, except for tfar
            const float* const loadA = q == 0 ? a.ox : (q == 1 ? a.oy : (q == 2 ? a.oz : a.tnear));
            const float* const loadB = q == 0 ? a.dx : (q == 1 ? a.dy : (q == 2 ? a.dz : a.tfar_in));
            const float keepFar = tfar;
            if (a.ray_aos) {
                const float* const rec = reinterpret_cast<const float*>(a.ray_aos + 2u * size_t(rayId));
                setRay(rec[q], q == 3u ? keepFar : rec[4u + min(q, 2u)], a.bvhs[outerBvh]);
            } else {
                const float vb = loadB[rayId];
                setRay(loadA[rayId], q == 3u ? keepFar : vb, a.bvhs[outerBvh]);
            }
            spBase = 0;
            curInst = PGPU_MISS_ID;
        } else if (st == 1 && sp == 0) {
            if (q == 0) {
                const uint32_t off = rayId << 2;
                if (ANY) {
                    if (a.init_miss)
                        at32(a.any_out, off) = 0u;
                } else {
                    at32(a.t_out, off) = tfar;
                    if (hitObj != PGPU_MISS_ID) {
                        at32(a.obj_out, off) = hitObj;
                        at32(a.leaf_out, off) = tag;
                        at32(a.u_out, off) = hitU;
                        at32(a.v_out, off) = hitV;
                        if (INST && a.inst_out)
                            at32(a.inst_out, off) = hitInst;
                    } else if (a.init_miss) {
                        at32(a.obj_out, off) = PGPU_MISS_ID;
                        at32(a.leaf_out, off) = 0u;
                        at32(a.u_out, off) = 0.0f;
                        at32(a.v_out, off) = 0.0f;
                        if (INST && a.inst_out)
                            at32(a.inst_out, off) = PGPU_MISS_ID;
                    }
                }
            }
            st = 0;
            shift = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        } // live rays
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

template <bool ANY, int TRI, bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(7, 7))) void bvh8TraceKernel(Bvh8TraceArgs a)
{
    bvh8TraceBody<ANY, TRI, STATS, false>(a);
}

// the same with instance leaves (a few more registers of per-ray state: one resident block less per CU)
template <bool ANY, int TRI, bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 7))) void bvh8TraceInstKernel(Bvh8TraceArgs a)
{
    bvh8TraceBody<ANY, TRI, STATS, true>(a);
}

// ---------------------------------------------------------------------------------------------------
// Pair layout: a wave64 carries 32 rays, one per pair of lanes. Lane p of a pair owns children 4p..4p+3 of the node
// its ray visits (loaded straight into registers: 8 x 16 bytes) and triangles 2p, 2p+1 of its leaf (two triangles
// per lane run through the packed-FP32 pipe together). The per-ray bookkeeping (pop, state, ranking, retire) is the
// same number of instructions as in the quad layout but serves twice as many rays. LDS holds only the 8 order slots
// and the stack window of every ray.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kPairRaysPerWave = 32;
constexpr uint32_t kPairSlotBytes = 80; // 8 x (child, tmin) + 16 bytes that spread the rays over the banks
constexpr uint64_t kPairLane0 = 0x5555555555555555ull;
constexpr uint64_t kPairLane1 = 0xAAAAAAAAAAAAAAAAull;

__device__ __forceinline__ int pairSwap(int v) { return quadPerm<1, 0, 3, 2>(v); }
__device__ __forceinline__ uint32_t pairOr(uint32_t v) { return v | uint32_t(pairSwap(int(v))); }
template <int L>
__device__ __forceinline__ float pairBcastF(float v) { return __int_as_float(quadPerm<L, L, 2 + L, 2 + L>(__float_as_int(v))); }
template <int L>
__device__ __forceinline__ int pairBcastI(int v) { return quadPerm<L, L, 2 + L, 2 + L>(v); }

__device__ __forceinline__ f2 mk2(float x, float y)
{
    f2 r;
    r.x = x;
    r.y = y;
    return r;
}

typedef float f4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 half2of(f4v_t v, int h) { return h ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1); }

struct TriHit2 {
    f2 t, u, v, ka, kb;
    bool pass[2], hit[2];
};

// triMT for two triangles at once (component i of every f2 belongs to triangle i): the same operations in the same
// order, issued as packed FP32 where the hardware has them (add, mul).
__device__ __forceinline__ TriHit2 triMT2(float ox, float oy, float oz, float dx, float dy, float dz, float tnear, float tfar,
    f2 ax, f2 ay, f2 az, f2 bx, f2 by, f2 bz, f2 cx, f2 cy, f2 cz)
{
    const f2 e1x = bx - ax, e1y = by - ay, e1z = bz - az;
    const f2 e2x = cx - ax, e2y = cy - ay, e2z = cz - az;
    const f2 hx = dy * e2z - e2y * dz;
    const f2 hy = dz * e2x - e2z * dx;
    const f2 hz = dx * e2y - e2x * dy;
    const f2 det = (e1x * hx + e1y * hy) + e1z * hz;
    f2 f;
    f.x = 1.0f / det.x;
    f.y = 1.0f / det.y;
    const f2 sx = ox - ax, sy = oy - ay, sz = oz - az;
    const f2 u = f * ((sx * hx + sy * hy) + sz * hz);
    const f2 qx = sy * e1z - e1y * sz;
    const f2 qy = sz * e1x - e1z * sx;
    const f2 qz = sx * e1y - e1x * sy;
    const f2 v = f * ((dx * qx + dy * qy) + dz * qz);
    const f2 t = f * ((e2x * qx + e2y * qy) + e2z * qz);
    const f2 uv = u + v;
    TriHit2 r;
    r.pass[0] = !((det.x > -0.000001f && det.x < 0.000001f) || (u.x < 0.0f || u.x > 1.0f) || (v.x < 0.0f || uv.x > 1.0f));
    r.pass[1] = !((det.y > -0.000001f && det.y < 0.000001f) || (u.y < 0.0f || u.y > 1.0f) || (v.y < 0.0f || uv.y > 1.0f));
    r.hit[0] = r.pass[0] && !intervalReject<0>(t.x, 0.0f, tnear, tfar);
    r.hit[1] = r.pass[1] && !intervalReject<0>(t.y, 0.0f, tnear, tfar);
    r.t = t;
    r.u = u;
    r.v = v;
    r.ka = t;
    r.kb = mk2(0.0f, 0.0f);
    return r;
}

template <bool ANY, int TRI, bool STATS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 5))) void bvh8TracePairKernel(Bvh8TraceArgs a)
{
    extern __shared__ __align__(16) unsigned char ldsGeneric[];
    lds_u8* const lds = (lds_u8*)ldsGeneric;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    const uint32_t r = lane >> 1, p = lane & 1u;
    const uint32_t W = a.stack_entries;

    // LDS: [4 waves x 32 rays x order slots][4 waves x 32 rays x stack window]
    lds_u8* const slotBase = lds + (wave * kPairRaysPerWave + r) * kPairSlotBytes;
    lds_u8* const stackBase = lds + kWavesPerBlock * kPairRaysPerWave * kPairSlotBytes + (wave * kPairRaysPerWave + r) * stackStrideBytes(W);
    unsigned long long* const spill = a.spill ? a.spill + (size_t(blockIdx.x * kWavesPerBlock + wave) * kPairRaysPerWave + r) * a.spill_entries : nullptr;
    const uint32_t pairLane0x4 = (lane & ~1u) << 2;
    const uint32_t belowMask = p ? 0xFFFFFFFFu : 0u;

    auto stackStore = [&](uint32_t idx, uint2 v) {
        if (idx < W) {
            ldsStoreU2(stackBase + idx * 8u, v);
        } else {
            __hip_atomic_store(&spill[idx - W], (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    auto stackLoad = [&](uint32_t idx) {
        if (idx < W)
            return ldsU2(stackBase + idx * 8u);
        const unsigned long long v = __hip_atomic_load(&spill[idx - W], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint2(uint32_t(v), uint32_t(v >> 32));
    };

    // per-ray state, replicated over the 2 lanes of the pair
    float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0, ix = 0, iy = 0, iz = 0, tnear = 0, tfar = 0;
    float hitU = 0, hitV = 0;
    uint32_t hitObj = PGPU_MISS_ID, rayId = 0, tag = 0, handle = 0;
    uint32_t shift = 0; // bits 4:0 octant bit offset, bit 8 exact slab test (as in the quad kernel)
    const unsigned char* nodesP = nullptr; // nodes + p * 16: this lane's half of every 32-byte array
    const unsigned char* trisP = nullptr;  // tris + p * 96: triangles 2p, 2p+1 of leaf slot 0
    // byte offset of the near-side plane array per axis (min array if 1/d >= 0 else max array); the far one is 32 bytes
    // away (offset ^ 32)
    uint32_t offNX = 0, offNY = 64, offNZ = 128;
    int sp = 0, hitMark = 0;
    int st = 0; // 0 idle, 1 traversing, 2 waiting at a popped leaf, 3 popped an inner node (inside one step)
    uint32_t chunkNext = 0, chunkEnd = 0; // wave-uniform
    uint32_t range = blockIdx.x & 7u;
    bool exhausted = a.n_work_dev != nullptr && blockIdx.x >= max(8u, (min(*a.n_work_dev, a.n_work) + 2u * kPairRaysPerWave * 4u - 1u) / (2u * kPairRaysPerWave * 4u));
    uint32_t statInner = 0, statLeaf = 0, statTri = 0, statIter = 0, statLeafPass = 0, statRefill = 0;

    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) f4v* gf4;
    typedef const __attribute__((address_space(1))) u4v* gu4;

    for (bool done = false; !done;) {
        // ------------------------------------------------------------ refill idle pairs
        const uint64_t idleMask = __ballot(st == 0);
        const uint32_t nIdle = uint32_t(__popcll(idleMask & kPairLane0));
        uint64_t liveMask = ~idleMask;
        if (!exhausted && (nIdle >= a.refill_min || nIdle == kPairRaysPerWave)) {
            for (uint32_t tried = 0; chunkNext >= chunkEnd && !exhausted;) {
                uint32_t c = 0;
                if (lane == 0)
                    c = atomicAdd(a.cursor + range * 8u, a.chunk);
                c = uint32_t(__builtin_amdgcn_readfirstlane(int(c)));
                // (n_work_dev: the list's length is known on the device only -- rounds queued ahead of the host's knowledge)
                uint32_t nWork = a.n_work, rangeSize = a.range_size;
                if (a.n_work_dev) {
                    nWork = min(*a.n_work_dev, a.n_work);
                    rangeSize = (((nWork + 7u) / 8u + a.chunk - 1u) / a.chunk) * a.chunk;
                }
                const uint32_t lo = range * rangeSize, hi = min(lo + rangeSize, nWork);
                if (lo < hi && c < hi - lo) {
                    chunkNext = lo + c;
                    chunkEnd = min(chunkNext + a.chunk, hi);
                } else if (++tried == 8u) {
                    exhausted = true;
                } else {
                    range = (range + 1u) & 7u;
                }
            }
            const uint64_t marks = idleMask & kPairLane1; // (the marker sits in lane 1 so that a pair does not count itself)
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(marks >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(marks), 0u));
            const uint32_t w = chunkNext + rank;
            const bool take = st == 0 && w < chunkEnd;
            chunkNext = uint32_t(__builtin_amdgcn_readfirstlane(int(min(chunkNext + nIdle, chunkEnd))));
            liveMask |= __ballot(take);
            if (take) {
                rayId = a.work_ray ? a.work_ray[w] : w;
                const uint32_t bvhId = a.work_bvh ? a.work_bvh[w] : a.fixed_bvh;
                tag = a.work_tag ? a.work_tag[w] : 0u;
                const Bvh8Dev bd = a.bvhs[min(bvhId, a.n_bvhs - 1u)];
                nodesP = bd.nodes + p * 16u;
                trisP = bd.tris + p * 96u;
                if (a.ray_aos) {
                    const float4 o = a.ray_aos[2u * size_t(rayId)], d = a.ray_aos[2u * size_t(rayId) + 1u];
                    ox = o.x;
                    oy = o.y;
                    oz = o.z;
                    tnear = o.w;
                    dx = d.x;
                    dy = d.y;
                    dz = d.z;
                } else {
                    ox = a.ox[rayId];
                    oy = a.oy[rayId];
                    oz = a.oz[rayId];
                    dx = a.dx[rayId];
                    dy = a.dy[rayId];
                    dz = a.dz[rayId];
                    tnear = a.tnear[rayId];
                }
                tfar = a.tfar_in[rayId];
                ix = 1.0f / dx; // wive_bvh8_impl.h:66-68: no zero guard
                iy = 1.0f / dy;
                iz = 1.0f / dz;
                // rays that can meet 0 * inf (or carry inf/NaN themselves) take the x86-exact slab test
                const float big = 3.4028234663852886e38f;
                const bool tame = fabsf(ix) <= big && ix != 0.0f && fabsf(ox) <= big && fabsf(iy) <= big && iy != 0.0f && fabsf(oy) <= big
                    && fabsf(iz) <= big && iz != 0.0f && fabsf(oz) <= big && tnear == tnear && tfar == tfar;
                shift = ((dx > 0.0f ? 1u : 0u) | (dy > 0.0f ? 2u : 0u) | (dz > 0.0f ? 4u : 0u)) * 3u; // :384-388
                shift |= ((tame ? 0u : 1u) | (bd.needs_exact != 0u ? 1u : 0u)) << 8;
                offNX = ix < 0.0f ? 32u : 0u;
                offNY = iy < 0.0f ? 96u : 64u;
                offNZ = iz < 0.0f ? 160u : 128u;
                hitObj = PGPU_MISS_ID;
                hitU = hitV = 0.0f;
                if (p == 0)
                    ldsStoreU2(stackBase, make_uint2(bd.root, __float_as_uint(0.0f))); // :80-82
                sp = 1;
                hitMark = 0;
                st = (bvhId < a.n_bvhs && bd.nodes != nullptr) ? 1 : 0;
            }
            if (STATS)
                statRefill += (lane == 0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (liveMask == 0) {
            done = exhausted;
        } else {
        // ------------------------------------------------------------ pop: the top 4 entries at once (2 per lane)
        {
            const bool run = st == 1;
            const int top = sp - 1;
            const int e0 = top - 2 * int(p), e1 = e0 - 1;
            const bool valid0 = run && e0 >= 0, valid1 = run && e1 >= 0;
            const bool deep = __ballot(uint32_t(sp) > W) != 0;
            uint2 ent0 = make_uint2(0u, 0u), ent1 = make_uint2(0u, 0u);
            if (valid0) {
                if (!deep) {
                    ent0 = ldsU2(stackBase + uint32_t(e0) * 8u);
                    ent1 = ldsU2(stackBase + uint32_t(max(e1, 0)) * 8u);
                } else {
                    ent0 = stackLoad(uint32_t(e0));
                    ent1 = stackLoad(uint32_t(max(e1, 0)));
                }
            }
            const bool ok0 = valid0 && (e0 >= hitMark || __uint_as_float(ent0.y) < tfar);
            const bool ok1 = valid1 && (e1 >= hitMark || __uint_as_float(ent1.y) < tfar);
            const uint32_t m = pairOr(((ok0 ? 1u : 0u) | (ok1 ? 2u : 0u)) << (2u * p));
            const uint32_t first = uint32_t(__builtin_ctz(m | 16u));
            const uint32_t cand = (first & 1u) ? ent1.x : ent0.x;
            const uint32_t popped = uint32_t(__builtin_amdgcn_ds_bpermute(int(pairLane0x4 + ((first >> 1) << 2)), int(cand)));
            const bool got = run && m != 0u;
            sp = run ? (m != 0u ? top - int(first) : max(sp - 4, 0)) : sp;
            hitMark = min(hitMark, sp);
            handle = got ? popped : handle;
            st = got ? ((popped >> 29) == 2u ? 3 : 2) : st;
        }
        const uint64_t pendMask = __ballot(st == 2);
        const uint64_t innerMask = __ballot(st == 3);
        const uint32_t leafPassU = uint32_t(pendMask != 0) & (uint32_t(uint32_t(__popcll(pendMask & kPairLane0)) >= a.leaf_min) | uint32_t(innerMask == 0));
        const bool leafPass = leafPassU != 0u;
        const uint32_t index = handle & 0x1FFFFFFFu;
        if (STATS) {
            statIter += (lane == 0);
            statLeafPass += (lane == 0 && leafPass);
        }

        // ------------------------------------------------------------ fetch: this lane's half of the node -> registers
        // (deliberately not initialised: the lanes that do not load do not use them either, and zeroing 32 registers
        // per step costs more than the slab test)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsometimes-uninitialized"
#pragma clang diagnostic ignored "-Wuninitialized"
        f4v nearX, farX, nearY, farY, nearZ, farZ;
        u4v child4, perm4;
        if (st == 3) {
            const unsigned char* nb = nodesP + size_t(index) * 256u;
            nearX = *(gf4)(nb + offNX);
            farX = *(gf4)(nb + (offNX ^ 32u));
            nearY = *(gf4)(nb + offNY);
            farY = *(gf4)(nb + (offNY ^ 32u));
            nearZ = *(gf4)(nb + offNZ);
            farZ = *(gf4)(nb + (offNZ ^ 32u));
            child4 = *(gu4)(nb + 192);
            perm4 = *(gu4)(nb + 224);
        }
#pragma clang diagnostic pop

        // ------------------------------------------------------------ leaf pass: <= 4 triangles, two per lane
        if (leafPass) {
            const bool isLeaf = st == 2;
            const uint32_t last = (handle >> 29) & 3u; // count - 1 (:378-382)
            const bool tri0 = isLeaf && 2u * p <= last, tri1 = isLeaf && 2u * p + 1u <= last;
            f2 tHit = mk2(__builtin_inff(), __builtin_inff()), uHit = mk2(0.0f, 0.0f), vHit = mk2(0.0f, 0.0f), tRaw = mk2(0.0f, 0.0f), ka = mk2(0.0f, 0.0f), kb = mk2(0.0f, 0.0f);
            uint32_t obj0 = 0, obj1 = 0, pass0 = 0, pass1 = 0;
            if (tri0) {
                gf4 t = (gf4)(trisP + size_t(index) * 48u);
                // (the record after the leaf's last one exists: the array ends in a zero record)
                const f4v a0 = t[0], a1 = t[1], a2 = t[2], b0 = t[3], b1 = t[4], b2 = t[5];
                obj0 = __float_as_uint(a0.w);
                obj1 = __float_as_uint(b0.w);
                if (TRI == PGPU_TRI_WATERTIGHT_ID) {
                    const TriHit h0 = triWatertight(ox, oy, oz, dx, dy, dz, tnear, tfar, make_float4(a0.x, a0.y, a0.z, 0.0f),
                        make_float4(a1.x, a1.y, a1.z, 0.0f), make_float4(a2.x, a2.y, a2.z, 0.0f));
                    const TriHit h1 = triWatertight(ox, oy, oz, dx, dy, dz, tnear, tfar, make_float4(b0.x, b0.y, b0.z, 0.0f),
                        make_float4(b1.x, b1.y, b1.z, 0.0f), make_float4(b2.x, b2.y, b2.z, 0.0f));
                    uHit = mk2(h0.u, h1.u);
                    vHit = mk2(h0.v, h1.v);
                    tRaw = mk2(h0.t, h1.t);
                    ka = mk2(h0.ka, h1.ka);
                    kb = mk2(h0.kb, h1.kb);
                    pass0 = h0.pass ? 1u : 0u;
                    pass1 = (h1.pass && tri1) ? 1u : 0u;
                    if (h0.hit)
                        tHit.x = h0.t;
                    if (h1.hit && tri1)
                        tHit.y = h1.t;
                } else {
                    const TriHit2 th = triMT2(ox, oy, oz, dx, dy, dz, tnear, tfar, mk2(a0.x, b0.x), mk2(a0.y, b0.y), mk2(a0.z, b0.z),
                        mk2(a1.x, b1.x), mk2(a1.y, b1.y), mk2(a1.z, b1.z), mk2(a2.x, b2.x), mk2(a2.y, b2.y), mk2(a2.z, b2.z));
                    uHit = th.u;
                    vHit = th.v;
                    tRaw = th.t;
                    ka = th.ka;
                    kb = th.kb;
                    pass0 = th.pass[0] ? 1u : 0u;
                    pass1 = (th.pass[1] && tri1) ? 1u : 0u;
                    if (th.hit[0])
                        tHit.x = th.t.x;
                    if (th.hit[1] && tri1)
                        tHit.y = th.t.y;
                }
                if (STATS)
                    statTri += 1u + (tri1 ? 1u : 0u);
            }
            if (STATS)
                statLeaf += (isLeaf && p == 0);
            // (see the quad kernel for why the fast path is "last object whose t equals the minimum" and when the
            // sequence has to be replayed)
            uint32_t winners;
            f2 tWin = tHit;
            if (TRI == PGPU_TRI_WATERTIGHT_ID || __ballot((pass0 != 0u && tRaw.x != tRaw.x) || (pass1 != 0u && tRaw.y != tRaw.y)) != 0) {
                float cur = tfar;
                uint32_t win = 0;
#define PANDORA_LEAF_STEP2(L, C, PASS, BIT)                                                                          \
    {                                                                                                                \
        const float kaI = pairBcastF<L>(ka.C), kbI = pairBcastF<L>(kb.C), tI = pairBcastF<L>(tRaw.C);                \
        const bool acc = pairBcastI<L>(int(PASS)) != 0 && !intervalReject<TRI>(kaI, kbI, tnear, cur);                \
        cur = acc ? tI : cur;                                                                                        \
        win = acc ? (1u << BIT) : win;                                                                               \
    }
                PANDORA_LEAF_STEP2(0, x, pass0, 0)
                PANDORA_LEAF_STEP2(0, y, pass1, 1)
                PANDORA_LEAF_STEP2(1, x, pass0, 2)
                PANDORA_LEAF_STEP2(1, y, pass1, 3)
#undef PANDORA_LEAF_STEP2
                winners = win;
                tWin = tRaw;
            } else {
                const float bestLocal = fminf(tHit.x, tHit.y);
                const float best = fminf(bestLocal, __int_as_float(pairSwap(__float_as_int(bestLocal))));
                const uint32_t mine = ((tHit.x == best ? 1u : 0u) | (tHit.y == best ? 2u : 0u)) << (2u * p);
                winners = best != __builtin_inff() ? pairOr(mine) : 0u;
            }
            const uint32_t widx = 31u - uint32_t(__builtin_clz(winners | 1u)); // (no winner: unused)
            const bool second = (widx & 1u) != 0u;
            const int srcAddr = int(pairLane0x4 + ((widx >> 1) << 2));
            const float wt = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(second ? tWin.y : tWin.x)));
            const float wu = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(second ? uHit.y : uHit.x)));
            const float wv = __int_as_float(__builtin_amdgcn_ds_bpermute(srcAddr, __float_as_int(second ? vHit.y : vHit.x)));
            const uint32_t wobj = uint32_t(__builtin_amdgcn_ds_bpermute(srcAddr, int(second ? obj1 : obj0)));
            if (isLeaf) {
                st = 1;
                if ((winners & 0xFu) != 0u) {
                    if (ANY) {
                        if (p == 0)
                            at32(a.any_out, rayId << 2) = 1u; // :193-196: first hit ends the ray
                        st = 0;
                        sp = 0;
                        shift = 0;
                    } else {
                        tfar = wt;
                        hitU = wu;
                        hitV = wv;
                        hitObj = wobj;
                        hitMark = sp; // everything still on the stack predates this hit
                    }
                }
            }
        }

        // ------------------------------------------------------------ inner node: slab test of this lane's four children
        if (st == 3) {
            float tmin[4], tmax[4];
            if (__ballot(shift >= 256u) == 0) {
                // near/far planes were picked through the load address (see the quad kernel)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f2 tnx = (half2of(nearX, h) - ox) * ix, tfx = (half2of(farX, h) - ox) * ix;
                    const f2 tny = (half2of(nearY, h) - oy) * iy, tfy = (half2of(farY, h) - oy) * iy;
                    const f2 tnz = (half2of(nearZ, h) - oz) * iz, tfz = (half2of(farZ, h) - oz) * iz;
                    tmin[2 * h] = fmaxf(tnear, fmaxf(fmaxf(tnx.x, tny.x), tnz.x));
                    tmin[2 * h + 1] = fmaxf(tnear, fmaxf(fmaxf(tnx.y, tny.y), tnz.y));
                    tmax[2 * h] = fminf(tfar, fminf(fminf(tfx.x, tfy.x), tfz.x));
                    tmax[2 * h + 1] = fminf(tfar, fminf(fminf(tfx.y, tfy.y), tfz.y));
                }
            } else {
                // which of the two loaded arrays is min and which is max
                const bool sxn = (offNX & 32u) != 0u, syn = (offNY & 32u) != 0u, szn = (offNZ & 32u) != 0u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float loX = sxn ? farX[i] : nearX[i], hiX = sxn ? nearX[i] : farX[i];
                    const float loY = syn ? farY[i] : nearY[i], hiY = syn ? nearY[i] : farY[i];
                    const float loZ = szn ? farZ[i] : nearZ[i], hiZ = szn ? nearZ[i] : farZ[i];
                    const float tx1 = (loX - ox) * ix, tx2 = (hiX - ox) * ix; // :212-217
                    const float ty1 = (loY - oy) * iy, ty2 = (hiY - oy) * iy;
                    const float tz1 = (loZ - oz) * iz, tz2 = (hiZ - oz) * iz;
                    const float txMin = min86(tx1, tx2), tyMin = min86(ty1, ty2), tzMin = min86(tz1, tz2);
                    const float txMax = max86(tx1, tx2), tyMax = max86(ty1, ty2), tzMax = max86(tz1, tz2);
                    tmin[i] = max86(tnear, max86(txMin, max86(tyMin, tzMin))); // :224
                    tmax[i] = min86(tfar, min86(txMax, min86(tyMax, tzMax)));  // :225
                }
            }
            // slot j (far -> near) takes the child named by the octant permutation (:229-232,235); a child that fails
            // tmin <= tmax (:233) travels as 0xFFFFFFFF
            ldsStoreU4(slotBase + p * 32u, tmin[0] <= tmax[0] ? child4.x : 0xFFFFFFFFu, __float_as_uint(tmin[0]),
                tmin[1] <= tmax[1] ? child4.y : 0xFFFFFFFFu, __float_as_uint(tmin[1]));
            ldsStoreU4(slotBase + p * 32u + 16u, tmin[2] <= tmax[2] ? child4.z : 0xFFFFFFFFu, __float_as_uint(tmin[2]),
                tmin[3] <= tmax[3] ? child4.w : 0xFFFFFFFFu, __float_as_uint(tmin[3]));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint2 s0 = ldsU2(slotBase + (__builtin_amdgcn_ubfe(perm4.x, shift, 3u) << 3));
            const uint2 s1 = ldsU2(slotBase + (__builtin_amdgcn_ubfe(perm4.y, shift, 3u) << 3));
            const uint2 s2 = ldsU2(slotBase + (__builtin_amdgcn_ubfe(perm4.z, shift, 3u) << 3));
            const uint2 s3 = ldsU2(slotBase + (__builtin_amdgcn_ubfe(perm4.w, shift, 3u) << 3));
            // compress (:234-237): slots are pushed in slot order; this lane's slots are 4p..4p+3
            const uint32_t k0 = s0.x != 0xFFFFFFFFu ? 1u : 0u, k1 = s1.x != 0xFFFFFFFFu ? 1u : 0u;
            const uint32_t k2 = s2.x != 0xFFFFFFFFu ? 1u : 0u, k3 = s3.x != 0xFFFFFFFFu ? 1u : 0u;
            const uint32_t tot = (k0 + k1) + (k2 + k3);
            const uint32_t other = uint32_t(pairSwap(int(tot)));
            const uint32_t at0 = uint32_t(sp) + (other & belowMask);
            const uint32_t at1 = at0 + k0, at2 = at1 + k1, at3 = at2 + k2;
            const int spNew = sp + int(tot + other);
            if (__ballot(uint32_t(spNew) > W) == 0) {
                if (k0)
                    ldsStoreU2(stackBase + at0 * 8u, s0);
                if (k1)
                    ldsStoreU2(stackBase + at1 * 8u, s1);
                if (k2)
                    ldsStoreU2(stackBase + at2 * 8u, s2);
                if (k3)
                    ldsStoreU2(stackBase + at3 * 8u, s3);
            } else {
                if (k0)
                    stackStore(at0, s0);
                if (k1)
                    stackStore(at1, s1);
                if (k2)
                    stackStore(at2, s2);
                if (k3)
                    stackStore(at3, s3);
            }
            sp = spNew;
            st = 1;
            if (STATS)
                statInner += (p == 0);
        }

        // ------------------------------------------------------------ retire rays whose stack ran dry
        if (st == 1 && sp == 0) {
            if (p == 0) {
                const uint32_t off = rayId << 2;
                if (ANY) {
                    if (a.init_miss)
                        at32(a.any_out, off) = 0u;
                } else {
                    at32(a.t_out, off) = tfar;
                    if (hitObj != PGPU_MISS_ID) {
                        at32(a.obj_out, off) = hitObj;
                        at32(a.leaf_out, off) = tag;
                        at32(a.u_out, off) = hitU;
                        at32(a.v_out, off) = hitV;
                    } else if (a.init_miss) {
                        at32(a.obj_out, off) = PGPU_MISS_ID;
                        at32(a.leaf_out, off) = 0u;
                        at32(a.u_out, off) = 0.0f;
                        at32(a.v_out, off) = 0.0f;
                    }
                }
            }
            st = 0;
            shift = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        } // live rays
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

size_t bvh8TraceLdsBytes(uint32_t stackEntries, uint32_t lanesPerRay)
{
    if (lanesPerRay == 2u)
        return kWavesPerBlock * kPairRaysPerWave * (kPairSlotBytes + stackStrideBytes(stackEntries));
    return kWavesPerBlock * kStagePerWave + kWavesPerBlock * kRaysPerWave * quadStackStrideBytes(stackEntries);
}

hipError_t launchBvh8Trace(const Bvh8TraceArgs& args, int kind, int triMode, bool stats, uint32_t blocks, hipStream_t stream)
{
    const dim3 grid(blocks), block(256);
    if ((args.n_work_dev || args.ray_aos) && (args.instanced ? args.lanes_per_ray == 2u : args.lanes_per_ray != 2u))
        return hipErrorInvalidValue; // (only the pair-layout kernel and the instanced quad kernel read their list length from device memory / rays as records)
    const size_t ldsBytes = bvh8TraceLdsBytes(args.stack_entries, args.lanes_per_ray);
#define PANDORA_LAUNCH(ANYV, TRIV, STATV)                                                                            \
    do {                                                                                                             \
        if (args.instanced)                                                                                          \
            hipLaunchKernelGGL((bvh8TraceInstKernel<ANYV, TRIV, STATV>), grid, block, ldsBytes, stream, args);       \
        else if (args.lanes_per_ray == 2u)                                                                           \
            hipLaunchKernelGGL((bvh8TracePairKernel<ANYV, TRIV, STATV>), grid, block, ldsBytes, stream, args);       \
        else                                                                                                         \
            hipLaunchKernelGGL((bvh8TraceKernel<ANYV, TRIV, STATV>), grid, block, ldsBytes, stream, args);           \
    } while (0)
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

} // namespace pandora_hip
