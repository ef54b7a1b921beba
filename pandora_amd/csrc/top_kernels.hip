// Top level of the two-level structure on MI355X: PauseableBVH4 traversal that parks rays at
// batching points, the SVDAG proxy-geometry early-out, and the binning of parked rays by leaf.
//
// Reference semantics:
//   PauseableBVH4::intersectT            pauseable_bvh/pauseable_bvh4.h:225-355
//   BatchingPoint::intersect/intersectAny offline_batching_acceleration_structure.h:269-305
//   SparseVoxelDAG::intersectScalar      src/svo/sparse_voxel_dag.cpp:350-567
//   per-leaf queues + batch execution    stream_framework task_graph.h:130-156,264-293 (replaced by
//                                        a histogram / scan / scatter over SoA records in HBM)
//
// One ray per lane. A kernel instance runs ONE episode per ray: from its saved (node, stack) token
// until the next batching point whose SVDAG it hits (the ray is parked: token saved, record appended)
// or until it leaves the tree. The reference never visits children outside validMask on purpose
// either, but it also never masks them; here they are masked.
//
// The SVDAG cast drops the fourth SIMD lane of the reference (the (x,y,z,1) padding lane whose
// 0x07FFFFFF mask only burns iterations); the oracle proves the boolean is unchanged.
#include "kernels.h"
#include "wave_ops.h"
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace pandora_hip {

namespace {

constexpr uint32_t kTopBlock = 256;
constexpr uint32_t kTopNodeBytes = 112;
constexpr uint32_t kTopLdsNodeBudget = 40 * 1024; // nodes staged in LDS when they fit
constexpr uint32_t kTopHistMax = 4096;             // batching points whose per-block counters fit LDS
constexpr int kCastDepth = 23;                     // CAST_STACK_DEPTH, sparse_voxel_dag.cpp:311

__device__ __forceinline__ float min86(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float max86(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float stdmin(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float stdmax(float a, float b) { return a < b ? b : a; }
__device__ __forceinline__ float exp2i(int e) { return __uint_as_float(uint32_t(e + 127) << 23); }

// sparse_voxel_dag.cpp:350-567 with three lanes, cut into "set up" and "one trip of the loop" so that the same code serves the
// cast inside the top-level episode (run to completion per lane) and the cast kernel (lanes take new requests as their casts
// finish). Results: 0 = still running, 1 = hit, 2 = miss.
// Where a cast reads its nodes: the shared array (index = offset there), or one DAG's image staged in LDS, whose nodes refer
// to each other by local offset and to nodes that did not fit by kDagGlobalFlag | offset in the shared array (everything
// below such a node is read from the shared array; its offsets carry no flag there, so the flag is added on the way).
struct GlobalDag {
    const uint32_t* __restrict__ words;
    __device__ __forceinline__ uint32_t node(uint32_t p) const { return words[p]; }
    __device__ __forceinline__ uint32_t child(uint32_t p, uint32_t k) const { return words[p + 1u + k]; }
};
struct StagedDag {
    const uint32_t* lds;
    const uint32_t* __restrict__ words;
    __device__ __forceinline__ uint32_t node(uint32_t p) const { return (p & kDagGlobalFlag) ? words[p & ~kDagGlobalFlag] : lds[p]; }
    __device__ __forceinline__ uint32_t child(uint32_t p, uint32_t k) const
    {
        return (p & kDagGlobalFlag) ? (words[(p & ~kDagGlobalFlag) + 1u + k] | kDagGlobalFlag) : lds[p + 1u + k];
    }
};

struct CastState {
    float tCoef[3], tBias[3], pos[3];
    float tMin;
    uint32_t idx, parent, desc, octBits, trips, tripCap;
    int scale;
};

template <typename Dag>
__device__ __forceinline__ int castInit(CastState& s, const Dag& words, uint32_t root, const SvdagDev& dag, float rox, float roy, float roz,
    float rdx, float rdy, float rdz)
{
    const float eps = 1.1920928955078125e-07f;
    float o[3] = { 1.0f + (dag.invExtent * (rox - dag.bmin[0])), 1.0f + (dag.invExtent * (roy - dag.bmin[1])), 1.0f + (dag.invExtent * (roz - dag.bmin[2])) };
    float d[3] = { rdx, rdy, rdz };
    uint32_t octMask = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (fabsf(d[i]) < eps)
            d[i] = copysignf(eps, d[i]);
        s.tCoef[i] = 1.0f / -fabsf(d[i]);
        s.tBias[i] = s.tCoef[i] * o[i];
        if (d[i] > 0.0f) {
            octMask |= 1u << i;
            s.tBias[i] = 3.0f * s.tCoef[i] - s.tBias[i];
        }
    }
    s.octBits = (7u ^ octMask) & 7u;
    const float tMin0 = stdmax(0.0f, stdmax(2.0f * s.tCoef[0] - s.tBias[0], stdmax(2.0f * s.tCoef[1] - s.tBias[1], 2.0f * s.tCoef[2] - s.tBias[2])));
    const float tMax0 = stdmin(s.tCoef[0] - s.tBias[0], stdmin(s.tCoef[1] - s.tBias[1], s.tCoef[2] - s.tBias[2]));
    if (tMin0 >= tMax0)
        return 2;
    s.tMin = tMin0;
    s.idx = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bool m = (1.5f * s.tCoef[i] - s.tBias[i]) > tMin0;
        s.idx |= uint32_t(m) << i;
        s.pos[i] = m ? 1.5f : 1.0f;
    }
    s.parent = root;
    // the descriptor word of `parent`: fetched when the cast arrives at a node (PUSH, POP), not in every trip -- an ADVANCE
    // that stays inside the node (the common trip in a sparse DAG) costs no memory access at all
    s.desc = words.node(s.parent);
    s.scale = kCastDepth - 1;
    // with NaN operands an ADVANCE makes no progress: after this many trips give up and do not cull (deviation 4)
    s.tripCap = (16u * dag.levels + 16u) << dag.levels;
    s.trips = 0;
    return 0;
}

// one trip of the loop (:425-540); `saved` is this lane's column of the parent stack. Precondition: s.scale < kCastDepth.
template <typename Dag>
__device__ __forceinline__ int castTrip(CastState& s, const Dag& words, uint32_t* saved, uint32_t savedStride)
{
    if (++s.trips > s.tripCap)
        return 1;
    float tCorner[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        tCorner[i] = s.pos[i] * s.tCoef[i] - s.tBias[i];
    const uint32_t child = 7u - ((s.idx & 7u) ^ s.octBits);
    const uint32_t valid = (s.desc >> 8) & 0xFFu, leaf = s.desc & 0xFFu;
    if (valid >> child & 1u) {
        // PUSH
        const float half = exp2i((s.scale - 1) - kCastDepth);
        saved[uint32_t(kCastDepth - 1 - s.scale) * savedStride] = s.parent;
        if ((valid & leaf) >> child & 1u)
            return 1;
        const uint32_t inner = valid & ~leaf;
        s.parent = words.child(s.parent, __popc(inner & ((1u << child) - 1u)));
        s.desc = words.node(s.parent);
        --s.scale;
        s.idx = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float tCenter = half * s.tCoef[i] + tCorner[i];
            const bool m = tCenter > s.tMin;
            s.idx |= uint32_t(m) << i;
            s.pos[i] = m ? s.pos[i] + half : s.pos[i];
        }
    } else {
        // ADVANCE
        const float size = exp2i(s.scale - kCastDepth);
        const float tcMax = stdmin(tCorner[0], stdmin(tCorner[1], tCorner[2]));
        uint32_t step = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bool st = tCorner[i] <= tcMax;
            step |= uint32_t(st) << i;
            s.pos[i] = st ? s.pos[i] - size : s.pos[i];
        }
        s.tMin = tcMax;
        s.idx ^= step;
        if ((s.idx & step) != 0) {
            // POP
            uint32_t diff = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (step >> i & 1u)
                    diff |= __float_as_uint(s.pos[i]) ^ __float_as_uint(s.pos[i] + size);
            s.scale = 31 - __clz(int(diff));
            if (s.scale < kCastDepth) {
                s.parent = saved[uint32_t(kCastDepth - 1 - s.scale) * savedStride];
                s.desc = words.node(s.parent);
            }
        }
        uint32_t sh[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sh[i] = __float_as_uint(s.pos[i]) >> s.scale;
            s.pos[i] = __uint_as_float(sh[i] << s.scale);
        }
        s.idx = (sh[0] & 1u) | ((sh[1] & 1u) << 1) | ((sh[2] & 1u) << 2);
    }
    return s.scale < kCastDepth ? 0 : 2;
}

__device__ bool svdagHit(const uint32_t* __restrict__ words, const SvdagDev& dag, float rox, float roy, float roz,
    float rdx, float rdy, float rdz, uint32_t* saved, uint32_t savedStride)
{
    CastState s;
    const GlobalDag g { words };
    int r = castInit(s, g, dag.root, dag, rox, roy, roz, rdx, rdy, rdz);
    while (r == 0)
        r = castTrip(s, g, saved, savedStride);
    return r == 1;
}

// simd4_avx2.ipp:412-430 (horizontalMinIndex): lane-wise two-round min, first lane equal to it, default 3
__device__ __forceinline__ uint32_t hminIndex4(const float v[4])
{
    float m2[4], m4[4];
    m2[0] = min86(v[0], v[2]);
    m2[1] = min86(v[1], v[3]);
    m2[2] = min86(v[2], v[0]);
    m2[3] = min86(v[3], v[1]);
    m4[0] = min86(m2[0], m2[1]);
    m4[1] = min86(m2[1], m2[0]);
    m4[2] = min86(m2[2], m2[3]);
    m4[3] = min86(m2[3], m2[2]);
    if (v[0] == m4[0])
        return 0;
    if (v[1] == m4[1])
        return 1;
    if (v[2] == m4[2])
        return 2;
    return 3;
}

// pauseable_bvh4.h:262-286: the four children's slab tests of one ray -- compare + select in the operand order of x86 minps /
// maxps, which is what the reference's simd4 executes (the order decides which operand a NaN replaces). An IEEE min3 / max3
// form for waves whose rays cannot produce a NaN (as in the bottom-level kernel) was measured: it needs two more registers than
// the 64 that keep 8 waves per SIMD resident, and this kernel lives on occupancy -- 7.24 ms per step against 7.08 ms.
__device__ __forceinline__ uint32_t slabs4(const float* nf, float ox, float oy, float oz, float ix, float iy, float iz, float tnear, float tfar, float tmin[4])
{
    uint32_t hitMask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float tx1 = (nf[0 + i] - ox) * ix, tx2 = (nf[12 + i] - ox) * ix;
        const float ty1 = (nf[4 + i] - oy) * iy, ty2 = (nf[16 + i] - oy) * iy;
        const float tz1 = (nf[8 + i] - oz) * iz, tz2 = (nf[20 + i] - oz) * iz;
        const float txMin = min86(tx1, tx2), tyMin = min86(ty1, ty2), tzMin = min86(tz1, tz2);
        const float txMax = max86(tx1, tx2), tyMax = max86(ty1, ty2), tzMax = max86(tz1, tz2);
        tmin[i] = max86(tnear, max86(txMin, max86(tyMin, tzMin)));
        const float tmax = min86(tfar, min86(txMax, min86(tyMax, tzMax)));
        hitMask |= uint32_t(tmin[i] <= tmax) << i;
    }
    return hitMask;
}

// the ray of an episode: from its 32-byte record when the trace keeps them (TopTraceArgs::ray_aos), else from the eight arrays
// -- and then also copied into a record when the trace wants them made (ray_aos_out)
__device__ __forceinline__ void loadRay(const TopTraceArgs& a, uint32_t ray, float& ox, float& oy, float& oz, float& dx, float& dy, float& dz, float& tnear)
{
    if (a.ray_aos) {
        const float4 o = a.ray_aos[2u * size_t(ray)], d = a.ray_aos[2u * size_t(ray) + 1u];
        ox = o.x;
        oy = o.y;
        oz = o.z;
        tnear = o.w;
        dx = d.x;
        dy = d.y;
        dz = d.z;
        return;
    }
    ox = a.ox[ray];
    oy = a.oy[ray];
    oz = a.oz[ray];
    dx = a.dx[ray];
    dy = a.dy[ray];
    dz = a.dz[ray];
    tnear = a.tnear[ray];
    // (written here, while the values are in hand: writing only the rays that park -- the only ones anybody reads again -- at the
    // end of the episode keeps seven more values alive through the traversal and was slower, 6.77 against 6.67 ms per step.
    // The fourth value of the second half is never read: the kernels take tfar from the state arrays.)
    if (a.ray_aos_out) {
        a.ray_aos_out[2u * size_t(ray)] = make_float4(ox, oy, oz, tnear);
        a.ray_aos_out[2u * size_t(ray) + 1u] = make_float4(dx, dy, dz, 0.0f);
    }
}
__device__ __forceinline__ void loadDirection(const TopTraceArgs& a, uint32_t ray, float& dx, float& dy, float& dz)
{
    if (a.ray_aos) {
        const float4 d = a.ray_aos[2u * size_t(ray) + 1u];
        dx = d.x;
        dy = d.y;
        dz = d.z;
        return;
    }
    dx = a.dx[ray];
    dy = a.dy[ray];
    dz = a.dz[ray];
}

// PHASED: a wave walks until every one of its rays has parked, left, or stands in front of a batching point's DAG, and only then
// runs the casts -- all of them together -- after which the rays whose cast missed walk on. In the plain form a cast runs inside
// the step that reached the DAG, so the rest of the wave waits for every cast of every step; phased, a wave runs a few cast
// phases per episode instead of one per step. The sequence of operations of each ray is the same.
template <bool ANY, bool NODES_IN_LDS, bool DEFER, bool PHASED = false>
__global__ __launch_bounds__(kTopBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void topTraverseKernel(TopTraceArgs a)
{
    // dynamic LDS: [dag_levels x 256 saved parents of the SVDAG cast][the top-level nodes, when they fit]
    // [2 SVDAG counters + one counter per batching point, when they fit]
    extern __shared__ __align__(16) unsigned char topLds[];
    uint32_t* const dagStack = reinterpret_cast<uint32_t*>(topLds);
    unsigned char* const ldsNodes = topLds + size_t(a.top.dag_levels) * kTopBlock * 4u;
    uint32_t* const ldsCount = reinterpret_cast<uint32_t*>(ldsNodes + (NODES_IN_LDS ? ((size_t(a.top.n_nodes) * kTopNodeBytes + 15u) & ~size_t(15)) : 16u));
    const bool countInLds = a.top.n_leaves <= kTopHistMax;
    uint32_t* const ldsReqCount = ldsCount + 2u + a.top.n_leaves; // DEFER, countInLds: cast requests per batching point
    const uint32_t nActive = a.n_active_dev ? min(*a.n_active_dev, a.n_active) : a.n_active;
    if (blockIdx.x * kTopBlock >= nActive)
        return; // (a launch sized before the list's length was known: nothing here for this block, skip the staging too)
    if (NODES_IN_LDS) {
        const uint32_t words = a.top.n_nodes * (kTopNodeBytes / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.top.nodes);
        uint32_t* dst = reinterpret_cast<uint32_t*>(ldsNodes);
        for (uint32_t i = threadIdx.x; i < words; i += kTopBlock)
            dst[i] = src[i];
    }
    for (uint32_t i = threadIdx.x; i < 2u + (countInLds ? a.top.n_leaves * (DEFER ? 2u : 1u) : 0u); i += kTopBlock)
        ldsCount[i] = 0u;
    __syncthreads();
    const unsigned char* nodeBase = NODES_IN_LDS ? ldsNodes : a.top.nodes;
    uint32_t nTests = 0, nCulled = 0;

    // (whole waves enter every trip: the parking step below uses wave-wide ballots)
    for (uint32_t w0 = blockIdx.x * kTopBlock; w0 < nActive; w0 += gridDim.x * kTopBlock) {
        const uint32_t w = w0 + threadIdx.x; // position in the round's list
        // (only_culled: this pass advances the rays whose cast missed in the last one and leaves the others alone)
        const bool live = w < nActive && (a.only_culled == 0u || a.parked_leaf[w] == kCastCulled);
        const uint32_t ray = live ? (a.active ? a.active[w] : w) : 0u;
        bool parked = false, deferred = false;
        uint32_t leafId = 0;
        // a shadow ray already occluded in its last leaf is done (offline_batching...h:246-247)
        if (live && !(ANY && a.any_out[ray] != 0u)) {
            float ox, oy, oz, dx, dy, dz, tnear;
            loadRay(a, ray, ox, oy, oz, dx, dy, dz, tnear);
            const float tfar = ANY ? a.tfar_in[ray] : a.tfar_state[ray]; // simdRay is rebuilt from the ray at every resume (:243-254)
            const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz; // :241
            uint32_t node = a.node[ray];
            uint64_t stack = a.stack[ray];
            uint32_t steps = 0;
            bool pending = false; // PHASED: the ray stands in front of the DAG of batching point `leafId`
            for (;;) { // (PHASED: walk phase, cast phase, walk phase, ...; otherwise once through)
            // (Ending a walk phase early, once only a half / third / quarter / eighth of the rays that began it still walk, so that
            // the rays that are ready do not wait for the stragglers: measured, within noise of waiting for all of them.)
            for (;;) {
                if (PHASED && pending)
                    break;
                if (++steps > a.step_cap)
                    break; // non-terminating traversal (kernels.h): the ray leaves the top level
                const float* nf = reinterpret_cast<const float*>(nodeBase + size_t(node) * kTopNodeBytes);
                const uint32_t* nu = reinterpret_cast<const uint32_t*>(nf);
                const uint32_t bits = nu[27];
                const uint32_t valid = bits & 0xFu, leafMask = (bits >> 4) & 0xFu;
                const int bitPos = 4 * int((bits >> 8) & 0xFFu);
                const uint64_t interest = (stack >> bitPos) & 0xFull;
                bool descended = false;
                if (interest != 0) {
                    float tmin[4];
                    const uint32_t hitMask = slabs4(nf, ox, oy, oz, ix, iy, iz, tnear, tfar, tmin);
                    const uint32_t toVisit = hitMask & uint32_t(interest) & valid;
                    if (toVisit != 0) {
                        uint32_t child;
                        if (ANY) {
                            child = (toVisit & 1u) ? 0u : ((toVisit & 2u) ? 1u : ((toVisit & 4u) ? 2u : 3u)); // bitScan4
                        } else {
                            float masked[4];
    #pragma unroll
                            for (int i = 0; i < 4; ++i)
                                masked[i] = (toVisit >> i & 1u) ? tmin[i] : 3.4028234663852886e38f;
                            child = hminIndex4(masked);
                        }
                        uint64_t rest = uint64_t(toVisit) ^ (1ull << child);
                        stack = stack ^ (interest << bitPos);
                        stack = stack | (rest << bitPos);
                        if ((valid & ~leafMask) >> child & 1u) {
                            node = nu[24] + __popc((valid & ~leafMask) & ((1u << child) - 1u));
                        } else {
                            const uint32_t leaf = nu[25] + __popc((valid & leafMask) & ((1u << child) - 1u));
                            const LeafDagDev lf = a.top.leafDags[leaf];
                            bool culled = false;
                            if (DEFER && lf.has_dag != 0u) {
                                // the cast kernel decides: parked here, or culled and on with the traversal from this token
                                deferred = true;
                                leafId = leaf;
                                break;
                            }
                            if (PHASED && lf.has_dag != 0u) {
                                pending = true; // (the cast runs when the whole wave has got this far; a miss resumes at this node)
                                leafId = leaf;
                                continue;
                            }
                            if (lf.has_dag != 0u) {
                                ++nTests;
                                const SvdagDev dag { lf.root, { lf.bmin[0], lf.bmin[1], lf.bmin[2] }, lf.invExtent, lf.levels, 0u, 0u };
                                if (!svdagHit(a.top.dagWords, dag, ox, oy, oz, dx, dy, dz, dagStack + threadIdx.x, kTopBlock)) {
                                    ++nCulled;
                                    culled = true;
                                }
                            }
                            if (!culled) {
                                parked = a.may_park != 0u;
                                leafId = leaf;
                                break;
                            }
                        }
                        descended = true;
                    }
                }
                if (descended)
                    continue;
                stack = stack | (0xFFFFFFFFFFFFFFFFull << bitPos);
                if (stack == 0xFFFFFFFFFFFFFFFFull)
                    break;
                node = nu[26];
            }
            if (!PHASED || !pending)
                break;
            {
                // the cast phase: every ray of the wave that stands in front of a DAG
                const LeafDagDev lf = a.top.leafDags[leafId];
                ++nTests;
                const SvdagDev dag { lf.root, { lf.bmin[0], lf.bmin[1], lf.bmin[2] }, lf.invExtent, lf.levels, 0u, 0u };
                pending = false;
                if (svdagHit(a.top.dagWords, dag, ox, oy, oz, dx, dy, dz, dagStack + threadIdx.x, kTopBlock)) {
                    parked = a.may_park != 0u;
                    break;
                }
                ++nCulled;
            }
            }
            if (parked || deferred) {
                a.node[ray] = node;
                a.stack[ray] = stack;
                if (a.visits && parked)
                    a.visits[ray] += 1u; // Ray::numTopLevelIntersections (:283)
            }
        }
        // park: the round's list keeps one entry per work item (the batching point, or none), one count per point
        if (live)
            a.parked_leaf[w] = parked ? leafId : (deferred ? (leafId | kCastRequestFlag) : PGPU_MISS_ID);
        if (DEFER && __ballot(deferred) != 0) {
            // one request count per batching point (they are binned by point for the cast kernel)
            const KeyGroup kg = groupByKey(leafId, deferred);
            if (deferred && kg.leader == (threadIdx.x & 63u)) {
                if (countInLds)
                    atomicAdd(&ldsReqCount[leafId], kg.size);
                else
                    atomicAdd(&a.req_leaf_count[leafId], kg.size);
            }
        }
        if (__ballot(parked) != 0) {
            const KeyGroup kg = groupByKey(leafId, parked);
            if (a.leaf_count && parked && kg.leader == (threadIdx.x & 63u)) {
                if (countInLds)
                    atomicAdd(&ldsCount[2u + leafId], kg.size);
                else
                    atomicAdd(&a.leaf_count[leafId], kg.size);
            }
        }
    }
    // one global update per block and counter instead of one per ray: same-address atomics serialise in L2
    if (nTests)
        atomicAdd(&ldsCount[0], nTests);
    if (nCulled)
        atomicAdd(&ldsCount[1], nCulled);
    __syncthreads();
    if (threadIdx.x < 2u && ldsCount[threadIdx.x] != 0u)
        atomicAdd(&a.counters[threadIdx.x], (unsigned long long)ldsCount[threadIdx.x]);
    if (countInLds && a.leaf_count)
        for (uint32_t i = threadIdx.x; i < a.top.n_leaves; i += kTopBlock)
            if (ldsCount[2u + i] != 0u)
                atomicAdd(&a.leaf_count[i], ldsCount[2u + i]);
    if (DEFER && countInLds)
        for (uint32_t i = threadIdx.x; i < a.top.n_leaves; i += kTopBlock)
            if (ldsReqCount[i] != 0u)
                atomicAdd(&a.req_leaf_count[i], ldsReqCount[i]);
}

// ---------------------------------------------------------------------------------------------------
// The same episode with lanes that never wait for their wave's slowest ray. Every lane is a small state machine -- idle,
// walking the BVH4, or inside an SVDAG cast -- and a wave repeats: give idle lanes the next rays of its chunk; then run ONE of
// the two phases, the one more of its lanes are waiting for: a BVH4 step for the walkers, or a few cast trips for the casters
// (the others sit the phase out; whichever side is left waiting grows until it is the majority, so nobody starves). In the
// lockstep kernel above a wave is as slow as its slowest ray and runs both code paths for every mix of lanes: 29 % of its
// VALU lane-cycles do work on incoherent rays (profiles/r2_pmc_summary_two_level.txt). The sequence of operations of each
// ray is the same in both kernels, so are the results.
// ---------------------------------------------------------------------------------------------------
template <bool ANY, bool NODES_IN_LDS>
__global__ __launch_bounds__(kTopBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void topTraverseLanesKernel(TopTraceArgs a)
{
    extern __shared__ __align__(16) unsigned char topLds[];
    uint32_t* const dagStack = reinterpret_cast<uint32_t*>(topLds);
    unsigned char* const ldsNodes = topLds + size_t(a.top.dag_levels) * kTopBlock * 4u;
    uint32_t* const ldsCount = reinterpret_cast<uint32_t*>(ldsNodes + (NODES_IN_LDS ? ((size_t(a.top.n_nodes) * kTopNodeBytes + 15u) & ~size_t(15)) : 16u));
    const bool countInLds = a.top.n_leaves <= kTopHistMax;
    const uint32_t nActive = a.n_active_dev ? min(*a.n_active_dev, a.n_active) : a.n_active;
    if (blockIdx.x * kTopBlock >= nActive)
        return;
    if (NODES_IN_LDS) {
        const uint32_t words = a.top.n_nodes * (kTopNodeBytes / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(a.top.nodes);
        uint32_t* dst = reinterpret_cast<uint32_t*>(ldsNodes);
        for (uint32_t i = threadIdx.x; i < words; i += kTopBlock)
            dst[i] = src[i];
    }
    for (uint32_t i = threadIdx.x; i < 2u + (countInLds ? a.top.n_leaves : 0u); i += kTopBlock)
        ldsCount[i] = 0u;
    __syncthreads();
    const unsigned char* nodeBase = NODES_IN_LDS ? ldsNodes : a.top.nodes;
    const GlobalDag dagWords { a.top.dagWords };
    uint32_t* const saved = dagStack + threadIdx.x;
    uint32_t nTests = 0, nCulled = 0;

    // work: chunks of 64 consecutive positions of the round's list, dealt round-robin over the waves of the grid
    const uint32_t wavesTotal = gridDim.x * (kTopBlock / 64u);
    uint32_t chunk = uint32_t(__builtin_amdgcn_readfirstlane(int(blockIdx.x * (kTopBlock / 64u) + (threadIdx.x >> 6))));
    uint32_t chunkNext = chunk * 64u, chunkEnd = min(chunkNext + 64u, nActive);

    int mode = 0; // 0 idle, 1 walking the BVH4, 2 inside an SVDAG cast, 3 back from a cast that missed: the ray is reloaded, then 1
    uint32_t w = 0, ray = 0, node = 0, steps = 0, castLeaf = 0;
    uint64_t stack = 0;
    // A lane is walking or casting, never both: the ray (origin, 1 / direction, interval) and the cast's per-axis state share
    // their registers -- the cast overwrites the ray and a lane whose cast missed reloads it (one 32-byte record + the far end)
    // before it walks on. Held side by side they cost 84 VGPRs = 5 waves per SIMD; shared, 64 = 8 waves, and this kernel lives
    // on occupancy.
    CastState cs {};
    float &ox = cs.tCoef[0], &oy = cs.tCoef[1], &oz = cs.tCoef[2], &ix = cs.tBias[0], &iy = cs.tBias[1], &iz = cs.tBias[2];
    float &tnear = cs.pos[0], &tfar = cs.pos[1];

    for (;;) {
        // ---------------------------------------------------------------- idle lanes take new rays
        const uint64_t idle = __ballot(mode == 0);
        const uint32_t nIdle = uint32_t(__popcll(idle));
        const bool more = chunkNext < nActive;
        if (nIdle == 64u && !more)
            break;
        if (more && (nIdle == 64u || nIdle >= a.lanes_refill_min)) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(idle >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idle), 0u));
            const uint32_t avail = chunkEnd - chunkNext;
            const bool take = mode == 0 && rank < avail;
            const uint32_t mine = chunkNext + rank;
            chunkNext += min(nIdle, avail);
            if (chunkNext >= chunkEnd) {
                chunk += wavesTotal;
                chunkNext = chunk * 64u;
                chunkEnd = min(chunkNext + 64u, nActive);
            }
            if (take) {
                w = mine;
                ray = a.active ? a.active[w] : w;
                if (ANY && a.any_out[ray] != 0u) {
                    a.parked_leaf[w] = PGPU_MISS_ID; // a shadow ray already occluded in its last leaf is done (offline_batching...h:246-247)
                } else {
                    node = a.node[ray];
                    stack = a.stack[ray];
                    steps = 0;
                    mode = 3;
                }
            }
        }
        if (__ballot(mode == 3) != 0) {
            if (mode == 3) {
                float dx, dy, dz;
                loadRay(a, ray, ox, oy, oz, dx, dy, dz, tnear);
                tfar = ANY ? a.tfar_in[ray] : a.tfar_state[ray]; // simdRay is rebuilt from the ray at every resume (:243-254)
                ix = 1.0f / dx; // :241
                iy = 1.0f / dy;
                iz = 1.0f / dz;
                mode = 1;
            }
        }

        const uint32_t nWalk = uint32_t(__popcll(__ballot(mode == 1))), nCast = uint32_t(__popcll(__ballot(mode == 2)));
        bool fin = false, parked = false;
        uint32_t leafId = 0;
        if (nWalk != 0u && nWalk >= nCast) {
            // ------------------------------------------------------------ one BVH4 step (pauseable_bvh4.h:262-352)
            if (mode == 1) {
                if (++steps > a.step_cap) {
                    fin = true; // non-terminating traversal (kernels.h): the ray leaves the top level
                } else {
                    const float* nf = reinterpret_cast<const float*>(nodeBase + size_t(node) * kTopNodeBytes);
                    const uint32_t* nu = reinterpret_cast<const uint32_t*>(nf);
                    const uint32_t bits = nu[27];
                    const uint32_t valid = bits & 0xFu, leafMask = (bits >> 4) & 0xFu;
                    const int bitPos = 4 * int((bits >> 8) & 0xFFu);
                    const uint64_t interest = (stack >> bitPos) & 0xFull;
                    bool descended = false;
                    if (interest != 0) {
                        float tmin[4];
                        const uint32_t hitMask = slabs4(nf, ox, oy, oz, ix, iy, iz, tnear, tfar, tmin);
                        const uint32_t toVisit = hitMask & uint32_t(interest) & valid;
                        if (toVisit != 0) {
                            uint32_t child;
                            if (ANY) {
                                child = (toVisit & 1u) ? 0u : ((toVisit & 2u) ? 1u : ((toVisit & 4u) ? 2u : 3u)); // bitScan4
                            } else {
                                float masked[4];
    #pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    masked[i] = (toVisit >> i & 1u) ? tmin[i] : 3.4028234663852886e38f;
                                child = hminIndex4(masked);
                            }
                            const uint64_t rest = uint64_t(toVisit) ^ (1ull << child);
                            stack = stack ^ (interest << bitPos);
                            stack = stack | (rest << bitPos);
                            descended = true;
                            if ((valid & ~leafMask) >> child & 1u) {
                                node = nu[24] + __popc((valid & ~leafMask) & ((1u << child) - 1u));
                            } else {
                                // BatchingPoint::intersect (offline_batching...h:269-305): the proxy geometry decides
                                const uint32_t leaf = nu[25] + __popc((valid & leafMask) & ((1u << child) - 1u));
                                const LeafDagDev lf = a.top.leafDags[leaf];
                                if (lf.has_dag != 0u) {
                                    ++nTests;
                                    const SvdagDev dag { lf.root, { lf.bmin[0], lf.bmin[1], lf.bmin[2] }, lf.invExtent, lf.levels, 0u, 0u };
                                    float dx, dy, dz;
                                    loadDirection(a, ray, dx, dy, dz);
                                    const int r = castInit(cs, dagWords, dag.root, dag, ox, oy, oz, dx, dy, dz);
                                    if (r == 0) {
                                        castLeaf = leaf;
                                        mode = 2;
                                    } else {
                                        ++nCulled; // the ray misses the DAG's cube: the traversal goes on at the same node
                                        mode = 3;  // (castInit has overwritten the ray)
                                    }
                                } else {
                                    fin = true;
                                    parked = a.may_park != 0u;
                                    leafId = leaf;
                                }
                            }
                        }
                    }
                    if (!descended) {
                        stack = stack | (0xFFFFFFFFFFFFFFFFull << bitPos);
                        if (stack == 0xFFFFFFFFFFFFFFFFull)
                            fin = true;
                        else
                            node = nu[26];
                    }
                }
            }
        } else if (nCast != 0u) {
            // ------------------------------------------------------------ a few trips of the casts (a trip is about a third of
            // a BVH4 step and a cast takes tens of them: several per turn keep the loop's bookkeeping off them)
            for (uint32_t k = 0; k < a.lanes_cast_trips; ++k) {
                if (mode == 2 && !fin) {
                    const int r = castTrip(cs, dagWords, saved, kTopBlock);
                    if (r == 1) {
                        fin = true;
                        parked = a.may_park != 0u;
                        leafId = castLeaf;
                    } else if (r == 2) {
                        ++nCulled;
                        mode = 3; // culled: the traversal goes on at the same node, once the ray is back in its registers
                    }
                }
                if (__ballot(mode == 2 && !fin) == 0)
                    break;
            }
        }

        // ---------------------------------------------------------------- rays whose episode is over
        if (fin) {
            a.parked_leaf[w] = parked ? leafId : PGPU_MISS_ID;
            if (parked) {
                a.node[ray] = node;
                a.stack[ray] = stack;
                if (a.visits)
                    a.visits[ray] += 1u; // Ray::numTopLevelIntersections (:283)
                if (a.leaf_count) {
                    if (countInLds)
                        atomicAdd(&ldsCount[2u + leafId], 1u);
                    else
                        atomicAdd(&a.leaf_count[leafId], 1u);
                }
            }
            mode = 0;
        }
    }
    // one global update per block and counter instead of one per ray: same-address atomics serialise in L2
    if (nTests)
        atomicAdd(&ldsCount[0], nTests);
    if (nCulled)
        atomicAdd(&ldsCount[1], nCulled);
    __syncthreads();
    if (threadIdx.x < 2u && ldsCount[threadIdx.x] != 0u)
        atomicAdd(&a.counters[threadIdx.x], (unsigned long long)ldsCount[threadIdx.x]);
    if (countInLds && a.leaf_count)
        for (uint32_t i = threadIdx.x; i < a.top.n_leaves; i += kTopBlock)
            if (ldsCount[2u + i] != 0u)
                atomicAdd(&a.leaf_count[i], ldsCount[2u + i]);
}

// ---------------------------------------------------------------------------------------------------
// The casts of one pass (see CastArgs in kernels.h): requests sorted by batching point, the point's DAG staged in LDS.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kCastTripsPerPass = 8; // trips between two looks at who is done
constexpr uint32_t kCastRefillMin = 12;   // idle lanes that make a refill worth its cost
constexpr uint32_t kCastChunk = 64;       // requests a wave draws at a time from the unit's cursor
constexpr uint32_t kCastUnit = 2048;      // requests of one batching point a block takes at a time

__global__ __launch_bounds__(kTopBlock) __attribute__((amdgpu_waves_per_eu(4, 8))) void svdagCastKernel(CastArgs a)
{
    // dynamic LDS: [the DAG image][dag_levels x 256 saved parents][cursor, hits of the run, tests, culled]
    extern __shared__ __align__(16) unsigned char castLds[];
    uint32_t* const image = reinterpret_cast<uint32_t*>(castLds);
    uint32_t* const dagStack = image + a.top.img_max_words;
    uint32_t* const ctl = dagStack + size_t(a.top.dag_levels) * kTopBlock; // [0] cursor [1] hits of this unit [2] tests [3] culled [4] unit
    if (threadIdx.x < 5u)
        ctl[threadIdx.x] = 0u;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t* const saved = dagStack + threadIdx.x;
    // Work units: the requests of one batching point in pieces of kCastUnit (svdagPlanKernel numbered them: unit_start[l] =
    // first unit of point l). Blocks draw units from a cursor until none is left, so a pass whose requests are spread thinly
    // over many points still keeps every CU busy -- each unit has a fixed cost (staging the image, filling and draining the
    // lanes) that is paid in parallel, not one after the other by the block that happens to own that stretch of the list.
    const uint32_t nUnits = a.plan[0];
    uint32_t nTests = 0, nCulled = 0;
    {
        for (;;) {
            __syncthreads(); // the previous unit is done with the image and the control words
            if (threadIdx.x == 0)
                ctl[4] = atomicAdd(&a.plan[1], 1u);
            __syncthreads();
            const uint32_t unit = ctl[4];
            if (unit >= nUnits)
                break;
            // the batching point that owns this unit: the last one whose first unit is <= unit (points without requests share
            // their successor's number and sort before it)
            uint32_t first = 0, last = a.top.n_leaves;
            while (last - first > 1u) {
                const uint32_t mid = (first + last) >> 1;
                if (a.unit_start[mid] <= unit)
                    first = mid;
                else
                    last = mid;
            }
            const uint32_t l = first;
            const uint32_t start = a.req_start[l], count = a.req_count[l];
            const uint32_t runLo = start + (unit - a.unit_start[l]) * kCastUnit, runHi = min(start + count, runLo + kCastUnit);
            const SvdagDev dag = a.top.dags[a.top.leaves[l].svdag];
            for (uint32_t i = threadIdx.x; i < dag.img_words; i += kTopBlock)
                image[i] = a.top.dagImages[dag.img_offset + i];
            if (threadIdx.x == 0) {
                ctl[0] = runLo;
                ctl[1] = 0u;
            }
            __syncthreads();
            const StagedDag words { image, a.top.dagWords };
            const uint32_t root = dag.img_words ? 0u : (dag.root | kDagGlobalFlag);
            CastState s {};
            bool busy = false;
            int result = 0; // of a busy lane: 0 running, 1 hit, 2 miss
            uint32_t w = 0, ray = 0, hits = 0;
            uint32_t chunkNext = 0, chunkEnd = 0; // wave-uniform
            bool exhausted = false;
            for (;;) {
                const uint64_t idleMask = __ballot(!busy);
                const uint32_t nIdle = uint32_t(__popcll(idleMask));
                if (!exhausted && (nIdle >= kCastRefillMin || nIdle == 64u)) {
                    if (chunkNext >= chunkEnd) {
                        uint32_t c = 0;
                        if (lane == 0)
                            c = atomicAdd(&ctl[0], kCastChunk);
                        c = uint32_t(__builtin_amdgcn_readfirstlane(int(c)));
                        if (c < runHi) {
                            chunkNext = c;
                            chunkEnd = min(c + kCastChunk, runHi);
                        } else {
                            exhausted = true;
                        }
                    }
                    if (!exhausted) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(idleMask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(idleMask), 0u));
                        const uint32_t at = chunkNext + rank;
                        const bool take = !busy && at < chunkEnd;
                        chunkNext = min(chunkNext + nIdle, chunkEnd);
                        if (take) {
                            w = a.req_list[at];
                            ray = a.active ? a.active[w] : w;
                            ++nTests;
                            busy = true;
                            result = castInit(s, words, root, dag, a.ox[ray], a.oy[ray], a.oz[ray], a.dx[ray], a.dy[ray], a.dz[ray]);
                        }
                    }
                }
                if (__ballot(busy) == 0) {
                    if (exhausted)
                        break;
                    continue;
                }
                for (uint32_t k = 0; k < kCastTripsPerPass; ++k)
                    if (busy && result == 0)
                        result = castTrip(s, words, saved, kTopBlock);
                if (busy && result == 1) {
                    a.parked_leaf[w] = l; // parked at this batching point
                    if (a.visits)
                        a.visits[ray] += 1u; // Ray::numTopLevelIntersections (a ray has at most one request per pass)
                    ++hits;
                    busy = false;
                } else if (busy && result == 2) {
                    a.parked_leaf[w] = kCastCulled;
                    ++nCulled;
                    busy = false;
                }
            }
            // hits of the run: wave sum, one LDS atomic per wave, one global atomic per block and run
            for (int o = 32; o > 0; o >>= 1)
                hits += uint32_t(__shfl_xor(int(hits), o));
            if (lane == 0 && hits)
                atomicAdd(&ctl[1], hits);
            __syncthreads();
            if (threadIdx.x == 0 && ctl[1] != 0u)
                atomicAdd(&a.leaf_count[l], ctl[1]);
        }
    }
    if (nTests)
        atomicAdd(&ctl[2], nTests);
    if (nCulled)
        atomicAdd(&ctl[3], nCulled);
    __syncthreads();
    if (threadIdx.x < 2u && ctl[2u + threadIdx.x] != 0u)
        atomicAdd(&a.counters[threadIdx.x], (unsigned long long)ctl[2u + threadIdx.x]);
}

// exclusive scan of leaf_count into leaf_start (single block; batching points number in the hundreds)
// (resetCount / cursor: rounds queued ahead of the host -- the scan leaves the counts zeroed for the next episode and clears the
// bottom-level kernel's work cursors, which saves two fill launches per round; the tail of a trace is a string of tiny rounds)
__global__ __launch_bounds__(1024) void leafScanKernel(const uint32_t* count, uint32_t* __restrict__ start,
    uint32_t* __restrict__ fill, uint32_t n, uint32_t* __restrict__ total, uint32_t* resetCount = nullptr /* may be `count` */, uint32_t* __restrict__ cursor = nullptr,
    uint32_t* __restrict__ totalHost = nullptr /* pinned host memory: the same total, without a copy of its own */)
{
    if (cursor && threadIdx.x < 64u)
        cursor[threadIdx.x] = 0u;
    __shared__ uint32_t part[1024];
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i)
        sum += count[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; ++i) {
        start[i] = run;
        fill[i] = 0;
        const uint32_t c = count[i];
        run += c;
        if (resetCount)
            resetCount[i] = 0u;
    }
    if (threadIdx.x == 1023) {
        *total = part[1023]; // rays parked in this round
        if (totalHost)
            *totalHost = part[1023];
    }
}

// scatter parked records into per-leaf contiguous ranges (the GPU form of the per-leaf queues)
// what a position of the round's list is binned under: its batching point, or (requests) the point under kCastRequestFlag
__device__ __forceinline__ uint32_t binKey(uint32_t v, bool requests)
{
    if (!requests)
        return v;
    return (v & kCastRequestFlag) ? (v & ~kCastRequestFlag) : PGPU_MISS_ID; // (PGPU_MISS_ID and kCastCulled stay out of range)
}

__global__ __launch_bounds__(256) void binByLeafKernel(const uint32_t* __restrict__ active, const uint32_t* __restrict__ parkedLeaf,
    uint32_t n, uint32_t nLeaves, const uint32_t* __restrict__ leafStart, uint32_t* __restrict__ leafFill,
    const TopLeafDev* __restrict__ leaves, uint32_t* __restrict__ workRay, uint32_t* __restrict__ workBvh, uint32_t* __restrict__ workLeaf, bool requests,
    const uint32_t* __restrict__ nDev)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (nDev)
        n = min(n, *nDev);
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t leaf = i < n ? binKey(parkedLeaf[i], requests) : PGPU_MISS_ID;
        const bool live = leaf < nLeaves; // (PGPU_MISS_ID = not parked; anything else out of range is ignored too)
        const KeyGroup kg = groupByKey(leaf, live);
        uint32_t base = 0;
        if (live && kg.leader == lane)
            base = atomicAdd(&leafFill[leaf], kg.size);
        base = uint32_t(__builtin_amdgcn_ds_bpermute(int(kg.leader << 2), int(base)));
        if (live) {
            const uint32_t at = leafStart[leaf] + base + kg.rank;
            workRay[at] = active ? active[i] : i;
            workBvh[at] = leaves[leaf].bvh;
            workLeaf[at] = leaf;
        }
    }
}

// Same, for scenes whose batching points fit LDS: a block owns a contiguous slice of the round's list, counts it per
// point in LDS, reserves one range per point it saw with a single global atomic, and hands out the slots from LDS.
__global__ __launch_bounds__(256) void binByLeafLdsKernel(const uint32_t* __restrict__ active, const uint32_t* __restrict__ parkedLeaf,
    uint32_t n, uint32_t nLeaves, const uint32_t* __restrict__ leafStart, uint32_t* __restrict__ leafFill,
    const TopLeafDev* __restrict__ leaves, uint32_t* __restrict__ workRay, uint32_t* __restrict__ workBvh, uint32_t* __restrict__ workLeaf, bool requests,
    const uint32_t* __restrict__ nDev)
{
    extern __shared__ uint32_t binLds[]; // [nLeaves] running count, [nLeaves] first slot of this block's range
    uint32_t* const cnt = binLds;
    uint32_t* const first = binLds + nLeaves;
    const uint32_t lane = threadIdx.x & 63u;
    if (nDev)
        n = min(n, *nDev);
    const uint32_t per = (((n + gridDim.x - 1u) / gridDim.x) + 255u) & ~255u;
    const uint32_t lo = min(blockIdx.x * per, n), hi = min(lo + per, n);
    for (uint32_t i = threadIdx.x; i < nLeaves; i += 256u)
        cnt[i] = 0u;
    __syncthreads();
    for (uint32_t i0 = lo; i0 < hi; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t leaf = i < hi ? binKey(parkedLeaf[i], requests) : PGPU_MISS_ID;
        const bool live = leaf < nLeaves;
        if (__ballot(live) != 0) {
            const KeyGroup kg = groupByKey(leaf, live);
            if (live && kg.leader == lane)
                atomicAdd(&cnt[leaf], kg.size);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nLeaves; i += 256u) {
        const uint32_t c = cnt[i];
        if (c != 0u) {
            first[i] = leafStart[i] + atomicAdd(&leafFill[i], c);
            cnt[i] = 0u;
        }
    }
    __syncthreads();
    for (uint32_t i0 = lo; i0 < hi; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t leaf = i < hi ? binKey(parkedLeaf[i], requests) : PGPU_MISS_ID;
        const bool live = leaf < nLeaves;
        if (__ballot(live) == 0)
            continue;
        const KeyGroup kg = groupByKey(leaf, live);
        uint32_t off = 0;
        if (live && kg.leader == lane)
            off = atomicAdd(&cnt[leaf], kg.size);
        off = uint32_t(__builtin_amdgcn_ds_bpermute(int(kg.leader << 2), int(off)));
        if (live) {
            const uint32_t at = first[leaf] + off + kg.rank;
            workRay[at] = active ? active[i] : i;
            workBvh[at] = leaves[leaf].bvh;
            workLeaf[at] = leaf;
        }
    }
}

// The next round's list for the top-level kernel: the rays that parked, in the order of the current list (an order-preserving
// compaction). Round 0 lists the rays by id, so every later list is ascending in ray id too, and the top-level kernel -- which
// reads twelve per-ray values through the list -- touches memory almost as coalesced as in round 0. (The bottom level needs the
// other order, by batching point: binByLeaf writes that list. Taking it for the next top-level episode as well made that
// kernel gather every value of every ray from a different cache line.) Two passes over fixed slices: count, then -- after a
// scan of the slice counts -- write.
__device__ __forceinline__ void compactSlice(uint32_t n, uint32_t& lo, uint32_t& hi)
{
    const uint32_t per = (((n + gridDim.x - 1u) / gridDim.x) + 255u) & ~255u;
    lo = min(blockIdx.x * per, n);
    hi = min(lo + per, n);
}

__global__ __launch_bounds__(256) void compactCountKernel(const uint32_t* __restrict__ parkedLeaf, uint32_t n, const uint32_t* __restrict__ nDev,
    uint32_t nLeaves, uint32_t* __restrict__ sliceCount)
{
    __shared__ uint32_t waveSum[4];
    if (nDev)
        n = min(n, *nDev);
    uint32_t lo, hi;
    compactSlice(n, lo, hi);
    uint32_t mine = 0;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256u)
        mine += parkedLeaf[i] < nLeaves ? 1u : 0u;
    for (int o = 32; o > 0; o >>= 1)
        mine += uint32_t(__shfl_xor(int(mine), o));
    if ((threadIdx.x & 63u) == 0u)
        waveSum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0)
        sliceCount[blockIdx.x] = waveSum[0] + waveSum[1] + waveSum[2] + waveSum[3];
}

__global__ __launch_bounds__(256) void compactWriteKernel(const uint32_t* __restrict__ active, const uint32_t* __restrict__ parkedLeaf, uint32_t n,
    const uint32_t* __restrict__ nDev, uint32_t nLeaves, const uint32_t* __restrict__ sliceStart, uint32_t* __restrict__ nextActive)
{
    __shared__ uint32_t waveSum[4];
    if (nDev)
        n = min(n, *nDev);
    uint32_t lo, hi;
    compactSlice(n, lo, hi);
    uint32_t base = sliceStart[blockIdx.x];
    const uint32_t wave = threadIdx.x >> 6;
    for (uint32_t i0 = lo; i0 < hi; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        const bool live = i < hi && parkedLeaf[i] < nLeaves;
        const uint64_t mask = __ballot(live);
        if ((threadIdx.x & 63u) == 0u)
            waveSum[wave] = uint32_t(__popcll(mask));
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t k = 0; k < 4u; ++k) {
            before += k < wave ? waveSum[k] : 0u;
            total += waveSum[k];
        }
        if (live)
            nextActive[base + before + __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u))] = active ? active[i] : i;
        base += total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void leafToBvhKernel(const uint32_t* __restrict__ workLeaf, const TopLeafDev* __restrict__ leaves, uint32_t nLeaves,
    uint32_t* __restrict__ workBvh, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t l = workLeaf[i];
        workBvh[i] = l < nLeaves ? leaves[l].bvh : PGPU_MISS_ID;
    }
}

__global__ __launch_bounds__(256) void initStateKernel(uint32_t n, uint32_t root, uint32_t* node, uint64_t* stack, const float* tfarIn,
    float* tState, uint32_t* obj, uint32_t* leaf, float* u, float* v, uint32_t* visits, uint32_t* anyOut)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        node[i] = root;
        stack[i] = 0xFFFFFFFFFFFFFFFFull; // pauseable_bvh4.h:202
        if (tState) {
            tState[i] = tfarIn[i];
            obj[i] = PGPU_MISS_ID;
            leaf[i] = 0u;
            u[i] = 0.0f;
            v[i] = 0.0f;
        }
        if (visits)
            visits[i] = 0u;
        if (anyOut)
            anyOut[i] = 0u;
    }
}

} // namespace

bool topNodesFitLds(uint32_t nNodes) { return size_t(nNodes) * kTopNodeBytes <= kTopLdsNodeBudget; }

size_t topTraverseLdsBytes(const TopDev& top)
{
    // (the second per-point counter array is used by the deferred-cast variant only; it is small)
    return size_t(top.dag_levels) * kTopBlock * 4u + (topNodesFitLds(top.n_nodes) ? ((size_t(top.n_nodes) * kTopNodeBytes + 15u) & ~size_t(15)) : 16u)
        + 8u + (top.n_leaves <= kTopHistMax ? size_t(top.n_leaves) * 8u : 0u);
}

hipError_t launchTopTraverse(const TopTraceArgs& a, int kind, uint32_t blocks, hipStream_t stream, bool defer)
{
    const bool inLds = topNodesFitLds(a.top.n_nodes);
    const dim3 grid(blocks), block(kTopBlock);
    const size_t ldsBytes = topTraverseLdsBytes(a.top);
#define PANDORA_TOP(ANYV, LDSV)                                                                                  \
    do {                                                                                                         \
        if (!defer && a.lanes_refill_min != 0u)                                                                  \
            hipLaunchKernelGGL((topTraverseLanesKernel<ANYV, LDSV>), grid, block, ldsBytes, stream, a);         \
        else if (defer)                                                                                          \
            hipLaunchKernelGGL((topTraverseKernel<ANYV, LDSV, true>), grid, block, ldsBytes, stream, a);        \
        else if (a.phased != 0u)                                                                                 \
            hipLaunchKernelGGL((topTraverseKernel<ANYV, LDSV, false, true>), grid, block, ldsBytes, stream, a); \
        else                                                                                                     \
            hipLaunchKernelGGL((topTraverseKernel<ANYV, LDSV, false>), grid, block, ldsBytes, stream, a);       \
    } while (0)
    if (kind == PGPU_ANY_ID) {
        if (inLds)
            PANDORA_TOP(true, true);
        else
            PANDORA_TOP(true, false);
    } else {
        if (inLds)
            PANDORA_TOP(false, true);
        else
            PANDORA_TOP(false, false);
    }
#undef PANDORA_TOP
    return hipGetLastError();
}

size_t svdagCastLdsBytes(const TopDev& top) { return (size_t(top.img_max_words) + size_t(top.dag_levels) * kTopBlock + 8u) * 4u; }

// exclusive scan of the request counts (first request per point, as leafScanKernel) and of the work units per point
__global__ __launch_bounds__(1024) void svdagPlanKernel(const uint32_t* __restrict__ count, uint32_t* __restrict__ start, uint32_t* __restrict__ fill,
    uint32_t* __restrict__ unitStart, uint32_t n, uint32_t* __restrict__ plan)
{
    __shared__ uint32_t part[1024], upart[1024];
    const uint32_t per = (n + 1023u) / 1024u;
    const uint32_t lo = min(threadIdx.x * per, n), hi = min(lo + per, n);
    uint32_t sum = 0, usum = 0;
    for (uint32_t i = lo; i < hi; ++i) {
        sum += count[i];
        usum += (count[i] + kCastUnit - 1u) / kCastUnit;
    }
    part[threadIdx.x] = sum;
    upart[threadIdx.x] = usum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u, u = threadIdx.x >= off ? upart[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        upart[threadIdx.x] += u;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum, urun = upart[threadIdx.x] - usum;
    for (uint32_t i = lo; i < hi; ++i) {
        start[i] = run;
        fill[i] = 0;
        unitStart[i] = urun;
        run += count[i];
        urun += (count[i] + kCastUnit - 1u) / kCastUnit;
    }
    if (threadIdx.x == 1023) {
        plan[0] = upart[1023]; // work units of the pass
        plan[1] = 0u;          // the cast kernel's unit cursor
    }
}

hipError_t launchSvdagPlan(const uint32_t* count, uint32_t* start, uint32_t* fill, uint32_t* unitStart, uint32_t n, uint32_t* plan, hipStream_t stream)
{
    hipLaunchKernelGGL(svdagPlanKernel, dim3(1), dim3(1024), 0, stream, count, start, fill, unitStart, n, plan);
    return hipGetLastError();
}

hipError_t launchSvdagCast(const CastArgs& a, uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(svdagCastKernel, dim3(blocks), dim3(kTopBlock), svdagCastLdsBytes(a.top), stream, a);
    return hipGetLastError();
}

hipError_t launchLeafScan(const uint32_t* count, uint32_t* start, uint32_t* fill, uint32_t n, uint32_t* total, hipStream_t stream, uint32_t* resetCount,
    uint32_t* cursor64, uint32_t* totalHost)
{
    hipLaunchKernelGGL(leafScanKernel, dim3(1), dim3(1024), 0, stream, count, start, fill, n, total, resetCount, cursor64, totalHost);
    return hipGetLastError();
}

hipError_t launchBinByLeaf(const uint32_t* active, const uint32_t* parkedLeaf, uint32_t nActive, uint32_t nLeaves, const uint32_t* leafStart,
    uint32_t* leafFill, const TopLeafDev* leaves, uint32_t* workRay, uint32_t* workBvh, uint32_t* workLeaf, uint32_t blocks, hipStream_t stream,
    bool requests, const uint32_t* nActiveDev)
{
    if (nLeaves <= kTopHistMax) {
        hipLaunchKernelGGL(binByLeafLdsKernel, dim3(blocks), dim3(256), size_t(nLeaves) * 8u, stream, active, parkedLeaf, nActive, nLeaves, leafStart,
            leafFill, leaves, workRay, workBvh, workLeaf, requests, nActiveDev);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(binByLeafKernel, dim3(blocks), dim3(256), 0, stream, active, parkedLeaf, nActive, nLeaves, leafStart, leafFill, leaves,
        workRay, workBvh, workLeaf, requests, nActiveDev);
    return hipGetLastError();
}

hipError_t launchCompactParked(const uint32_t* active, const uint32_t* parkedLeaf, uint32_t n, const uint32_t* nDev, uint32_t nLeaves, uint32_t* sliceCount,
    uint32_t* sliceStart, uint32_t* scratchFill, uint32_t* scratchTotal, uint32_t* nextActive, uint32_t slices, hipStream_t stream)
{
    hipLaunchKernelGGL(compactCountKernel, dim3(slices), dim3(256), 0, stream, parkedLeaf, n, nDev, nLeaves, sliceCount);
    hipLaunchKernelGGL(leafScanKernel, dim3(1), dim3(1024), 0, stream, sliceCount, sliceStart, scratchFill, slices, scratchTotal, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    hipLaunchKernelGGL(compactWriteKernel, dim3(slices), dim3(256), 0, stream, active, parkedLeaf, n, nDev, nLeaves, sliceStart, nextActive);
    return hipGetLastError();
}

hipError_t launchLeafToBvh(const uint32_t* workLeaf, const TopLeafDev* leaves, uint32_t nLeaves, uint32_t* workBvh, uint32_t n, uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(leafToBvhKernel, dim3(blocks), dim3(256), 0, stream, workLeaf, leaves, nLeaves, workBvh, n);
    return hipGetLastError();
}

hipError_t launchInitState(uint32_t n, uint32_t root, uint32_t* node, uint64_t* stack, const float* tfarIn, float* tState, uint32_t* obj,
    uint32_t* leaf, float* u, float* v, uint32_t* visits, uint32_t* anyOut, uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(initStateKernel, dim3(blocks), dim3(256), 0, stream, n, root, node, stack, tfarIn, tState, obj, leaf, u, v, visits, anyOut);
    return hipGetLastError();
}

} // namespace pandora_hip
