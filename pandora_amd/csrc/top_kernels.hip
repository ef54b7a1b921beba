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

// sparse_voxel_dag.cpp:350-567 with three lanes. `saved` is this lane's column of the parent stack.
__device__ bool svdagHit(const uint32_t* __restrict__ words, const SvdagDev& dag, float rox, float roy, float roz,
    float rdx, float rdy, float rdz, uint32_t* saved, uint32_t savedStride)
{
    const float eps = 1.1920928955078125e-07f;
    float o[3] = { 1.0f + (dag.invExtent * (rox - dag.bmin[0])), 1.0f + (dag.invExtent * (roy - dag.bmin[1])), 1.0f + (dag.invExtent * (roz - dag.bmin[2])) };
    float d[3] = { rdx, rdy, rdz };
    float tCoef[3], tBias[3], pos[3];
    uint32_t octMask = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (fabsf(d[i]) < eps)
            d[i] = copysignf(eps, d[i]);
        tCoef[i] = 1.0f / -fabsf(d[i]);
        tBias[i] = tCoef[i] * o[i];
        if (d[i] > 0.0f) {
            octMask |= 1u << i;
            tBias[i] = 3.0f * tCoef[i] - tBias[i];
        }
    }
    const uint32_t octBits = (7u ^ octMask) & 7u;
    const float tMin0 = stdmax(0.0f, stdmax(2.0f * tCoef[0] - tBias[0], stdmax(2.0f * tCoef[1] - tBias[1], 2.0f * tCoef[2] - tBias[2])));
    const float tMax0 = stdmin(tCoef[0] - tBias[0], stdmin(tCoef[1] - tBias[1], tCoef[2] - tBias[2]));
    if (tMin0 >= tMax0)
        return false;
    float tMin = tMin0;
    uint32_t idx = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bool m = (1.5f * tCoef[i] - tBias[i]) > tMin0;
        idx |= uint32_t(m) << i;
        pos[i] = m ? 1.5f : 1.0f;
    }
    uint32_t parent = dag.root;
    int scale = kCastDepth - 1;
    // with NaN operands an ADVANCE makes no progress: after this many trips give up and do not cull (deviation 4)
    const uint32_t tripCap = (16u * dag.levels + 16u) << dag.levels;
    uint32_t trips = 0;
    while (scale < kCastDepth) {
        if (++trips > tripCap)
            return true;
        float tCorner[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            tCorner[i] = pos[i] * tCoef[i] - tBias[i];
        const uint32_t child = 7u - ((idx & 7u) ^ octBits);
        const uint32_t desc = words[parent];
        const uint32_t valid = (desc >> 8) & 0xFFu, leaf = desc & 0xFFu;
        if (valid >> child & 1u) {
            // PUSH
            const float half = exp2i((scale - 1) - kCastDepth);
            saved[uint32_t(kCastDepth - 1 - scale) * savedStride] = parent;
            if ((valid & leaf) >> child & 1u)
                return true;
            const uint32_t inner = valid & ~leaf;
            parent = words[parent + 1u + __popc(inner & ((1u << child) - 1u))];
            --scale;
            idx = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float tCenter = half * tCoef[i] + tCorner[i];
                const bool m = tCenter > tMin;
                idx |= uint32_t(m) << i;
                pos[i] = m ? pos[i] + half : pos[i];
            }
        } else {
            // ADVANCE
            const float size = exp2i(scale - kCastDepth);
            const float tcMax = stdmin(tCorner[0], stdmin(tCorner[1], tCorner[2]));
            uint32_t step = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const bool s = tCorner[i] <= tcMax;
                step |= uint32_t(s) << i;
                pos[i] = s ? pos[i] - size : pos[i];
            }
            tMin = tcMax;
            idx ^= step;
            if ((idx & step) != 0) {
                // POP
                uint32_t diff = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (step >> i & 1u)
                        diff |= __float_as_uint(pos[i]) ^ __float_as_uint(pos[i] + size);
                scale = 31 - __clz(int(diff));
                if (scale < kCastDepth)
                    parent = saved[uint32_t(kCastDepth - 1 - scale) * savedStride];
            }
            uint32_t sh[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sh[i] = __float_as_uint(pos[i]) >> scale;
                pos[i] = __uint_as_float(sh[i] << scale);
            }
            idx = (sh[0] & 1u) | ((sh[1] & 1u) << 1) | ((sh[2] & 1u) << 2);
        }
    }
    return false;
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

// Many rays of a wave usually want the same counter (coherent rays park at the same batching point), and
// same-address atomics from single lanes serialise in L2. Lanes with equal keys therefore elect a leader that adds
// the group's size once; every lane learns its leader, its rank inside the group and the group size.
struct KeyGroup {
    uint32_t leader, rank, size;
};
__device__ __forceinline__ KeyGroup groupByKey(uint32_t key, bool active)
{
    KeyGroup kg;
    kg.leader = threadIdx.x & 63u;
    kg.rank = 0;
    kg.size = 1;
    uint64_t todo = __ballot(active);
    while (todo != 0) {
        const int l = __builtin_ctzll(todo);
        const uint32_t k = uint32_t(__builtin_amdgcn_readlane(int(key), l));
        const uint64_t same = __ballot(active && key == k);
        if (active && key == k) {
            kg.leader = uint32_t(l);
            kg.rank = __builtin_amdgcn_mbcnt_hi(uint32_t(same >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(same), 0u));
            kg.size = uint32_t(__popcll(same));
        }
        todo &= ~same;
    }
    return kg;
}

template <bool ANY, bool NODES_IN_LDS>
__global__ __launch_bounds__(kTopBlock) __attribute__((amdgpu_waves_per_eu(8, 8))) void topTraverseKernel(TopTraceArgs a)
{
    // dynamic LDS: [dag_levels x 256 saved parents of the SVDAG cast][the top-level nodes, when they fit]
    // [2 SVDAG counters + one counter per batching point, when they fit]
    extern __shared__ __align__(16) unsigned char topLds[];
    uint32_t* const dagStack = reinterpret_cast<uint32_t*>(topLds);
    unsigned char* const ldsNodes = topLds + size_t(a.top.dag_levels) * kTopBlock * 4u;
    uint32_t* const ldsCount = reinterpret_cast<uint32_t*>(ldsNodes + (NODES_IN_LDS ? ((size_t(a.top.n_nodes) * kTopNodeBytes + 15u) & ~size_t(15)) : 16u));
    const bool countInLds = a.top.n_leaves <= kTopHistMax;
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
    uint32_t nTests = 0, nCulled = 0;

    // (whole waves enter every trip: the parking step below uses wave-wide ballots)
    for (uint32_t w0 = blockIdx.x * kTopBlock; w0 < a.n_active; w0 += gridDim.x * kTopBlock) {
        const uint32_t w = w0 + threadIdx.x;
        const bool live = w < a.n_active;
        const uint32_t ray = live ? (a.active ? a.active[w] : w) : 0u;
        bool parked = false;
        uint32_t leafId = 0;
        // a shadow ray already occluded in its last leaf is done (offline_batching...h:246-247)
        if (live && !(ANY && a.any_out[ray] != 0u)) {
            const float ox = a.ox[ray], oy = a.oy[ray], oz = a.oz[ray];
            const float dx = a.dx[ray], dy = a.dy[ray], dz = a.dz[ray];
            const float tnear = a.tnear[ray];
            const float tfar = ANY ? a.tfar_in[ray] : a.tfar_state[ray]; // simdRay is rebuilt from the ray at every resume (:243-254)
            const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz; // :241
            uint32_t node = a.node[ray];
            uint64_t stack = a.stack[ray];
            uint32_t steps = 0;
            for (;;) {
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
                            const TopLeafDev lf = a.top.leaves[leaf];
                            bool culled = false;
                            if (lf.svdag != PGPU_MISS_ID) {
                                ++nTests;
                                const SvdagDev dag = a.top.dags[lf.svdag];
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
            if (parked) {
                a.node[ray] = node;
                a.stack[ray] = stack;
                if (a.visits)
                    a.visits[ray] += 1u; // Ray::numTopLevelIntersections (:283)
            }
        }
        // park: the round's list keeps one entry per work item (the batching point, or none), one count per point
        if (live)
            a.parked_leaf[w] = parked ? leafId : PGPU_MISS_ID;
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
}

// exclusive scan of leaf_count into leaf_start (single block; batching points number in the hundreds)
__global__ __launch_bounds__(1024) void leafScanKernel(const uint32_t* __restrict__ count, uint32_t* __restrict__ start,
    uint32_t* __restrict__ fill, uint32_t n, uint32_t* __restrict__ total)
{
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
        run += count[i];
    }
    if (threadIdx.x == 1023)
        *total = part[1023]; // rays parked in this round
}

// scatter parked records into per-leaf contiguous ranges (the GPU form of the per-leaf queues)
__global__ __launch_bounds__(256) void binByLeafKernel(const uint32_t* __restrict__ active, const uint32_t* __restrict__ parkedLeaf,
    uint32_t n, uint32_t nLeaves, const uint32_t* __restrict__ leafStart, uint32_t* __restrict__ leafFill,
    const TopLeafDev* __restrict__ leaves, uint32_t* __restrict__ workRay, uint32_t* __restrict__ workBvh, uint32_t* __restrict__ workLeaf)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t leaf = i < n ? parkedLeaf[i] : PGPU_MISS_ID;
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
    const TopLeafDev* __restrict__ leaves, uint32_t* __restrict__ workRay, uint32_t* __restrict__ workBvh, uint32_t* __restrict__ workLeaf)
{
    extern __shared__ uint32_t binLds[]; // [nLeaves] running count, [nLeaves] first slot of this block's range
    uint32_t* const cnt = binLds;
    uint32_t* const first = binLds + nLeaves;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t per = (((n + gridDim.x - 1u) / gridDim.x) + 255u) & ~255u;
    const uint32_t lo = min(blockIdx.x * per, n), hi = min(lo + per, n);
    for (uint32_t i = threadIdx.x; i < nLeaves; i += 256u)
        cnt[i] = 0u;
    __syncthreads();
    for (uint32_t i0 = lo; i0 < hi; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t leaf = i < hi ? parkedLeaf[i] : PGPU_MISS_ID;
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
        const uint32_t leaf = i < hi ? parkedLeaf[i] : PGPU_MISS_ID;
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
    return size_t(top.dag_levels) * kTopBlock * 4u + (topNodesFitLds(top.n_nodes) ? ((size_t(top.n_nodes) * kTopNodeBytes + 15u) & ~size_t(15)) : 16u)
        + 8u + (top.n_leaves <= kTopHistMax ? size_t(top.n_leaves) * 4u : 0u);
}

hipError_t launchTopTraverse(const TopTraceArgs& a, int kind, uint32_t blocks, hipStream_t stream)
{
    const bool inLds = topNodesFitLds(a.top.n_nodes);
    const dim3 grid(blocks), block(kTopBlock);
    const size_t ldsBytes = topTraverseLdsBytes(a.top);
    if (kind == PGPU_ANY_ID) {
        if (inLds)
            hipLaunchKernelGGL((topTraverseKernel<true, true>), grid, block, ldsBytes, stream, a);
        else
            hipLaunchKernelGGL((topTraverseKernel<true, false>), grid, block, ldsBytes, stream, a);
    } else {
        if (inLds)
            hipLaunchKernelGGL((topTraverseKernel<false, true>), grid, block, ldsBytes, stream, a);
        else
            hipLaunchKernelGGL((topTraverseKernel<false, false>), grid, block, ldsBytes, stream, a);
    }
    return hipGetLastError();
}

hipError_t launchLeafScan(const uint32_t* count, uint32_t* start, uint32_t* fill, uint32_t n, uint32_t* total, hipStream_t stream)
{
    hipLaunchKernelGGL(leafScanKernel, dim3(1), dim3(1024), 0, stream, count, start, fill, n, total);
    return hipGetLastError();
}

hipError_t launchBinByLeaf(const uint32_t* active, const uint32_t* parkedLeaf, uint32_t nActive, uint32_t nLeaves, const uint32_t* leafStart,
    uint32_t* leafFill, const TopLeafDev* leaves, uint32_t* workRay, uint32_t* workBvh, uint32_t* workLeaf, uint32_t blocks, hipStream_t stream)
{
    if (nLeaves <= kTopHistMax) {
        hipLaunchKernelGGL(binByLeafLdsKernel, dim3(blocks), dim3(256), size_t(nLeaves) * 8u, stream, active, parkedLeaf, nActive, nLeaves, leafStart,
            leafFill, leaves, workRay, workBvh, workLeaf);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(binByLeafKernel, dim3(blocks), dim3(256), 0, stream, active, parkedLeaf, nActive, nLeaves, leafStart, leafFill, leaves,
        workRay, workBvh, workLeaf);
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
