// Streamed leaves (BASELINE config 5): the copy side of the HBM leaf cache of top_abi.hip.
//
// The reference loads the geometry of a batching point from its disk cache when the point's queue is flushed
// (CachedBVH::doMakeResident, offline_bvh_cache.cpp:127-148). Here the image of a point's BVH8 (nodes, triangle records,
// instance table) lives in pinned host memory and one kernel per launch group pulls every image the group needs over PCIe:
// a group of a sparse round is made of many small images (tens to hundreds of KB), and one hipMemcpyAsync per image costs
// more in submission than in transfer.
#include "kernels.h"
#include <hip/hip_runtime.h>

namespace pandora_hip {

namespace {

constexpr uint32_t kChunk16 = 1024; // 16-byte pieces per chunk: 16 KiB

// Every block takes chunks round-robin; a chunk belongs to the last segment whose chunk0 is not above it.
__global__ __launch_bounds__(256) void leafGatherKernel(const LeafCopySeg* __restrict__ segs, uint32_t nSegs, uint32_t nChunks)
{
    for (uint32_t chunk = blockIdx.x; chunk < nChunks; chunk += gridDim.x) {
        uint32_t lo = 0, hi = nSegs; // (wave-uniform search)
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (segs[mid].chunk0 <= chunk)
                lo = mid;
            else
                hi = mid;
        }
        const LeafCopySeg s = segs[lo];
        const uint64_t first = uint64_t(chunk - uint32_t(s.chunk0)) * kChunk16, n16 = s.bytes >> 4;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        const u4* const src = static_cast<const u4*>(s.src) + first;
        u4* const dst = static_cast<u4*>(s.dst) + first;
        const uint32_t n = uint32_t(n16 - first < kChunk16 ? n16 - first : kChunk16);
        // four loads in flight per lane before the first store: the reads cross PCIe
        u4 v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = threadIdx.x + k * 256u;
            if (i < n)
                v[k] = __builtin_nontemporal_load(src + i);
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t i = threadIdx.x + k * 256u;
            if (i < n)
                dst[i] = v[k];
        }
    }
}

__global__ void leafTableFixKernel(const unsigned char* __restrict__ fixes64, uint32_t nFixes, Bvh8Dev* __restrict__ table, uint32_t* __restrict__ arrivalGroup)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nFixes) {
        const LeafTableFix f = *reinterpret_cast<const LeafTableFix*>(fixes64 + size_t(i) * 64u);
        table[f.id] = f.entry;
        if (arrivalGroup)
            arrivalGroup[f.id] = f.group;
    }
}

__global__ void leafTableViewKernel(const Bvh8Dev* __restrict__ table, const uint32_t* __restrict__ leafFlag, const uint32_t* __restrict__ arrivalGroup,
    uint32_t group, Bvh8Dev* __restrict__ view, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        Bvh8Dev e = table[i];
        if (leafFlag[i] != 0u && arrivalGroup[i] != group) {
            e.nodes = nullptr;
            e.tris = nullptr;
            e.inst = nullptr;
        }
        view[i] = e;
    }
}

// Rays that stay parked for another round (their batching point's image is not worth loading yet, see top_abi.hip): the
// counts of the held rays join the round's per-point counts before the scan ...
__global__ void addCountsKernel(uint32_t* __restrict__ count, const uint32_t* __restrict__ add, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        count[i] += add[i];
}

// ... and after the round the binned list (sorted by batching point) is split: the rays of the points that were traced form
// the next round's list, the rays of the others the next held list. dst[l] = where point l's rays go; bit 31 = held.
// (One thread per list position, the position's point by binary search over the starts: a block per point leaves the few points
// that hold most of a round's rays to one block each -- 0.2 ms per launch on the benchmark landscape.)
__global__ __launch_bounds__(256) void splitByLeafKernel(const uint32_t* __restrict__ workRay, const uint32_t* __restrict__ leafStart,
    const uint32_t* __restrict__ leafCount, const uint32_t* __restrict__ dst, uint32_t nLeaves, uint32_t nItems, uint32_t* __restrict__ nextActive,
    uint32_t* __restrict__ heldRay, uint32_t* __restrict__ heldLeaf)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < nItems; i += gridDim.x * 256u) {
        // the last point whose range starts at or before i (empty points share their start with the next one)
        uint32_t lo = 0, hi = nLeaves;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (leafStart[mid] <= i)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t l = lo;
        const uint32_t at = i - leafStart[l];
        if (at >= leafCount[l])
            continue; // (cannot happen for a list that the starts and counts describe)
        const uint32_t d = dst[l], to = (d & 0x7FFFFFFFu) + at, ray = workRay[i];
        if (d & 0x80000000u) {
            heldRay[to] = ray;
            heldLeaf[to] = l;
        } else {
            nextActive[to] = ray;
        }
    }
}

} // namespace

hipError_t launchAddCounts(uint32_t* count, const uint32_t* add, uint32_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(addCountsKernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, count, add, n);
    return hipGetLastError();
}

hipError_t launchSplitByLeaf(const uint32_t* workRay, const uint32_t* leafStart, const uint32_t* leafCount, const uint32_t* dst, uint32_t nLeaves, uint32_t nItems,
    uint32_t* nextActive, uint32_t* heldRay, uint32_t* heldLeaf, uint32_t blocks, hipStream_t stream)
{
    if (nItems == 0u || nLeaves == 0u)
        return hipSuccess;
    const uint32_t need = (nItems + 255u) / 256u;
    hipLaunchKernelGGL(splitByLeafKernel, dim3(blocks < need ? blocks : need), dim3(256), 0, stream, workRay, leafStart, leafCount, dst, nLeaves, nItems,
        nextActive, heldRay, heldLeaf);
    return hipGetLastError();
}

hipError_t launchLeafGather(const LeafCopySeg* segs, uint32_t nSegs, uint32_t nChunks, uint32_t blocks, hipStream_t stream)
{
    if (nSegs == 0u || nChunks == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(leafGatherKernel, dim3(blocks < nChunks ? blocks : nChunks), dim3(256), 0, stream, segs, nSegs, nChunks);
    return hipGetLastError();
}

hipError_t launchLeafTableFix(const unsigned char* fixes64, uint32_t nFixes, Bvh8Dev* table, hipStream_t stream, uint32_t* arrivalGroup)
{
    if (nFixes == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(leafTableFixKernel, dim3((nFixes + 63u) / 64u), dim3(64), 0, stream, fixes64, nFixes, table, arrivalGroup);
    return hipGetLastError();
}

hipError_t launchLeafTableView(const Bvh8Dev* table, const uint32_t* leafFlag, const uint32_t* arrivalGroup, uint32_t group, Bvh8Dev* view, uint32_t n,
    hipStream_t stream)
{
    if (n == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(leafTableViewKernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, table, leafFlag, arrivalGroup, group, view, n);
    return hipGetLastError();
}

} // namespace pandora_hip
