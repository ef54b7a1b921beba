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

__global__ void leafTableFixKernel(const unsigned char* __restrict__ fixes64, uint32_t nFixes, Bvh8Dev* __restrict__ table)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nFixes) {
        const LeafTableFix f = *reinterpret_cast<const LeafTableFix*>(fixes64 + size_t(i) * 64u);
        table[f.id] = f.entry;
    }
}

} // namespace

hipError_t launchLeafGather(const LeafCopySeg* segs, uint32_t nSegs, uint32_t nChunks, uint32_t blocks, hipStream_t stream)
{
    if (nSegs == 0u || nChunks == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(leafGatherKernel, dim3(blocks < nChunks ? blocks : nChunks), dim3(256), 0, stream, segs, nSegs, nChunks);
    return hipGetLastError();
}

hipError_t launchLeafTableFix(const unsigned char* fixes64, uint32_t nFixes, Bvh8Dev* table, hipStream_t stream)
{
    if (nFixes == 0u)
        return hipSuccess;
    hipLaunchKernelGGL(leafTableFixKernel, dim3((nFixes + 63u) / 64u), dim3(64), 0, stream, fixes64, nFixes, table);
    return hipGetLastError();
}

} // namespace pandora_hip
