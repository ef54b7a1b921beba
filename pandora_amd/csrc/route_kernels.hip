// Device side of the leaf-sharded loop (pandora_amd/sharded.py): packing the rays a top-level episode
// parked into per-destination record ranges for the all-to-all, unpacking received records into a work list, packing
// and merging the results. A record is 9 dwords: ox oy oz dx dy dz tnear tfar leaf (tfar = the ray's current t for
// closest hit); a result is {t, obj, u, v} (closest) or {hit} (any).
#include "kernels.h"
#include "wave_ops.h"
#include <hip/hip_runtime.h>

namespace pandora_hip {

namespace {

constexpr uint32_t kMaxRanks = 64;

__device__ __forceinline__ uint32_t destOf(const uint32_t* __restrict__ parkedLeaf, const uint32_t* __restrict__ owner, uint32_t nLeaves, uint32_t i,
    uint32_t n, uint32_t* leaf)
{
    *leaf = i < n ? parkedLeaf[i] : PGPU_MISS_ID;
    return *leaf < nLeaves ? owner[*leaf] : PGPU_MISS_ID;
}

// counts[d] += parked work items whose batching point is owned by rank d
__global__ __launch_bounds__(256) void routeCountKernel(const uint32_t* __restrict__ parkedLeaf, uint32_t n, const uint32_t* __restrict__ owner,
    uint32_t nLeaves, uint32_t world, uint32_t* __restrict__ counts)
{
    __shared__ uint32_t cnt[kMaxRanks];
    if (threadIdx.x < kMaxRanks)
        cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {
        uint32_t leaf;
        const uint32_t dest = destOf(parkedLeaf, owner, nLeaves, i0 + threadIdx.x, n, &leaf);
        const bool live = dest < world;
        if (__ballot(live) != 0) {
            const KeyGroup kg = groupByKey(dest, live);
            if (live && kg.leader == (threadIdx.x & 63u))
                atomicAdd(&cnt[dest], kg.size);
        }
    }
    __syncthreads();
    if (threadIdx.x < world && cnt[threadIdx.x] != 0u)
        atomicAdd(&counts[threadIdx.x], cnt[threadIdx.x]);
}

struct RouteStarts {
    uint32_t v[kMaxRanks];
};

// A block owns a contiguous slice of the round's list: counts it per destination in LDS, reserves one range per
// destination with a single global atomic, hands out the slots from LDS and writes the records.
__global__ __launch_bounds__(256) void routeScatterKernel(const float* __restrict__ ox, const float* __restrict__ oy, const float* __restrict__ oz,
    const float* __restrict__ dx, const float* __restrict__ dy, const float* __restrict__ dz, const float* __restrict__ tnear,
    const float* __restrict__ tfar, const uint32_t* __restrict__ active, const uint32_t* __restrict__ parkedLeaf, uint32_t n,
    const uint32_t* __restrict__ owner, uint32_t nLeaves, uint32_t world, RouteStarts starts, uint32_t* __restrict__ fill,
    float* __restrict__ records, uint32_t* __restrict__ sentSlot, uint32_t* __restrict__ sentLeaf)
{
    __shared__ uint32_t cnt[kMaxRanks], first[kMaxRanks];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t per = (((n + gridDim.x - 1u) / gridDim.x) + 255u) & ~255u;
    const uint32_t lo = min(blockIdx.x * per, n), hi = min(lo + per, n);
    if (threadIdx.x < kMaxRanks)
        cnt[threadIdx.x] = 0u;
    __syncthreads();
    for (uint32_t i0 = lo; i0 < hi; i0 += 256u) {
        uint32_t leaf;
        const uint32_t dest = destOf(parkedLeaf, owner, nLeaves, i0 + threadIdx.x, hi, &leaf);
        const bool live = dest < world;
        if (__ballot(live) != 0) {
            const KeyGroup kg = groupByKey(dest, live);
            if (live && kg.leader == lane)
                atomicAdd(&cnt[dest], kg.size);
        }
    }
    __syncthreads();
    if (threadIdx.x < world) {
        const uint32_t c = cnt[threadIdx.x];
        first[threadIdx.x] = c ? starts.v[threadIdx.x] + atomicAdd(&fill[threadIdx.x], c) : 0u;
        cnt[threadIdx.x] = 0u;
    }
    __syncthreads();
    for (uint32_t i0 = lo; i0 < hi; i0 += 256u) {
        const uint32_t i = i0 + threadIdx.x;
        uint32_t leaf;
        const uint32_t dest = destOf(parkedLeaf, owner, nLeaves, i, hi, &leaf);
        const bool live = dest < world;
        if (__ballot(live) == 0)
            continue;
        const KeyGroup kg = groupByKey(dest, live);
        uint32_t off = 0;
        if (live && kg.leader == lane)
            off = atomicAdd(&cnt[dest], kg.size);
        off = uint32_t(__builtin_amdgcn_ds_bpermute(int(kg.leader << 2), int(off)));
        if (live) {
            const uint32_t at = first[dest] + off + kg.rank;
            const uint32_t ray = active ? active[i] : i;
            float* r = records + size_t(at) * 9u;
            r[0] = ox[ray];
            r[1] = oy[ray];
            r[2] = oz[ray];
            r[3] = dx[ray];
            r[4] = dy[ray];
            r[5] = dz[ray];
            r[6] = tnear[ray];
            r[7] = tfar[ray];
            r[8] = __uint_as_float(leaf);
            sentSlot[at] = ray;
            sentLeaf[at] = leaf;
        }
    }
}

// received records -> SoA rays [8][k] + the batching point of every work item + a fresh miss record
__global__ __launch_bounds__(256) void unpackRecordsKernel(const float* __restrict__ records, uint32_t k, float* __restrict__ soa,
    uint32_t* __restrict__ workLeaf, float* __restrict__ t, uint32_t* __restrict__ obj, uint32_t* __restrict__ anyHit)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < k; i += gridDim.x * 256u) {
        const float* r = records + size_t(i) * 9u;
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c)
            soa[size_t(c) * k + i] = r[c];
        workLeaf[i] = __float_as_uint(r[8]);
        if (t) {
            t[i] = r[7];
            obj[i] = PGPU_MISS_ID;
        }
        if (anyHit)
            anyHit[i] = 0u;
    }
}

// result rows: {hit} (any hit), {t, obj, u, v} (closest hit), or {t, obj, u, v, inst} when the scene has instanced leaves
__global__ __launch_bounds__(256) void packResultsKernel(uint32_t k, const float* __restrict__ t, const uint32_t* __restrict__ obj,
    const float* __restrict__ u, const float* __restrict__ v, const uint32_t* __restrict__ anyHit, const uint32_t* __restrict__ inst,
    float* __restrict__ results)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < k; i += gridDim.x * 256u) {
        if (anyHit) {
            results[i] = __uint_as_float(anyHit[i]);
        } else if (inst) {
            float* r = results + size_t(i) * 5u;
            r[0] = t[i];
            r[1] = __uint_as_float(obj[i]);
            r[2] = u[i];
            r[3] = v[i];
            r[4] = __uint_as_float(inst[i]);
        } else {
            float4 r;
            r.x = t[i];
            r.y = __uint_as_float(obj[i]);
            r.z = u[i];
            r.w = v[i];
            reinterpret_cast<float4*>(results)[i] = r;
        }
    }
}

// home side: a returned result that found something closer replaces the ray's hit record
__global__ __launch_bounds__(256) void mergeResultsKernel(uint32_t k, const float* __restrict__ results, const uint32_t* __restrict__ sentSlot,
    const uint32_t* __restrict__ sentLeaf, float* __restrict__ t, uint32_t* __restrict__ obj, uint32_t* __restrict__ leaf, float* __restrict__ u,
    float* __restrict__ v, uint32_t* __restrict__ anyHit, uint32_t* __restrict__ inst, uint32_t rowFloats)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < k; i += gridDim.x * 256u) {
        const uint32_t ray = sentSlot[i];
        if (anyHit) {
            if (__float_as_uint(results[i]) != 0u)
                anyHit[ray] = 1u;
        } else if (rowFloats == 5u) {
            const float* r = results + size_t(i) * 5u;
            if (__float_as_uint(r[1]) != PGPU_MISS_ID) {
                t[ray] = r[0];
                obj[ray] = __float_as_uint(r[1]);
                leaf[ray] = sentLeaf[i];
                u[ray] = r[2];
                v[ray] = r[3];
                if (inst)
                    inst[ray] = __float_as_uint(r[4]);
            }
        } else {
            const float4 r = reinterpret_cast<const float4*>(results)[i];
            if (__float_as_uint(r.y) != PGPU_MISS_ID) {
                t[ray] = r.x;
                obj[ray] = __float_as_uint(r.y);
                leaf[ray] = sentLeaf[i];
                u[ray] = r.z;
                v[ray] = r.w;
            }
        }
    }
}

// counts[key] += 1 for every key < nKeys: the leaf histogram of received work. Coherent batches put most of their rays on a
// handful of keys, and global atomics on one address serialise across the chip (the same effect that made the first top-level
// kernel 60x slower): a block therefore counts in LDS (wave-aggregated) and flushes its non-zero counters once, one global
// atomic each. IN_LDS = false (more keys than fit): wave-aggregated global atomics.
template <bool IN_LDS>
__global__ __launch_bounds__(256) void keyCountKernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t nKeys, uint32_t* __restrict__ counts)
{
    extern __shared__ uint32_t ldsCounts[];
    if (IN_LDS) {
        for (uint32_t k = threadIdx.x; k < nKeys; k += 256u)
            ldsCounts[k] = 0u;
        __syncthreads();
    }
    for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += gridDim.x * 256u) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t key = i < n ? keys[i] : PGPU_MISS_ID;
        const bool live = key < nKeys;
        if (__ballot(live) != 0) {
            const KeyGroup kg = groupByKey(key, live);
            if (live && kg.leader == (threadIdx.x & 63u))
                atomicAdd(IN_LDS ? &ldsCounts[key] : &counts[key], kg.size);
        }
    }
    if (IN_LDS) {
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < nKeys; k += 256u) {
            const uint32_t c = ldsCounts[k];
            if (c)
                atomicAdd(&counts[k], c);
        }
    }
}

} // namespace

uint32_t routeMaxRanks() { return kMaxRanks; }

hipError_t launchRouteCount(const uint32_t* parkedLeaf, uint32_t n, const uint32_t* owner, uint32_t nLeaves, uint32_t world, uint32_t* counts,
    uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(routeCountKernel, dim3(blocks), dim3(256), 0, stream, parkedLeaf, n, owner, nLeaves, world, counts);
    return hipGetLastError();
}

hipError_t launchRouteScatter(const float* const rays[8], const uint32_t* active, const uint32_t* parkedLeaf, uint32_t n, const uint32_t* owner,
    uint32_t nLeaves, uint32_t world, const uint32_t* startsHost, uint32_t* fill, float* records, uint32_t* sentSlot, uint32_t* sentLeaf,
    uint32_t blocks, hipStream_t stream)
{
    RouteStarts s {};
    for (uint32_t i = 0; i < world && i < kMaxRanks; ++i)
        s.v[i] = startsHost[i];
    hipLaunchKernelGGL(routeScatterKernel, dim3(blocks), dim3(256), 0, stream, rays[0], rays[1], rays[2], rays[3], rays[4], rays[5], rays[6], rays[7],
        active, parkedLeaf, n, owner, nLeaves, world, s, fill, records, sentSlot, sentLeaf);
    return hipGetLastError();
}

hipError_t launchUnpackRecords(const float* records, uint32_t k, float* soa, uint32_t* workLeaf, float* t, uint32_t* obj, uint32_t* anyHit,
    uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(unpackRecordsKernel, dim3(blocks), dim3(256), 0, stream, records, k, soa, workLeaf, t, obj, anyHit);
    return hipGetLastError();
}

hipError_t launchPackResults(uint32_t k, const float* t, const uint32_t* obj, const float* u, const float* v, const uint32_t* anyHit, float* results,
    uint32_t blocks, hipStream_t stream, const uint32_t* inst)
{
    hipLaunchKernelGGL(packResultsKernel, dim3(blocks), dim3(256), 0, stream, k, t, obj, u, v, anyHit, inst, results);
    return hipGetLastError();
}

hipError_t launchMergeResults(uint32_t k, const float* results, const uint32_t* sentSlot, const uint32_t* sentLeaf, float* t, uint32_t* obj,
    uint32_t* leaf, float* u, float* v, uint32_t* anyHit, uint32_t blocks, hipStream_t stream, uint32_t* inst, uint32_t rowFloats)
{
    hipLaunchKernelGGL(mergeResultsKernel, dim3(blocks), dim3(256), 0, stream, k, results, sentSlot, sentLeaf, t, obj, leaf, u, v, anyHit, inst, rowFloats);
    return hipGetLastError();
}

hipError_t launchKeyCount(const uint32_t* keys, uint32_t n, uint32_t nKeys, uint32_t* counts, uint32_t blocks, hipStream_t stream)
{
    if (nKeys <= 8192u)
        hipLaunchKernelGGL(keyCountKernel<true>, dim3(blocks), dim3(256), size_t(nKeys) * 4u, stream, keys, n, nKeys, counts);
    else
        hipLaunchKernelGGL(keyCountKernel<false>, dim3(blocks), dim3(256), 0, stream, keys, n, nKeys, counts);
    return hipGetLastError();
}

} // namespace pandora_hip
