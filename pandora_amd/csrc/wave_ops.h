// Wave64 / quad primitives shared by the kernels: DPP quad_perm data movement (what the reference does with vpermps / shuffles
// inside one SIMD register, projects/simd) and the leader election that keeps same-address atomics off single lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pandora_hip {

constexpr uint64_t kQuadLane0 = 0x1111111111111111ull;
constexpr uint64_t kQuadLane3 = 0x8888888888888888ull;

// value of lane `(lane & ~3) | perm[lane & 3]` (v_*_dpp quad_perm: folded into the consuming VALU instruction)
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ int quadPerm(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xF, 0xF, true);
}
template <int K>
__device__ __forceinline__ float quadBcast(float v) { return __int_as_float(quadPerm<K, K, K, K>(__float_as_int(v))); }
template <int K>
__device__ __forceinline__ uint32_t quadBcastU(uint32_t v) { return uint32_t(quadPerm<K, K, K, K>(int(v))); }
__device__ __forceinline__ uint32_t quadOr(uint32_t v)
{
    v |= uint32_t(quadPerm<1, 0, 3, 2>(int(v)));
    v |= uint32_t(quadPerm<2, 3, 0, 1>(int(v)));
    return v;
}
__device__ __forceinline__ uint32_t quadAdd(uint32_t v)
{
    v += uint32_t(quadPerm<1, 0, 3, 2>(int(v)));
    v += uint32_t(quadPerm<2, 3, 0, 1>(int(v)));
    return v;
}
__device__ __forceinline__ float quadMin(float v)
{
    v = fminf(v, __int_as_float(quadPerm<1, 0, 3, 2>(__float_as_int(v))));
    v = fminf(v, __int_as_float(quadPerm<2, 3, 0, 1>(__float_as_int(v))));
    return v;
}

// Many lanes of a wave often want the same counter (coherent rays park at the same batching point, records go to the same
// rank), and same-address atomics from single lanes serialise in L2. Lanes with equal keys therefore elect a leader that adds
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

} // namespace pandora_hip
