// Probe (development aid): cost of a wave's 16-byte-per-lane loads as a function of how the lanes' addresses
// fall into cache lines. Decides how the bottom-level kernels should spread a 256-byte node over lanes.
//   mode 0: each quad of lanes reads 64 contiguous bytes of one node (16 nodes per instruction)
//   mode 1: each pair of lanes reads 32 contiguous bytes of one node (32 nodes per instruction)
//   mode 2: the wave reads 1 KiB contiguous
//   mode 3: every lane reads 16 bytes of its own node (64 nodes per instruction)
//   mode 4: each group of 8 lanes reads a whole 128-byte line (8 nodes per instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash(uint32_t x)
{
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned char* nodes, uint32_t nNodesMask, uint32_t iters, uint32_t* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waveId = (blockIdx.x * 256u + threadIdx.x) >> 6;
    u4 acc = u4(0u);
    for (uint32_t it = 0; it < iters; ++it) {
        uint32_t group, within, step;
        if (MODE == 0) { group = lane >> 2; within = (lane & 3u) * 16u; step = 64u; }
        else if (MODE == 1) { group = lane >> 1; within = (lane & 1u) * 16u; step = 32u; }
        else if (MODE == 2) { group = 0; within = lane * 16u; step = 0u; }
        else if (MODE == 3) { group = lane; within = 0u; step = 16u; }
        else { group = lane >> 3; within = (lane & 7u) * 16u; step = 128u; }
        const uint32_t node = hash((waveId * 977u + it) * 64u + group) & nNodesMask;
        const unsigned char* p = nodes + size_t(node) * 256u + within;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // 4 loads per iteration; (mode 0: the whole node; mode 1: half of it; mode 4: two iterations' worth)
            const uint32_t off = MODE == 2 ? uint32_t(k) * 1024u : ((uint32_t(k) * step) & 255u);
            const u4 v = *reinterpret_cast<const u4*>(p + off);
            acc ^= v;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u)
        out[0] = 1;
}

template <int MODE>
static void run(const unsigned char* d, uint32_t nNodes, uint32_t* o, const char* name)
{
    const uint32_t iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 0, 0, d, nNodes - 1, 100u, o);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE>), dim3(blocks), dim3(256), 0, 0, d, nNodes - 1, iters, o);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = double(blocks) * 4 * iters * 4; // wave-level load instructions
    const double bytes = instr * 1024.0;
    printf("  %-46s %8.3f ms  %7.2f TB/s  %6.2f ns per wave-load per CU (%.1f clk @2.4GHz)\n", name, ms, bytes / ms / 1e9,
        ms * 1e6 / (instr / 256.0), ms * 1e6 / (instr / 256.0) * 2.4);
}

int main()
{
    uint32_t* o;
    hipMalloc(&o, 64);
    for (uint32_t nNodes : { 32u, 8192u, 131072u, 4194304u }) {
        unsigned char* d;
        const size_t bytes = size_t(nNodes) * 256 + 8192;
        hipMalloc(&d, bytes);
        hipMemset(d, 1, bytes);
        printf("%u nodes (%.1f MiB)\n", nNodes, bytes / 1048576.0);
        run<2>(d, nNodes, o, "wave reads 1 KiB contiguous");
        run<4>(d, nNodes, o, "8 lanes per 128-byte line (8 nodes/instr)");
        run<0>(d, nNodes, o, "quad per 64 bytes (16 nodes/instr)");
        run<1>(d, nNodes, o, "pair per 32 bytes (32 nodes/instr)");
        run<3>(d, nNodes, o, "lane per 16 bytes (64 nodes/instr)");
        hipFree(d);
    }
    return 0;
}
