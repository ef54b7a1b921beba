// Probe (development aid): where does the immediate offset of global_load_lds_dwordx4 land in LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const uint32_t* src, uint32_t* out)
{
    extern __shared__ __align__(16) unsigned char lds[];
    for (uint32_t i = threadIdx.x; i < 4096; i += 64)
        reinterpret_cast<uint32_t*>(lds)[i] = 0xDEADBEEFu;
    __syncthreads();
    const unsigned char* g = reinterpret_cast<const unsigned char*>(src) + threadIdx.x * 16;
    // offset 64: global side reads src + lane*16 + 64; does the LDS side also move by 64?
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
        (__attribute__((address_space(3))) void*)(lds + 4096), 16, 64, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 4096; i += 64)
        out[i] = reinterpret_cast<uint32_t*>(lds)[i];
}
int main()
{
    std::vector<uint32_t> h(4096);
    for (uint32_t i = 0; i < 4096; ++i) h[i] = i;
    uint32_t *d, *o;
    hipMalloc(&d, 16384); hipMalloc(&o, 16384);
    hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 16384, 0, d, o);
    hipMemcpy(h.data(), o, 16384, hipMemcpyDeviceToHost);
    uint32_t first = 0xFFFFFFFFu, last = 0;
    for (uint32_t i = 0; i < 4096; ++i) if (h[i] != 0xDEADBEEFu) { if (first == 0xFFFFFFFFu) first = i; last = i; }
    printf("first written dword %u (byte %u) value %u ; last dword %u value %u\n", first, first * 4, h[first], last, h[last]);
    printf("expected lds base byte 4096: imm offset %s the LDS address\n", first * 4 == 4096 ? "DOES NOT move" : (first * 4 == 4160 ? "ALSO moves" : "??"));
    return 0;
}
