// Film accumulation: Sensor::addPixelContribution / Sensor::Pixel, graphics_core/sensor.cpp:36-49, :86-99,
// sensor.h:29-35. Three u64 per pixel in 40.24 fixed point; integer adds commute, so the film does not depend on the
// order in which rays finish and sums over GPUs with an integer all-reduce (pandora_amd/sharded.py: film_allreduce).
//
// The fixed-point type is cnl::fixed_point<uint64_t, -24> (CNL, a vcpkg dependency that is not in the container):
// construction from float scales by 2^24 and converts to the integer representation (truncation); conversion back is
// float(rep) * 2^-24. Restated from that definition: parity unpinned.
#include "context.h"
#include <algorithm>

using namespace pandora_hip;

namespace {

// one channel of Pixel(color): float -> 40.24. Negative values wrap (two's complement: adding them subtracts).
__device__ __forceinline__ unsigned long long toFixed(float c) { return (unsigned long long)(long long)(c * 16777216.0f); }

__global__ __launch_bounds__(256) void filmAddKernel(unsigned long long* film, uint32_t nPixels, const uint32_t* pixel, const float* r,
    const float* g, const float* b, uint32_t n)
{
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t p = pixel[i];
        const float cr = r[i], cg = g[i], cb = b[i];
        // sensor.cpp:41-42: a contribution with a NaN or inf channel is dropped whole; so is one beyond the format's
        // range (|c| >= 2^39, undefined in the reference) and one that names a pixel outside the film
        const float lim = 549755813888.0f;
        const bool ok = p < nPixels && fabsf(cr) < lim && fabsf(cg) < lim && fabsf(cb) < lim; // (false for NaN)
        if (ok) {
            unsigned long long* px = film + size_t(p) * 3u;
            __hip_atomic_fetch_add(px + 0, toFixed(cr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(px + 1, toFixed(cg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(px + 2, toFixed(cb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ __launch_bounds__(256) void filmResolveKernel(const unsigned long long* film, uint32_t nValues, float* out)
{
    const size_t stride = size_t(gridDim.x) * blockDim.x; // (nValues is close to 2^32 for the largest film: 64-bit index)
    for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < nValues; i += stride)
        out[i] = float(film[i]) * 5.9604644775390625e-08f; // sensor.cpp:96-99 (2^-24)
}

} // namespace

extern "C" {

int pgpu_film_add(pgpu_ctx* c, unsigned long long* film, size_t n_pixels, const uint32_t* pixel, const float* r, const float* g,
    const float* b, size_t n, void* stream_)
{
    if (!c)
        return PGPU_ERR_INVALID;
    if (!film || n_pixels == 0 || n_pixels > 0x55555555u || n > 0x3FFFFFFFu || (n && (!pixel || !r || !g || !b)))
        return fail(c, PGPU_ERR_INVALID, "pgpu_film_add: null argument, empty film or more than 2^30 contributions");
    if (n == 0)
        return PGPU_OK;
    if (hipSetDevice(c->device) != hipSuccess)
        return fail(c, PGPU_ERR_HIP, "pgpu_film_add: hipSetDevice");
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    const uint32_t blocks = uint32_t(std::min<size_t>((n + 255) / 256, size_t(c->numCUs) * 8));
    hipLaunchKernelGGL(filmAddKernel, dim3(blocks), dim3(256), 0, stream, film, uint32_t(n_pixels), pixel, r, g, b, uint32_t(n));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PGPU_OK : fail(c, PGPU_ERR_HIP, "pgpu_film_add: launch", e);
}

int pgpu_film_resolve(pgpu_ctx* c, const unsigned long long* film, size_t n_pixels, float* rgb_out, void* stream_)
{
    if (!c)
        return PGPU_ERR_INVALID;
    if (!film || !rgb_out || n_pixels == 0 || n_pixels > 0x55555555u)
        return fail(c, PGPU_ERR_INVALID, "pgpu_film_resolve: null argument or empty film");
    if (hipSetDevice(c->device) != hipSuccess)
        return fail(c, PGPU_ERR_HIP, "pgpu_film_resolve: hipSetDevice");
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    const uint32_t nValues = uint32_t(n_pixels * 3);
    const uint32_t blocks = uint32_t(std::min<size_t>((size_t(nValues) + 255) / 256, size_t(c->numCUs) * 8));
    hipLaunchKernelGGL(filmResolveKernel, dim3(blocks), dim3(256), 0, stream, film, nValues, rgb_out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? PGPU_OK : fail(c, PGPU_ERR_HIP, "pgpu_film_resolve: launch", e);
}

} // extern "C"
