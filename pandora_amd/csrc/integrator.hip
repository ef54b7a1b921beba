// The benchmark integrator's ray sources and film splat on the device (include/pandora_hip.h has the
// reference lines each step follows). HBM-bound streaming kernels: one ray per lane, SoA in, SoA out, no shared state
// except the film's 64-bit integer atomics.
#include "context.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>

#pragma clang fp contract(off)

using namespace pandora_hip;

namespace {

// samplers/rng/pcg.cpp:32-39
__device__ __forceinline__ uint32_t pcgU32(unsigned long long& state, unsigned long long inc)
{
    const unsigned long long old = state;
    state = old * 0x5851F42D4C957F2Dull + inc;
    const uint32_t xorshifted = uint32_t(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = uint32_t(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
}
// :23-30
__device__ __forceinline__ void pcgSeed(unsigned long long initseq, unsigned long long& state, unsigned long long& inc)
{
    state = 0;
    inc = (initseq << 1u) | 1u;
    pcgU32(state, inc);
    state += 0x853C49E6748FEA9Bull;
    pcgU32(state, inc);
}
// :58-61
__device__ __forceinline__ float pcgFloat(unsigned long long& state, unsigned long long inc)
{
    const float f = float(pcgU32(state, inc)) * 2.3283064365386962890625e-10f; // 0x1p-32f
    const float lim = 1.0f - 1.1920928955078125e-07f;
    return lim < f ? lim : f;
}

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 normalize(V3 v)
{
    const float inv = 1.0f / sqrtf(dot(v, v)); // glm::normalize = v * inversesqrt(dot(v, v)); sqrtf is the correctly rounded form (__fsqrt_rn is not, on gfx950)
    return { v.x * inv, v.y * inv, v.z * inv };
}

struct CameraArgs {
    pgpu_camera cam;
    uint32_t width, height, spp;
    unsigned long long seed, first;
    uint32_t n;
    pgpu_rays_out rays;
    pgpu_paths paths;
};

__global__ __launch_bounds__(256) void cameraRaysKernel(CameraArgs a)
{
    const float resx = float(a.width), resy = float(a.height);
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < a.n; k += gridDim.x * 256u) {
        const unsigned long long i = a.first + k; // sampler_integrator.cpp:145-160
        const unsigned long long pixelIndex = i / a.spp;
        const uint32_t x = uint32_t(pixelIndex % a.width), y = uint32_t(pixelIndex / a.width);
        unsigned long long st, inc;
        pcgSeed(a.seed + i, st, inc);
        const float r0 = pcgFloat(st, inc), r1 = pcgFloat(st, inc);
        const float cx = (float(x) + r0) / resx - 0.5f, cy = (float(y) + r1) / resy - 0.5f; // perspective_camera.cpp:39
        const float* c = a.cam.lens_center;
        float o[3], d[3];
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {
            o[ax] = (c[ax] + cx * c[3 + ax]) + cy * c[6 + ax];
            const float sc = (c[9 + ax] + cx * c[12 + ax]) + cy * c[15 + ax];
            d[ax] = sc - o[ax];
        }
        const V3 dir = normalize({ d[0], d[1], d[2] });
        a.rays.ox[k] = o[0];
        a.rays.oy[k] = o[1];
        a.rays.oz[k] = o[2];
        a.rays.dx[k] = dir.x;
        a.rays.dy[k] = dir.y;
        a.rays.dz[k] = dir.z;
        a.rays.tnear[k] = 0.0f;
        a.rays.tfar[k] = 3.4028234663852886e38f;
        a.paths.rng_state[k] = st;
        a.paths.rng_inc[k] = inc;
        a.paths.pixel[k] = y * a.width + x;
        a.paths.wr[k] = 1.0f;
        a.paths.wg[k] = 1.0f;
        a.paths.wb[k] = 1.0f;
    }
}

struct BounceArgs {
    const Bvh8Dev* bvhs;
    const TopLeafDev* leaves; // nullptr: single level, fixedBvh
    uint32_t fixedBvh, nBvhs, nLeaves, n;
    const float *dx, *dy, *dz;
    const uint32_t *obj, *leaf;
    const float *u, *v;
    float albedo[3];
    pgpu_rays_out rays;
    pgpu_paths paths;
};

// A path that ended (it missed -- its sky contribution was added when the miss was splatted, path_integrator.cpp:113-127 -- or
// its bounce failed, :109-112): a ray that cannot hit anything AND a weight of zero, so that the miss of the dead ray adds nothing
// when the next depth is splatted. The reference ends the path; here its slot stays in the batch.
__device__ __forceinline__ void deadRay(const pgpu_rays_out& r, const pgpu_paths& p, uint32_t k)
{
    p.wr[k] = 0.0f;
    p.wg[k] = 0.0f;
    p.wb[k] = 0.0f;
    r.ox[k] = 0.0f;
    r.oy[k] = 0.0f;
    r.oz[k] = 0.0f;
    r.dx[k] = 1.0f;
    r.dy[k] = 0.0f;
    r.dz[k] = 0.0f;
    r.tnear[k] = 1.0f;
    r.tfar[k] = 0.0f;
}

__global__ __launch_bounds__(256) void lambertBounceKernel(BounceArgs a)
{
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < a.n; k += gridDim.x * 256u) {
        const uint32_t obj = a.obj[k];
        uint32_t bvhId = a.fixedBvh;
        if (a.leaves && obj != PGPU_MISS_ID) {
            const uint32_t l = a.leaf[k];
            bvhId = l < a.nLeaves ? a.leaves[l].bvh : PGPU_MISS_ID;
        }
        const float* shade = (obj != PGPU_MISS_ID && bvhId < a.nBvhs) ? a.bvhs[bvhId].shade : nullptr;
        if (!shade) {
            deadRay(a.rays, a.paths, k);
            continue;
        }
        const float* tp = shade + size_t(obj) * 9u;
        const V3 p0 { tp[0], tp[1], tp[2] }, p1 { tp[3], tp[4], tp[5] }, p2 { tp[6], tp[7], tp[8] };
        const V3 dir { a.dx[k], a.dy[k], a.dz[k] };
        const V3 ndir { -dir.x, -dir.y, -dir.z };
        // triangle_intersect.cpp:458-460
        V3 ng = normalize(cross({ p1.x - p0.x, p1.y - p0.y, p1.z - p0.z }, { p2.x - p0.x, p2.y - p0.y, p2.z - p0.z }));
        if (dot(ng, ndir) < 0.0f)
            ng = { -ng.x, -ng.y, -ng.z };
        // triangle_shading.cpp:14-21
        const float uu = a.u[k], vv = a.v[k];
        const float b0 = 1.0f - uu - vv;
        const V3 pos { (b0 * p0.x + uu * p1.x) + vv * p2.x, (b0 * p0.y + uu * p1.y) + vv * p2.y, (b0 * p0.z + uu * p1.z) + vv * p2.z };
        // material.cpp:5-20
        const V3 helper = ng.x > 0.5f ? V3 { 0.0f, 1.0f, 0.0f } : V3 { 1.0f, 0.0f, 0.0f };
        const V3 ss = normalize(cross(ng, helper));
        const V3 ts = normalize(cross(ng, ss));
        // sampling.h:8-49
        unsigned long long st = a.paths.rng_state[k];
        const unsigned long long inc = a.paths.rng_inc[k];
        const float r0 = pcgFloat(st, inc), r1 = pcgFloat(st, inc);
        const float ox = 2.0f * r0 - 1.0f, oy = 2.0f * r1 - 1.0f;
        float dxl = 0.0f, dyl = 0.0f;
        if (!(ox == 0.0f && oy == 0.0f)) {
            const float pi = 3.14159265358979323846264338327950288f;
            float theta, r;
            if (fabsf(ox) > fabsf(oy)) {
                r = ox;
                theta = pi / 4.0f * (oy / ox);
            } else {
                r = oy;
                theta = pi / 2.0f - pi / 4.0f * (ox / oy);
            }
            dxl = r * cosf(theta);
            dyl = r * sinf(theta);
        }
        const float zz = 1.0f - dxl * dxl - dyl * dyl;
        float wz = sqrtf(0.0f > zz ? 0.0f : zz);
        const float woz = dot(ndir, ng);
        if (woz < 0.0f)
            wz = -wz;
        const V3 wi { (ss.x * dxl + ts.x * dyl) + ng.x * wz, (ss.y * dxl + ts.y * dyl) + ng.y * wz, (ss.z * dxl + ts.z * dyl) + ng.z * wz }; // material.cpp:41-47
        const float invPi = 0.318309886183790671537767526745028724f;
        const float pdf = (woz * wz > 0.0f) ? fabsf(wz) * invPi : 0.0f;
        const unsigned long long left = pcgU32(st, inc), right = pcgU32(st, inc); // path_integrator.cpp:151
        unsigned long long nst, ninc;
        pcgSeed((left << 32) | right, nst, ninc);
        a.paths.rng_state[k] = nst;
        a.paths.rng_inc[k] = ninc;
        if (!(pdf > 0.0f) || (a.albedo[0] == 0.0f && a.albedo[1] == 0.0f && a.albedo[2] == 0.0f)) {
            deadRay(a.rays, a.paths, k);
            continue;
        }
        const float adot = fabsf(dot(wi, ng));
        a.paths.wr[k] = a.paths.wr[k] * (a.albedo[0] * invPi) * adot / pdf; // :152
        a.paths.wg[k] = a.paths.wg[k] * (a.albedo[1] * invPi) * adot / pdf;
        a.paths.wb[k] = a.paths.wb[k] * (a.albedo[2] * invPi) * adot / pdf;
        const bool front = dot(ng, wi) > 0.0f; // interaction.cpp:39-46
        const float eps = 0.000005f;
        a.rays.ox[k] = front ? pos.x + ng.x * eps : pos.x - ng.x * eps;
        a.rays.oy[k] = front ? pos.y + ng.y * eps : pos.y - ng.y * eps;
        a.rays.oz[k] = front ? pos.z + ng.z * eps : pos.z - ng.z * eps;
        a.rays.dx[k] = wi.x;
        a.rays.dy[k] = wi.y;
        a.rays.dz[k] = wi.z;
        a.rays.tnear[k] = 0.0f;
        a.rays.tfar[k] = 3.4028234663852886e38f;
    }
}

__device__ __forceinline__ unsigned long long toFixed(float c) { return (unsigned long long)(long long)(c * 16777216.0f); }

__global__ __launch_bounds__(256) void splatMissesKernel(unsigned long long* film, uint32_t nPixels, const uint32_t* obj, pgpu_paths paths, uint32_t n,
    float skyR, float skyG, float skyB)
{
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < n; k += gridDim.x * 256u) {
        if (obj[k] != PGPU_MISS_ID)
            continue;
        const uint32_t p = paths.pixel[k];
        const float cr = paths.wr[k] * skyR, cg = paths.wg[k] * skyG, cb = paths.wb[k] * skyB; // path_integrator.cpp:113-125
        const float lim = 549755813888.0f;
        if (p < nPixels && fabsf(cr) < lim && fabsf(cg) < lim && fabsf(cb) < lim) { // sensor.cpp:41-42 (as pgpu_film_add)
            unsigned long long* px = film + size_t(p) * 3u;
            __hip_atomic_fetch_add(px + 0, toFixed(cr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(px + 1, toFixed(cg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(px + 2, toFixed(cb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- pgpu_compact_paths: the paths that are still alive, in their order, at the front of another set of arrays
__global__ __launch_bounds__(256) void liveFlagsKernel(const float* __restrict__ tnear, const float* __restrict__ tfar, uint32_t n, uint32_t* __restrict__ flags)
{
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k <= n; k += gridDim.x * 256u)
        flags[k] = (k < n && tnear[k] <= tfar[k]) ? 1u : 0u; // (one element more: the exclusive scan then also yields the total)
}

struct CompactArgs {
    pgpu_rays in;
    pgpu_paths pin;
    pgpu_rays_out out;
    pgpu_paths pout;
    const uint32_t *flags, *places;
    uint32_t n;
};

__global__ __launch_bounds__(256) void compactPathsKernel(CompactArgs a)
{
    for (uint32_t k = blockIdx.x * 256u + threadIdx.x; k < a.n; k += gridDim.x * 256u) {
        if (a.flags[k] == 0u)
            continue;
        const uint32_t d = a.places[k];
        if (d >= a.n)
            continue; // (cannot happen: places count the live rays before k)
        a.out.ox[d] = a.in.ox[k];
        a.out.oy[d] = a.in.oy[k];
        a.out.oz[d] = a.in.oz[k];
        a.out.dx[d] = a.in.dx[k];
        a.out.dy[d] = a.in.dy[k];
        a.out.dz[d] = a.in.dz[k];
        a.out.tnear[d] = a.in.tnear[k];
        a.out.tfar[d] = a.in.tfar[k];
        a.pout.rng_state[d] = a.pin.rng_state[k];
        a.pout.rng_inc[d] = a.pin.rng_inc[k];
        a.pout.pixel[d] = a.pin.pixel[k];
        a.pout.wr[d] = a.pin.wr[k];
        a.pout.wg[d] = a.pin.wg[k];
        a.pout.wb[d] = a.pin.wb[k];
    }
}

uint32_t gridFor(const pgpu_ctx* c, size_t n) { return uint32_t(std::max<size_t>(1, std::min<size_t>((n + 255) / 256, size_t(c->numCUs) * 8))); }

bool pathsOk(const pgpu_paths* p) { return p && p->rng_state && p->rng_inc && p->pixel && p->wr && p->wg && p->wb; }
bool raysOutOk(const pgpu_rays_out* r) { return r && r->ox && r->oy && r->oz && r->dx && r->dy && r->dz && r->tnear && r->tfar; }

} // namespace

extern "C" {

int pgpu_camera_rays(pgpu_ctx* c, const pgpu_camera* cam, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint64_t first, size_t n,
    const pgpu_rays_out* rays, const pgpu_paths* paths, void* stream_)
{
    if (!c || !cam || !raysOutOk(rays) || !pathsOk(paths) || width == 0 || height == 0 || spp == 0)
        return fail(c, PGPU_ERR_INVALID, "pgpu_camera_rays: null argument or empty image");
    if (n > 0x3FFFFFFFull || first + n > uint64_t(width) * height * spp)
        return fail(c, PGPU_ERR_INVALID, "pgpu_camera_rays: sample range beyond width * height * spp (or more than 2^30 rays)");
    if (n == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    CameraArgs a {};
    a.cam = *cam;
    a.width = width;
    a.height = height;
    a.spp = spp;
    a.seed = seed;
    a.first = first;
    a.n = uint32_t(n);
    a.rays = *rays;
    a.paths = *paths;
    hipLaunchKernelGGL(cameraRaysKernel, dim3(gridFor(c, n)), dim3(256), 0, stream, a);
    HIP_TRY(c, hipGetLastError());
    return PGPU_OK;
}

int pgpu_lambert_bounce(pgpu_ctx* c, uint32_t bvh_id, const pgpu_rays* rays_in, const pgpu_hits* hits, size_t n, const float albedo[3],
    const pgpu_rays_out* rays, const pgpu_paths* paths, void* stream_)
{
    if (!c || !rays_in || !rays_in->dx || !rays_in->dy || !rays_in->dz || !hits || !hits->obj || !hits->u || !hits->v || !albedo || !raysOutOk(rays) || !pathsOk(paths))
        return fail(c, PGPU_ERR_INVALID, "pgpu_lambert_bounce: null argument");
    if (n > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_lambert_bounce: batch too large");
    const bool twoLevel = bvh_id == PGPU_NO_BVH;
    if (twoLevel && (!c->haveTop || !hits->leaf))
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_lambert_bounce: two-level hits need the uploaded top level and hits->leaf");
    if (!twoLevel && bvh_id >= c->bvhs.size())
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_lambert_bounce: unknown bvh id");
    if (c->hasInstances)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_lambert_bounce: instanced scenes are not supported by the benchmark integrator's bounce");
    for (size_t i = 0; i < c->bvhs.size(); ++i)
        if ((twoLevel || i == bvh_id) && !c->bvhs[i].shade)
            return fail(c, PGPU_ERR_NO_SCENE, "pgpu_lambert_bounce: a BVH was uploaded without shading data (pgpu_set_keep_shading before pgpu_upload_bvh8; staged BVHs have none)");
    if (n == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = syncBvhTable(c);
    if (rc != PGPU_OK)
        return rc;
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    BounceArgs a {};
    a.bvhs = c->bvhTable;
    a.leaves = twoLevel ? c->top.leaves : nullptr;
    a.fixedBvh = twoLevel ? 0u : bvh_id;
    a.nBvhs = uint32_t(c->bvhs.size());
    a.nLeaves = twoLevel ? c->top.n_leaves : 0u;
    a.n = uint32_t(n);
    a.dx = rays_in->dx;
    a.dy = rays_in->dy;
    a.dz = rays_in->dz;
    a.obj = hits->obj;
    a.leaf = hits->leaf;
    a.u = hits->u;
    a.v = hits->v;
    for (int i = 0; i < 3; ++i)
        a.albedo[i] = albedo[i];
    a.rays = *rays;
    a.paths = *paths;
    hipLaunchKernelGGL(lambertBounceKernel, dim3(gridFor(c, n)), dim3(256), 0, stream, a);
    HIP_TRY(c, hipGetLastError());
    return PGPU_OK;
}

int pgpu_splat_misses(pgpu_ctx* c, unsigned long long* film, size_t n_pixels, const uint32_t* obj, const pgpu_paths* paths, size_t n, const float sky[3],
    void* stream_)
{
    if (!c || !film || !obj || !sky || !pathsOk(paths) || n_pixels == 0 || n_pixels > 0x55555555u || n > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_splat_misses: null argument, empty film or batch too large");
    if (n == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    hipLaunchKernelGGL(splatMissesKernel, dim3(gridFor(c, n)), dim3(256), 0, stream, film, uint32_t(n_pixels), obj, *paths, uint32_t(n), sky[0], sky[1], sky[2]);
    HIP_TRY(c, hipGetLastError());
    return PGPU_OK;
}
int pgpu_compact_paths(pgpu_ctx* c, const pgpu_rays* rays_in, const pgpu_paths* paths_in, size_t n, const pgpu_rays_out* rays_out, const pgpu_paths* paths_out,
    size_t* n_live, void* stream_)
{
    if (!c || !rays_in || !rays_in->ox || !rays_in->oy || !rays_in->oz || !rays_in->dx || !rays_in->dy || !rays_in->dz || !rays_in->tnear || !rays_in->tfar
        || !pathsOk(paths_in) || !raysOutOk(rays_out) || !pathsOk(paths_out) || !n_live)
        return fail(c, PGPU_ERR_INVALID, "pgpu_compact_paths: null argument");
    if (n > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_compact_paths: batch too large");
    if (rays_out->ox == rays_in->ox || paths_out->pixel == paths_in->pixel)
        return fail(c, PGPU_ERR_INVALID, "pgpu_compact_paths: the output arrays must not be the input arrays");
    *n_live = 0;
    if (n == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    const size_t words = alignUp((n + 1) * 4);
    size_t scanBytes = 0;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, static_cast<uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), int(n) + 1, stream));
    const int rc = ensure(c, c->compactWork, 2 * words + alignUp(scanBytes));
    if (rc != PGPU_OK)
        return rc;
    unsigned char* base = static_cast<unsigned char*>(c->compactWork.ptr);
    uint32_t* flags = reinterpret_cast<uint32_t*>(base);
    uint32_t* places = reinterpret_cast<uint32_t*>(base + words);
    hipLaunchKernelGGL(liveFlagsKernel, dim3(gridFor(c, n + 1)), dim3(256), 0, stream, rays_in->tnear, rays_in->tfar, uint32_t(n), flags);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(base + 2 * words, scanBytes, flags, places, int(n) + 1, stream));
    CompactArgs a {};
    a.in = *rays_in;
    a.pin = *paths_in;
    a.out = *rays_out;
    a.pout = *paths_out;
    a.flags = flags;
    a.places = places;
    a.n = uint32_t(n);
    hipLaunchKernelGGL(compactPathsKernel, dim3(gridFor(c, n)), dim3(256), 0, stream, a);
    HIP_TRY(c, hipGetLastError());
    uint32_t live = 0;
    HIP_TRY(c, hipMemcpyAsync(&live, places + n, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    if (live > n)
        return fail(c, PGPU_ERR_HIP, "pgpu_compact_paths: more live paths than paths (internal error)");
    *n_live = live;
    return PGPU_OK;
}
}
