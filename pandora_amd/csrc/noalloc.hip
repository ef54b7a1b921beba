// pgpu_trace_bvh8_noalloc: the single-level trace for callers that cannot allocate (code under HIP graph capture, tight
// serving loops, harnesses that forbid hipMalloc). It has no counterpart in the reference. Same kernels and same results as
// pgpu_trace_bvh8 with PGPU_MEM_DEVICE; what differs is where the kernels' scratch lives: the work cursors and the overflow
// ("spill") of the per-ray traversal stacks sit in a static device array instead of buffers the context allocates on demand.
//
// `scratch` below is a host-side stand-in for the device array, not a device address: host code never passes the name to a
// kernel or a HIP call. Each GPU holds its own copy of the array at its own address; pgpu_init resolves that address once per
// device (hipGetSymbolAddress with the device current) and host code uses the resolved pointer from then on.
//
// Every call on one GPU shares that GPU's array: two calls in flight on one GPU at the same time race (the same rule the
// header states for a context, here per device). The cursors are cleared on the stream ahead of every launch; the spill area
// needs no clearing (an entry is written before it is read).
#include "context.h"
#include <algorithm>
#include <mutex>

using namespace pandora_hip;

namespace {

constexpr size_t kCursorBytes = 256;               // 8 work cursors, 32 bytes apart
constexpr size_t kSpillBytes = size_t(64) << 20;   // overflow of the per-ray stacks of every resident ray group
constexpr size_t kScratchBytes = kCursorBytes + kSpillBytes;
static_assert(kScratchBytes == PGPU_NOALLOC_SCRATCH_BYTES, "include/pandora_hip.h states the size of the static scratch");

static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
std::mutex g_lock;
unsigned char* g_scratchPtr[kMaxDevices] = {}; // per device: the address of that GPU's copy of `scratch`

} // namespace

namespace pandora_hip {

// Called by pgpu_init with the context's device current.
int initStaticScratch(pgpu_ctx* c)
{
    if (c->device < 0 || c->device >= kMaxDevices)
        return fail(c, PGPU_ERR_INVALID, "pgpu_init: device id beyond the static scratch table (64 devices)");
    std::lock_guard<std::mutex> hold(g_lock);
    if (!g_scratchPtr[c->device]) {
        unsigned char* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(c, PGPU_ERR_HIP, "pgpu_init: hipGetSymbolAddress(scratch)", e);
        g_scratchPtr[c->device] = p;
    }
    c->staticScratch = g_scratchPtr[c->device];
    return PGPU_OK;
}

} // namespace pandora_hip

extern "C" {

int pgpu_trace_bvh8_noalloc(pgpu_ctx* c, uint32_t bvh_id, int kind, int tri_mode, const pgpu_rays* rays, size_t n, const pgpu_hits* hits,
    uint32_t* any_hit, void* stream_)
{
    if (!c || !rays)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8_noalloc: null argument");
    if (bvh_id >= c->bvhs.size())
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_bvh8_noalloc: unknown bvh id");
    if (bvh_id < c->staged.size() && c->staged[bvh_id].host)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_bvh8_noalloc: this BVH is staged in host memory (pgpu_stage_bvh8); only pgpu_trace streams it");
    if (kind != PGPU_CLOSEST && kind != PGPU_ANY)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8_noalloc: kind must be PGPU_CLOSEST or PGPU_ANY");
    if (tri_mode != PGPU_TRI_MT && tri_mode != PGPU_TRI_WATERTIGHT)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8_noalloc: unknown triangle test");
    if (kind == PGPU_CLOSEST && (!hits || !hits->t || !hits->obj || !hits->leaf || !hits->u || !hits->v))
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8_noalloc: closest-hit needs t, obj, leaf, u, v arrays");
    if (kind == PGPU_ANY && !any_hit)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8_noalloc: any-hit needs the any_hit array");
    if (n > PGPU_NOALLOC_MAX_RAYS)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8_noalloc: more than PGPU_NOALLOC_MAX_RAYS (2^30) rays in one batch");
    const float* rp[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tnear, rays->tfar };
    for (const float* p : rp)
        if (!p && n)
            return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8_noalloc: null ray array");
    if (!c->staticScratch)
        return fail(c, PGPU_ERR_HIP, "pgpu_trace_bvh8_noalloc: the static scratch was not resolved at pgpu_init");
    if (c->bvhTableDirty || !c->bvhTable)
        return fail(c, PGPU_ERR_NO_SCENE,
            "pgpu_trace_bvh8_noalloc: the device BVH table is out of date (a BVH was staged or the leaf cache was reset since the last trace); "
            "bringing it up to date may allocate -- call pgpu_trace_bvh8 or pgpu_trace once first");
    if (n == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;

    unsigned char* const base = c->staticScratch; // this device's copy of `scratch`
    Bvh8TraceArgs a {};
    a.bvhs = c->bvhTable;
    a.fixed_bvh = bvh_id;
    a.n_work = uint32_t(n);
    a.ox = rp[0];
    a.oy = rp[1];
    a.oz = rp[2];
    a.dx = rp[3];
    a.dy = rp[4];
    a.dz = rp[5];
    a.tnear = rp[6];
    a.tfar_in = rp[7];
    if (kind == PGPU_CLOSEST) {
        a.t_out = hits->t;
        a.obj_out = hits->obj;
        a.leaf_out = hits->leaf;
        a.u_out = hits->u;
        a.v_out = hits->v;
        a.inst_out = hits->inst;
    }
    a.any_out = any_hit;
    a.cursor = reinterpret_cast<uint32_t*>(base);
    a.chunk = c->chunk;
    a.init_miss = 1;
    a.stats = c->stats ? c->statsDev : nullptr;

    // The launch plan of pgpu_trace_bvh8, with the grid clamped to what the static spill area serves: every resident ray
    // group owns spill_entries 8-byte slots of it.
    uint32_t blocks = 1;
    planBvh8LaunchShape(c, c->bvhs[bvh_id].depth, n, a, &blocks);
    if (a.spill_entries) {
        const size_t perBlock = size_t(256u / a.lanes_per_ray) * a.spill_entries * 8u;
        const size_t fit = kSpillBytes / perBlock;
        if (fit == 0)
            return fail(c, PGPU_ERR_DEPTH, "pgpu_trace_bvh8_noalloc: the tree's traversal stack does not fit the static scratch");
        blocks = uint32_t(std::min<size_t>(blocks, fit));
        a.spill = reinterpret_cast<unsigned long long*>(base + kCursorBytes);
    }
    HIP_TRY(c, hipMemsetAsync(base, 0, kCursorBytes, stream));
    if (a.inst_out && !c->hasInstances)
        HIP_TRY(c, hipMemsetAsync(a.inst_out, 0xFF, n * 4, stream));
    if (c->stats)
        HIP_TRY(c, hipMemsetAsync(c->statsDev, 0, 64, stream));
    // Timing events only outside stream capture: an event recorded into a graph cannot be timed, and pgpu_last_kernel_ms would
    // keep a stale figure -- under capture it reads 0 (no timing available there).
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess)
        capturing = hipStreamCaptureStatusNone;
    const bool timed = capturing == hipStreamCaptureStatusNone;
    if (timed)
        HIP_TRY(c, hipEventRecord(c->ev0, stream));
    else
        c->lastKernelMs = 0.0f;
    c->timingStale = !timed; // (pgpu_sync must not refresh the figure from events an earlier call recorded)
    HIP_TRY(c, launchBvh8Trace(a, kind, tri_mode, c->stats, blocks, stream));
    if (timed)
        HIP_TRY(c, hipEventRecord(c->ev1, stream));
    return PGPU_OK;
}
}
