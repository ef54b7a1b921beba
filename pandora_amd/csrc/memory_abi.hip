// Memory, copies and streams through the C ABI, for hosts that have no HIP of their own: the C++ adapter
// (include/pandora_amd/gpu_acceleration_structure.hpp) stages rays in pinned memory and moves them with these while a trace
// call is in flight on the same context. They keep no state in the context (only the error text on failure).
#include "context.h"

using namespace pandora_hip;

extern "C" {

int pgpu_host_alloc(pgpu_ctx* c, size_t bytes, void** out)
{
    if (!c || !out)
        return PGPU_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return PGPU_OK;
}

int pgpu_host_free(pgpu_ctx* c, void* p)
{
    if (!c)
        return PGPU_ERR_INVALID;
    if (p)
        HIP_TRY(c, hipHostFree(p));
    return PGPU_OK;
}

int pgpu_device_alloc(pgpu_ctx* c, size_t bytes, void** out)
{
    if (!c || !out)
        return PGPU_ERR_INVALID;
    *out = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMalloc(out, bytes ? bytes : 1));
    return PGPU_OK;
}

int pgpu_device_free(pgpu_ctx* c, void* p)
{
    if (!c)
        return PGPU_ERR_INVALID;
    if (p) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipFree(p));
    }
    return PGPU_OK;
}

int pgpu_stream_create(pgpu_ctx* c, void** stream)
{
    if (!c || !stream)
        return PGPU_ERR_INVALID;
    hipStream_t s = nullptr;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); // copies overlap the context's own (blocking) stream
    *stream = s;
    return PGPU_OK;
}

int pgpu_stream_destroy(pgpu_ctx* c, void* stream)
{
    if (!c)
        return PGPU_ERR_INVALID;
    if (stream) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamDestroy(static_cast<hipStream_t>(stream)));
    }
    return PGPU_OK;
}

int pgpu_copy(pgpu_ctx* c, void* dst, const void* src, size_t bytes, void* stream)
{
    if (!c || !stream || (bytes && (!dst || !src)))
        return PGPU_ERR_INVALID;
    if (!bytes)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, static_cast<hipStream_t>(stream)));
    return PGPU_OK;
}

int pgpu_stream_sync(pgpu_ctx* c, void* stream)
{
    if (!c || !stream)
        return PGPU_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PGPU_OK;
}

} // extern "C"
