// Two-level path (top-level PauseableBVH4 + SVDAG early-out + per-leaf BVH8): placeholder entry
// points until top_kernels.hip lands; they fail loudly instead of pretending.
#include "context.h"
extern "C" {
int pgpu_upload_svdags(pgpu_ctx* c, const uint32_t*, size_t, const pgpu_svdag_desc*, size_t)
{
    if (c) c->err = "pgpu_upload_svdags: not built yet";
    return PGPU_ERR_NO_SCENE;
}
int pgpu_upload_top_bvh4(pgpu_ctx* c, const void*, size_t, uint32_t, const pgpu_leaf*, size_t)
{
    if (c) c->err = "pgpu_upload_top_bvh4: not built yet";
    return PGPU_ERR_NO_SCENE;
}
int pgpu_trace(pgpu_ctx* c, int, int, const pgpu_rays*, size_t, int, const pgpu_hits*, uint32_t*, pgpu_counters*, void*)
{
    if (c) c->err = "pgpu_trace: not built yet";
    return PGPU_ERR_NO_SCENE;
}
}
