// C ABI of the leaf-sharded loop's device-side record handling (pandora_amd/sharded.py drives it between all-to-alls).
#include "context.h"
#include <algorithm>
#include <cstring>
#include <vector>

using namespace pandora_hip;

namespace {

uint32_t gridFor(const pgpu_ctx* c, size_t n) { return uint32_t(std::max<size_t>(1, std::min<size_t>((n + 255) / 256, size_t(c->numCUs) * 8))); }

} // namespace

namespace pandora_hip {

// scratch of traceLeafRecords for k records: SoA rays 8k, t/u/v 3k, obj/leaf/any 3k, workLeaf, sorted workRay/workBvh/workLeaf 3k, inst,
// per-leaf count/start/fill, total
struct LeafRecordsLayout {
    size_t oSoa, oT, oU, oV, oObj, oLeaf, oAny, oWl, oSr, oSb, oSl, oInst, oCnt, oStart, oFill, oTotal, bytes;
};
static LeafRecordsLayout leafRecordsLayout(const pgpu_ctx* c, size_t k)
{
    const size_t nl = c->top.n_leaves;
    LeafRecordsLayout L {};
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += alignUp(bytes);
        return at;
    };
    L.oSoa = carve(k * 32), L.oT = carve(k * 4), L.oU = carve(k * 4), L.oV = carve(k * 4), L.oObj = carve(k * 4), L.oLeaf = carve(k * 4), L.oAny = carve(k * 4);
    L.oWl = carve(k * 4), L.oSr = carve(k * 4), L.oSb = carve(k * 4), L.oSl = carve(k * 4), L.oInst = carve(k * 4);
    L.oCnt = carve(nl * 4), L.oStart = carve(nl * 4), L.oFill = carve(nl * 4), L.oTotal = carve(256);
    L.bytes = off;
    return L;
}

int reserveLeafRecords(pgpu_ctx* c, size_t k, DeviceBuffer* spill)
{
    int rc = ensure(c, c->routeWork, leafRecordsLayout(c, std::max<size_t>(k, 1)).bytes);
    if (rc != PGPU_OK)
        return rc;
    Bvh8TraceArgs ba {};
    ba.work_ray = reinterpret_cast<const uint32_t*>(c->routeWork.ptr); // (a work list: selects the layout the real launch will use)
    ba.chunk = c->chunk;
    uint32_t blocks = 1;
    return planBvh8Launch(c, deepestBvh(c), std::max<size_t>(k, 1), ba, &blocks, spill);
}

// The owner side of one round: k received records through the batching points they name -> k results, same order.
// (Arguments are validated by the callers.)
int traceLeafRecords(pgpu_ctx* c, int kind, int tri_mode, const float* records, size_t k, float* results, hipStream_t stream, bool instRows, uint32_t* cursor,
    DeviceBuffer* spill)
{
    const size_t nl = c->top.n_leaves;
    const LeafRecordsLayout L = leafRecordsLayout(c, k);
    const size_t off = L.bytes, oSoa = L.oSoa, oT = L.oT, oU = L.oU, oV = L.oV, oObj = L.oObj, oLeaf = L.oLeaf, oAny = L.oAny, oWl = L.oWl, oSr = L.oSr,
                 oSb = L.oSb, oSl = L.oSl, oInst = L.oInst, oCnt = L.oCnt, oStart = L.oStart, oFill = L.oFill, oTotal = L.oTotal;
    if (!cursor)
        cursor = c->cursor;
    int rc = ensure(c, c->routeWork, off);
    if (rc != PGPU_OK)
        return rc;
    unsigned char* wb = static_cast<unsigned char*>(c->routeWork.ptr);
    float* soa = reinterpret_cast<float*>(wb + oSoa);
    float *t = reinterpret_cast<float*>(wb + oT), *u = reinterpret_cast<float*>(wb + oU), *v = reinterpret_cast<float*>(wb + oV);
    uint32_t *obj = reinterpret_cast<uint32_t*>(wb + oObj), *leaf = reinterpret_cast<uint32_t*>(wb + oLeaf), *any = reinterpret_cast<uint32_t*>(wb + oAny);
    uint32_t *wl = reinterpret_cast<uint32_t*>(wb + oWl), *sr = reinterpret_cast<uint32_t*>(wb + oSr), *sb = reinterpret_cast<uint32_t*>(wb + oSb),
             *sl = reinterpret_cast<uint32_t*>(wb + oSl), *inst = reinterpret_cast<uint32_t*>(wb + oInst);
    uint32_t *cnt = reinterpret_cast<uint32_t*>(wb + oCnt), *start = reinterpret_cast<uint32_t*>(wb + oStart), *fill = reinterpret_cast<uint32_t*>(wb + oFill),
             *total = reinterpret_cast<uint32_t*>(wb + oTotal);
    {
        const int rt = syncBvhTable(c);
        if (rt != PGPU_OK)
            return rt;
    }
    const uint32_t blocks = gridFor(c, k);
    const bool closest = kind == PGPU_CLOSEST;
    HIP_TRY(c, launchUnpackRecords(records, uint32_t(k), soa, wl, closest ? t : nullptr, obj, closest ? nullptr : any, blocks, stream));
    // sort the work by batching point so that the rays of one leaf are adjacent (histogram, scan, scatter)
    HIP_TRY(c, hipMemsetAsync(cnt, 0, nl * 4, stream));
    HIP_TRY(c, launchKeyCount(wl, uint32_t(k), uint32_t(nl), cnt, blocks, stream));
    HIP_TRY(c, launchLeafScan(cnt, start, fill, uint32_t(nl), total, stream));
    Bvh8TraceArgs ba {};
    ba.bvhs = c->bvhTable;
    ba.work_ray = sr;
    ba.work_bvh = sb;
    ba.work_tag = sl;
    ba.n_work = uint32_t(k); // (work items whose leaf id was out of range were not binned: the tail of the list is skipped below)
    ba.ox = soa;
    ba.oy = soa + k;
    ba.oz = soa + 2 * k;
    ba.dx = soa + 3 * k;
    ba.dy = soa + 4 * k;
    ba.dz = soa + 5 * k;
    ba.tnear = soa + 6 * k;
    ba.tfar_in = closest ? t : soa + 7 * k;
    ba.t_out = t;
    ba.obj_out = obj;
    ba.leaf_out = leaf;
    ba.u_out = u;
    ba.v_out = v;
    ba.any_out = any;
    if (instRows && closest) {
        ba.inst_out = inst; // (written by hits only: a miss row's value is never read)
        HIP_TRY(c, hipMemsetAsync(inst, 0xFF, k * 4, stream));
    }
    ba.cursor = cursor;
    ba.chunk = c->chunk;
    ba.init_miss = 0;
    ba.stats = c->stats ? c->statsDev : nullptr;
    const uint32_t deepest = deepestBvh(c);
    // every record names a valid leaf when it comes from pgpu_route_pack; a foreign record with a bad leaf id would leave
    // a hole at the end of the sorted list, so mark the tail as "no bvh" first
    HIP_TRY(c, hipMemsetAsync(sb, 0xFF, k * 4, stream));
    HIP_TRY(c, hipMemsetAsync(sr, 0, k * 4, stream));
    HIP_TRY(c, launchBinByLeaf(nullptr, wl, uint32_t(k), uint32_t(nl), start, fill, c->top.leaves, sr, sb, sl, blocks, stream));
    if (c->nStaged > 0) {
        // some batching points are staged in host memory: the sorted list is traced in launch groups of what fits the HBM cache
        // together (top_abi.hip); the per-point counts cut the list, so the host has to see them
        std::vector<uint32_t> hCount(nl), hStart(nl), order;
        HIP_TRY(c, hipMemcpyAsync(hCount.data(), cnt, nl * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        uint32_t run = 0;
        for (size_t l = 0; l < nl; ++l) {
            hStart[l] = run;
            run += hCount[l];
            if (hCount[l] != 0u)
                order.push_back(uint32_t(l));
        }
        rc = traceLeafListStreamed(c, ba, kind, tri_mode, deepest, order, hStart.data(), hCount.data(), cursor, spill, stream);
        if (rc != PGPU_OK)
            return rc;
    } else {
        HIP_TRY(c, hipMemsetAsync(cursor, 0, 256, stream));
        uint32_t nBlocks = 1;
        rc = planBvh8Launch(c, deepest, k, ba, &nBlocks, spill);
        if (rc != PGPU_OK)
            return rc;
        HIP_TRY(c, launchBvh8Trace(ba, kind, tri_mode, c->stats, nBlocks, stream));
    }
    HIP_TRY(c, launchPackResults(uint32_t(k), t, obj, u, v, closest ? nullptr : any, results, blocks, stream, instRows && closest ? inst : nullptr));
    return PGPU_OK;
}

} // namespace pandora_hip

extern "C" {

int pgpu_route_pack(pgpu_ctx* c, const pgpu_rays* rays, const float* t_state, const uint32_t* active, const uint32_t* parked_leaf, size_t n_active,
    const uint32_t* owner_of_leaf, uint32_t world, uint32_t* dest_counts_host, float* records, uint32_t* sent_slot, uint32_t* sent_leaf, void* stream_)
{
    if (!c || !dest_counts_host || (n_active && (!rays || !parked_leaf || !owner_of_leaf || !records || !sent_slot || !sent_leaf)))
        return fail(c, PGPU_ERR_INVALID, "pgpu_route_pack: null argument");
    if (!c->haveTop)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_route_pack: no top-level structure uploaded");
    if (world == 0 || world > routeMaxRanks())
        return fail(c, PGPU_ERR_INVALID, "pgpu_route_pack: between 1 and 64 ranks");
    if (n_active > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_route_pack: batch too large");
    std::memset(dest_counts_host, 0, world * sizeof(uint32_t));
    if (n_active == 0)
        return PGPU_OK;
    const float* rp[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tnear, t_state ? t_state : rays->tfar };
    for (const float* p : rp)
        if (!p)
            return fail(c, PGPU_ERR_INVALID, "pgpu_route_pack: null ray array");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    int rc = ensure(c, c->blockWork, alignUp(n_active * 4) + 1024);
    if (rc != PGPU_OK)
        return rc;
    // the two small counter arrays sit behind the work-item scratch
    uint32_t* dCounts = reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(c->blockWork.ptr) + c->blockWork.bytes - 1024);
    uint32_t* dFill = dCounts + 64;
    HIP_TRY(c, hipMemsetAsync(dCounts, 0, 512, stream));
    const uint32_t blocks = gridFor(c, n_active);
    HIP_TRY(c, launchRouteCount(parked_leaf, uint32_t(n_active), owner_of_leaf, c->top.n_leaves, world, dCounts, blocks, stream));
    HIP_TRY(c, hipMemcpyAsync(dest_counts_host, dCounts, world * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    uint32_t starts[64] = { 0 };
    uint32_t run = 0;
    for (uint32_t d = 0; d < world; ++d) {
        starts[d] = run;
        run += dest_counts_host[d];
    }
    if (run == 0)
        return PGPU_OK;
    HIP_TRY(c, launchRouteScatter(rp, active, parked_leaf, uint32_t(n_active), owner_of_leaf, c->top.n_leaves, world, starts, dFill, records, sent_slot,
                   sent_leaf, blocks, stream));
    return PGPU_OK;
}

int pgpu_trace_leaf_records(pgpu_ctx* c, int kind, int tri_mode, const float* records, size_t k, float* results, void* stream_)
{
    if (!c || (k && (!records || !results)))
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaf_records: null argument");
    if (!c->haveTop)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_leaf_records: no top-level structure uploaded");
    if (kind != PGPU_CLOSEST && kind != PGPU_ANY)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaf_records: kind must be PGPU_CLOSEST or PGPU_ANY");
    if (tri_mode != PGPU_TRI_MT && tri_mode != PGPU_TRI_WATERTIGHT)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaf_records: unknown triangle test");
    if (k > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaf_records: batch too large");
    if (k == 0)
        return PGPU_OK;
    if (c->bvhs.empty())
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_leaf_records: no bottom-level BVH is resident");
    if (c->nStaged > 0)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_leaf_records: this context holds BVHs staged in host memory (pgpu_stage_bvh8); only pgpu_trace streams them");
    if (c->hasInstances)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_leaf_records: this context holds instanced BVHs; instance hits are not routed between ranks yet");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    HIP_TRY(c, hipEventRecord(c->ev0, stream));
    const int rc = traceLeafRecords(c, kind, tri_mode, records, k, results, stream);
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    c->timingStale = false;
    return rc;
}

int pgpu_route_merge(pgpu_ctx* c, int kind, const float* results, const uint32_t* sent_slot, const uint32_t* sent_leaf, size_t k,
    const pgpu_hits* hits, uint32_t* any_hit, void* stream_)
{
    if (!c || (k && (!results || !sent_slot || !sent_leaf)))
        return fail(c, PGPU_ERR_INVALID, "pgpu_route_merge: null argument");
    if (kind != PGPU_CLOSEST && kind != PGPU_ANY)
        return fail(c, PGPU_ERR_INVALID, "pgpu_route_merge: kind must be PGPU_CLOSEST or PGPU_ANY");
    if (kind == PGPU_CLOSEST && (!hits || !hits->t || !hits->obj || !hits->leaf || !hits->u || !hits->v))
        return fail(c, PGPU_ERR_INVALID, "pgpu_route_merge: closest-hit needs t, obj, leaf, u, v arrays");
    if (kind == PGPU_ANY && !any_hit)
        return fail(c, PGPU_ERR_INVALID, "pgpu_route_merge: any-hit needs the any_hit array");
    if (k > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_route_merge: batch too large");
    if (k == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    const bool closest = kind == PGPU_CLOSEST;
    HIP_TRY(c, launchMergeResults(uint32_t(k), results, sent_slot, sent_leaf, closest ? hits->t : nullptr, closest ? hits->obj : nullptr,
                   closest ? hits->leaf : nullptr, closest ? hits->u : nullptr, closest ? hits->v : nullptr, closest ? nullptr : any_hit, gridFor(c, k), stream));
    return PGPU_OK;
}
}
