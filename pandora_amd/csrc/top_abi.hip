// Two-level entry points of the C ABI: SVDAG / top-level upload and the wavefront loop
//   top-level episode -> bin parked rays by leaf -> per-leaf bottom-level traversal -> resume
// that replaces the reference's TaskGraph flush loop (task_graph.cpp:19-56) around
// OfflineBatchingAccelerationStructure (offline_batching_acceleration_structure.h:158-260,406-437).
#include "context.h"
#include "trace_ranges.h"
#include <limits>
#include <cmath>
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <thread>

using namespace pandora_hip;

namespace {

template <typename T>
int uploadOwned(pgpu_ctx* c, const T* host, size_t n, T** dev)
{
    void* p = nullptr;
    HIP_TRY(c, hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16)));
    c->owned.push_back(p);
    if (n)
        HIP_TRY(c, hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<T*>(p);
    return PGPU_OK;
}

// Builds the (empty) HBM cache for staged BVHs: one arena of pgpu_set_leaf_cache bytes (default: everything staged fits), used
// as a ring -- images of different sizes are written one behind the other, the oldest make room.
int ensureLeafCache(pgpu_ctx* c)
{
    if (c->nStaged == 0)
        return PGPU_OK;
    // (per-point arrays the host exchanges with the device every round: sized by the top level, which may have been replaced
    // since the cache was built)
    const size_t words = 3 * size_t(c->top.n_leaves) + 64;
    if (words > c->leafHostCap) {
        HIP_TRY(c, hipDeviceSynchronize());
        if (c->leafHostBuf)
            HIP_TRY(c, hipHostFree(c->leafHostBuf));
        c->leafHostBuf = nullptr;
        c->leafHostCap = 0;
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->leafHostBuf), words * 4));
        c->leafHostCap = words;
    }
    if (c->cacheArena)
        return PGPU_OK;
    size_t largest = 0, total = 0;
    for (const StagedBvh& s : c->staged)
        if (s.host) {
            largest = std::max(largest, alignUp(s.imageBytes()));
            total += alignUp(s.imageBytes());
        }
    size_t bytes = std::max<size_t>(total, 256);
    if (c->cacheBudget) {
        if (c->cacheBudget < largest)
            return fail(c, PGPU_ERR_INVALID, "pgpu_trace: the leaf cache (pgpu_set_leaf_cache) is smaller than the largest staged BVH");
        bytes = std::min(bytes, c->cacheBudget & ~size_t(255));
    }
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->cacheArena), bytes));
    c->cacheBytes = bytes;
    c->ringWrite = 0;
    c->ring.clear();
    if (c->launchEvents.empty()) {
        c->launchEvents.resize(64);
        for (hipEvent_t& e : c->launchEvents)
            HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    if (!c->copyStream)
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copyStream, hipStreamNonBlocking));
    if (!c->copyDone)
        HIP_TRY(c, hipEventCreateWithFlags(&c->copyDone, hipEventDisableTiming));
    // room for the largest possible launch group (every staged BVH arrives and every one leaves), twice
    const size_t cap = 4 * c->nStaged + 64;
    if (cap > c->copyTableCap) {
        HIP_TRY(c, hipDeviceSynchronize());
        if (c->copyTableHost)
            HIP_TRY(c, hipHostFree(c->copyTableHost));
        if (c->copyTableDev)
            HIP_TRY(c, hipFree(c->copyTableDev));
        c->copyTableHost = c->copyTableDev = nullptr;
        c->copyTableCap = 0;
        HIP_TRY(c, hipHostMalloc(&c->copyTableHost, cap * 64));
        HIP_TRY(c, hipMalloc(&c->copyTableDev, cap * 64));
        c->copyTableCap = cap;
        c->copyTableNext = 0;
    }
    return PGPU_OK;
}

// What one launch group of the streaming path brings in and throws out.
struct LeafGroupPlan {
    std::vector<uint32_t> arrived, evicted, touched;
    std::vector<int> waitLaunches; // launch events the copies must wait for: the last readers of the space being overwritten
};

// Gives staged BVH `id` a place in the ring for launch group `group` (no-op if it is resident). Returns false -- and changes
// nothing -- when that needs the space of an image the same group uses: the group is complete then.
bool ringPlace(pgpu_ctx* c, uint32_t id, uint64_t group, LeafGroupPlan& plan)
{
    StagedBvh& s = c->staged[id];
    if (s.resident) {
        if (s.pinGroup != group)
            plan.touched.push_back(id);
        s.pinGroup = group;
        return true;
    }
    const size_t bytes = alignUp(s.imageBytes()), cap = c->cacheBytes;
    if (c->ring.empty())
        c->ringWrite = 0;
    const size_t w = c->ringWrite;
    const bool wrap = w + bytes > cap;
    auto overlaps = [&](const StagedBvh& r, size_t lo, size_t hi) { return r.offset < hi && r.offset + alignUp(r.imageBytes()) > lo; };
    // the images in the way are the oldest ones: a prefix of the ring
    size_t victims = 0;
    for (; victims < c->ring.size(); ++victims) {
        const StagedBvh& r = c->staged[c->ring[victims]];
        const bool inTheWay = wrap ? (overlaps(r, w, cap) || overlaps(r, 0, bytes)) : overlaps(r, w, w + bytes);
        if (!inTheWay)
            break;
        if (r.pinGroup == group)
            return false;
    }
    for (size_t k = 0; k < victims; ++k) { // evict (CachedBVH::doEvict): the device table entry is nulled too, so a work item that names it is dropped
        const uint32_t v = c->ring.front();
        c->ring.pop_front();
        StagedBvh& r = c->staged[v];
        r.resident = false;
        if (r.lastLaunch >= 0)
            plan.waitLaunches.push_back(r.lastLaunch);
        r.lastLaunch = -1;
        c->bvhs[v].nodes = nullptr;
        c->bvhs[v].tris = nullptr;
        c->bvhs[v].inst = nullptr;
        plan.evicted.push_back(v);
    }
    s.offset = wrap ? 0 : w;
    c->ringWrite = s.offset + bytes;
    s.resident = true;
    s.pinGroup = group;
    s.lastLaunch = -1;
    unsigned char* dst = c->cacheArena + s.offset;
    c->bvhs[id].nodes = dst;
    c->bvhs[id].tris = dst + s.nodeBytes;
    c->bvhs[id].inst = s.instBytes ? dst + s.nodeBytes + s.triBytes : nullptr;
    c->ring.push_back(id);
    plan.arrived.push_back(id);
    plan.touched.push_back(id);
    c->counters.leaf_loads += 1;
    c->counters.streamed_bytes += s.imageBytes();
    return true;
}

// Queues the copies of a launch group on the copy stream and the changes of the device BVH table on the compute stream (behind
// the copies, and behind the previous launch, which may still read the entries of the images that leave).
int commitLeafGroup(pgpu_ctx* c, const LeafGroupPlan& plan, hipStream_t stream, uint64_t group)
{
    if (plan.arrived.empty() && plan.evicted.empty())
        return PGPU_OK;
    const TraceRange range_("leaf streaming: copy group"); // the reference's static-data load (task_graph.h:249-259, offline_batching...h:165-175)
    for (int ev : plan.waitLaunches)
        HIP_TRY(c, hipStreamWaitEvent(c->copyStream, c->launchEvents[size_t(ev)], 0));
    const size_t nSeg = c->leafCopyMode == 0 ? plan.arrived.size() : 0, nFix = plan.arrived.size() + plan.evicted.size();
    if (c->copyTableNext + nSeg + nFix > c->copyTableCap) { // (the table is reused from its start: everything that read it must be done)
        HIP_TRY(c, hipStreamSynchronize(c->copyStream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        c->copyTableNext = 0;
    }
    unsigned char* const host = static_cast<unsigned char*>(c->copyTableHost) + c->copyTableNext * 64;
    unsigned char* const dev = static_cast<unsigned char*>(c->copyTableDev) + c->copyTableNext * 64;
    c->copyTableNext += nSeg + nFix;
    uint64_t chunks = 0;
    for (size_t i = 0; i < plan.arrived.size(); ++i) {
        const uint32_t id = plan.arrived[i];
        const StagedBvh& s = c->staged[id];
        if (c->leafCopyMode == 0) {
            LeafCopySeg seg {};
            seg.src = s.host;
            seg.dst = c->cacheArena + s.offset;
            seg.bytes = (s.imageBytes() + 15) & ~size_t(15); // (the pinned image and its place in the arena are 256-byte multiples)
            seg.chunk0 = chunks;
            chunks += (seg.bytes + 16383) / 16384;
            std::memcpy(host + i * 64, &seg, 64);
        } else {
            HIP_TRY(c, hipMemcpyAsync(c->cacheArena + s.offset, s.host, s.imageBytes(), hipMemcpyHostToDevice, c->copyStream));
        }
    }
    size_t f = 0;
    auto putFix = [&](uint32_t id) {
        LeafTableFix fix {};
        fix.entry = c->bvhs[id];
        fix.id = id;
        fix.group = c->staged[id].resident ? uint32_t(group) : 0u;
        std::memcpy(host + (nSeg + f++) * 64, &fix, sizeof(fix));
    };
    for (uint32_t id : plan.evicted)
        if (!c->staged[id].resident) // (an image evicted and brought back within one group keeps its new entry)
            putFix(id);
    for (uint32_t id : plan.arrived)
        if (c->staged[id].resident)
            putFix(id);
    HIP_TRY(c, hipMemcpyAsync(dev, host, (nSeg + f) * 64, hipMemcpyHostToDevice, c->copyStream));
    if (nSeg)
        HIP_TRY(c, launchLeafGather(reinterpret_cast<const LeafCopySeg*>(dev), uint32_t(nSeg), uint32_t(chunks), uint32_t(c->numCUs) * 4u, c->copyStream));
    HIP_TRY(c, hipEventRecord(c->copyDone, c->copyStream));
    HIP_TRY(c, hipStreamWaitEvent(stream, c->copyDone, 0));
    uint32_t* const arrivalGroup = c->tableView.ptr ? reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(c->tableView.ptr) + c->bvhTableCap * (sizeof(Bvh8Dev) + 4)) : nullptr;
    HIP_TRY(c, launchLeafTableFix(dev + nSeg * 64, uint32_t(f), c->bvhTable, stream, arrivalGroup));
    return PGPU_OK;
}

// The view of the BVH table for launches that trace only their own group's arrivals (kernels.h launchLeafTableView): the view
// itself, which ids a batching point names, and the group every id arrived with (rebuilt whenever the cache is: nothing is
// resident then).
int prepareTableView(pgpu_ctx* c)
{
    const size_t cap = c->bvhTableCap;
    const size_t bytes = cap * (sizeof(Bvh8Dev) + 8);
    if (c->tableView.bytes >= bytes && c->tableView.ptr)
        return PGPU_OK;
    int rc = ensure(c, c->tableView, bytes);
    if (rc != PGPU_OK)
        return rc;
    return PGPU_OK;
}
int fillTableView(pgpu_ctx* c, hipStream_t stream)
{
    const size_t cap = c->bvhTableCap;
    std::vector<uint32_t> flag(2 * cap, 0u); // [named by a batching point?][arrival group = 0: none]
    for (const TopLeafDev& l : c->topLeavesHost)
        if (l.bvh < cap)
            flag[l.bvh] = 1u;
    // (images that are resident from an earlier trace keep group 0: no arrivals-only launch of this trace brought them in)
    HIP_TRY(c, hipMemcpyAsync(static_cast<unsigned char*>(c->tableView.ptr) + cap * sizeof(Bvh8Dev), flag.data(), flag.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipStreamSynchronize(stream)); // (pageable source)
    return PGPU_OK;
}

} // namespace

namespace pandora_hip {

int prepareLeafCache(pgpu_ctx* c) { return ensureLeafCache(c); }

int prepareLeafTableView(pgpu_ctx* c, hipStream_t stream)
{
    const int rc = prepareTableView(c);
    return rc == PGPU_OK ? fillTableView(c, stream) : rc;
}

// The loops that do not let rays wait (pgpu_trace_sharded: the rays parked at this rank's own points, and the records other
// ranks sent): a point whose queue is worth its image -- a ray per leafHoldBytesPerRay bytes -- is loaded (launch groups of what
// fits the cache together), the rays of the others traverse their image where it lies, in host memory.
int traceLeafListStreamed(pgpu_ctx* c, const Bvh8TraceArgs& ba, int kind, int tri_mode, uint32_t deepest, const std::vector<uint32_t>& points, const uint32_t* leafStart,
    const uint32_t* leafCount, uint32_t* cursor, DeviceBuffer* spill, hipStream_t stream)
{
    std::vector<uint32_t> load, inPlace;
    for (uint32_t l : points) {
        const uint32_t id = c->topLeavesHost[l].bvh;
        const StagedBvh* im = (id < c->staged.size() && c->staged[id].host) ? &c->staged[id] : nullptr;
        const bool sparse = im && !im->resident && c->leafInPlaceBytesPerRay != 0u && c->leafHoldBytesPerRay != 0u && c->tableView.ptr
            && uint64_t(leafCount[l]) * c->leafHoldBytesPerRay < im->imageBytes() && uint64_t(leafCount[l]) * c->leafInPlaceBytesPerRay < im->imageBytes();
        (sparse ? inPlace : load).push_back(l);
    }
    int rc = load.empty() ? PGPU_OK : traceLeafGroups(c, ba, kind, tri_mode, deepest, load, leafStart, leafCount, cursor, spill, stream);
    if (rc == PGPU_OK && !inPlace.empty()) {
        std::sort(inPlace.begin(), inPlace.end());
        rc = traceLeavesInPlace(c, ba, kind, tri_mode, deepest, inPlace, leafStart, leafCount, cursor, spill, stream);
    }
    return rc;
}

// Bottom-level traversal of a work list that is sorted by batching point, through BVHs that may be staged in host memory: the
// points of `visitOrder` (ascending or descending slots) whose images fit the HBM cache together form one launch over the
// contiguous stretch of the list they span; the copies of a group run on the copy stream, behind the last launch that read the
// space they overwrite and ahead of the launch that needs them. Work items of points that are not in the cache (not in
// visitOrder, evicted) are dropped by the kernel: their table entry is null. `ba`: the launch arguments with work_ray / work_bvh /
// work_tag at the list's start.
int traceLeafGroups(pgpu_ctx* c, Bvh8TraceArgs ba, int kind, int tri_mode, uint32_t deepest, const std::vector<uint32_t>& visitOrder, const uint32_t* leafStart,
    const uint32_t* leafCount, uint32_t* cursor, DeviceBuffer* spill, hipStream_t stream, bool arrivalsOnly)
{
    if (arrivalsOnly && !c->tableView.ptr)
        return fail(c, PGPU_ERR_INVALID, "leaf cache: no table view (internal)");
    const uint32_t* const listRay = ba.work_ray;
    const uint32_t* const listBvh = ba.work_bvh;
    const uint32_t* const listTag = ba.work_tag;
    ba.cursor = cursor;
    size_t at = 0;
    while (at < visitOrder.size()) {
        if (uint32_t(c->groupClock + 1) == 0u)
            ++c->groupClock; // (the low 32 bits name the group on the device; 0 means "none")
        const uint64_t group = ++c->groupClock;
        LeafGroupPlan plan;
        uint32_t lo = 0xFFFFFFFFu, hi = 0;
        size_t members = 0;
        for (; at < visitOrder.size(); ++at, ++members) {
            const uint32_t l = visitOrder[at];
            const uint32_t id = c->topLeavesHost[l].bvh;
            if (id < c->staged.size() && c->staged[id].host && !ringPlace(c, id, group, plan))
                break; // the cache is full of BVHs this launch needs: launch, then go on
            lo = std::min(lo, leafStart[l]);
            hi = std::max(hi, leafStart[l] + leafCount[l]);
        }
        if (members == 0)
            return fail(c, PGPU_ERR_INVALID, "leaf cache: no room for a single BVH (internal)");
        int rc = commitLeafGroup(c, plan, stream, group);
        if (rc != PGPU_OK)
            return rc;
        if (arrivalsOnly) {
            Bvh8Dev* const view = static_cast<Bvh8Dev*>(c->tableView.ptr);
            const uint32_t* const leafFlag = reinterpret_cast<const uint32_t*>(static_cast<unsigned char*>(c->tableView.ptr) + c->bvhTableCap * sizeof(Bvh8Dev));
            HIP_TRY(c, launchLeafTableView(c->bvhTable, leafFlag, leafFlag + c->bvhTableCap, uint32_t(group), view, uint32_t(c->bvhs.size()), stream));
            ba.bvhs = view;
        }
        HIP_TRY(c, hipMemsetAsync(cursor, 0, 256, stream));
        ba.work_ray = listRay + lo;
        ba.work_bvh = listBvh + lo;
        ba.work_tag = listTag + lo;
        ba.n_work = hi - lo;
        uint32_t blocks = 1;
        rc = planBvh8Launch(c, deepest, hi - lo, ba, &blocks, spill);
        if (rc != PGPU_OK)
            return rc;
        {
            const TraceRange r_("round: bottom level, one group of cached batching points");
            HIP_TRY(c, launchBvh8Trace(ba, kind, tri_mode, c->stats, blocks, stream));
        }
        if (!plan.touched.empty()) {
            const size_t ev = c->launchEventNext++ % c->launchEvents.size();
            HIP_TRY(c, hipEventRecord(c->launchEvents[ev], stream));
            for (uint32_t id : plan.touched)
                if (c->staged[id].resident)
                    c->staged[id].lastLaunch = int(ev);
        }
    }
    return PGPU_OK;
}

int traceLeavesInPlace(pgpu_ctx* c, Bvh8TraceArgs ba, int kind, int tri_mode, uint32_t deepest, const std::vector<uint32_t>& points, const uint32_t* leafStart,
    const uint32_t* leafCount, uint32_t* cursor, DeviceBuffer* spill, hipStream_t stream)
{
    if (points.empty())
        return PGPU_OK;
    if (!c->tableView.ptr)
        return fail(c, PGPU_ERR_INVALID, "leaf cache: no table view (internal)");
    const TraceRange range_("round: bottom level, sparse batching points traced in host memory");
    if (uint32_t(c->groupClock + 1) == 0u)
        ++c->groupClock;
    const uint64_t group = ++c->groupClock; // (a group nothing arrives with: the view hides every BVH a batching point names)
    const size_t n = points.size();
    if (c->copyTableNext + n > c->copyTableCap) { // (the table is reused from its start: everything that read it must be done)
        HIP_TRY(c, hipStreamSynchronize(c->copyStream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        c->copyTableNext = 0;
    }
    unsigned char* const host = static_cast<unsigned char*>(c->copyTableHost) + c->copyTableNext * 64;
    unsigned char* const dev = static_cast<unsigned char*>(c->copyTableDev) + c->copyTableNext * 64;
    c->copyTableNext += n;
    uint32_t lo = 0xFFFFFFFFu, hi = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t l = points[i];
        const uint32_t id = c->topLeavesHost[l].bvh;
        if (id >= c->staged.size() || !c->staged[id].host || c->staged[id].resident)
            return fail(c, PGPU_ERR_INVALID, "leaf cache: a point traced in place must be staged and not resident (internal)");
        const StagedBvh& s = c->staged[id];
        LeafTableFix fix {};
        fix.entry = c->bvhs[id];
        fix.entry.nodes = s.host; // (pinned memory: the device reads it at the same address, as the gather kernel does)
        fix.entry.tris = s.host + s.nodeBytes;
        fix.entry.inst = s.instBytes ? s.host + s.nodeBytes + s.triBytes : nullptr;
        fix.id = id;
        fix.group = 0u;
        std::memcpy(host + i * 64, &fix, sizeof(fix));
        lo = std::min(lo, leafStart[l]);
        hi = std::max(hi, leafStart[l] + leafCount[l]);
    }
    HIP_TRY(c, hipMemcpyAsync(dev, host, n * 64, hipMemcpyHostToDevice, stream));
    Bvh8Dev* const view = static_cast<Bvh8Dev*>(c->tableView.ptr);
    const uint32_t* const leafFlag = reinterpret_cast<const uint32_t*>(static_cast<unsigned char*>(c->tableView.ptr) + c->bvhTableCap * sizeof(Bvh8Dev));
    HIP_TRY(c, launchLeafTableView(c->bvhTable, leafFlag, leafFlag + c->bvhTableCap, uint32_t(group), view, uint32_t(c->bvhs.size()), stream));
    HIP_TRY(c, launchLeafTableFix(dev, uint32_t(n), view, stream, nullptr));
    HIP_TRY(c, hipMemsetAsync(cursor, 0, 256, stream));
    ba.cursor = cursor;
    ba.bvhs = view;
    ba.work_ray += lo;
    ba.work_bvh += lo;
    ba.work_tag += lo;
    ba.n_work = hi - lo;
    uint32_t blocks = 1;
    const int rc = planBvh8Launch(c, deepest, hi - lo, ba, &blocks, spill);
    if (rc != PGPU_OK)
        return rc;
    HIP_TRY(c, launchBvh8Trace(ba, kind, tri_mode, c->stats, blocks, stream));
    return PGPU_OK;
}

} // namespace pandora_hip

// one record per batching point with its DAG's descriptor inlined (LeafDagDev); needs the leaves, and the DAGs they name
static int rebuildLeafDags(pgpu_ctx* c)
{
    if (c->topLeavesHost.empty())
        return PGPU_OK;
    std::vector<LeafDagDev> ld(c->topLeavesHost.size());
    for (size_t i = 0; i < ld.size(); ++i) {
        LeafDagDev& r = ld[i];
        std::memset(&r, 0, sizeof(r));
        const uint32_t s = c->topLeavesHost[i].svdag;
        // pgpu_set_svdag_policy(ctx, 1): the proxy test exists to save a load; a point whose BVH sits in this GPU's
        // HBM is cheaper to visit than to test
        const uint32_t b = c->topLeavesHost[i].bvh;
        const bool costly = b == PGPU_NO_BVH || (b < c->staged.size() && c->staged[b].host != nullptr);
        if (s != PGPU_MISS_ID && s < c->dagsHost.size() && (c->svdagPolicy == 0 || costly)) {
            const SvdagDev& d = c->dagsHost[s];
            r.root = d.root;
            for (int a = 0; a < 3; ++a)
                r.bmin[a] = d.bmin[a];
            r.invExtent = d.invExtent;
            r.levels = d.levels;
            r.has_dag = 1u;
        }
    }
    LeafDagDev* dl = nullptr;
    const int rc = uploadOwned(c, ld.data(), ld.size(), &dl);
    if (rc != PGPU_OK)
        return rc;
    // the previous table (every policy change and every upload builds a new one) is released once nothing queued can read it
    if (c->top.leafDags) {
        void* const old = const_cast<LeafDagDev*>(c->top.leafDags);
        HIP_TRY(c, hipDeviceSynchronize());
        auto it = std::find(c->owned.begin(), c->owned.end(), old);
        if (it != c->owned.end()) {
            c->owned.erase(it);
            HIP_TRY(c, hipFree(old));
        }
    }
    c->top.leafDags = dl;
    return PGPU_OK;
}

extern "C" {

int pgpu_set_top_variant(pgpu_ctx* c, int variant, int refill_min, int cast_trips)
{
    if (!c || variant < 0 || variant > 4 || refill_min < 0 || refill_min > 64 || cast_trips < 0 || cast_trips > 64)
        return PGPU_ERR_INVALID;
    c->topVariant = variant;
    c->topRefillMin = refill_min ? uint32_t(refill_min) : 16u;
    c->topCastTrips = cast_trips ? uint32_t(cast_trips) : 4u;
    return PGPU_OK;
}

int pgpu_set_svdag_policy(pgpu_ctx* c, int policy)
{
    if (!c || policy < 0 || policy > 1)
        return fail(c, PGPU_ERR_INVALID, "pgpu_set_svdag_policy: 0 (test every batching point that has an SVDAG) or 1 (only points whose geometry is not resident here)");
    HIP_TRY(c, hipSetDevice(c->device));
    c->svdagPolicy = policy;
    return rebuildLeafDags(c);
}

int pgpu_set_svdag_defer(pgpu_ctx* c, uint32_t min_rays, uint32_t passes)
{
    if (!c)
        return PGPU_ERR_INVALID;
    if (min_rays != 0u && (passes < 1u || passes > 8u))
        return fail(c, PGPU_ERR_INVALID, "pgpu_set_svdag_defer: passes must be 1..8");
    c->deferMin = min_rays;
    if (min_rays != 0u)
        c->deferPasses = passes;
    return PGPU_OK;
}

int pgpu_upload_svdags(pgpu_ctx* c, const uint32_t* words, size_t n_words, const pgpu_svdag_desc* dags, size_t n_dags)
{
    const TraceRange range_("pgpu_upload_svdags");
    if (!c || !words || !dags || n_words == 0 || n_dags == 0)
        return fail(c, PGPU_ERR_INVALID, "pgpu_upload_svdags: null or empty argument");
    HIP_TRY(c, hipSetDevice(c->device));
    // every offset stored in the array must stay inside it (nodes are read unchecked on the device)
    for (size_t d = 0; d < n_dags; ++d)
        if (dags[d].root >= n_words)
            return fail(c, PGPU_ERR_INVALID, "pgpu_upload_svdags: root offset outside the node array");
    // Walk every node reachable from the roots once: offsets must stay inside the array, the graph must be
    // acyclic, and the height of the tallest DAG sizes the cast's saved-parent stack in LDS.
    std::vector<uint8_t> state(n_words, 0), height(n_words, 0); // state: 0 new, 1 on the walk's path, 2 done
    std::vector<std::pair<uint32_t, uint32_t>> path;            // (node, next child to look at)
    uint32_t levels = 1;
    for (size_t d = 0; d < n_dags; ++d) {
        if (state[dags[d].root] == 0)
            path.emplace_back(dags[d].root, 0u);
        while (!path.empty()) {
            const uint32_t at = path.back().first;
            const uint32_t desc = words[at], inner = ((desc >> 8) & 0xFFu) & ~(desc & 0xFFu);
            const uint32_t nKids = uint32_t(__builtin_popcount(inner));
            if (path.back().second == 0) {
                state[at] = 1;
                if (nKids && size_t(at) + nKids >= n_words)
                    return fail(c, PGPU_ERR_INVALID, "pgpu_upload_svdags: child table runs past the node array");
            }
            if (path.back().second < nKids) {
                const uint32_t kid = words[at + 1 + path.back().second++];
                if (kid >= n_words)
                    return fail(c, PGPU_ERR_INVALID, "pgpu_upload_svdags: child offset outside the node array");
                if (state[kid] == 1)
                    return fail(c, PGPU_ERR_INVALID, "pgpu_upload_svdags: the node graph has a cycle");
                if (state[kid] == 0)
                    path.emplace_back(kid, 0u);
                continue;
            }
            uint32_t h = 0;
            for (uint32_t k = 0; k < nKids; ++k)
                h = std::max<uint32_t>(h, height[words[at + 1 + k]]);
            if (h + 1 > 23)
                return fail(c, PGPU_ERR_DEPTH, "pgpu_upload_svdags: DAG taller than CAST_STACK_DEPTH (23)");
            height[at] = uint8_t(h + 1);
            state[at] = 2;
            path.pop_back();
        }
        levels = std::max<uint32_t>(levels, height[dags[d].root]);
    }
    std::vector<SvdagDev> dev(n_dags);
    // Per-DAG images for the cast kernel (TopDev::dagImages): breadth first from the root, local offsets, at most
    // kDagImageMaxWords words; what does not fit is referred to in the shared array.
    std::vector<uint32_t> images, order;
    std::vector<int32_t> localOf(n_words, -1);
    uint32_t imgMax = 0;
    for (size_t d = 0; d < n_dags; ++d) {
        dev[d].root = dags[d].root;
        for (int a = 0; a < 3; ++a)
            dev[d].bmin[a] = dags[d].bounds_min[a];
        dev[d].invExtent = dags[d].inv_extent;
        dev[d].levels = height[dags[d].root];
        order.clear();
        order.push_back(dags[d].root);
        localOf[dags[d].root] = 0;
        uint32_t used = 0;
        auto nodeWords = [&](uint32_t at) {
            const uint32_t desc = words[at];
            return 1u + uint32_t(__builtin_popcount(((desc >> 8) & 0xFFu) & ~(desc & 0xFFu)));
        };
        used = nodeWords(dags[d].root);
        if (used > kDagImageMaxWords) { // (cannot happen: a node has at most 9 words)
            localOf[dags[d].root] = -1;
            order.clear();
            used = 0;
        }
        for (size_t i = 0; i < order.size(); ++i) {
            const uint32_t at = order[i], nk = nodeWords(at) - 1u;
            for (uint32_t k = 0; k < nk; ++k) {
                const uint32_t kid = words[at + 1 + k];
                if (localOf[kid] >= 0)
                    continue;
                const uint32_t sz = nodeWords(kid);
                if (used + sz > kDagImageMaxWords)
                    continue; // stays in the shared array
                localOf[kid] = int32_t(used);
                used += sz;
                order.push_back(kid);
            }
        }
        dev[d].img_offset = uint32_t(images.size());
        dev[d].img_words = used;
        images.resize(images.size() + used);
        uint32_t* img = images.data() + dev[d].img_offset;
        for (uint32_t at : order) {
            const uint32_t base = uint32_t(localOf[at]), nk = nodeWords(at) - 1u;
            img[base] = words[at];
            for (uint32_t k = 0; k < nk; ++k) {
                const uint32_t kid = words[at + 1 + k];
                img[base + 1 + k] = localOf[kid] >= 0 ? uint32_t(localOf[kid]) : (kid | kDagGlobalFlag);
            }
        }
        for (uint32_t at : order)
            localOf[at] = -1;
        imgMax = std::max(imgMax, used);
    }
    if (n_words >= kDagGlobalFlag)
        return fail(c, PGPU_ERR_INVALID, "pgpu_upload_svdags: more than 2^31 words");
    uint32_t *dw = nullptr, *di = nullptr;
    SvdagDev* dd = nullptr;
    int rc = uploadOwned(c, words, n_words, &dw);
    if (rc != PGPU_OK)
        return rc;
    rc = uploadOwned(c, images.data(), images.size(), &di);
    if (rc != PGPU_OK)
        return rc;
    c->top.dagImages = di;
    c->top.img_max_words = imgMax;
    rc = uploadOwned(c, dev.data(), n_dags, &dd);
    if (rc != PGPU_OK)
        return rc;
    c->top.dagWords = dw;
    c->top.dags = dd;
    c->top.n_dags = uint32_t(n_dags);
    c->top.dag_levels = levels;
    c->dagsHost = dev;
    return rebuildLeafDags(c);
}

int pgpu_upload_top_bvh4(pgpu_ctx* c, const void* nodes112, size_t n_nodes, uint32_t root, const pgpu_leaf* leaves, size_t n_leaves)
{
    const TraceRange range_("pgpu_upload_top_bvh4");
    if (!c || !nodes112 || !leaves || n_nodes == 0 || n_leaves == 0 || root >= n_nodes)
        return fail(c, PGPU_ERR_INVALID, "pgpu_upload_top_bvh4: null or empty argument");
    if (n_leaves < 2)
        return fail(c, PGPU_ERR_SINGLE_LEAF, "pgpu_upload_top_bvh4: a single batching point is not supported (pauseable_bvh4.h:180)");
    HIP_TRY(c, hipSetDevice(c->device));
    const unsigned char* nb = static_cast<const unsigned char*>(nodes112);
    for (size_t i = 0; i < n_nodes; ++i) {
        const uint32_t* u = reinterpret_cast<const uint32_t*>(nb + i * 112);
        const uint32_t bits = u[27], valid = bits & 0xFu, leaf = (bits >> 4) & 0xFu, depth = (bits >> 8) & 0xFFu;
        if (depth > 15)
            return fail(c, PGPU_ERR_DEPTH, "pgpu_upload_top_bvh4: node deeper than 15 (64-bit traversal stack)");
        const uint32_t nInner = uint32_t(__builtin_popcount(valid & ~leaf)), nLeaf = uint32_t(__builtin_popcount(valid & leaf));
        if ((nInner && size_t(u[24]) + nInner > n_nodes) || (nLeaf && size_t(u[25]) + nLeaf > n_leaves) || u[26] >= n_nodes)
            return fail(c, PGPU_ERR_INVALID, "pgpu_upload_top_bvh4: child, leaf or parent handle out of range");
    }
    std::vector<TopLeafDev> lv(n_leaves);
    bool remote = false;
    for (size_t i = 0; i < n_leaves; ++i) {
        if (leaves[i].bvh_id != PGPU_NO_BVH && leaves[i].bvh_id >= c->bvhs.size())
            return fail(c, PGPU_ERR_INVALID, "pgpu_upload_top_bvh4: leaf names an unknown bvh id");
        remote = remote || leaves[i].bvh_id == PGPU_NO_BVH;
        if (leaves[i].svdag != PGPU_NO_SVDAG && leaves[i].svdag >= c->top.n_dags)
            return fail(c, PGPU_ERR_INVALID, "pgpu_upload_top_bvh4: leaf names an unknown svdag");
        lv[i].bvh = leaves[i].bvh_id;
        lv[i].svdag = leaves[i].svdag;
    }
    unsigned char* dn = nullptr;
    TopLeafDev* dl = nullptr;
    int rc = uploadOwned(c, nb, n_nodes * 112, &dn);
    if (rc != PGPU_OK)
        return rc;
    rc = uploadOwned(c, lv.data(), n_leaves, &dl);
    if (rc != PGPU_OK)
        return rc;
    c->top.nodes = dn;
    c->top.leaves = dl;
    c->top.root = root;
    c->top.n_nodes = uint32_t(n_nodes);
    c->top.n_leaves = uint32_t(n_leaves);
    c->haveTop = true;
    c->topLeavesHost = lv;
    c->topHasRemoteLeaves = remote;
    return rebuildLeafDags(c);
}

// One batch through the round loop on one context (pgpu_trace below runs a large batch as several of these side by side).
static int traceBatch(pgpu_ctx* c, int kind, int tri_mode, const pgpu_rays* rays, size_t n, int mem,
    const pgpu_hits* hits, uint32_t* any_hit, pgpu_counters* counters, void* stream_)
{
    const TraceRange range_("pgpu_trace");
    if (!c || !rays)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace: null argument");
    if (!c->haveTop)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace: no top-level structure uploaded");
    if (c->topHasRemoteLeaves)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace: some batching points are not resident on this GPU (PGPU_NO_BVH); drive pgpu_top_episode / pgpu_trace_leaves from a sharded loop");
    if (kind != PGPU_CLOSEST && kind != PGPU_ANY)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace: kind must be PGPU_CLOSEST or PGPU_ANY");
    if (tri_mode != PGPU_TRI_MT && tri_mode != PGPU_TRI_WATERTIGHT)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace: unknown triangle test");
    if (mem != PGPU_MEM_HOST && mem != PGPU_MEM_DEVICE)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace: mem must be PGPU_MEM_HOST or PGPU_MEM_DEVICE");
    if (kind == PGPU_CLOSEST && (!hits || !hits->t || !hits->obj || !hits->leaf || !hits->u || !hits->v))
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace: closest-hit needs t, obj, leaf, u, v arrays");
    if (kind == PGPU_ANY && !any_hit)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace: any-hit needs the any_hit array");
    if (n > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace: batch too large");
    const float* rp[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tnear, rays->tfar };
    for (const float* p : rp)
        if (!p && n)
            return fail(c, PGPU_ERR_INVALID, "pgpu_trace: null ray array");
    if (counters)
        std::memset(counters, 0, sizeof(*counters));
    if (n == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rt = syncBvhTable(c);
        if (rt != PGPU_OK)
            return rt;
    }
    hipStream_t stream = (mem == PGPU_MEM_DEVICE && stream_) ? static_cast<hipStream_t>(stream_) : c->stream;
    const bool wantVisits = hits && hits->visits;

    // ---- carve the work buffer
    const size_t nl = c->top.n_leaves;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += alignUp(bytes);
        return at;
    };
    const size_t oNode = carve(n * 4), oStack = carve(n * 8), oWorkRayAlt = carve(n * 4), oParkedLeaf = carve(n * 4);
    const size_t oWorkRay = carve(n * 4), oWorkBvh = carve(n * 4), oWorkLeaf = carve(n * 4);
    const size_t oLeafCount = carve(nl * 4), oLeafStart = carve(nl * 4), oLeafFill = carve(nl * 4);
    const size_t oParkedCount = carve(256), oCounters = carve(256);
    const size_t oRayAos = carve(n * 32);
    const size_t oIdsA = carve(n * 4), oIdsB = carve(n * 4), oSliceCount = carve(4096 * 4), oSliceStart = carve(4096 * 4), oSliceFill = carve(4096 * 4);
    // deferred casts: the requests sorted by batching point, and per point their count / first / fill + the total
    const size_t oReqList = carve(n * 4), oReqCount = carve(nl * 4), oReqStart = carve(nl * 4), oReqFill = carve(nl * 4), oUnitStart = carve(nl * 4), oReqTotal = carve(256);
    size_t oRays = 0, oHits = 0;
    if (mem == PGPU_MEM_HOST) {
        oRays = carve(n * 8 * 4);
        oHits = carve(n * 8 * 4);
    }
    // streamed leaves: rays held at batching points whose image is not loaded yet (ray, point; double buffered), the list of the
    // rays that go on, and two per-point arrays the host fills every round
    size_t oHeld = 0, oNextActive = 0, oHeldCount = 0, oSplitDst = 0, oWorkBvhEarly = 0;
    if (c->nStaged > 0) {
        oWorkBvhEarly = carve(n * 4);
        oHeld = carve(n * 4 * 4);
        oNextActive = carve(n * 4);
        oHeldCount = carve(2 * nl * 4); // held counts, then the split destinations (sent together)
        oSplitDst = oHeldCount + nl * 4;
    }
    int rc = ensure(c, c->work, off);
    if (rc != PGPU_OK)
        return rc;
    unsigned char* wb = static_cast<unsigned char*>(c->work.ptr);
    uint32_t* dNode = reinterpret_cast<uint32_t*>(wb + oNode);
    uint64_t* dStack = reinterpret_cast<uint64_t*>(wb + oStack);
    uint32_t* dWorkRayAlt = reinterpret_cast<uint32_t*>(wb + oWorkRayAlt); // the work list is double buffered: binning reads one, writes the other
    uint32_t* dParkedLeaf = reinterpret_cast<uint32_t*>(wb + oParkedLeaf);
    uint32_t* dWorkRay = reinterpret_cast<uint32_t*>(wb + oWorkRay);
    uint32_t* dWorkBvh = reinterpret_cast<uint32_t*>(wb + oWorkBvh);
    uint32_t* dWorkLeaf = reinterpret_cast<uint32_t*>(wb + oWorkLeaf);
    uint32_t* dLeafCount = reinterpret_cast<uint32_t*>(wb + oLeafCount);
    uint32_t* dLeafStart = reinterpret_cast<uint32_t*>(wb + oLeafStart);
    uint32_t* dLeafFill = reinterpret_cast<uint32_t*>(wb + oLeafFill);
    uint32_t* dParkedCount = reinterpret_cast<uint32_t*>(wb + oParkedCount);
    uint32_t* dIds[2] = { reinterpret_cast<uint32_t*>(wb + oIdsA), reinterpret_cast<uint32_t*>(wb + oIdsB) }; // the top level's lists, by ray id
    float4* dRayAos = reinterpret_cast<float4*>(wb + oRayAos);
    uint32_t* dSliceCount = reinterpret_cast<uint32_t*>(wb + oSliceCount);
    uint32_t* dSliceStart = reinterpret_cast<uint32_t*>(wb + oSliceStart);
    uint32_t* dSliceFill = reinterpret_cast<uint32_t*>(wb + oSliceFill);
    unsigned long long* dCounters = reinterpret_cast<unsigned long long*>(wb + oCounters);
    uint32_t* dReqList = reinterpret_cast<uint32_t*>(wb + oReqList);
    uint32_t* dReqCount = reinterpret_cast<uint32_t*>(wb + oReqCount);
    uint32_t* dReqStart = reinterpret_cast<uint32_t*>(wb + oReqStart);
    uint32_t* dReqFill = reinterpret_cast<uint32_t*>(wb + oReqFill);
    uint32_t* dUnitStart = reinterpret_cast<uint32_t*>(wb + oUnitStart);
    uint32_t* dReqTotal = reinterpret_cast<uint32_t*>(wb + oReqTotal);

    float* dT = nullptr;
    uint32_t *dObj = nullptr, *dLeaf = nullptr, *dVisits = nullptr, *dAny = nullptr, *dInst = nullptr;
    const bool wantInst = kind == PGPU_CLOSEST && hits && hits->inst;
    float *dU = nullptr, *dV = nullptr;
    if (mem == PGPU_MEM_HOST) {
        float* dr = reinterpret_cast<float*>(wb + oRays);
        for (int i = 0; i < 8; ++i)
            HIP_TRY(c, hipMemcpyAsync(dr + size_t(i) * n, rp[i], n * 4, hipMemcpyHostToDevice, stream));
        for (int i = 0; i < 8; ++i)
            rp[i] = dr + size_t(i) * n;
        float* dh = reinterpret_cast<float*>(wb + oHits);
        dT = dh;
        dObj = reinterpret_cast<uint32_t*>(dh + n);
        dLeaf = reinterpret_cast<uint32_t*>(dh + 2 * n);
        dU = dh + 3 * n;
        dV = dh + 4 * n;
        dVisits = wantVisits ? reinterpret_cast<uint32_t*>(dh + 5 * n) : nullptr;
        dAny = reinterpret_cast<uint32_t*>(dh + 6 * n);
        dInst = wantInst ? reinterpret_cast<uint32_t*>(dh + 7 * n) : nullptr;
    } else {
        dInst = wantInst ? hits->inst : nullptr;
        if (kind == PGPU_CLOSEST) {
            dT = hits->t;
            dObj = hits->obj;
            dLeaf = hits->leaf;
            dU = hits->u;
            dV = hits->v;
        }
        dVisits = wantVisits ? hits->visits : nullptr;
        dAny = any_hit;
    }

    const uint32_t streamBlocks = uint32_t(std::min<size_t>((n + 255) / 256, size_t(c->numCUs) * 8));
    HIP_TRY(c, hipMemsetAsync(dCounters, 0, 256, stream));
    if (c->stats)
        HIP_TRY(c, hipMemsetAsync(c->statsDev, 0, 64, stream));
    if (dInst)
        HIP_TRY(c, hipMemsetAsync(dInst, 0xFF, n * 4, stream)); // a hit through an instance overwrites it, a later plain hit resets it
    HIP_TRY(c, hipEventRecord(c->ev0, stream));
    HIP_TRY(c, launchInitState(uint32_t(n), c->top.root, dNode, dStack, rp[7], kind == PGPU_CLOSEST ? dT : nullptr, dObj, dLeaf, dU, dV,
                   dVisits, kind == PGPU_ANY ? dAny : nullptr, streamBlocks, stream));

    TopTraceArgs ta {};
    ta.top = c->top;
    ta.ox = rp[0];
    ta.oy = rp[1];
    ta.oz = rp[2];
    ta.dx = rp[3];
    ta.dy = rp[4];
    ta.dz = rp[5];
    ta.tnear = rp[6];
    ta.tfar_in = rp[7];
    ta.tfar_state = dT;
    ta.node = dNode;
    ta.stack = dStack;
    ta.visits = dVisits;
    ta.any_out = dAny;
    ta.parked_leaf = dParkedLeaf;
    ta.leaf_count = dLeafCount;
    ta.step_cap = uint32_t(std::min<uint64_t>(0xFFFFFFF0ull, 8ull * (uint64_t(c->top.n_nodes) + c->top.n_leaves) + 64ull));
    ta.counters = dCounters;
    ta.lanes_refill_min = c->topVariant == 2 ? c->topRefillMin : 0u; // (variant 3: set per round below)
    ta.lanes_cast_trips = c->topCastTrips;
    ta.phased = (c->topVariant == 0 || c->topVariant == 4) ? 1u : 0u; // (0 = automatic: the phased form; 1 = casts inside the step)
    // the next episode's list: the binning's (rays of one batching point adjacent: coherent waves) or the parked rays in ray-id order
    // (coalesced reads; a kernel whose lanes are independent has no coherence to lose)
    const bool listById = c->topVariant == 2 || c->topVariant == 3;

    Bvh8TraceArgs ba {};
    ba.bvhs = c->bvhTable;
    ba.work_ray = dWorkRay;
    ba.work_bvh = dWorkBvh;
    ba.work_tag = dWorkLeaf;
    ba.ox = rp[0];
    ba.oy = rp[1];
    ba.oz = rp[2];
    ba.dx = rp[3];
    ba.dy = rp[4];
    ba.dz = rp[5];
    ba.tnear = rp[6];
    ba.tfar_in = kind == PGPU_CLOSEST ? dT : rp[7];
    ba.t_out = dT;
    ba.obj_out = dObj;
    ba.leaf_out = dLeaf;
    ba.u_out = dU;
    ba.v_out = dV;
    ba.any_out = dAny;
    ba.inst_out = dInst;
    ba.cursor = c->cursor;
    ba.chunk = c->chunk;
    ba.init_miss = 0;
    const uint32_t deepest = deepestBvh(c);
    ba.stats = c->stats ? c->statsDev : nullptr;

    const bool streaming = c->nStaged > 0;
    rc = ensureLeafCache(c);
    if (rc != PGPU_OK)
        return rc;
    std::vector<uint32_t> hLeafCount, hLeafStart, visitOrder, inPlace;
    std::vector<uint8_t> traced, inCache;
    uint32_t* dHeldRay[2] = { nullptr, nullptr };
    uint32_t* dHeldLeaf[2] = { nullptr, nullptr };
    uint32_t *dNextActive = nullptr, *dHeldCount = nullptr, *dSplitDst = nullptr, *dWorkBvhEarly = nullptr;
    if (streaming) {
        dWorkBvhEarly = reinterpret_cast<uint32_t*>(wb + oWorkBvhEarly);
        uint32_t* h = reinterpret_cast<uint32_t*>(wb + oHeld);
        dHeldRay[0] = h, dHeldLeaf[0] = h + n, dHeldRay[1] = h + 2 * n, dHeldLeaf[1] = h + 3 * n;
        dNextActive = reinterpret_cast<uint32_t*>(wb + oNextActive);
        dHeldCount = reinterpret_cast<uint32_t*>(wb + oHeldCount);
        dSplitDst = reinterpret_cast<uint32_t*>(wb + oSplitDst);
    }
    uint32_t nHeld = 0, heldBuf = 0;   // rays waiting at batching points that were not traced in the round they arrived
    uint64_t guardRounds = 0;          // rounds since rays were last held (the pause budget of a ray counts from there)
    uint64_t leafLoads0 = c->counters.leaf_loads = 0, streamed0 = c->counters.streamed_bytes = 0;
    (void)leafLoads0;
    (void)streamed0;
    uint64_t rounds = 0, leafVisits = 0;
    uint32_t nActive = uint32_t(n);
    const uint32_t* active = nullptr; // round 0: every ray, identity order
    // blocks of 256 per CU the top-level kernel can keep resident: LDS (160 KiB) and registers (64 VGPRs: 8 waves per SIMD)
    const uint32_t topPerCU = uint32_t(std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / ((topTraverseLdsBytes(c->top) + 1023u) & ~size_t(1023)))));
    const uint32_t topMaxBlocks = uint32_t(c->numCUs) * topPerCU;
    CastArgs ca {};
    ca.top = c->top;
    ca.ox = rp[0];
    ca.oy = rp[1];
    ca.oz = rp[2];
    ca.dx = rp[3];
    ca.dy = rp[4];
    ca.dz = rp[5];
    ca.parked_leaf = dParkedLeaf;
    ca.visits = dVisits;
    ca.leaf_count = dLeafCount;
    ca.counters = dCounters;
    ca.req_list = dReqList;
    ca.req_start = dReqStart;
    ca.req_count = dReqCount;
    ca.unit_start = dUnitStart;
    ca.plan = dReqTotal;
    ta.req_leaf_count = dReqCount;
    const uint32_t castPerCU = uint32_t(std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / ((svdagCastLdsBytes(c->top) + 1023u) & ~size_t(1023)))));
    // Rounds queued ahead of the host (resident scenes, casts inside the episode). A round's kernels take the length of their
    // work list from device memory -- the previous round's parked count -- and their launch size from the newest count the host
    // has seen (counts never grow), so the host reads a round's count a whole round late, after the next round is queued: the GPU
    // never waits for that read. The round queued behind the first empty one runs empty (a few launches of nothing).
    // (The pair-layout kernel and the instanced variant of the quad kernel take their list length from device memory: in the plain
    // quad kernel, which is bound by instruction issue, the extra scalar state cost 3 % on the single-level benchmark. A forced
    // quad layout therefore runs round by round.)
    const bool pairLists = c->hasInstances || c->lanesPerRay == 0u || c->lanesPerRay == 2u; // (instanced scenes: the quad kernel's instanced variant does the same)
    // streamed scenes: a round's bottom level runs in two passes (see the loop below) when the kernel can take its list length from
    // device memory; pgpu_set_round_lookahead(ctx, 0) restores the single pass behind the host's decision
    const bool earlyPass = streaming && pairLists && c->roundLookahead;
    if (earlyPass)
        for (hipEvent_t& e : c->roundEv)
            if (!e)
                HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (streaming) { // (the second pass of a round and the launches that trace in host memory go through a view of the BVH table)
        rc = prepareTableView(c);
        if (rc == PGPU_OK)
            rc = fillTableView(c, stream);
        if (rc != PGPU_OK)
            return rc;
    }
    if (!streaming && c->deferMin == 0u && c->roundLookahead && pairLists) {
        if (c->roundCountsCap < nl + 8) {
            if (c->roundCounts)
                HIP_TRY(c, hipHostFree(c->roundCounts));
            c->roundCounts = nullptr;
            c->roundCountsCap = 0;
            HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->roundCounts), (nl + 8) * sizeof(uint32_t), hipHostMallocDefault));
            c->roundCountsCap = nl + 8;
        }
        for (hipEvent_t& e : c->roundEv)
            if (!e)
                HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        uint32_t bound = uint32_t(n); // no round from here on has more work items than this
        for (size_t r = 0; r + 4 < c->roundCountsCap; ++r) {
            uint32_t* const countOut = dParkedCount + (r & 1u);
            const uint32_t* const countIn = r ? dParkedCount + ((r - 1) & 1u) : nullptr;
            if (r == 0) // (later rounds: the scan of the round before left the counts zeroed)
                HIP_TRY(c, hipMemsetAsync(dLeafCount, 0, nl * 4, stream));
            ta.active = active;
            ta.n_active = bound;
            ta.n_active_dev = countIn;
            // the first episode reads the rays from the caller's arrays (its list is the identity: coalesced) and leaves them behind
            // as 32-byte records; everything after it -- lists in batching-point order -- reads one sector per ray instead of eight
            ta.ray_aos = r ? dRayAos : nullptr;
            ta.ray_aos_out = r ? nullptr : dRayAos;
            ba.ray_aos = dRayAos;
            ta.may_park = r < nl ? 1u : 0u; // a ray pauses at most once per batching point
            ta.only_culled = 0u;
            if (c->topVariant == 3) // lockstep waves for the first episode (every ray from the root, coalesced reads), independent lanes for the resumed rays
                ta.lanes_refill_min = r ? c->topRefillMin : 0u;
            const uint32_t listBlocks = std::max(1u, std::min(uint32_t(c->numCUs) * 8u, (bound + 255u) / 256u));
            {
                const TraceRange r_("round: top-level episode");
                HIP_TRY(c, launchTopTraverse(ta, kind, std::max(1u, std::min(topMaxBlocks, (bound + 255u) / 256u)), stream));
            }
            HIP_TRY(c, launchLeafScan(dLeafCount, dLeafStart, dLeafFill, uint32_t(nl), countOut, stream, dLeafCount, c->cursor, &c->roundCounts[r]));
            // The parked rays by batching point, for the bottom level (binning). The next top-level episode takes the same list when
            // its waves run in lockstep -- rays of one batching point walk the same nodes and cast through the same DAG -- and the
            // parked rays in the order of the current list, i.e. by ascending ray id, when its lanes are independent anyway
            // (launchCompactParked): coalesced reads are then worth more than coherent waves.
            uint32_t* const workRayOut = (active == dWorkRay) ? dWorkRayAlt : dWorkRay;
            HIP_TRY(c, launchBinByLeaf(active, dParkedLeaf, bound, uint32_t(nl), dLeafStart, dLeafFill, c->top.leaves, workRayOut, dWorkBvh, dWorkLeaf,
                           listBlocks, stream, false, countIn));
            uint32_t* nextActive = workRayOut;
            if (listById) {
                nextActive = dIds[r & 1u];
                HIP_TRY(c, launchCompactParked(active, dParkedLeaf, bound, countIn, uint32_t(nl), dSliceCount, dSliceStart, dSliceFill, dParkedCount + 8,
                               nextActive, std::max(1u, std::min(4096u, (bound + 1023u) / 1024u)), stream));
            }
            HIP_TRY(c, hipEventRecord(c->roundEv[r & 1u], stream)); // (the scan wrote the round's count to roundCounts[r], pinned)
            ba.work_ray = workRayOut;
            ba.n_work = bound;
            ba.n_work_dev = countOut;
            ba.chunk = c->chunk;
            uint32_t blocks = 1;
            rc = planBvh8Launch(c, deepest, bound, ba, &blocks);
            if (rc != PGPU_OK)
                return rc;
            {
                const TraceRange r_("round: bottom level (per-point BVH8)"); // the reference's leaf task kernel (task_graph.h:259)
                HIP_TRY(c, launchBvh8Trace(ba, kind, tri_mode, c->stats, blocks, stream));
            }
            active = nextActive;
            if (r >= 1) {
                const TraceRange w_("round: wait for the previous round's parked count");
                HIP_TRY(c, hipEventSynchronize(c->roundEv[(r - 1) & 1u]));
                const uint32_t parked = c->roundCounts[r - 1];
                if (parked == 0)
                    break; // round r - 1 parked nothing: the traversal was over, round r ran empty
                ++rounds;
                leafVisits += parked;
                bound = parked;
            }
        }
        nActive = 0; // (skips the loop below)
    }
    bool firstEpisode = true;
    while (nActive > 0 || nHeld > 0) {
        HIP_TRY(c, hipMemsetAsync(dLeafCount, 0, nl * 4, stream));
        ta.active = active;
        ta.n_active = nActive;
        ta.n_active_dev = nullptr;
        ta.may_park = guardRounds < nl ? 1u : 0u; // a ray pauses at most once per batching point
        ++guardRounds;
        if (nActive > 0) {
        // the rays as 32-byte records (see the loop above): written by the first episode, read by the later ones and by the
        // pair-layout bottom level (streamed scenes; the deferred-cast passes of the first round read the caller's arrays)
        ta.ray_aos = (pairLists && !firstEpisode) ? dRayAos : nullptr;
        ta.ray_aos_out = (pairLists && firstEpisode) ? dRayAos : nullptr;
        ba.ray_aos = pairLists ? dRayAos : nullptr;
        ba.n_work_dev = nullptr;
        firstEpisode = false;
        const uint32_t topBlocks = std::max(1u, std::min(topMaxBlocks, (nActive + 255u) / 256u));
        // pgpu_set_svdag_defer (off unless asked for: measured slower than casts inside the episode).
        // Large passes with proxy geometry: the episode stops at every batching point that has an SVDAG, the requests are binned
        // by point and cast by a kernel that stages the point's DAG in LDS, and the rays whose cast missed (culled) go on --
        // twice; whatever is still on its way after that finishes with the casts inside the episode, as small passes do from
        // the start. (Inside an episode every trip of every lane reads 4 bytes of some DAG: a cache line per lane and trip.)
        const bool defer = c->deferMin != 0u && c->top.n_dags > 0 && ta.may_park != 0u && nActive >= c->deferMin;
        ta.only_culled = 0u;
        if (defer) {
            const uint32_t binBlocks = std::max(1u, std::min(uint32_t(c->numCUs) * 8u, (nActive + 255u) / 256u));
            const uint32_t castBlocks = std::max(1u, std::min(uint32_t(c->numCUs) * castPerCU, uint32_t(nActive / 2048u + nl)));
            ca.active = active;
            for (uint32_t pass = 0; pass < c->deferPasses; ++pass) {
                HIP_TRY(c, hipMemsetAsync(dReqCount, 0, nl * 4, stream));
                ta.only_culled = pass ? 1u : 0u;
                HIP_TRY(c, launchTopTraverse(ta, kind, topBlocks, stream, true));
                HIP_TRY(c, launchSvdagPlan(dReqCount, dReqStart, dReqFill, dUnitStart, uint32_t(nl), dReqTotal, stream));
                // (positions, not ray ids: the cast kernel updates parked_leaf[position]; the bvh / leaf outputs are scratch here)
                HIP_TRY(c, launchBinByLeaf(nullptr, dParkedLeaf, nActive, uint32_t(nl), dReqStart, dReqFill, c->top.leaves, dReqList, dWorkBvh, dWorkLeaf,
                               binBlocks, stream, true));
                HIP_TRY(c, launchSvdagCast(ca, castBlocks, stream));
            }
            ta.only_culled = 1u;
        }
        {
            const TraceRange r_("round: top-level episode");
            HIP_TRY(c, launchTopTraverse(ta, kind, topBlocks, stream));
        }
        } // nActive > 0
        if (nHeld > 0) // the rays still waiting from earlier rounds are part of this round's queues
            HIP_TRY(c, launchAddCounts(dLeafCount, dHeldCount, uint32_t(nl), stream));
        // The scan and the binning do not depend on the host knowing the count: queue them behind the traversal,
        // then fetch the number of parked rays (it sizes the bottom-level launch and ends the loop).
        HIP_TRY(c, launchLeafScan(dLeafCount, dLeafStart, dLeafFill, uint32_t(nl), dParkedCount, stream));
        uint32_t* const workRayOut = (active == dWorkRay) ? dWorkRayAlt : dWorkRay; // binning reads the current list
        if (nActive > 0)
            HIP_TRY(c, launchBinByLeaf(active, dParkedLeaf, nActive, uint32_t(nl), dLeafStart, dLeafFill, c->top.leaves, workRayOut, dWorkBvh, dWorkLeaf,
                           std::max(1u, std::min(uint32_t(c->numCUs) * 8u, (nActive + 255u) / 256u)), stream));
        if (nHeld > 0)
            HIP_TRY(c, launchBinByLeaf(dHeldRay[heldBuf], dHeldLeaf[heldBuf], nHeld, uint32_t(nl), dLeafStart, dLeafFill, c->top.leaves, workRayOut, dWorkBvh,
                           dWorkLeaf, std::max(1u, std::min(uint32_t(c->numCUs) * 8u, (nHeld + 255u) / 256u)), stream));
        uint32_t parked = 0;
        const bool firstPassRan = streaming && earlyPass && nActive > 0;
        int firstPassEvent = -1;
        if (streaming) {
            // one read per round: the per-point counts (pinned); the total and the starts follow from them
            uint32_t* const pc = c->leafHostBuf;
            HIP_TRY(c, hipMemcpyAsync(pc, dLeafCount, nl * 4, hipMemcpyDeviceToHost, stream));
            if (earlyPass) {
                HIP_TRY(c, hipEventRecord(c->roundEv[0], stream)); // (the host waits for the counts, not for what is queued behind them)
                if (nActive > 0) {
                    // First pass of the round, queued BEFORE the host looks at anything: the rays the episode has just parked, in
                    // the order of its list, through the BVH table as it stands -- batching points whose image is in the cache (or
                    // was never staged) are traced, the work items of the others are dropped by the kernel (null table entry) and
                    // are traced by the second pass, out of the binned list, once their image has come in. (Rays held from earlier
                    // rounds wait at points that are not in the cache: they are not in this pass.) While it runs the host reads the
                    // counts, decides what to load, and the copies go out: the round's loads overlap its own traversal instead of
                    // standing in front of it. The bottom level does not need its list sorted by point.
                    const TraceRange r_("round: bottom level, first pass (batching points already in the cache)");
                    HIP_TRY(c, launchLeafToBvh(dParkedLeaf, c->top.leaves, uint32_t(nl), dWorkBvhEarly, nActive,
                                   std::max(1u, std::min(uint32_t(c->numCUs) * 8u, (nActive + 255u) / 256u)), stream));
                    HIP_TRY(c, hipMemsetAsync(c->cursor, 0, 256, stream));
                    Bvh8TraceArgs ea = ba;
                    ea.bvhs = c->bvhTable;
                    ea.work_ray = active;
                    ea.work_bvh = dWorkBvhEarly;
                    ea.work_tag = dParkedLeaf;
                    ea.n_work = nActive;
                    ea.n_work_dev = nullptr;
                    ea.ray_aos = (c->hasInstances || active != nullptr) ? dRayAos : nullptr; // (identity list of a plain scene: the quad kernel, caller's arrays)
                    ea.chunk = c->chunk;
                    uint32_t blocks = 1;
                    rc = planBvh8Launch(c, deepest, nActive, ea, &blocks);
                    if (rc != PGPU_OK)
                        return rc;
                    HIP_TRY(c, launchBvh8Trace(ea, kind, tri_mode, c->stats, blocks, stream));
                    // (which images it reads is known once the counts are here: they are marked below, before anything is placed)
                    firstPassEvent = int(c->launchEventNext++ % c->launchEvents.size());
                    HIP_TRY(c, hipEventRecord(c->launchEvents[size_t(firstPassEvent)], stream));
                }
                HIP_TRY(c, hipEventSynchronize(c->roundEv[0]));
            } else {
                HIP_TRY(c, hipStreamSynchronize(stream));
            }
            hLeafCount.assign(pc, pc + nl);
            hLeafStart.resize(nl);
            for (size_t l = 0; l < nl; ++l) {
                hLeafStart[l] = parked;
                parked += hLeafCount[l];
            }
        } else {
            HIP_TRY(c, hipMemcpyAsync(&parked, dParkedCount, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(c, hipStreamSynchronize(stream));
        }
        if (parked == 0)
            break;
        ++rounds;
        if (!streaming)
            leafVisits += parked;
        ba.work_ray = workRayOut;
        uint32_t nextActiveCount = parked, nextHeld = 0;
        const uint32_t* nextActive = workRayOut;
        if (!streaming) {
            HIP_TRY(c, hipMemsetAsync(c->cursor, 0, 256, stream));
            ba.n_work = parked;
            uint32_t blocks = 1;
            rc = planBvh8Launch(c, deepest, parked, ba, &blocks);
            if (rc != PGPU_OK)
                return rc;
            HIP_TRY(c, launchBvh8Trace(ba, kind, tri_mode, c->stats, blocks, stream));
        } else {
            // Leaves streamed from host memory: the work list is sorted by batching point, so consecutive points whose
            // BVHs fit the cache together form one launch; the copies of the next group run on the copy stream while
            // this group traces (an image is overwritten only behind the last launch that read it). Odd rounds walk the
            // points downwards: what the previous round brought in last is used first.
            const uint32_t* workRayAll = workRayOut;
            // Which queues are flushed this round. The reference drains the fullest queue first and lets the others fill
            // (task_graph.cpp:19-56): an image that is not in the cache is loaded once its point holds one ray per
            // leafHoldBytesPerRay bytes of image; the rays of the other points stay parked. When nothing qualifies, the points
            // with the most rays per byte are taken, one cache-full of them. After 4 n_leaves + 64 rounds nothing is held any more.
            traced.assign(nl, 0);
            const bool mayHold = c->leafHoldBytesPerRay != 0u && rounds < 4 * uint64_t(nl) + 64;
            auto stagedImage = [&](size_t l) -> const StagedBvh* {
                const uint32_t id = c->topLeavesHost[l].bvh;
                return (id < c->staged.size() && c->staged[id].host) ? &c->staged[id] : nullptr;
            };
            size_t nTraced = 0, nWaiting = 0;
            inCache.assign(nl, 0); // points the first pass has traced already (with earlyPass): nothing to load, nothing to launch
            for (size_t l = 0; l < nl; ++l) {
                if (hLeafCount[l] == 0)
                    continue;
                const StagedBvh* im = stagedImage(l);
                inCache[l] = firstPassRan && (!im || im->resident);
                if (inCache[l] && im) // the first pass reads this image: it is not overwritten before that launch is done
                    c->staged[c->topLeavesHost[l].bvh].lastLaunch = firstPassEvent;
                if (!mayHold || !im || im->resident || uint64_t(hLeafCount[l]) * c->leafHoldBytesPerRay >= im->imageBytes()) {
                    traced[l] = 1;
                    ++nTraced;
                } else {
                    ++nWaiting;
                }
            }
            // Waiting only pays while rays can still arrive: the rays that go on this round (at most that many join the queues
            // next round) against the rays that wait. Once more rays wait than move -- or nothing moves at all -- the queues will
            // not fill any further: the waiting points with the most rays per byte are flushed too, up to one cache-full of
            // images per round in all. (Without this the tail of a trace is a string of rounds in which a few rays bounce
            // between cached points while tens of thousands sit in queues that never reach their threshold: 89 rounds, against 47
            // with a ratio of 4 on the benchmark landscape -- 1 halves the rounds again but loads a quarter more bytes and is slower,
            // profiles/r3_landscape_drain_sweep.txt.)
            uint64_t raysOn = 0, raysWaiting = 0;
            size_t loadBytes = 0;
            for (size_t l = 0; l < nl; ++l) {
                if (hLeafCount[l] == 0)
                    continue;
                if (traced[l]) {
                    raysOn += hLeafCount[l];
                    const StagedBvh* im = stagedImage(l);
                    if (im && !im->resident)
                        loadBytes += alignUp(im->imageBytes());
                } else {
                    raysWaiting += hLeafCount[l];
                }
            }
            // (prefetch: while many rays move, a round's launches are long and the copy engine idles -- room to bring in waiting
            // points early, leafPrefetchBytesPerRay bytes of image per moving ray, fullest queues first)
            const size_t prefetchBudget = size_t(double(c->leafPrefetchBytesPerRay) * double(raysOn));
            const bool drain = nTraced == 0 || double(raysWaiting) > c->leafDrainRatio * double(raysOn);
            inPlace.clear();
            if (nWaiting && (drain || loadBytes < prefetchBudget)) {
                const size_t limit = drain ? c->cacheBytes : std::min(c->cacheBytes, prefetchBudget);
                visitOrder.clear();
                for (size_t l = 0; l < nl; ++l)
                    if (hLeafCount[l] != 0 && !traced[l])
                        visitOrder.push_back(uint32_t(l));
                std::sort(visitOrder.begin(), visitOrder.end(), [&](uint32_t x, uint32_t y) {
                    const double rx = double(hLeafCount[x]) / double(stagedImage(x)->imageBytes()), ry = double(hLeafCount[y]) / double(stagedImage(y)->imageBytes());
                    return rx > ry || (rx == ry && x < y);
                });
                // A flushed queue with only a few rays is not worth its image: those rays traverse the image where it lies, in pinned
                // host memory (a visit reads a few KB over PCIe, the image is hundreds). In the tail of a trace that is most queues
                // -- 77 % of the benchmark landscape's loads served 0.15 % of its visits -- and since tracing in place has no cache
                // to fit, ALL of them go at once instead of one cache-full per round.
                for (uint32_t l : visitOrder) {
                    const size_t b = alignUp(stagedImage(l)->imageBytes());
                    const bool sparse = drain && c->leafInPlaceBytesPerRay != 0u && uint64_t(hLeafCount[l]) * c->leafInPlaceBytesPerRay < stagedImage(l)->imageBytes();
                    if (sparse) {
                        inPlace.push_back(l);
                    } else {
                        if (nTraced && loadBytes + b > limit)
                            continue; // (fuller queues first; the sparse ones behind it in the order still go in place)
                        loadBytes += b;
                    }
                    traced[l] = 1;
                    ++nTraced;
                    --nWaiting;
                }
                std::sort(inPlace.begin(), inPlace.end());
            }
            visitOrder.clear();
            uint64_t tracedRays = 0;
            {
                size_t ip = 0;
                for (size_t l = 0; l < nl; ++l)
                    if (traced[l]) {
                        const bool here = ip < inPlace.size() && inPlace[ip] == l;
                        ip += here ? 1 : 0;
                        if (!inCache[l] && !here)
                            visitOrder.push_back(uint32_t(l)); // second pass: the points whose image has to come in first
                        tracedRays += hLeafCount[l];
                    }
            }
            leafVisits += tracedRays;
            if (std::getenv("PANDORA_TRACE_ROUNDS"))
                std::fprintf(stderr, "round %llu: parked %u (active %u held %u) traced rays %llu, points traced %zu (second pass %zu, in host memory %zu) waiting %zu\n",
                    (unsigned long long)rounds, parked, nActive, nHeld, (unsigned long long)tracedRays, nTraced, visitOrder.size(), inPlace.size(), nWaiting);
            if (nWaiting) { // split the binned list: the traced points' rays go on, the others wait (one small kernel; skipped when nothing waits)
                uint32_t* const hHeldCount = c->leafHostBuf + nl; // (pinned; read by the copies queued at the end of the round)
                uint32_t* const hSplitDst = c->leafHostBuf + 2 * nl;
                std::fill(hHeldCount, hHeldCount + nl, 0u);
                uint32_t a0 = 0, h0 = 0;
                for (size_t l = 0; l < nl; ++l) {
                    if (traced[l] || hLeafCount[l] == 0) {
                        hSplitDst[l] = a0;
                        a0 += hLeafCount[l];
                    } else {
                        hSplitDst[l] = h0 | 0x80000000u;
                        hHeldCount[l] = hLeafCount[l];
                        h0 += hLeafCount[l];
                    }
                }
                nextActiveCount = a0;
                nextHeld = h0;
                nextActive = dNextActive;
                guardRounds = 0;
            }
            if (rounds % 2 == 0) // (`rounds` was incremented above: the first round walks upwards)
                std::reverse(visitOrder.begin(), visitOrder.end());
            ba.work_ray = workRayAll;
            ba.work_bvh = dWorkBvh;
            ba.work_tag = dWorkLeaf;
            ba.chunk = c->chunk;
            rc = traceLeafGroups(c, ba, kind, tri_mode, deepest, visitOrder, hLeafStart.data(), hLeafCount.data(), c->cursor, nullptr, stream, firstPassRan);
            if (rc == PGPU_OK)
                rc = traceLeavesInPlace(c, ba, kind, tri_mode, deepest, inPlace, hLeafStart.data(), hLeafCount.data(), c->cursor, nullptr, stream);
            if (rc != PGPU_OK)
                return rc;
            ba.bvhs = c->bvhTable;
            ba.work_bvh = dWorkBvh;
            ba.work_tag = dWorkLeaf;
            if (nextHeld) {
                // (pinned sources, rewritten only after the next round's synchronisation)
                HIP_TRY(c, hipMemcpyAsync(dHeldCount, c->leafHostBuf + nl, 2 * nl * 4, hipMemcpyHostToDevice, stream));
                HIP_TRY(c, launchSplitByLeaf(workRayOut, dLeafStart, dLeafCount, dSplitDst, uint32_t(nl), parked, dNextActive, dHeldRay[heldBuf ^ 1u],
                               dHeldLeaf[heldBuf ^ 1u], uint32_t(c->numCUs) * 4u, stream));
                heldBuf ^= 1u;
            }
        }
        active = nextActive;
        nActive = nextActiveCount;
        nHeld = nextHeld;
    }
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    c->timingStale = false;

    if (mem == PGPU_MEM_HOST) {
        if (kind == PGPU_CLOSEST) {
            void* dst[5] = { hits->t, hits->obj, hits->leaf, hits->u, hits->v };
            const void* src[5] = { dT, dObj, dLeaf, dU, dV };
            for (int i = 0; i < 5; ++i)
                HIP_TRY(c, hipMemcpyAsync(dst[i], src[i], n * 4, hipMemcpyDeviceToHost, stream));
        } else {
            HIP_TRY(c, hipMemcpyAsync(any_hit, dAny, n * 4, hipMemcpyDeviceToHost, stream));
        }
        if (wantVisits)
            HIP_TRY(c, hipMemcpyAsync(hits->visits, dVisits, n * 4, hipMemcpyDeviceToHost, stream));
        if (wantInst)
            HIP_TRY(c, hipMemcpyAsync(hits->inst, dInst, n * 4, hipMemcpyDeviceToHost, stream));
    }
    unsigned long long cnt[2] = { 0, 0 };
    HIP_TRY(c, hipMemcpyAsync(cnt, dCounters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    HIP_TRY(c, hipEventElapsedTime(&c->lastKernelMs, c->ev0, c->ev1));
    const uint64_t leafLoads = c->counters.leaf_loads, streamedBytes = c->counters.streamed_bytes;
    c->counters = pgpu_counters {};
    c->counters.leaf_loads = leafLoads;
    c->counters.streamed_bytes = streamedBytes;
    c->counters.svdag_tests = cnt[0];
    c->counters.svdag_culled = cnt[1];
    c->counters.leaf_visits = leafVisits;
    c->counters.rounds = rounds;
    if (c->stats) {
        const int rs = readKernelStats(c);
        if (rs != PGPU_OK)
            return rs;
    }
    if (counters)
        *counters = c->counters;
    return PGPU_OK;
}

// ---- a large batch as parts side by side ---------------------------------------------------------------------------------
// A round of the loop above is a chain of kernels of opposite character -- the top-level episode is bound by memory latency
// (few waves per CU, every lane chasing its own nodes), the bottom level by instruction issue -- and each of them ends in a
// tail in which most CUs idle. Two chains over half the rays each, on two streams, fill each other's gaps: the GPU runs the
// one's episode next to the other's bottom level (profiles/r3_overlap_halves.txt). Rays do not interact, so the results are the
// same whichever way a batch is cut. A part runs on a "lane": a context of its own (stream, work buffers, cursors, pinned
// counts) that shares the scene -- the device pointers -- with the context it belongs to and owns none of it.
namespace {
pgpu_ctx* newLane(pgpu_ctx* c)
{
    pgpu_ctx* l = new (std::nothrow) pgpu_ctx();
    if (!l)
        return nullptr;
    l->device = c->device;
    l->numCUs = c->numCUs;
    hipError_t e = hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking);
    if (e == hipSuccess)
        e = hipEventCreate(&l->ev0);
    if (e == hipSuccess)
        e = hipEventCreate(&l->ev1);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&l->cursor), 512);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&l->statsDev), 64);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&l->topCounters), 64);
    if (e != hipSuccess) {
        pgpu_shutdown(l); // (owns nothing of the scene yet)
        return nullptr;
    }
    return l;
}

// the scene and the settings as the owning context has them now (pointers and small host tables: no device work)
void refreshLane(pgpu_ctx* l, const pgpu_ctx* c)
{
    l->blocksPerCU = c->blocksPerCU;
    l->chunk = c->chunk;
    l->refillMin = c->refillMin;
    l->lanesPerRay = c->lanesPerRay;
    l->leafMin = c->leafMin;
    l->stackWindow = c->stackWindow;
    l->svdagPolicy = c->svdagPolicy;
    l->hasInstances = c->hasInstances;
    l->bvhs = c->bvhs;
    l->bvhTable = c->bvhTable;
    l->bvhTableCap = c->bvhTableCap;
    l->bvhTableDirty = false;
    l->top = c->top;
    l->haveTop = c->haveTop;
    l->topVariant = c->topVariant;
    l->topRefillMin = c->topRefillMin;
    l->topCastTrips = c->topCastTrips;
    l->roundLookahead = true;
}
}

} // extern "C"
void pandora_hip::destroyLanes(pgpu_ctx* c)
{
    for (pgpu_ctx* l : c->lanes) {
        l->bvhTable = nullptr; // the scene belongs to c
        l->bvhs.clear();
        pgpu_shutdown(l);
    }
    c->lanes.clear();
}
extern "C" {

int pgpu_set_batch_parts(pgpu_ctx* c, int parts)
{
    if (!c || parts < 0 || parts > 4)
        return PGPU_ERR_INVALID;
    c->batchParts = parts;
    return PGPU_OK;
}

int pgpu_trace(pgpu_ctx* c, int kind, int tri_mode, const pgpu_rays* rays, size_t n, int mem,
    const pgpu_hits* hits, uint32_t* any_hit, pgpu_counters* counters, void* stream_)
{
    // Which batches are cut: resident scenes on the look-ahead loop (the parts then never wait for the host inside a round), no
    // per-kernel statistics (one set of counters), and enough rays that each part still fills the GPU. Everything else -- and
    // every argument error -- goes through traceBatch as it is.
    size_t parts = c ? (c->batchParts ? size_t(c->batchParts) : (n >= kBatchPartsAutoMin ? 2u : 1u)) : 1u;
    const bool pairLists = c && (c->hasInstances || c->lanesPerRay == 0u || c->lanesPerRay == 2u);
    const bool argsOk = c && rays && (kind == PGPU_CLOSEST || kind == PGPU_ANY) && (mem == PGPU_MEM_HOST || mem == PGPU_MEM_DEVICE)
        && (kind == PGPU_ANY ? any_hit != nullptr : (hits && hits->t && hits->obj && hits->leaf && hits->u && hits->v))
        && rays->ox && rays->oy && rays->oz && rays->dx && rays->dy && rays->dz && rays->tnear && rays->tfar && n <= 0x3FFFFFFFull;
    if (!argsOk || parts < 2 || n < parts * 65536 || !c->haveTop || c->topHasRemoteLeaves || c->nStaged > 0 || c->deferMin != 0u || !c->roundLookahead
        || !pairLists || c->stats)
        return traceBatch(c, kind, tri_mode, rays, n, mem, hits, any_hit, counters, stream_);

    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rt = syncBvhTable(c);
        if (rt != PGPU_OK)
            return rt;
    }
    while (c->lanes.size() + 1 < parts) {
        pgpu_ctx* l = newLane(c);
        if (!l)
            return fail(c, PGPU_ERR_HIP, "pgpu_trace: could not set up a lane for a part of the batch");
        c->lanes.push_back(l);
    }
    hipStream_t stream = (mem == PGPU_MEM_DEVICE && stream_) ? static_cast<hipStream_t>(stream_) : c->stream;
    if (!c->lanesFork)
        HIP_TRY(c, hipEventCreateWithFlags(&c->lanesFork, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->lanesFork, stream)); // whatever produced the rays on the caller's stream comes first
    const size_t per = ((n + parts - 1) / parts + 255) & ~size_t(255);
    struct Part {
        pgpu_rays rays;
        pgpu_hits hits;
        uint32_t* any = nullptr;
        size_t n = 0;
        pgpu_counters counters {};
        int rc = PGPU_OK;
    };
    std::vector<Part> part(parts);
    for (size_t p = 0; p < parts; ++p) {
        const size_t at = std::min(n, p * per);
        Part& q = part[p];
        q.n = std::min(n - at, per);
        q.rays = pgpu_rays { rays->ox + at, rays->oy + at, rays->oz + at, rays->dx + at, rays->dy + at, rays->dz + at, rays->tnear + at, rays->tfar + at };
        q.hits = pgpu_hits {};
        if (hits) {
            q.hits.t = hits->t ? hits->t + at : nullptr;
            q.hits.obj = hits->obj ? hits->obj + at : nullptr;
            q.hits.leaf = hits->leaf ? hits->leaf + at : nullptr;
            q.hits.u = hits->u ? hits->u + at : nullptr;
            q.hits.v = hits->v ? hits->v + at : nullptr;
            q.hits.visits = hits->visits ? hits->visits + at : nullptr;
            q.hits.inst = hits->inst ? hits->inst + at : nullptr;
        }
        q.any = any_hit ? any_hit + at : nullptr;
    }
    std::vector<std::thread> threads;
    for (size_t p = 1; p < parts; ++p) {
        pgpu_ctx* l = c->lanes[p - 1];
        refreshLane(l, c);
        HIP_TRY(c, hipStreamWaitEvent(l->stream, c->lanesFork, 0));
    }
    for (size_t p = 1; p < parts; ++p)
        threads.emplace_back([&, p] {
            Part& q = part[p];
            pgpu_ctx* l = c->lanes[p - 1];
            q.rc = traceBatch(l, kind, tri_mode, &q.rays, q.n, mem, hits ? &q.hits : nullptr, q.any, &q.counters, l->stream);
        });
    part[0].rc = traceBatch(c, kind, tri_mode, &part[0].rays, part[0].n, mem, hits ? &part[0].hits : nullptr, part[0].any, &part[0].counters, stream_);
    for (std::thread& t : threads)
        t.join(); // (traceBatch returns with its stream drained: every part's results are in place)
    pgpu_counters sum {};
    float ms = c->lastKernelMs;
    for (size_t p = 0; p < parts; ++p) {
        const Part& q = part[p];
        if (q.rc != PGPU_OK) {
            if (p > 0)
                c->err = c->lanes[p - 1]->err;
            return q.rc;
        }
        sum.svdag_tests += q.counters.svdag_tests;
        sum.svdag_culled += q.counters.svdag_culled;
        sum.leaf_visits += q.counters.leaf_visits;
        sum.rounds = std::max(sum.rounds, q.counters.rounds);
        if (p > 0)
            ms = std::max(ms, c->lanes[p - 1]->lastKernelMs);
    }
    c->lastKernelMs = ms; // (the parts start together: the longest one is the trace)
    c->counters = sum;
    if (counters)
        *counters = sum;
    return PGPU_OK;
}

int pgpu_ray_state_init(pgpu_ctx* c, size_t n, const float* tfar, uint32_t* node, uint64_t* stack, const pgpu_hits* hits, uint32_t* any_hit, void* stream_)
{
    if (!c || !node || !stack || n > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_ray_state_init: null argument or batch too large");
    if (!c->haveTop)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_ray_state_init: no top-level structure uploaded");
    if (hits && (!tfar || !hits->t || !hits->obj || !hits->leaf || !hits->u || !hits->v))
        return fail(c, PGPU_ERR_INVALID, "pgpu_ray_state_init: closest-hit state needs tfar, t, obj, leaf, u, v");
    if (n == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    const uint32_t blocks = uint32_t(std::min<size_t>((n + 255) / 256, size_t(c->numCUs) * 8));
    HIP_TRY(c, launchInitState(uint32_t(n), c->top.root, node, stack, tfar, hits ? hits->t : nullptr, hits ? hits->obj : nullptr,
                   hits ? hits->leaf : nullptr, hits ? hits->u : nullptr, hits ? hits->v : nullptr, hits ? hits->visits : nullptr, any_hit, blocks, stream));
    return PGPU_OK;
}

int pgpu_top_episode(pgpu_ctx* c, int kind, const pgpu_rays* rays, const uint32_t* active, size_t n_active, uint32_t* node, uint64_t* stack,
    const float* t_state, uint32_t* visits, uint32_t* any_hit, uint32_t* parked_leaf, int may_park, void* stream_)
{
    const TraceRange range_("pgpu_top_episode");
    if (!c || !rays || !node || !stack || !parked_leaf)
        return fail(c, PGPU_ERR_INVALID, "pgpu_top_episode: null argument");
    if (!c->haveTop)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_top_episode: no top-level structure uploaded");
    if (kind != PGPU_CLOSEST && kind != PGPU_ANY)
        return fail(c, PGPU_ERR_INVALID, "pgpu_top_episode: kind must be PGPU_CLOSEST or PGPU_ANY");
    if ((kind == PGPU_CLOSEST && !t_state) || (kind == PGPU_ANY && !any_hit))
        return fail(c, PGPU_ERR_INVALID, "pgpu_top_episode: closest hit needs t_state, any hit needs any_hit");
    const float* rp[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tnear, rays->tfar };
    for (const float* p : rp)
        if (!p)
            return fail(c, PGPU_ERR_INVALID, "pgpu_top_episode: null ray array");
    if (n_active > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_top_episode: batch too large");
    if (n_active == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    TopTraceArgs ta {};
    ta.top = c->top;
    ta.active = active;
    ta.n_active = uint32_t(n_active);
    ta.ox = rp[0];
    ta.oy = rp[1];
    ta.oz = rp[2];
    ta.dx = rp[3];
    ta.dy = rp[4];
    ta.dz = rp[5];
    ta.tnear = rp[6];
    ta.tfar_in = rp[7];
    ta.tfar_state = t_state;
    ta.node = node;
    ta.stack = stack;
    ta.visits = visits;
    ta.any_out = any_hit;
    ta.parked_leaf = parked_leaf;
    ta.leaf_count = nullptr;
    ta.counters = c->topCounters;
    ta.step_cap = uint32_t(std::min<uint64_t>(0xFFFFFFF0ull, 8ull * (uint64_t(c->top.n_nodes) + c->top.n_leaves) + 64ull));
    ta.may_park = may_park ? 1u : 0u;
    ta.phased = c->topVariant != 1 ? 1u : 0u;
    const uint32_t perCU = uint32_t(std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / ((topTraverseLdsBytes(c->top) + 1023u) & ~size_t(1023)))));
    const uint32_t blocks = std::max(1u, std::min(uint32_t(c->numCUs) * perCU, uint32_t((n_active + 255) / 256)));
    HIP_TRY(c, launchTopTraverse(ta, kind, blocks, stream));
    c->topCountersDirty = true;
    return PGPU_OK;
}

int pgpu_trace_leaves(pgpu_ctx* c, int kind, int tri_mode, const pgpu_rays* rays, const uint32_t* work_ray, const uint32_t* work_leaf,
    size_t n_work, const pgpu_hits* hits, uint32_t* any_hit, void* stream_)
{
    const TraceRange range_("pgpu_trace_leaves");
    if (!c || !rays || !work_ray || !work_leaf)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaves: null argument");
    if (!c->haveTop)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_leaves: no top-level structure uploaded");
    if (kind != PGPU_CLOSEST && kind != PGPU_ANY)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaves: kind must be PGPU_CLOSEST or PGPU_ANY");
    if (tri_mode != PGPU_TRI_MT && tri_mode != PGPU_TRI_WATERTIGHT)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaves: unknown triangle test");
    if (kind == PGPU_CLOSEST && (!hits || !hits->t || !hits->obj || !hits->leaf || !hits->u || !hits->v))
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaves: closest-hit needs t, obj, leaf, u, v arrays");
    if (kind == PGPU_ANY && !any_hit)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaves: any-hit needs the any_hit array");
    const float* rp[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tnear, rays->tfar };
    for (const float* p : rp)
        if (!p)
            return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaves: null ray array");
    if (n_work > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_leaves: batch too large");
    if (n_work == 0)
        return PGPU_OK;
    if (c->bvhs.empty())
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_leaves: no bottom-level BVH is resident");
    if (c->nStaged > 0)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_leaves: this context holds BVHs staged in host memory (pgpu_stage_bvh8); only pgpu_trace streams them");
    HIP_TRY(c, hipSetDevice(c->device));
    {
        const int rt = syncBvhTable(c);
        if (rt != PGPU_OK)
            return rt;
    }
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    int rc = ensure(c, c->blockWork, alignUp(n_work * 4));
    if (rc != PGPU_OK)
        return rc;
    uint32_t* dWorkBvh = static_cast<uint32_t*>(c->blockWork.ptr);
    HIP_TRY(c, launchLeafToBvh(work_leaf, c->top.leaves, c->top.n_leaves, dWorkBvh, uint32_t(n_work),
                   uint32_t(std::min<size_t>((n_work + 255) / 256, size_t(c->numCUs) * 8)), stream));
    Bvh8TraceArgs ba {};
    ba.bvhs = c->bvhTable;
    ba.work_ray = work_ray;
    ba.work_bvh = dWorkBvh;
    ba.work_tag = work_leaf;
    ba.n_work = uint32_t(n_work);
    ba.ox = rp[0];
    ba.oy = rp[1];
    ba.oz = rp[2];
    ba.dx = rp[3];
    ba.dy = rp[4];
    ba.dz = rp[5];
    ba.tnear = rp[6];
    ba.tfar_in = kind == PGPU_CLOSEST ? hits->t : rp[7];
    if (kind == PGPU_CLOSEST) {
        ba.t_out = hits->t;
        ba.obj_out = hits->obj;
        ba.leaf_out = hits->leaf;
        ba.u_out = hits->u;
        ba.v_out = hits->v;
        ba.inst_out = hits->inst;
    }
    ba.any_out = any_hit;
    ba.cursor = c->cursor;
    ba.chunk = c->chunk;
    ba.init_miss = 0;
    ba.stats = c->stats ? c->statsDev : nullptr;
    const uint32_t deepest = deepestBvh(c);
    HIP_TRY(c, hipMemsetAsync(c->cursor, 0, 256, stream));
    uint32_t blocks = 1;
    rc = planBvh8Launch(c, deepest, n_work, ba, &blocks);
    if (rc != PGPU_OK)
        return rc;
    HIP_TRY(c, hipEventRecord(c->ev0, stream));
    HIP_TRY(c, launchBvh8Trace(ba, kind, tri_mode, c->stats, blocks, stream));
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    c->timingStale = false;
    return PGPU_OK;
}
}
