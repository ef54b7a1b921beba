// Several GPUs: the round loop of pgpu_trace with the batching points sharded over the ranks' HBM (BASELINE
// configs[3]) and the film all-reduce. See include/pandora_hip.h ("several GPUs") for the protocol.
//
// What it replaces in the reference: the per-leaf MPMC queues a ray is enqueued into when it reaches a batching point
// (OfflineBatchingAccelerationStructure, offline_batching_acceleration_structure.h:269-305; tasking::TaskGraph::enqueue,
// stream/task_graph.h:130-156) and the flush loop that drains the fullest queue after loading its geometry from the disk
// cache (task_graph.cpp:19-56, offline_batching...h:158-260). Here a leaf's geometry is resident on its owner GPU and a
// leaf's "queue" is whatever the exchange delivers to that GPU in a round.
//
// Round (every rank, in lockstep; everything on one stream):
//   top-level episode for the active rays (counts per batching point in LDS)          topTraverseKernel
//   scan + bin the parked rays by batching point (next round's active list)           leafScanKernel, binByLeafKernel
//   count the parked rays per owner rank                                              routeCountKernel
//   all-gather of the count vectors -> the world x world matrix, read by the host     <- the ONE host synchronisation
//   bottom-level traversal of the binned list in place: points owned elsewhere are skipped by the kernel (no BVH here)
//   records of the rays parked at other ranks' points -> all-to-all -> owner traces them -> results all-to-all back -> merge
// until the matrix is all zero. The transport (three collectives) is RCCL -- grouped ncclSend / ncclRecv, one message per
// peer and direction, i.e. one per xGMI link -- or supplied by the caller.
#include "context.h"
#include "trace_ranges.h"
#include <algorithm>
#include <cstring>
#include <dlfcn.h>
#include <rccl/rccl.h>

using namespace pandora_hip;

namespace pandora_hip {

// RCCL is resolved at run time: the library has no link-time dependency on it, and a process that already holds an RCCL
// (torch ships one under the same soname) keeps exactly one copy.
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*getUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*groupStart)() = nullptr;
    ncclResult_t (*groupEnd)() = nullptr;
    ncclResult_t (*send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*allGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*allReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*getErrorString)(ncclResult_t) = nullptr;
};

} // namespace pandora_hip

// (declared next to pgpu_ctx in context.h)
struct Comm {
    int rank = 0, world = 1;
    pgpu_transport transport {};
    bool ownsRccl = false;
    ncclComm_t nccl = nullptr;
    std::string transportError;
};

namespace {

RcclApi g_rccl;

const char* loadRccl()
{
    if (g_rccl.lib)
        return nullptr;
    void* lib = nullptr;
    for (const char* name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" })
        if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr)
            break;
    if (!lib)
        return "librccl.so.1 not found";
    RcclApi a;
    a.lib = lib;
#define PANDORA_SYM(field, name)                                        \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name));    \
    if (!a.field)                                                       \
        return "librccl: symbol " name " missing";
    PANDORA_SYM(getUniqueId, "ncclGetUniqueId")
    PANDORA_SYM(commInitRank, "ncclCommInitRank")
    PANDORA_SYM(commDestroy, "ncclCommDestroy")
    PANDORA_SYM(groupStart, "ncclGroupStart")
    PANDORA_SYM(groupEnd, "ncclGroupEnd")
    PANDORA_SYM(send, "ncclSend")
    PANDORA_SYM(recv, "ncclRecv")
    PANDORA_SYM(allGather, "ncclAllGather")
    PANDORA_SYM(allReduce, "ncclAllReduce")
    PANDORA_SYM(getErrorString, "ncclGetErrorString")
#undef PANDORA_SYM
    g_rccl = a;
    return nullptr;
}

// ---- the RCCL transport (user = Comm*)
int rcclFail(Comm* cm, const char* what, ncclResult_t r)
{
    cm->transportError = std::string(what) + ": " + (g_rccl.getErrorString ? g_rccl.getErrorString(r) : "?");
    return 1;
}

int rcclAllGatherU32(void* user, const uint32_t* send, uint32_t* recv, size_t n, void* stream)
{
    Comm* cm = static_cast<Comm*>(user);
    const ncclResult_t r = g_rccl.allGather(send, recv, n, ncclUint32, cm->nccl, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : rcclFail(cm, "ncclAllGather", r);
}

// all-to-all(v) as one group of point-to-point messages: one per peer and direction, so every xGMI link carries only its
// own pair's traffic
int rcclExchange(void* user, const void* send, const uint32_t* sendCounts, void* recv, const uint32_t* recvCounts, size_t rowBytes, void* stream_)
{
    Comm* cm = static_cast<Comm*>(user);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const unsigned char* s = static_cast<const unsigned char*>(send);
    unsigned char* d = static_cast<unsigned char*>(recv);
    ncclResult_t r = g_rccl.groupStart();
    if (r != ncclSuccess)
        return rcclFail(cm, "ncclGroupStart", r);
    size_t so = 0, ro = 0;
    for (int p = 0; p < cm->world && r == ncclSuccess; ++p) {
        const size_t sb = size_t(sendCounts[p]) * rowBytes, rb = size_t(recvCounts[p]) * rowBytes;
        if (p == cm->rank) {
            if (sb && hipMemcpyAsync(d + ro, s + so, sb, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                r = ncclUnhandledCudaError;
        } else {
            if (sb)
                r = g_rccl.send(s + so, sb, ncclChar, p, cm->nccl, stream);
            if (rb && r == ncclSuccess)
                r = g_rccl.recv(d + ro, rb, ncclChar, p, cm->nccl, stream);
        }
        so += sb;
        ro += rb;
    }
    const ncclResult_t e = g_rccl.groupEnd();
    if (r != ncclSuccess)
        return rcclFail(cm, "ncclSend / ncclRecv", r);
    return e == ncclSuccess ? 0 : rcclFail(cm, "ncclGroupEnd", e);
}

int rcclAllReduceU64(void* user, unsigned long long* buf, size_t n, void* stream)
{
    Comm* cm = static_cast<Comm*>(user);
    const ncclResult_t r = g_rccl.allReduce(buf, buf, n, ncclUint64, ncclSum, cm->nccl, static_cast<hipStream_t>(stream));
    return r == ncclSuccess ? 0 : rcclFail(cm, "ncclAllReduce", r);
}

int transportFail(pgpu_ctx* c, const char* what)
{
    std::string msg = what;
    if (c->comm && !c->comm->transportError.empty())
        msg += " (" + c->comm->transportError + ")";
    c->err = msg;
    return PGPU_ERR_HIP;
}

uint32_t gridFor(const pgpu_ctx* c, size_t n) { return uint32_t(std::max<size_t>(1, std::min<size_t>((n + 255) / 256, size_t(c->numCUs) * 8))); }

} // namespace

namespace pandora_hip {
void destroyComm(pgpu_ctx* c)
{
    if (!c->comm)
        return;
    if (c->comm->ownsRccl && c->comm->nccl && g_rccl.commDestroy)
        (void)g_rccl.commDestroy(c->comm->nccl);
    delete c->comm;
    c->comm = nullptr;
}
}

extern "C" {

int pgpu_comm_unique_id(void* out128)
{
    if (!out128)
        return PGPU_ERR_INVALID;
    if (loadRccl() != nullptr)
        return PGPU_ERR_HIP;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    ncclUniqueId id;
    if (g_rccl.getUniqueId(&id) != ncclSuccess)
        return PGPU_ERR_HIP;
    std::memcpy(out128, &id, sizeof(id));
    return PGPU_OK;
}

int pgpu_comm_init(pgpu_ctx* c, int rank, int world, const void* unique_id128)
{
    if (!c || !unique_id128 || world < 1 || world > int(routeMaxRanks()) || rank < 0 || rank >= world)
        return fail(c, PGPU_ERR_INVALID, "pgpu_comm_init: null argument, or rank / world out of range (1..64 ranks)");
    if (const char* why = loadRccl())
        return fail(c, PGPU_ERR_HIP, why);
    HIP_TRY(c, hipSetDevice(c->device));
    destroyComm(c);
    Comm* cm = new Comm();
    cm->rank = rank;
    cm->world = world;
    ncclUniqueId id;
    std::memcpy(&id, unique_id128, sizeof(id));
    const ncclResult_t r = g_rccl.commInitRank(&cm->nccl, world, id, rank);
    if (r != ncclSuccess) {
        c->err = std::string("pgpu_comm_init: ncclCommInitRank: ") + g_rccl.getErrorString(r);
        delete cm;
        return PGPU_ERR_HIP;
    }
    cm->ownsRccl = true;
    cm->transport.user = cm;
    cm->transport.allgather_u32 = rcclAllGatherU32;
    cm->transport.exchange = rcclExchange;
    cm->transport.allreduce_u64 = rcclAllReduceU64;
    c->comm = cm;
    return PGPU_OK;
}

int pgpu_comm_init_transport(pgpu_ctx* c, int rank, int world, const pgpu_transport* t)
{
    if (!c || !t || !t->allgather_u32 || !t->exchange || !t->allreduce_u64 || world < 1 || world > int(routeMaxRanks()) || rank < 0 || rank >= world)
        return fail(c, PGPU_ERR_INVALID, "pgpu_comm_init_transport: null argument, or rank / world out of range (1..64 ranks)");
    destroyComm(c);
    Comm* cm = new Comm();
    cm->rank = rank;
    cm->world = world;
    cm->transport = *t;
    c->comm = cm;
    return PGPU_OK;
}

int pgpu_comm_destroy(pgpu_ctx* c)
{
    if (!c)
        return PGPU_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    destroyComm(c);
    return PGPU_OK;
}

// Runs every collective of the transport once on small buffers and checks what comes back (collective: every rank calls it).
int pgpu_comm_selftest(pgpu_ctx* c)
{
    if (!c || !c->comm)
        return fail(c, PGPU_ERR_INVALID, "pgpu_comm_selftest: no communicator");
    HIP_TRY(c, hipSetDevice(c->device));
    Comm* cm = c->comm;
    const uint32_t world = uint32_t(cm->world), rank = uint32_t(cm->rank);
    const size_t rows = 3; // rows every rank sends to every rank
    // layout: [0] all-gather in (4 u32), [1] all-gather out, [2] exchange out, [3] exchange in, [4] all-reduce (8 u64), [5] send-to-self in / out
    const size_t seg = alignUp(std::max<size_t>(size_t(world) * rows * 16, 4096));
    int rc = ensure(c, c->shardRecv, 6 * seg);
    if (rc != PGPU_OK)
        return rc;
    unsigned char* base = static_cast<unsigned char*>(c->shardRecv.ptr);
    std::vector<uint32_t> h(seg / 4);
    hipStream_t stream = c->stream;
    auto segment = [&](int i) { return base + size_t(i) * seg; };
    // all-gather: rank r contributes {r, r + 100, r + 200, r + 300}
    for (uint32_t i = 0; i < 4; ++i)
        h[i] = rank + 100u * i;
    HIP_TRY(c, hipMemcpyAsync(segment(0), h.data(), 16, hipMemcpyHostToDevice, stream));
    if (cm->transport.allgather_u32(cm->transport.user, reinterpret_cast<uint32_t*>(segment(0)), reinterpret_cast<uint32_t*>(segment(1)), 4, stream) != 0)
        return fail(c, PGPU_ERR_HIP, ("pgpu_comm_selftest: all-gather failed (" + cm->transportError + ")").c_str());
    HIP_TRY(c, hipMemcpyAsync(h.data(), segment(1), size_t(world) * 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    for (uint32_t r = 0; r < world; ++r)
        for (uint32_t i = 0; i < 4; ++i)
            if (h[r * 4 + i] != r + 100u * i)
                return fail(c, PGPU_ERR_HIP, "pgpu_comm_selftest: all-gather returned wrong data");
    // exchange: `rows` 16-byte rows to every rank, row j for rank d = {rank, d, j, 7}
    std::vector<uint32_t> counts(world, uint32_t(rows));
    for (uint32_t d = 0; d < world; ++d)
        for (uint32_t j = 0; j < rows; ++j) {
            uint32_t* row = &h[(d * rows + j) * 4];
            row[0] = rank, row[1] = d, row[2] = j, row[3] = 7u;
        }
    HIP_TRY(c, hipMemcpyAsync(segment(2), h.data(), size_t(world) * rows * 16, hipMemcpyHostToDevice, stream));
    if (cm->transport.exchange(cm->transport.user, segment(2), counts.data(), segment(3), counts.data(), 16, stream) != 0)
        return fail(c, PGPU_ERR_HIP, ("pgpu_comm_selftest: exchange failed (" + cm->transportError + ")").c_str());
    HIP_TRY(c, hipMemcpyAsync(h.data(), segment(3), size_t(world) * rows * 16, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    for (uint32_t s = 0; s < world; ++s)
        for (uint32_t j = 0; j < rows; ++j) {
            const uint32_t* row = &h[(s * rows + j) * 4];
            if (row[0] != s || row[1] != rank || row[2] != j || row[3] != 7u)
                return fail(c, PGPU_ERR_HIP, "pgpu_comm_selftest: exchange returned wrong data");
        }
    // all-reduce: every rank contributes {1, rank, 2^40, ...}
    unsigned long long in[8], out[8];
    for (int i = 0; i < 8; ++i)
        in[i] = i == 0 ? 1ull : (i == 1 ? rank : (1ull << 40) + uint64_t(i));
    HIP_TRY(c, hipMemcpyAsync(segment(4), in, sizeof(in), hipMemcpyHostToDevice, stream));
    if (cm->transport.allreduce_u64(cm->transport.user, reinterpret_cast<unsigned long long*>(segment(4)), 8, stream) != 0)
        return fail(c, PGPU_ERR_HIP, ("pgpu_comm_selftest: all-reduce failed (" + cm->transportError + ")").c_str());
    HIP_TRY(c, hipMemcpyAsync(out, segment(4), sizeof(out), hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    for (int i = 0; i < 8; ++i) {
        const unsigned long long want = i == 0 ? world : (i == 1 ? uint64_t(world) * (world - 1) / 2 : uint64_t(world) * ((1ull << 40) + uint64_t(i)));
        if (out[i] != want)
            return fail(c, PGPU_ERR_HIP, "pgpu_comm_selftest: all-reduce returned a wrong sum");
    }
    // RCCL only: a point-to-point message to this rank itself (the exchange copies its own block without RCCL, so this is
    // what makes ncclSend / ncclRecv run when there is a single rank)
    if (cm->ownsRccl) {
        for (uint32_t i = 0; i < 256; ++i)
            h[i] = 0xC0FFEE00u + i;
        HIP_TRY(c, hipMemcpyAsync(segment(5), h.data(), 1024, hipMemcpyHostToDevice, stream));
        ncclResult_t r = g_rccl.groupStart();
        if (r == ncclSuccess)
            r = g_rccl.send(segment(5), 1024, ncclChar, cm->rank, cm->nccl, stream);
        if (r == ncclSuccess)
            r = g_rccl.recv(segment(5) + 2048, 1024, ncclChar, cm->rank, cm->nccl, stream);
        const ncclResult_t e = g_rccl.groupEnd();
        if (r != ncclSuccess || e != ncclSuccess)
            return fail(c, PGPU_ERR_HIP, (std::string("pgpu_comm_selftest: ncclSend / ncclRecv to self: ") + g_rccl.getErrorString(r != ncclSuccess ? r : e)).c_str());
        HIP_TRY(c, hipMemcpyAsync(h.data(), segment(5) + 2048, 1024, hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        for (uint32_t i = 0; i < 256; ++i)
            if (h[i] != 0xC0FFEE00u + i)
                return fail(c, PGPU_ERR_HIP, "pgpu_comm_selftest: ncclSend / ncclRecv to self returned wrong data");
    }
    return PGPU_OK;
}

int pgpu_set_leaf_owners(pgpu_ctx* c, const uint32_t* owner_of_leaf, size_t n_leaves)
{
    if (!c || !owner_of_leaf)
        return fail(c, PGPU_ERR_INVALID, "pgpu_set_leaf_owners: null argument");
    if (!c->haveTop || n_leaves != c->top.n_leaves)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_set_leaf_owners: one owner per batching point of the uploaded top level");
    const int world = c->comm ? c->comm->world : 1, rank = c->comm ? c->comm->rank : 0;
    std::vector<uint32_t> remote(n_leaves);
    for (size_t i = 0; i < n_leaves; ++i) {
        if (owner_of_leaf[i] >= uint32_t(world))
            return fail(c, PGPU_ERR_INVALID, "pgpu_set_leaf_owners: owner rank beyond the communicator (call pgpu_comm_init first)");
        const bool mine = owner_of_leaf[i] == uint32_t(rank);
        if (mine != (c->topLeavesHost[i].bvh != PGPU_NO_BVH))
            return fail(c, PGPU_ERR_INVALID, "pgpu_set_leaf_owners: a batching point's BVH must be resident on its owner and only there");
        remote[i] = mine ? PGPU_MISS : owner_of_leaf[i];
    }
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->leafOwnerDev || c->leafOwnerHost.size() < n_leaves) {
        void *a = nullptr, *b = nullptr;
        HIP_TRY(c, hipMalloc(&a, std::max<size_t>(n_leaves * 4, 16)));
        c->owned.push_back(a);
        HIP_TRY(c, hipMalloc(&b, std::max<size_t>(n_leaves * 4, 16)));
        c->owned.push_back(b);
        c->leafOwnerDev = static_cast<uint32_t*>(a);
        c->leafOwnerRemoteDev = static_cast<uint32_t*>(b);
    }
    HIP_TRY(c, hipMemcpy(c->leafOwnerDev, owner_of_leaf, n_leaves * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->leafOwnerRemoteDev, remote.data(), n_leaves * 4, hipMemcpyHostToDevice));
    c->leafOwnerHost.assign(owner_of_leaf, owner_of_leaf + n_leaves);
    return PGPU_OK;
}

int pgpu_trace_sharded(pgpu_ctx* c, int kind, int tri_mode, const pgpu_rays* rays, size_t n, const pgpu_hits* hits, uint32_t* any_hit,
    pgpu_counters* counters, void* stream_)
{
    if (!c || !rays)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: null argument");
    if (!c->haveTop)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_sharded: no top-level structure uploaded");
    if (c->leafOwnerHost.size() != c->top.n_leaves)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_sharded: no owner table for this top level (pgpu_set_leaf_owners)");
    if (kind != PGPU_CLOSEST && kind != PGPU_ANY)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: kind must be PGPU_CLOSEST or PGPU_ANY");
    if (tri_mode != PGPU_TRI_MT && tri_mode != PGPU_TRI_WATERTIGHT)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: unknown triangle test");
    if (kind == PGPU_CLOSEST && (!hits || !hits->t || !hits->obj || !hits->leaf || !hits->u || !hits->v))
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: closest-hit needs t, obj, leaf, u, v arrays");
    if (kind == PGPU_ANY && !any_hit)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: any-hit needs the any_hit array");
    if (n > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: batch too large");
    const float* rp[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tnear, rays->tfar };
    for (const float* p : rp)
        if (!p && n)
            return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: null ray array");
    if (counters)
        std::memset(counters, 0, sizeof(*counters));
    const uint32_t world = c->comm ? uint32_t(c->comm->world) : 1u, rank = c->comm ? uint32_t(c->comm->rank) : 0u;
    if (world > 1 && !c->comm)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: no communicator");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = syncBvhTable(c);
    if (rc != PGPU_OK)
        return rc;
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    const bool closest = kind == PGPU_CLOSEST;
    const size_t nl = c->top.n_leaves;
    const size_t nb = std::max<size_t>(n, 1); // a rank without rays still takes part in every round

    // ---- persistent work buffers (grown on demand, never per round)
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t at = off;
        off += alignUp(bytes);
        return at;
    };
    const size_t oNode = carve(nb * 4), oStack = carve(nb * 8), oParked = carve(nb * 4), oWorkA = carve(nb * 4), oWorkB = carve(nb * 4);
    const size_t oWorkBvh = carve(nb * 4), oWorkLeaf = carve(nb * 4), oSlot = carve(nb * 4), oSentLeaf = carve(nb * 4);
    const size_t oLeafCount = carve(nl * 4), oLeafStart = carve(nl * 4), oLeafFill = carve(nl * 4), oTotal = carve(256);
    const size_t oCounts = carve(512), oFill = carve(256), oMatrix = carve(size_t(world) * (world + 1u) * 4);
    // the rays as 32-byte records by ray id, as in pgpu_trace (Bvh8TraceArgs::ray_aos): pair-layout lists only
    const bool rayRecords = c->hasInstances || c->lanesPerRay == 0u || c->lanesPerRay == 2u; // (pair kernel, or the quad kernel's instanced variant)
    const size_t oRayAos = carve(rayRecords ? nb * 32 : 0);
    rc = ensure(c, c->shardWork, off);
    if (rc != PGPU_OK)
        return rc;
    rc = ensure(c, c->shardSend, alignUp(nb * 36) + alignUp(nb * 20)); // records out, results back (20 bytes each with instances)
    if (rc != PGPU_OK)
        return rc;
    unsigned char* wb = static_cast<unsigned char*>(c->shardWork.ptr);
    uint32_t* dNode = reinterpret_cast<uint32_t*>(wb + oNode);
    uint64_t* dStack = reinterpret_cast<uint64_t*>(wb + oStack);
    uint32_t* dParked = reinterpret_cast<uint32_t*>(wb + oParked);
    uint32_t* dWork[2] = { reinterpret_cast<uint32_t*>(wb + oWorkA), reinterpret_cast<uint32_t*>(wb + oWorkB) };
    uint32_t* dWorkBvh = reinterpret_cast<uint32_t*>(wb + oWorkBvh);
    uint32_t* dWorkLeaf = reinterpret_cast<uint32_t*>(wb + oWorkLeaf);
    uint32_t* dSlot = reinterpret_cast<uint32_t*>(wb + oSlot);
    uint32_t* dSentLeaf = reinterpret_cast<uint32_t*>(wb + oSentLeaf);
    uint32_t* dLeafCount = reinterpret_cast<uint32_t*>(wb + oLeafCount);
    uint32_t* dLeafStart = reinterpret_cast<uint32_t*>(wb + oLeafStart);
    uint32_t* dLeafFill = reinterpret_cast<uint32_t*>(wb + oLeafFill);
    uint32_t* dTotal = reinterpret_cast<uint32_t*>(wb + oTotal);
    uint32_t* dCounts = reinterpret_cast<uint32_t*>(wb + oCounts);
    uint32_t* dFill = reinterpret_cast<uint32_t*>(wb + oFill);
    uint32_t* dMatrix = reinterpret_cast<uint32_t*>(wb + oMatrix);
    float4* dRayAos = rayRecords ? reinterpret_cast<float4*>(wb + oRayAos) : nullptr;
    float* dSendRec = static_cast<float*>(c->shardSend.ptr);
    float* dResBack = reinterpret_cast<float*>(static_cast<unsigned char*>(c->shardSend.ptr) + alignUp(nb * 36));

    uint32_t* dVisits = hits ? hits->visits : nullptr;
    uint32_t* dInst = closest ? hits->inst : nullptr;
    if (dInst && n)
        HIP_TRY(c, hipMemsetAsync(dInst, 0xFF, n * 4, stream)); // a hit through an instance overwrites it, a later plain hit resets it
    HIP_TRY(c, hipMemsetAsync(c->topCounters, 0, 64, stream));
    if (c->stats)
        HIP_TRY(c, hipMemsetAsync(c->statsDev, 0, 64, stream));
    HIP_TRY(c, hipEventRecord(c->ev0, stream));
    if (n)
        HIP_TRY(c, launchInitState(uint32_t(n), c->top.root, dNode, dStack, rp[7], closest ? hits->t : nullptr, closest ? hits->obj : nullptr,
                       closest ? hits->leaf : nullptr, closest ? hits->u : nullptr, closest ? hits->v : nullptr, dVisits, closest ? nullptr : any_hit,
                       gridFor(c, n), stream));

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
    ta.tfar_state = closest ? hits->t : nullptr;
    ta.node = dNode;
    ta.stack = dStack;
    ta.visits = dVisits;
    ta.any_out = any_hit;
    ta.parked_leaf = dParked;
    ta.leaf_count = dLeafCount;
    ta.step_cap = uint32_t(std::min<uint64_t>(0xFFFFFFF0ull, 8ull * (uint64_t(c->top.n_nodes) + c->top.n_leaves) + 64ull));
    ta.phased = c->topVariant != 1 ? 1u : 0u; // (the casts of a wave gathered into phases, top_kernels.hip)
    ta.counters = c->topCounters;

    Bvh8TraceArgs ba {};
    ba.bvhs = c->bvhTable;
    ba.work_bvh = dWorkBvh;
    ba.work_tag = dWorkLeaf;
    ba.ox = rp[0];
    ba.oy = rp[1];
    ba.oz = rp[2];
    ba.dx = rp[3];
    ba.dy = rp[4];
    ba.dz = rp[5];
    ba.tnear = rp[6];
    ba.tfar_in = closest ? hits->t : rp[7];
    if (closest) {
        ba.t_out = hits->t;
        ba.obj_out = hits->obj;
        ba.leaf_out = hits->leaf;
        ba.u_out = hits->u;
        ba.v_out = hits->v;
        ba.inst_out = dInst;
    }
    ba.any_out = any_hit;
    ba.cursor = c->cursor;
    ba.init_miss = 0;
    ba.stats = c->stats ? c->statsDev : nullptr;
    const uint32_t deepest = deepestBvh(c);

    const uint32_t topPerCU = uint32_t(std::max<size_t>(1, std::min<size_t>(8, (160u * 1024u) / ((topTraverseLdsBytes(c->top) + 1023u) & ~size_t(1023)))));
    const uint32_t topMaxBlocks = uint32_t(c->numCUs) * topPerCU;
    std::vector<uint32_t> matrix(size_t(world) * world, 0u), gathered(size_t(world) * (world + 1u), 0u), sendCounts(world), recvCounts(world), starts(64, 0u);
    uint64_t rounds = 0, leafVisits = 0, routed = 0, exchangeBytes = 0;
    uint32_t nActive = uint32_t(n);
    const uint32_t* active = nullptr; // round 0: every ray, identity order
    int flip = 0;

    // ---- Failures. A rank that leaves the loop alone strands its peers inside the round's collectives. So (1) everything a round
    // could fail to allocate is reserved before the first round, from what the ranks tell each other up front; (2) a rank that
    // fails -- there, or in a launch later on -- keeps taking part: it raises a flag that travels with the next count rows, and
    // every rank leaves the loop in that same round (the failing rank with its own error, the others with "a peer failed").
    // Only a failing transport call ends the loop at once: there is nothing left to tell the peers with.
    // Batching points staged in host memory (pgpu_stage_bvh8; BASELINE configs[4]): the owner streams their images through its HBM
    // cache as pgpu_trace does -- both of a round's bottom-level passes (the rays parked here in place, the received records) run
    // as launch groups of what fits the cache together. Every point that has work in a round is loaded in that round: rays are
    // not held back for fuller queues here (pgpu_set_leaf_hold applies to pgpu_trace only), a record's result goes home in the
    // round it arrived.
    const bool streamed = c->nStaged > 0;
    c->counters.leaf_loads = c->counters.streamed_bytes = 0;
    std::vector<uint32_t> hLeafStart, visitOrder;
    int pending = PGPU_OK; // this rank's first soft failure
    auto soft = [&](int code) {
        if (code != PGPU_OK && pending == PGPU_OK)
            pending = code;
        return code == PGPU_OK;
    };
    auto softHip = [&](hipError_t e, const char* what) { return soft(e == hipSuccess ? PGPU_OK : fail(c, PGPU_ERR_HIP, what, e)); };

    // Up front, one all-gather: every rank's ray count and whether it holds instanced leaves. Result rows are {hit}, {t, obj, u, v},
    // or with a fifth value -- the instance a hit went through -- when any rank of the job holds instances: every rank must use the
    // same row size. No round can deliver more records to a rank than the other ranks have rays.
    size_t resultBytes = closest ? 16 : 4;
    bool instRows = c->hasInstances;
    size_t recvCap = 1;
    if (world > 1) {
        const uint32_t mine[2] = { uint32_t(n), c->hasInstances ? 1u : 0u };
        HIP_TRY(c, hipMemcpyAsync(dCounts, mine, 8, hipMemcpyHostToDevice, stream));
        if (c->comm->transport.allgather_u32(c->comm->transport.user, dCounts, dMatrix, 2, stream) != 0)
            return transportFail(c, "pgpu_trace_sharded: all-gather of the ray counts failed");
        HIP_TRY(c, hipMemcpyAsync(gathered.data(), dMatrix, size_t(world) * 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        uint64_t others = 0;
        for (uint32_t r = 0; r < world; ++r) {
            if (r != rank)
                others += gathered[2u * r];
            instRows = instRows || gathered[2u * r + 1u] != 0u;
        }
        recvCap = size_t(std::max<uint64_t>(others, 1));
        // (soft: a rank that cannot get the memory says so in the first round's rows)
        if (soft(ensure(c, c->shardRecv, alignUp(recvCap * 36) + alignUp(recvCap * 20))))
            soft(reserveLeafRecords(c, recvCap, &c->spill));
        if (!c->auxStream)
            softHip(hipStreamCreateWithFlags(&c->auxStream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
        for (hipEvent_t* e : { &c->auxFork, &c->auxJoin })
            if (!*e)
                softHip(hipEventCreateWithFlags(e, hipEventDisableTiming), "hipEventCreateWithFlags");
    }
    if (closest)
        resultBytes = instRows ? 20 : 16;
    if (streamed)
        if (soft(prepareLeafCache(c))) // (the arena, the copy stream, the pinned per-point arrays)
            soft(prepareLeafTableView(c, stream)); // (launches that trace sparse queues in host memory see the BVH table through a view)
    {
        // the in-place trace of the rays parked at this rank's own points: at most n work items, overflow buffer of its own
        Bvh8TraceArgs probe = ba;
        probe.work_ray = dWork[0];
        probe.chunk = c->chunk;
        uint32_t blocks = 1;
        soft(planBvh8Launch(c, deepest, nb, probe, &blocks, &c->spillB));
    }
    float* const dRecvRec = static_cast<float*>(c->shardRecv.ptr);
    float* const dResOut = c->shardRecv.ptr ? reinterpret_cast<float*>(static_cast<unsigned char*>(c->shardRecv.ptr) + alignUp(recvCap * 36)) : nullptr;
    uint32_t* const cursorLocal = c->cursor + 64; // (the record trace keeps c->cursor)
    // a record carries the ray with tfar = its current t (closest hit), so the owner only reports something closer
    const float* recordSrc[8] = { rp[0], rp[1], rp[2], rp[3], rp[4], rp[5], rp[6], closest ? hits->t : rp[7] };
    int peerFailed = 0;
    for (;;) {
        const TraceRange roundRange("sharded round");
        uint32_t* const workOut = dWork[flip];
        if (c->debugFailRound >= 0 && rounds == uint64_t(c->debugFailRound))
            soft(fail(c, PGPU_ERR_HIP, "pgpu_trace_sharded: failure injected by pgpu_debug_fail_sharded"));
        bool ok = pending == PGPU_OK;
        ok = ok && softHip(hipMemsetAsync(dLeafCount, 0, nl * 4, stream), "hipMemsetAsync");
        ok = ok && softHip(hipMemsetAsync(dCounts, 0, 512, stream), "hipMemsetAsync"); // counts per owner, then this rank's failure flag
        ok = ok && softHip(hipMemsetAsync(dFill, 0, 256, stream), "hipMemsetAsync");
        if (ok && nActive) {
            const TraceRange r_("sharded round: top-level episode + bin + count per owner");
            ta.active = active;
            ta.n_active = nActive;
            ta.may_park = rounds < nl ? 1u : 0u; // a ray pauses at most once per batching point
            ta.ray_aos = active ? dRayAos : nullptr; // the first episode (identity list) reads the caller's arrays and writes the records
            ta.ray_aos_out = active ? nullptr : dRayAos;
            ok = softHip(launchTopTraverse(ta, kind, std::max(1u, std::min(topMaxBlocks, (nActive + 255u) / 256u)), stream), "top-level episode")
                && softHip(launchLeafScan(dLeafCount, dLeafStart, dLeafFill, uint32_t(nl), dTotal, stream), "leaf scan")
                && softHip(launchBinByLeaf(active, dParked, nActive, uint32_t(nl), dLeafStart, dLeafFill, c->top.leaves, workOut, dWorkBvh, dWorkLeaf,
                               gridFor(c, nActive), stream), "binning")
                && softHip(launchRouteCount(dParked, nActive, c->leafOwnerDev, uint32_t(nl), world, dCounts, gridFor(c, nActive), stream), "route count");
        }
        // the one host synchronisation of the round: who sends how many records to whom (row r = rank r's counts per owner;
        // the diagonal counts the rays that parked at their own rank's points; behind each row: has that rank failed?)
        if (pending != PGPU_OK) {
            // (a failed rank promises nothing: a zero row with the flag up)
            std::vector<uint32_t> row(world + 1u, 0u);
            row[world] = 1u;
            (void)hipMemcpyAsync(dCounts, row.data(), row.size() * 4, hipMemcpyHostToDevice, stream);
            (void)hipStreamSynchronize(stream);
        }
        if (world > 1) {
            if (c->comm->transport.allgather_u32(c->comm->transport.user, dCounts, dMatrix, world + 1u, stream) != 0)
                return transportFail(c, "pgpu_trace_sharded: all-gather of the record counts failed");
            HIP_TRY(c, hipMemcpyAsync(gathered.data(), dMatrix, gathered.size() * 4, hipMemcpyDeviceToHost, stream));
        } else {
            HIP_TRY(c, hipMemcpyAsync(gathered.data(), dCounts, 8, hipMemcpyDeviceToHost, stream));
        }
        if (streamed && pending == PGPU_OK && c->leafHostBuf) // (the per-point counts of the binned list: they cut it into launch groups)
            HIP_TRY(c, hipMemcpyAsync(c->leafHostBuf, dLeafCount, nl * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        uint64_t total = 0;
        for (uint32_t r = 0; r < world; ++r) {
            for (uint32_t p = 0; p < world; ++p)
                total += matrix[size_t(r) * world + p] = gathered[size_t(r) * (world + 1u) + p];
            peerFailed += (r != rank && gathered[size_t(r) * (world + 1u) + world] != 0u) ? 1 : 0;
        }
        if (pending != PGPU_OK || peerFailed)
            break; // every rank sees the same flags: nobody is left inside a collective
        if (total == 0)
            break;
        ++rounds;
        uint32_t parked = 0, sendTotal = 0, recvTotal = 0;
        for (uint32_t p = 0; p < world; ++p) {
            parked += matrix[size_t(rank) * world + p];
            sendCounts[p] = p == rank ? 0u : matrix[size_t(rank) * world + p];
            recvCounts[p] = p == rank ? 0u : matrix[size_t(p) * world + rank];
            starts[p] = sendTotal;
            sendTotal += sendCounts[p];
            recvTotal += recvCounts[p];
        }
        leafVisits += parked;
        routed += sendTotal;
        if (recvTotal > recvCap) // (cannot happen: the other ranks do not have that many rays; a wrong count must not overrun the buffer)
            return fail(c, PGPU_ERR_INVALID, "pgpu_trace_sharded: a peer announced more records than it has rays");
        // Rays parked at this rank's own batching points: traced in place from the binned list (the kernel skips the work items
        // whose batching point has no BVH here -- those travel as records). With peers, on a second stream: it touches other rays'
        // results than the records and their merge do, so it runs next to the exchange and the trace of the received records.
        const bool fork = world > 1 && !streamed && c->auxStream && c->auxFork && c->auxJoin && (sendTotal || recvTotal);
        hipStream_t localStream = fork ? c->auxStream : stream;
        bool forked = false;
        if (parked) {
            const TraceRange r_("sharded round: in-place trace of the rays parked at this rank's points");
            if (fork) {
                forked = softHip(hipEventRecord(c->auxFork, stream), "hipEventRecord") && softHip(hipStreamWaitEvent(c->auxStream, c->auxFork, 0), "hipStreamWaitEvent");
                if (!forked)
                    localStream = stream;
            }
            ba.work_ray = workOut;
            ba.n_work = parked;
            ba.chunk = c->chunk;
            ba.ray_aos = dRayAos;
            ba.cursor = cursorLocal;
            uint32_t blocks = 1;
            if (streamed) {
                // this rank's points that have rays, up and down alternately (what the last round brought in last is used first)
                const uint32_t* const hCount = c->leafHostBuf;
                hLeafStart.resize(nl);
                visitOrder.clear();
                uint32_t run = 0;
                for (size_t l = 0; l < nl; ++l) {
                    hLeafStart[l] = run;
                    run += hCount[l];
                    if (hCount[l] != 0u && c->leafOwnerHost[l] == rank)
                        visitOrder.push_back(uint32_t(l));
                }
                if (rounds % 2 == 0)
                    std::reverse(visitOrder.begin(), visitOrder.end());
                ba.work_bvh = dWorkBvh;
                ba.work_tag = dWorkLeaf;
                soft(traceLeafListStreamed(c, ba, kind, tri_mode, deepest, visitOrder, hLeafStart.data(), hCount, cursorLocal, &c->spillB, stream));
            } else if (softHip(hipMemsetAsync(cursorLocal, 0, 256, localStream), "hipMemsetAsync") && soft(planBvh8Launch(c, deepest, parked, ba, &blocks, &c->spillB)))
                softHip(launchBvh8Trace(ba, kind, tri_mode, c->stats, blocks, localStream), "bottom-level trace");
            if (forked)
                softHip(hipEventRecord(c->auxJoin, c->auxStream), "hipEventRecord");
        }
        if (world > 1) {
            // (from here to the end of the round a failure is soft: the exchanges are still called with the promised counts)
            if (sendTotal) {
                const TraceRange r_("sharded round: pack records");
                softHip(launchRouteScatter(recordSrc, active, dParked, nActive, c->leafOwnerRemoteDev, uint32_t(nl), world, starts.data(), dFill, dSendRec, dSlot,
                            dSentLeaf, gridFor(c, nActive), stream), "record scatter");
            }
            // (closest hit: a record carries the ray's current t as tfar -- the in-place trace touches other rays only)
            {
                const TraceRange r_("sharded round: record exchange (all-to-all)");
                if (c->comm->transport.exchange(c->comm->transport.user, dSendRec, sendCounts.data(), dRecvRec, recvCounts.data(), 36, stream) != 0)
                    return transportFail(c, "pgpu_trace_sharded: record exchange failed");
            }
            if (recvTotal) {
                const TraceRange r_("sharded round: trace of the received records");
                soft(traceLeafRecords(c, kind, tri_mode, dRecvRec, recvTotal, dResOut, stream, instRows, c->cursor, &c->spill));
            }
            {
                const TraceRange r_("sharded round: result exchange (all-to-all)");
                if (c->comm->transport.exchange(c->comm->transport.user, dResOut, recvCounts.data(), dResBack, sendCounts.data(), resultBytes, stream) != 0)
                    return transportFail(c, "pgpu_trace_sharded: result exchange failed");
            }
            if (forked) // the merge and the next episode see the in-place trace's results
                softHip(hipStreamWaitEvent(stream, c->auxJoin, 0), "hipStreamWaitEvent");
            if (sendTotal)
                softHip(launchMergeResults(sendTotal, dResBack, dSlot, dSentLeaf, closest ? hits->t : nullptr, closest ? hits->obj : nullptr,
                            closest ? hits->leaf : nullptr, closest ? hits->u : nullptr, closest ? hits->v : nullptr, closest ? nullptr : any_hit,
                            gridFor(c, sendTotal), stream, dInst, closest && instRows ? 5u : 4u), "result merge");
            exchangeBytes += (uint64_t(sendTotal) + recvTotal) * (36 + resultBytes);
        }
        active = workOut;
        nActive = parked;
        flip ^= 1;
    }
    if (c->auxStream)
        (void)hipStreamSynchronize(c->auxStream);
    if (pending != PGPU_OK) {
        (void)hipStreamSynchronize(stream);
        return pending; // (pgpu_last_error holds what failed)
    }
    if (peerFailed) {
        (void)hipStreamSynchronize(stream);
        return fail(c, PGPU_ERR_PEER, "pgpu_trace_sharded: another rank of the job failed; every rank left the loop in the same round");
    }
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    c->timingStale = false;
    unsigned long long cnt[2] = { 0, 0 };
    HIP_TRY(c, hipMemcpyAsync(cnt, c->topCounters, sizeof(cnt), hipMemcpyDeviceToHost, stream));
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
    c->counters.records_routed = routed;
    c->counters.exchange_bytes = exchangeBytes;
    c->topCountersDirty = false;
    if (c->stats) {
        const int rs = readKernelStats(c);
        if (rs != PGPU_OK)
            return rs;
    }
    if (counters)
        *counters = c->counters;
    return PGPU_OK;
}

int pgpu_debug_fail_sharded(pgpu_ctx* c, int after_rounds)
{
    if (!c)
        return PGPU_ERR_INVALID;
    c->debugFailRound = after_rounds;
    return PGPU_OK;
}

int pgpu_film_allreduce(pgpu_ctx* c, unsigned long long* film, size_t n_pixels, void* stream_)
{
    if (!c || !film || n_pixels == 0 || n_pixels > 0x55555555u)
        return fail(c, PGPU_ERR_INVALID, "pgpu_film_allreduce: null argument or empty film");
    if (!c->comm || c->comm->world == 1)
        return PGPU_OK; // one rank: the film is the sum already
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    if (c->comm->transport.allreduce_u64(c->comm->transport.user, film, n_pixels * 3, stream) != 0)
        return transportFail(c, "pgpu_film_allreduce: all-reduce failed");
    return PGPU_OK;
}
}
