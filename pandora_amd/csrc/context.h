// Device context behind the opaque pgpu_ctx of include/pandora_hip.h.
#pragma once
#include "kernels.h"
#include "pandora_hip.h"
#include <deque>
#include <string>
#include <vector>

struct HostMesh {
    std::vector<float> pos;
    std::vector<uint32_t> idx;
};

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
};

// A bottom-level BVH kept in pinned host memory and copied into the HBM leaf cache when rays reach it
// (BASELINE config 5; the reference's CachedBVH::doMakeResident / doEvict, offline_bvh_cache.cpp:127-148).
struct StagedBvh {
    unsigned char* host = nullptr; // pinned: the 256-byte nodes, then (256-aligned) the 48-byte triangle records, then the instance table
    size_t nodeBytes = 0, triBytes = 0, instBytes = 0;
    bool resident = false;
    size_t offset = 0;      // where in the cache arena (resident only)
    int lastLaunch = -1;    // index into launchEvents: recorded on the compute stream after the last launch that read the image
    uint64_t pinGroup = 0;  // the launch group (a counter) that needs the image: it is not evicted while that group is being formed
    size_t imageBytes() const { return nodeBytes + triBytes + instBytes; }
};

struct Comm;

struct pgpu_ctx {
    int device = 0;
    int numCUs = 256;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    float lastKernelMs = 0.0f;
    bool timingStale = false; // the last entry point recorded no timing events (stream capture): ev0 / ev1 belong to an earlier call
    bool stats = false;
    int blocksPerCU = 0; // 0 = as many as LDS / registers allow
    uint32_t chunk = 0; // work items a wave draws at a time, 0 = automatic
    uint32_t refillMin = 0; // 0: automatic
    uint32_t lanesPerRay = 0; // bottom-level kernel layout: 4 = 16 rays per wave, 2 = 32 rays per wave, 0 = automatic (planBvh8LaunchShape)
    uint32_t leafMin = 0; // 0: automatic
    int stackWindow = 0;     // LDS stack entries per ray, 0 = automatic
    DeviceBuffer spill;      // global overflow of the per-ray stacks
    unsigned char* staticScratch = nullptr; // this device's copy of the static scratch array of noalloc.hip (resolved at pgpu_init)

    int svdagPolicy = 0;       // pgpu_set_svdag_policy: 0 = every batching point with an SVDAG is tested (the reference), 1 = only the points whose
                               // geometry is not in this GPU's HBM (staged in host memory, or owned by another rank)
    uint32_t deferMin = 0;     // pgpu_set_svdag_defer: two-level passes of at least this many rays (0 = never) run their SVDAG casts as
    uint32_t deferPasses = 2;  // a kernel of their own this many times per round; the episode kernel finishes the rest with casts inside
    bool keepShading = false;  // uploads also keep the vertices per leaf object on the device (pgpu_set_keep_shading)
    bool hasInstances = false; // some BVH holds instance leaves: the bottom-level launches use the kernel variant that follows them
    std::vector<HostMesh> meshes;
    std::vector<pandora_hip::Bvh8Dev> bvhs; // host mirror of the device table
    std::vector<void*> owned;               // every hipMalloc'd scene buffer
    pandora_hip::Bvh8Dev* bvhTable = nullptr;
    size_t bvhTableCap = 0;
    bool bvhTableDirty = true;

    uint32_t* cursor = nullptr;          // work cursors (one u32 per launch slot)
    unsigned long long* statsDev = nullptr; // inner visits, leaf pops, tri tests
    pgpu_counters counters {};

    // staging for host-resident ray batches: 8 ray arrays + 5 hit arrays + any/visits
    DeviceBuffer stageRays, stageHits;

    // two-level state (top_kernels.hip)
    pandora_hip::TopDev top {};
    bool haveTop = false;
    bool topHasRemoteLeaves = false;
    bool topCountersDirty = false; // pgpu_top_episode ran since the counters were last read
    DeviceBuffer work; // per-ray traversal state + queues, grown on demand
    DeviceBuffer blockWork; // scratch of the building-block entry points (work item -> bvh id)
    DeviceBuffer routeWork; // scratch of pgpu_trace_leaf_records
    DeviceBuffer compactWork; // scratch of pgpu_compact_paths (flags, places, scan storage)
    unsigned long long* topCounters = nullptr; // SVDAG tests / culled of pgpu_top_episode calls (device)

    // several GPUs (sharded_abi.hip)
    Comm* comm = nullptr;
    std::vector<uint32_t> leafOwnerHost;
    uint32_t* leafOwnerDev = nullptr;       // owner rank per batching point
    uint32_t* leafOwnerRemoteDev = nullptr; // the same with this rank's own points marked "no destination"
    DeviceBuffer shardWork, shardSend, shardRecv;
    // pgpu_trace_sharded traces the rays parked at this rank's own batching points on a second stream, next to the record exchange
    // and the trace of the received records: that launch has cursors and a stack-overflow buffer of its own
    hipStream_t auxStream = nullptr;
    hipEvent_t auxFork = nullptr, auxJoin = nullptr;
    DeviceBuffer spillB;
    int debugFailRound = -1; // pgpu_debug_fail_sharded: this rank fails (softly) when it has completed that many rounds

    // streamed leaves
    std::vector<StagedBvh> staged;               // indexed by bvh id; host == nullptr for HBM-resident ones
    std::vector<pandora_hip::TopLeafDev> topLeavesHost;
    size_t nStaged = 0, cacheBudget = 0;
    // The cache is a ring: images of different sizes are written one behind the other and the oldest ones make room (first in,
    // first out). `ring` = the resident BVH ids, oldest first.
    unsigned char* cacheArena = nullptr;
    size_t cacheBytes = 0, ringWrite = 0;
    std::deque<uint32_t> ring;
    uint64_t groupClock = 0;
    uint32_t leafHoldBytesPerRay = 128; // pgpu_set_leaf_hold: an image that is not in the cache is loaded once its point holds a ray per this many bytes (0: always)
    uint32_t leafInPlaceBytesPerRay = 128; // pgpu_set_leaf_in_place: flushed queues with fewer rays than one per this many bytes of image are traced in host memory (0: never)
    float leafPrefetchBytesPerRay = 0.0f; // waiting points are loaded early up to this many bytes of image per moving ray and round (PANDORA_LEAF_PREFETCH)
    float leafDrainRatio = 1.0f; // waiting queues are flushed as well once they hold more than this many times the rays that still move (PANDORA_LEAF_DRAIN)
    int leafCopyMode = 0;   // pgpu_set_leaf_copy: 0 = one gather kernel per launch group reads the pinned images, 1 = one hipMemcpyAsync per image
    void* copyTableHost = nullptr; // pinned: the gather kernel's segments and the BVH-table entries that changed, one run per launch group
    void* copyTableDev = nullptr;
    size_t copyTableCap = 0, copyTableNext = 0; // in 64-byte records
    uint32_t* leafHostBuf = nullptr; // pinned: per-point counts read back every round, and the two per-point arrays the host sends (3 n_leaves + 64 words)
    size_t leafHostCap = 0;
    // the launches that trace only what their group brought in (pgpu_trace's second pass of a round, see top_abi.hip) use a view
    // of the BVH table: [view table][per id: named by a batching point?][per id: the group it arrived with], bvhTableCap entries each
    DeviceBuffer tableView;
    hipStream_t copyStream = nullptr;
    hipEvent_t copyDone = nullptr;
    std::vector<pandora_hip::SvdagDev> dagsHost; // host copy of the descriptors (for the per-point records, LeafDagDev)
    // top-level kernel (pgpu_set_top_variant): 0 automatic, 1 lockstep waves, 2 lanes refilled as their rays finish
    int topVariant = 0;
    uint32_t topRefillMin = 16, topCastTrips = 4;
    // pgpu_trace queues rounds ahead of the host (pgpu_set_round_lookahead): per-round parked counts in pinned memory
    bool roundLookahead = true;
    uint32_t* roundCounts = nullptr;
    size_t roundCountsCap = 0;
    hipEvent_t roundEv[2] = { nullptr, nullptr };
    // pgpu_trace runs a large batch as parts side by side (pgpu_set_batch_parts): the contexts of the parts beyond the first.
    // They share this context's scene and own only their streams and work buffers (top_abi.hip)
    int batchParts = 0; // 0 = automatic
    std::vector<pgpu_ctx*> lanes;
    hipEvent_t lanesFork = nullptr;
    std::vector<hipEvent_t> launchEvents; // ring: one event per bottom-level launch of the streaming path (a later record of the
    size_t launchEventNext = 0;           // same ring entry is a later point of the compute stream, so waiting on it stays safe)
};

struct pgpu_host_svdags;
namespace pandora_hip {
// wraps finished node / leaf-index arrays as the object pgpu_build_bvh8 returns (pandora_amd/host/bvh_builder.cpp)
pgpu_host_bvh8* makeHostBvh8(const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices, uint32_t root, int depth);
// wraps a finished DAG array as the object pgpu_build_svdags returns (pandora_amd/host/svdag_builder.cpp)
pgpu_host_svdags* makeHostSvdags(std::vector<uint32_t>&& words, std::vector<pgpu_svdag_desc>&& descs, size_t wordsBeforeCompression);
// Shared by every translation unit of the C ABI (defined in abi.hip).
// fail: records the message pgpu_last_error returns and hands back `code`.
int fail(pgpu_ctx* c, int code, const char* what, hipError_t e = hipSuccess);
// ensure: grows a scratch buffer (never shrinks); the old contents are lost.
int ensure(pgpu_ctx* c, DeviceBuffer& b, size_t bytes);
// syncBvhTable: re-uploads the device copy of c->bvhs if an entry changed since the last upload.
int syncBvhTable(pgpu_ctx* c);
// dropLeafCache: waits for the device, empties and frees the HBM cache of staged BVHs (it is rebuilt at the next pgpu_trace).
int dropLeafCache(pgpu_ctx* c);
inline size_t alignUp(size_t v) { return (v + 255) & ~size_t(255); }
inline uint32_t deepestBvh(const pgpu_ctx* c)
{
    uint32_t d = 0;
    for (const Bvh8Dev& b : c->bvhs)
        d = b.depth > d ? b.depth : d;
    return d;
}
// Fills the stack window / spill fields of a bottom-level launch for trees of the given depth and returns
// the grid size for n_work items (abi.hip).
int readKernelStats(pgpu_ctx* c);
// (spill: the overflow buffer to use and grow; default = the context's)
int planBvh8Launch(pgpu_ctx* c, uint32_t depth, size_t n_work, Bvh8TraceArgs& args, uint32_t* blocks, DeviceBuffer* spill = nullptr);
// The same plan without the spill buffer: args.spill stays null, args.spill_entries says how many 8-byte entries every
// ray group of every block needs (the caller provides the memory).
void planBvh8LaunchShape(const pgpu_ctx* c, uint32_t depth, size_t n_work, Bvh8TraceArgs& args, uint32_t* blocks);
int initStaticScratch(pgpu_ctx* c); // noalloc.hip
// top_abi.hip: the HBM cache of staged BVHs (built on first use), and the bottom level over a work list sorted by batching point
// when some BVHs are staged -- launch groups by what fits the cache together (see there)
int prepareLeafCache(pgpu_ctx* c);
// (arrivalsOnly: every launch sees, of the BVHs batching points name, only the images its own group brought in)
int traceLeafGroups(pgpu_ctx* c, Bvh8TraceArgs ba, int kind, int tri_mode, uint32_t deepest, const std::vector<uint32_t>& visitOrder, const uint32_t* leafStart,
    const uint32_t* leafCount, uint32_t* cursor, DeviceBuffer* spill, hipStream_t stream, bool arrivalsOnly = false);
// the rays of the given batching points (staged, not in the cache) through their images where they lie in pinned host memory: one
// launch over the stretch of the list the points span, through a view of the BVH table in which only they are visible
int traceLeavesInPlace(pgpu_ctx* c, Bvh8TraceArgs ba, int kind, int tri_mode, uint32_t deepest, const std::vector<uint32_t>& points, const uint32_t* leafStart,
    const uint32_t* leafCount, uint32_t* cursor, DeviceBuffer* spill, hipStream_t stream);
int prepareLeafTableView(pgpu_ctx* c, hipStream_t stream); // (once per trace, before traceLeavesInPlace / traceLeafListStreamed)
int traceLeafListStreamed(pgpu_ctx* c, const Bvh8TraceArgs& ba, int kind, int tri_mode, uint32_t deepest, const std::vector<uint32_t>& points, const uint32_t* leafStart,
    const uint32_t* leafCount, uint32_t* cursor, DeviceBuffer* spill, hipStream_t stream);
// route_abi.hip. instRows: closest-hit result rows have five values (the instance of the hit last), see launchPackResults
// (cursor / spill: for a launch that runs next to another bottom-level launch of the same context; default = the context's)
int traceLeafRecords(pgpu_ctx* c, int kind, int tri_mode, const float* records, size_t k, float* results, hipStream_t stream, bool instRows = false,
    uint32_t* cursor = nullptr, DeviceBuffer* spill = nullptr);
// grows the scratch of traceLeafRecords (and the given overflow buffer) for up to k records, so that the call itself cannot fail on memory
int reserveLeafRecords(pgpu_ctx* c, size_t k, DeviceBuffer* spill);
void destroyComm(pgpu_ctx* c); // sharded_abi.hip
void destroyLanes(pgpu_ctx* c); // top_abi.hip
constexpr size_t kBatchPartsAutoMin = size_t(1) << 21; // batches of at least this many rays are cut in two unless told otherwise
}

#define HIP_TRY(ctx, call)                                            \
    do {                                                              \
        hipError_t e_ = (call);                                       \
        if (e_ != hipSuccess)                                         \
            return pandora_hip::fail((ctx), PGPU_ERR_HIP, #call, e_); \
    } while (0)
