// Internal device-side data contract shared by the kernels and the C-ABI implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pandora_hip {

constexpr uint32_t PGPU_MISS_ID = 0xFFFFFFFFu;
constexpr int PGPU_ANY_ID = 1;
constexpr int PGPU_TRI_WATERTIGHT_ID = 1;

// One bottom-level BVH8 resident in HBM.
//   nodes: the reference's 256-byte nodes, verbatim (wive_bvh8.h:76-86).
//   tris : one 48-byte record per slot of the reference's leaf-index array, i.e. record s belongs to
//          leafIndex[s]: {p0.xyz, leaf-object id}, {p1.xyz, 0}, {p2.xyz, 0}. A leaf handle with index h
//          and count c therefore owns the c consecutive records starting at h -- the three pointer
//          hops of the reference (leaf index -> OfflineBVHLeaf -> index triple -> vertices,
//          wive_bvh8_impl.h:272-280, offline_bvh_cache.cpp:42-53) collapse into one contiguous read.
struct Bvh8Dev {
    const unsigned char* nodes;
    const unsigned char* tris;
    uint32_t root;
    uint32_t n_nodes;
    uint32_t n_slots;
    uint32_t depth;
    uint32_t needs_exact; // some child box is unordered or non-finite: always use the x86-exact slab test
    uint32_t n_inst;      // entries of `inst`
    const unsigned char* inst; // InstDev table of the instance leaves (nullptr: every leaf object is a triangle)
    const float* shade;        // 9 floats per leaf object: the triangle's vertices, for the benchmark integrator's bounce kernel
                               // (nullptr unless pgpu_set_keep_shading was on at upload)
};

// One instance (OfflineBVHLeaf::Instance, offline_bvh_cache.h:18-39): the BVH it shows and the world-to-local transform
// (rows 0..2 of Transform::m_inverseMatrix, row-major). The triangle record of an instance slot is a degenerate triangle
// that carries the leaf-object id in p0.w and 0x80000000 | index into this table in p1.w.
struct InstDev {
    float m[12];
    uint32_t child;
    uint32_t pad[3];
};

struct Bvh8TraceArgs {
    const Bvh8Dev* bvhs;
    const uint32_t* work_ray; // work item -> ray id (nullptr: identity)
    const uint32_t* work_bvh; // work item -> bvh id (nullptr: fixed_bvh)
    const uint32_t* work_tag; // work item -> value stored in leaf_out on a hit (nullptr: 0)
    uint32_t fixed_bvh;
    uint32_t n_bvhs;   // work items naming a bvh id beyond this are skipped (geometry not resident here)
    uint32_t n_work;
    // optional, pair layout only: the rays as 32-byte records {ox, oy, oz, tnear}{dx, dy, dz, tfar} by ray id (written by the
    // top-level kernel's first episode). A work list sorted by batching point visits ray ids in random order: the eight arrays
    // cost eight memory sectors per ray, the record one. tfar is still read from tfar_in (it is the ray's current t there).
    const float4* ray_aos;
    const uint32_t* n_work_dev; // optional, pair layout only: the real number of work items (<= n_work, which then only sizes the launch)
    const float *ox, *oy, *oz, *dx, *dy, *dz, *tnear, *tfar_in;
    float* t_out;
    uint32_t* obj_out;
    uint32_t* leaf_out;
    float *u_out, *v_out;
    uint32_t* any_out;
    uint32_t* inst_out; // instance (leaf-object id in the traced BVH) a hit went through, PGPU_MISS_ID = none; may be nullptr
    uint32_t instanced; // 1: some BVH of the table holds instance leaves -> the kernel variant that follows them (quad layout)
    uint32_t* cursor; // 8 work cursors (one per XCD group of blocks), 32 bytes apart, zeroed before launch
    uint32_t range_size; // work items per cursor
    uint32_t chunk;
    uint32_t init_miss; // 1: write miss records too (single level); 0: only touch rays that hit
    uint32_t stack_entries; // stack entries per ray kept in LDS (the window)
    uint32_t spill_entries; // entries per ray group in the global spill buffer (0: the window holds the worst case)
    unsigned long long* spill;
    uint32_t refill_min;    // refill idle ray groups once this many are idle
    uint32_t leaf_min;      // run the leaf (triangle) pass once this many ray groups of the wave wait for it
    uint32_t lanes_per_ray; // 4: quad kernel (16 rays per wave), 2: pair kernel (32 rays per wave)
    unsigned long long* stats;
};

// Top level: PauseableBVH4 nodes (112 bytes, pauseable_bvh4.h:74-98), its leaves (batching points)
// and the shared SVDAG array (sparse_voxel_dag.h:40-78).
struct SvdagDev {
    uint32_t root;
    float bmin[3];
    float invExtent;
    uint32_t levels; // height of this DAG (bounds the loop trips of a cast that makes progress)
    uint32_t img_offset, img_words; // this DAG's private image inside TopDev::dagImages (see there)
};
struct TopLeafDev {
    uint32_t bvh;
    uint32_t svdag; // PGPU_MISS_ID: no proxy geometry
};
// what an episode needs to know when it arrives at a batching point, in one 32-byte record per point (leaf -> svdag id -> descriptor
// were two dependent loads in a kernel that waits for little else)
struct LeafDagDev {
    uint32_t root;
    float bmin[3];
    float invExtent;
    uint32_t levels;
    uint32_t has_dag; // 0: no proxy geometry, the ray parks without a test
    uint32_t pad;
};
struct TopDev {
    const unsigned char* nodes;
    const TopLeafDev* leaves;
    const LeafDagDev* leafDags; // per batching point (rebuilt whenever the top level or the SVDAGs are uploaded)
    const uint32_t* dagWords;
    const SvdagDev* dags;
    uint32_t root, n_nodes, n_leaves, n_dags;
    uint32_t dag_levels; // height of the tallest SVDAG = entries of the cast's saved-parent stack
    uint32_t img_max_words; // the largest image (sizes the cast kernel's LDS)
    // Per-DAG images for the cast kernel, which stages the DAG of the batching point it works on in LDS: the nodes of ONE
    // DAG, breadth first from its root (index 0), in the array's own format (descriptor word + one offset per inner child)
    // but with offsets local to the image. The shared array hash-conses subtrees across DAGs, so one DAG's nodes are
    // scattered over it; the image is contiguous and at most kDagImageMaxWords long -- nodes that do not fit stay in the
    // shared array and are referred to as 0x80000000 | offset there.
    const uint32_t* dagImages;
};
constexpr uint32_t kDagImageMaxWords = 6144; // 24 KiB
constexpr uint32_t kDagGlobalFlag = 0x80000000u;
constexpr uint32_t kCastRequestFlag = 0x80000000u; // parked_leaf value of a ray waiting for its cast: flag | batching point
constexpr uint32_t kCastCulled = 0xFFFFFFFEu;      // parked_leaf value of a ray whose cast missed: it goes on with its traversal

struct TopTraceArgs {
    TopDev top;
    const uint32_t* active; // ray ids to advance (nullptr: identity)
    uint32_t n_active;
    const uint32_t* n_active_dev; // optional: the real number of work items (<= n_active, which then only sizes the launch)
    // the rays as 32-byte records by ray id (see Bvh8TraceArgs::ray_aos): read from ray_aos when set, else from the eight arrays;
    // ray_aos_out (the first episode of a trace, whose list is the identity): every ray that is read is also written as a record
    const float4* ray_aos;
    float4* ray_aos_out;
    const float *ox, *oy, *oz, *dx, *dy, *dz, *tnear, *tfar_in;
    const float* tfar_state; // closest-hit: current ray.tfar
    uint32_t* node;          // PauseableBVHInsertHandle.first  (pauseable_bvh.h:8)
    uint64_t* stack;         // PauseableBVHInsertHandle.second
    uint32_t* visits;        // may be nullptr
    uint32_t* any_out;       // any-hit: 1 once occluded
    uint32_t* parked_leaf;   // per work item of this round: batching point it parked at, or PGPU_MISS_ID
    // Termination guards (deviation 4): the reference's loop does not terminate on every ray -- when every
    // candidate child has tmin = +inf, blend(FLT_MAX, tmin) + horizontalMinIndex (pauseable_bvh4.h:297-299) picks a child
    // that is not a candidate and the XOR at :304 then sets its stack bit.
    uint32_t step_cap; // loop trips per episode before the ray is made to leave the top level
    uint32_t may_park; // 0: a ray that reaches a batching point leaves instead of pausing (it has paused n_leaves times)
    uint32_t* leaf_count;
    unsigned long long* counters; // [0] svdag tests, [1] svdag culled
    // Deferred casts (launchTopTraverse with defer = true, then scan + bin of the requests + launchSvdagCast): the episode
    // stops at a batching point that has proxy geometry instead of casting through it -- token saved, parked_leaf = the point
    // with kCastRequestFlag, one count per point in req_leaf_count. only_culled: advance only the positions whose cast missed
    // in the previous pass (parked_leaf == kCastCulled); the others are left as they are.
    uint32_t only_culled;
    uint32_t* req_leaf_count;
    // topTraverseLanesKernel (lanes refilled as their rays finish) instead of the lockstep kernel: idle lanes of a wave before
    // it fetches new rays (0 = use the lockstep kernel), cast trips per turn of the cast phase
    uint32_t lanes_refill_min, lanes_cast_trips;
    uint32_t phased; // the lockstep kernel with its SVDAG casts gathered into phases of the whole wave (top_kernels.hip)
};

// The SVDAG casts of one pass as a kernel of their own. Requests (positions of the round's list) arrive sorted by batching
// point; a block takes a contiguous slice of them and, for every batching point in its slice, stages that point's DAG image in
// LDS and runs the casts from there -- every lane takes the next request of the run when its cast is done, so a wave does
// uniform work and no cast touches L1 / L2 (inside a top-level episode every trip of every lane reads 4 bytes of some DAG
// somewhere: a 64-byte line per lane and trip, which is what bounds the episode kernel). A hit leaves the ray parked
// (parked_leaf = the point, counted in leaf_count, visits + 1); a miss sets parked_leaf = kCastCulled.
struct CastArgs {
    TopDev top;
    const uint32_t* active; // position -> ray id (nullptr: identity)
    const float *ox, *oy, *oz, *dx, *dy, *dz;
    uint32_t* parked_leaf;
    uint32_t* visits; // may be nullptr
    const uint32_t* req_list;  // positions, sorted by batching point
    const uint32_t* req_start; // per batching point: first request
    const uint32_t* req_count; // per batching point: number of requests
    const uint32_t* unit_start; // per batching point: first work unit (svdagPlanKernel)
    uint32_t* plan;             // [0] work units of the pass, [1] the unit cursor
    uint32_t* leaf_count;
    unsigned long long* counters; // [0] svdag tests, [1] svdag culled
};

bool topNodesFitLds(uint32_t nNodes);
hipError_t launchTopTraverse(const TopTraceArgs& a, int kind, uint32_t blocks, hipStream_t stream, bool defer = false);
hipError_t launchSvdagCast(const CastArgs& a, uint32_t blocks, hipStream_t stream);
hipError_t launchSvdagPlan(const uint32_t* count, uint32_t* start, uint32_t* fill, uint32_t* unitStart, uint32_t n, uint32_t* plan, hipStream_t stream);
size_t svdagCastLdsBytes(const TopDev& top);
size_t topTraverseLdsBytes(const TopDev& top);
// resetCount (= count, or null): the counts are zeroed once they have been scanned; cursor64 (or null): 64 words cleared as well;
// totalHost (or null): pinned host memory that receives the total too
hipError_t launchLeafScan(const uint32_t* count, uint32_t* start, uint32_t* fill, uint32_t n, uint32_t* total, hipStream_t stream,
    uint32_t* resetCount = nullptr, uint32_t* cursor64 = nullptr, uint32_t* totalHost = nullptr);
// requests = true: bins the positions whose parked_leaf carries kCastRequestFlag (by the batching point under the flag)
hipError_t launchBinByLeaf(const uint32_t* active, const uint32_t* parkedLeaf, uint32_t nActive, uint32_t nLeaves, const uint32_t* leafStart,
    uint32_t* leafFill, const TopLeafDev* leaves, uint32_t* workRay, uint32_t* workBvh, uint32_t* workLeaf, uint32_t blocks, hipStream_t stream,
    bool requests = false, const uint32_t* nActiveDev = nullptr);
// nextActive = the entries of `active` (nullptr: identity) whose parkedLeaf names a batching point, in order; `slices` (<= 4096)
// fixed slices of the list are counted, scanned (sliceCount / sliceStart / scratchFill: `slices` words each) and written
hipError_t launchCompactParked(const uint32_t* active, const uint32_t* parkedLeaf, uint32_t n, const uint32_t* nDev, uint32_t nLeaves, uint32_t* sliceCount,
    uint32_t* sliceStart, uint32_t* scratchFill, uint32_t* scratchTotal, uint32_t* nextActive, uint32_t slices, hipStream_t stream);
hipError_t launchLeafToBvh(const uint32_t* workLeaf, const TopLeafDev* leaves, uint32_t nLeaves, uint32_t* workBvh, uint32_t n, uint32_t blocks, hipStream_t stream);
hipError_t launchInitState(uint32_t n, uint32_t root, uint32_t* node, uint64_t* stack, const float* tfarIn, float* tState, uint32_t* obj,
    uint32_t* leaf, float* u, float* v, uint32_t* visits, uint32_t* anyOut, uint32_t blocks, hipStream_t stream);

// route_kernels.hip: record packing for the leaf-sharded loop
uint32_t routeMaxRanks();
hipError_t launchRouteCount(const uint32_t* parkedLeaf, uint32_t n, const uint32_t* owner, uint32_t nLeaves, uint32_t world, uint32_t* counts,
    uint32_t blocks, hipStream_t stream);
hipError_t launchRouteScatter(const float* const rays[8], const uint32_t* active, const uint32_t* parkedLeaf, uint32_t n, const uint32_t* owner,
    uint32_t nLeaves, uint32_t world, const uint32_t* startsHost, uint32_t* fill, float* records, uint32_t* sentSlot, uint32_t* sentLeaf,
    uint32_t blocks, hipStream_t stream);
hipError_t launchUnpackRecords(const float* records, uint32_t k, float* soa, uint32_t* workLeaf, float* t, uint32_t* obj, uint32_t* anyHit,
    uint32_t blocks, hipStream_t stream);
// (inst != nullptr / rowFloats == 5: closest-hit rows carry the instance a hit went through as a fifth value)
hipError_t launchPackResults(uint32_t k, const float* t, const uint32_t* obj, const float* u, const float* v, const uint32_t* anyHit, float* results,
    uint32_t blocks, hipStream_t stream, const uint32_t* inst = nullptr);
hipError_t launchMergeResults(uint32_t k, const float* results, const uint32_t* sentSlot, const uint32_t* sentLeaf, float* t, uint32_t* obj,
    uint32_t* leaf, float* u, float* v, uint32_t* anyHit, uint32_t blocks, hipStream_t stream, uint32_t* inst = nullptr, uint32_t rowFloats = 4);
hipError_t launchKeyCount(const uint32_t* keys, uint32_t n, uint32_t nKeys, uint32_t* counts, uint32_t blocks, hipStream_t stream);

size_t bvh8TraceLdsBytes(uint32_t stackEntries, uint32_t lanesPerRay);
inline uint32_t stackEntriesForDepth(uint32_t depth)
{
    const uint32_t need = 7u * depth + 8u; // each inner visit pops 1 and pushes <= 8 (wive_bvh8_impl.h:97-102)
    return need < 16u ? 16u : ((need + 7u) & ~7u);
}
// LDS window of the per-ray stack: the whole worst case when it is small, otherwise a window sized so that
// 7 blocks fit a CU; deeper entries spill to global memory.
inline uint32_t stackWindow(uint32_t need, int requested)
{
    const uint32_t cap = requested > 0 ? uint32_t(requested) : 12u; // (12 entries = 96 bytes per ray: 7 blocks of the quad kernel fill 154 of a CU's 160 KiB)
    return need < cap ? need : cap;
}
// blocks of 256 threads per CU: requested, or (0) as many as LDS (160 KiB) and registers allow
inline uint32_t blocksPerCU(int requested, uint32_t stackEntries, uint32_t lanesPerRay)
{
    const uint32_t byLds = uint32_t((160u * 1024u) / ((bvh8TraceLdsBytes(stackEntries, lanesPerRay) + 1023u) & ~size_t(1023))); // LDS is handed out in coarse granules
    const uint32_t byRegs = lanesPerRay == 2u ? 4u : 7u; // <= 128 / <= 72 VGPRs
    const uint32_t cap = byLds < byRegs ? byLds : byRegs;
    if (requested <= 0)
        return cap < 1u ? 1u : cap;
    return uint32_t(requested) < cap ? uint32_t(requested) : cap;
}
hipError_t launchBvh8Trace(const Bvh8TraceArgs& args, int kind, int triMode, bool stats, uint32_t blocks, hipStream_t stream);

// Streamed leaves (top_abi.hip): what one launch group brings into the HBM leaf cache. 64-byte records, written by the host into
// pinned memory and copied to the device in one piece: first the segments of the gather kernel, then the BVH-table entries that
// changed (images that arrived, images that were evicted).
struct LeafCopySeg {
    const void* src; // pinned host memory
    void* dst;       // cache arena
    uint64_t bytes;  // a multiple of 16
    uint64_t chunk0; // index of the segment's first 16 KiB chunk among all chunks of the group
    uint64_t pad[4];
};
struct LeafTableFix {
    Bvh8Dev entry;
    uint32_t id;
    uint32_t group; // the launch group (its low 32 bits) an arriving image belongs to; 0 for an image that leaves
};
static_assert(sizeof(LeafCopySeg) == 64 && sizeof(LeafTableFix) <= 64, "one 64-byte record each");
hipError_t launchLeafGather(const LeafCopySeg* segs, uint32_t nSegs, uint32_t nChunks, uint32_t blocks, hipStream_t stream);
hipError_t launchAddCounts(uint32_t* count, const uint32_t* add, uint32_t n, hipStream_t stream);
hipError_t launchSplitByLeaf(const uint32_t* workRay, const uint32_t* leafStart, const uint32_t* leafCount, const uint32_t* dst, uint32_t nLeaves, uint32_t nItems,
    uint32_t* nextActive, uint32_t* heldRay, uint32_t* heldLeaf, uint32_t blocks, hipStream_t stream);
// (arrivalGroup, optional: per BVH id the launch group it arrived with, see launchLeafTableView)
hipError_t launchLeafTableFix(const unsigned char* fixes64, uint32_t nFixes, Bvh8Dev* table, hipStream_t stream, uint32_t* arrivalGroup = nullptr);
// view[i] = table[i], except that a BVH a batching point names (leafFlag[i] != 0) is there only if it arrived with launch group
// `group`: the table a launch uses that must trace the images its group brought in and nothing that an earlier launch of the round
// has traced already (prototypes of instances, which no batching point names, stay visible)
hipError_t launchLeafTableView(const Bvh8Dev* table, const uint32_t* leafFlag, const uint32_t* arrivalGroup, uint32_t group, Bvh8Dev* view, uint32_t n,
    hipStream_t stream);

} // namespace pandora_hip
