/*
 * pandora_hip.h -- C ABI of libpandora_hip.so, the MI355X (gfx950) implementation of Pandora's
 * ray-traversal hot path. This is the only way host code reaches the GPU.
 *
 * The reference (mathijs727/pandora) has no FFI layer; its seam is the C++ interface
 *     AccelerationStructure<HitRayState, AnyHitRayState>::intersect / intersectAny
 *         projects/pandora/include/pandora/traversal/acceleration_structure.h:6-11
 * implemented by OfflineBatchingAccelerationStructure (+Builder)
 *         projects/pandora/include/pandora/traversal/offline_batching_acceleration_structure.h:30-126
 * The header-only adapter include/pandora_amd/gpu_acceleration_structure.hpp keeps that interface
 * and is the intended caller of everything below. Each entry point names what it replaces.
 *
 * Conventions: every function returns PGPU_OK (0) or a negative PGPU_ERR_* code and never throws
 * (the reference reports errors by C++ exceptions, utility/error_handling.h:7-37);
 * pgpu_last_error() returns a human-readable message for the last failure on that context.
 * All pointers are plain host pointers unless a parameter says "device". No torch types.
 */
#ifndef PANDORA_HIP_H
#define PANDORA_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    PGPU_OK = 0,
    PGPU_ERR_INVALID = -1,     /* bad argument (null pointer, out-of-range id, size mismatch) */
    PGPU_ERR_HIP = -2,         /* a HIP runtime call failed; see pgpu_last_error */
    PGPU_ERR_SINGLE_LEAF = -3, /* top level needs >= 2 batching points (pauseable_bvh4.h:180) */
    PGPU_ERR_DEPTH = -4,       /* tree deeper than the traversal stacks allow */
    PGPU_ERR_NO_SCENE = -5,    /* trace called before the structure it needs was uploaded */
    PGPU_ERR_NO_DEVICE = -6,
    PGPU_ERR_PEER = -7         /* pgpu_trace_sharded: another rank of the job failed; all ranks left the loop in the same round */
};

enum { PGPU_CLOSEST = 0, PGPU_ANY = 1 };           /* intersect / intersectAny */
enum { PGPU_TRI_MT = 0, PGPU_TRI_WATERTIGHT = 1 }; /* triangle_intersect.cpp:425-462 / :340-423 */
enum { PGPU_MEM_HOST = 0, PGPU_MEM_DEVICE = 1 };   /* where ray / hit arrays live */
#define PGPU_MISS 0xFFFFFFFFu
#define PGPU_NO_SVDAG 0xFFFFFFFFu
#define PGPU_NO_BVH 0xFFFFFFFFu /* pgpu_leaf.bvh_id of a batching point whose geometry lives on another GPU */

typedef struct pgpu_ctx pgpu_ctx;

/* ------------------------------------------------------------------ host-side construction
 * Replaces the rtcBuildBVH calls of wive_bvh8_build8_impl.h:87-104 (bottom level) and
 * pauseable_bvh4.h:160-176 (top level) and the scene cut of batching.cpp:38-265. Output arrays are
 * in the reference's layouts: 256-byte nodes (wive_bvh8.h:76-86), u32 leaf-index array and 3+29 bit
 * handles (wive_bvh8_impl.h:333-382); 112-byte nodes (pauseable_bvh4.h:74-98). No GPU needed. */
typedef struct pgpu_host_bvh8 pgpu_host_bvh8;
typedef struct pgpu_host_bvh4 pgpu_host_bvh4;

/* BVH8 over the triangles prim_ids[0..n_prims) of an indexed mesh (prim_ids NULL = all of them,
 * n_prims = n_triangles). Leaf object j is triangle prim_ids[j]. max_leaf in 1..4. */
int pgpu_build_bvh8(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* prim_ids, size_t n_prims, uint32_t max_leaf, pgpu_host_bvh8** out);
const void* pgpu_host_bvh8_nodes(const pgpu_host_bvh8*, size_t* n_nodes);
const uint32_t* pgpu_host_bvh8_leaf_indices(const pgpu_host_bvh8*, size_t* n_indices);
uint32_t pgpu_host_bvh8_root(const pgpu_host_bvh8*);
int pgpu_host_bvh8_depth(const pgpu_host_bvh8*); /* nested inner nodes on the longest path */
void pgpu_host_bvh8_free(pgpu_host_bvh8*);
/* The BVH8s of many triangle groups (batching points) at once, built on the GPU of `ctx` (pandora_amd/csrc/bvh_build.hip; the
 * reference builds one tree per batching point with Embree on the CPU, wive_bvh8_build8_impl.h:87-104, offline_bvh_cache.cpp:273-309):
 * Morton order per group, one binary tree over all groups by locally-ordered clustering along that order (boxes and the collapse's
 * cost tables made with the nodes; PANDORA_BVH8_DEVICE_TREE=lbvh: a radix tree with a bottom-up pass instead), collapse into 8-wide
 * nodes by those tables, the reference's 256-byte nodes and octant words written by the device. Group g = triangles group_prims[group_offsets[g] .. group_offsets[g+1])
 * of the mesh (host arrays; offsets start at 0 and increase); out[g] receives the same kind of object pgpu_build_bvh8 returns
 * (leaf object j of a group = its j-th triangle) and is freed with pgpu_host_bvh8_free. On the benchmark atrium a diffuse
 * ray pays 6 % more node and leaf visits in such a tree than in pgpu_build_bvh8's (4 % fewer rays per second); the build costs 10 ms
 * of GPU time for 11.4 M triangles in 256 groups (the host builder: 0.6 s on 16 threads). Node numbering is breadth first per group and does
 * not depend on the run or on the other groups of the call. A group whose device tree would nest deeper than 12 inner nodes is
 * built by pgpu_build_bvh8 instead; *n_host_built (may be NULL) says how many were.
 * pgpu_last_kernel_ms = device time of the build. max_leaf in 1..4, at most 2^20 groups and 2^31 - 2 triangles in all. */
int pgpu_build_bvh8_device(pgpu_ctx*, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* group_prims, const uint64_t* group_offsets, size_t n_groups, uint32_t max_leaf, pgpu_host_bvh8** out,
    uint32_t* n_host_built);
/* The same build with the trees left where they were made (needs no pgpu_upload_mesh): nodes, 48-byte triangle records and, with
 * pgpu_set_keep_shading, the vertices per leaf object stay in HBM and join the context's BVH table as if pgpu_upload_bvh8 had been
 * called for every group with leaf objects (mesh, group_prims[j]); bvh_ids[g] receives group g's id. Nothing of the trees crosses
 * PCIe. A group too deep for the device builder is built by pgpu_build_bvh8 and uploaded. */
int pgpu_build_upload_bvh8_device(pgpu_ctx*, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* group_prims, const uint64_t* group_offsets, size_t n_groups, uint32_t max_leaf, uint32_t* bvh_ids,
    uint32_t* n_host_built);
/* A BVH of the context (resident or staged) back in the arrays pgpu_upload_bvh8 takes: sizes first, then nodes256 (n_nodes x 256
 * bytes) and leaf_indices (n_indices words). For trees built on the device, for checks, and for pgpu_bvh8_serialize. */
int pgpu_bvh8_info(pgpu_ctx*, uint32_t bvh_id, size_t* n_nodes, size_t* n_indices, uint32_t* root_handle, int* depth);
int pgpu_download_bvh8(pgpu_ctx*, uint32_t bvh_id, void* nodes256, uint32_t* leaf_indices);
/* pgpu_partition_prims (below) on the GPU of `ctx`: the mesh in Morton order, a binary radix tree over it, and the subtrees of at most
 * max_prims triangles under a parent that holds more -- the cut of detail::createSubScenes (batching.cpp:38-265) on an LBVH. Groups are
 * numbered along the Morton curve; pgpu_last_kernel_ms = device time. */
int pgpu_partition_prims_device(pgpu_ctx*, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    size_t max_prims, uint32_t* group_of_prim, uint32_t* n_groups);
/* BVH8 over arbitrary leaf objects given by their bounds (n_objects x {min xyz, max xyz}); leaf object j = box j. An object
 * with alone[j] != 0 gets a leaf of its own (instances; alone may be NULL). What LRUBVHSceneCache::createBVH builds over
 * the mix of instanced sub-scenes and triangles of a sub-scene (offline_bvh_cache.cpp:273-309). */
int pgpu_build_bvh8_boxes(const float* bounds6, size_t n_objects, const uint8_t* alone, uint32_t max_leaf, pgpu_host_bvh8** out);

/* BVH4 over batching-point bounds (n_leaves x {min xyz, max xyz}); leaf_order[k] = input leaf stored
 * at final leaf slot k (the reference moves leaves into m_leafs in this order, :461-467). */
int pgpu_build_top_bvh4(const float* leaf_bounds, size_t n_leaves, pgpu_host_bvh4** out);
const void* pgpu_host_bvh4_nodes(const pgpu_host_bvh4*, size_t* n_nodes);
const uint32_t* pgpu_host_bvh4_leaf_order(const pgpu_host_bvh4*, size_t* n_leaves);
uint32_t pgpu_host_bvh4_root(const pgpu_host_bvh4*);
int pgpu_host_bvh4_depth(const pgpu_host_bvh4*);
void pgpu_host_bvh4_free(pgpu_host_bvh4*);

/* Groups triangles into batching points of <= max_prims primitives (primitivesPerBatchingPoint). */
int pgpu_partition_prims(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    size_t max_prims, uint32_t* group_of_prim, uint32_t* n_groups);

/* The same cut at the level the reference makes it -- over scene objects (shapes, instanced sub-scenes) given by their bounds
 * and primitive counts: detail::createSubScenes, batching.cpp:38-265 (BVH2 over the objects, cut where a subtree holds
 * <= primitivesPerBatchingPoint primitives, :161-224). weights may be NULL (1 each). An object heavier than max_weight ends
 * up alone: split such shapes first, as the reference does with splitLargeSceneObjects at primitivesPerBatchingPoint / 8
 * (batching.cpp:358-422, offline_batching_acceleration_structure.cpp:57) -- pgpu_partition_prims over the shape's triangles. */
int pgpu_partition_objects(const float* bounds6, const uint64_t* weights, size_t n_objects, uint64_t max_weight, uint32_t* group_of_object,
    uint32_t* n_groups);

/* Voxelises each group of triangles into a resolution^3 grid over the group's bounds
 * (TriangleShape::voxelize, triangle_intersect.cpp:224-329), builds one SVO per group
 * (sparse_voxel_dag.cpp:39-133) and merges them into one shared DAG array (compressDAGs, :135-226).
 * group_prims/group_offsets: CSR lists of triangle ids per group. */
typedef struct {
    uint32_t root;         /* m_rootNodeOffset into the shared array */
    float bounds_min[3];   /* m_boundsMin */
    float inv_extent;      /* m_invBoundsExtent (all three components equal) */
    uint32_t resolution;
} pgpu_svdag_desc;
typedef struct pgpu_host_svdags pgpu_host_svdags;
int pgpu_build_svdags(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* group_prims, const uint64_t* group_offsets, size_t n_groups, uint32_t resolution,
    pgpu_host_svdags** out);
/* The same on the GPU of `ctx` (pandora_amd/csrc/svdag_build.hip): one thread per triangle voxelises, an occupancy pyramid in
 * Morton order gives the octrees of all batching points at once, and the subtrees are hash-consed level by level in a device
 * hash table. Same arguments (host arrays) and the same kind of result object; every DAG expands to the same voxels, has the
 * same bounds and leaf promotions, and the merged array has the same number of words as pgpu_build_svdags' -- the order of
 * the nodes inside the array is the builder's own. resolution 4..256, at most 65535 batching points. */
int pgpu_build_svdags_device(pgpu_ctx*, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* group_prims, const uint64_t* group_offsets, size_t n_groups, uint32_t resolution,
    pgpu_host_svdags** out);
const uint32_t* pgpu_host_svdags_words(const pgpu_host_svdags*, size_t* n_words);
const pgpu_svdag_desc* pgpu_host_svdags_descs(const pgpu_host_svdags*, size_t* n_dags);
size_t pgpu_host_svdags_words_before_compression(const pgpu_host_svdags*);
void pgpu_host_svdags_free(pgpu_host_svdags*);
/* Expands DAG `dag` back into a dense bit grid (z-major, x fastest, 1 bit per voxel packed LSB first
 * in u32 words, same addressing as VoxelGrid::index, voxel_grid.cpp:149-159). For validation. */
int pgpu_svdag_expand(const uint32_t* words, size_t n_words, const pgpu_svdag_desc* dag, uint32_t* grid_bits);
int pgpu_voxelize(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* prims, size_t n_prims, uint32_t resolution, float bounds_out[6], uint32_t* grid_bits);
/* Adds `base` to every child offset of a word array that pgpu_build_svdags / pgpu_build_svdags_device produced (nodes back to
 * back: a descriptor word, then one absolute offset per inner child, sparse_voxel_dag.h:40-78), so that arrays built apart --
 * each rank of a sharded build makes the DAGs of its own batching points -- can be laid one behind the other and uploaded as
 * one (the roots in the descriptors move by the same amount; the caller adds it). PGPU_ERR_INVALID when the words do not parse
 * as such a sequence. The reference builds all DAGs in one process (compressDAGs, sparse_voxel_dag.cpp:135-226): what is lost
 * here is only the sharing of equal subtrees ACROSS arrays. */
int pgpu_svdag_rebase(uint32_t* words, size_t n_words, uint32_t base);

/* ------------------------------------------------------------------ device context
 * One context per GPU, one process per GPU. Replaces the TaskGraph/LRU-cache plumbing the reference
 * threads through its builder (offline_batching_acceleration_structure.h:106-126). A context is not thread-safe, and its
 * calls share work cursors and scratch buffers: two traces of one context must not be in flight at the same time, not
 * even on different streams (queue them on one stream, or use one context per concurrent batch).
 *
 * Streams. Every call that takes `void* stream` queues its work on that hipStream_t; NULL means the context's own
 * stream. That stream is a *blocking* stream (hipStreamDefault): it is ordered against the legacy null stream in both
 * directions, so a caller whose other work runs on the null stream (torch's default stream has handle 0, which reads as
 * NULL here) needs no events around library calls. A caller that passes one of its own streams gets plain stream
 * order on it; work on other non-blocking streams must be ordered with events by the caller. */
int pgpu_init(int device_id, pgpu_ctx** out);
void pgpu_shutdown(pgpu_ctx*);
const char* pgpu_last_error(const pgpu_ctx*);
const char* pgpu_version(void);

/* Geometry of one shape: indexed triangles, TriangleShape m_positions / m_indices (triangle.h:78-87). */
int pgpu_upload_mesh(pgpu_ctx*, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    uint32_t* mesh_id);

/* One bottom-level BVH8 in the reference layout (what CachedBVH deserialises, offline_bvh_cache.cpp:138-148)
 * plus its leaf-object table: leaf object j is primitive objs[j].prim of mesh objs[j].mesh
 * (OfflineBVHLeaf::Primitive, offline_bvh_cache.h:18-39). */
typedef struct {
    uint32_t mesh; /* PGPU_INSTANCE: the leaf object is instance number `prim` of the table given with it */
    uint32_t prim;
} pgpu_leafobj;
int pgpu_upload_bvh8(pgpu_ctx*, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, uint32_t* bvh_id);

/* Instanced leaves: OfflineBVHLeaf::Instance (offline_bvh_cache.h:18-39, .cpp:21-29,54-61,74-77; made by
 * LRUBVHSceneCache::createBVH, :278-297). A leaf object may be a whole other BVH (an already uploaded bvh_id) under a
 * transform. world_to_local = rows 0..2 of Transform::m_inverseMatrix = glm::inverse(localToWorld), row-major 3x4.
 * Semantics: the ray is taken to the instance's space with Transform::transformToLocal (graphics_core/transform.cpp:40-45:
 * origin = inverse * (o, 1), direction = inverse * (d, 0), tnear / tfar unchanged) and the instance's BVH is traversed with
 * it; a hit shortens the world ray's tfar to the same t. The reference's own offline path forgets that transform (it hands the
 * nested traversal the world-space ray) -- its Embree-backed structures do it as described
 * (batching_acceleration_structure.h:411-422), and that is what is implemented: REF-UNPINNED.
 * Limits: one level (the instanced BVH holds no instances itself), an instance sits alone in its leaf (pgpu_build_bvh8_boxes
 * with the `alone` flag builds such trees), instanced BVHs must be resident (pgpu_upload_bvh8; the BVH that holds the
 * instances may be staged). pgpu_trace_sharded routes instance hits (every rank must hold every prototype; result rows then
 * carry the instance as a fifth value); the building blocks pgpu_trace_leaves / pgpu_trace_leaf_records refuse such contexts.
 * Hit records: obj = leaf object inside the INSTANCED BVH, pgpu_hits.inst = leaf object (the instance) inside the traced BVH. */
#define PGPU_INSTANCE 0xFFFFFFFFu
typedef struct {
    uint32_t bvh_id;
    float world_to_local[12];
} pgpu_instance;
int pgpu_upload_bvh8_instanced(pgpu_ctx*, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, const pgpu_instance* instances, size_t n_instances, uint32_t* bvh_id);
int pgpu_stage_bvh8_instanced(pgpu_ctx*, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, const pgpu_instance* instances, size_t n_instances, uint32_t* bvh_id);

/* Leaves streamed from host memory (BASELINE config 5; CachedBVH::doMakeResident / doEvict, offline_bvh_cache.cpp:127-148):
 * pgpu_stage_bvh8 takes the same arguments as pgpu_upload_bvh8 but keeps the device image in pinned host memory; the
 * two-level loop brings it into an HBM cache when rays are parked at its batching point. The cache is one arena used as a
 * ring: images of different sizes are written one behind the other and the oldest ones make room (first in, first out); the
 * images a round needs are visited in alternating order (up, then down), so that a round starts with what the previous one
 * brought in last. The images of one launch group are pulled over PCIe by a single kernel that reads the pinned memory
 * (pgpu_set_leaf_copy(ctx, 1): one hipMemcpyAsync per image instead), on a copy stream, overlapped with the traversal of
 * the previous group. Rays wait: like the reference, which drains its fullest queue first and lets the others fill
 * (task_graph.cpp:19-56), a round loads an image that is not in the cache only once its batching point holds one ray per
 * `image_bytes_per_ray` bytes of image (pgpu_set_leaf_hold; default 128, 0 = every point with a ray is loaded every round);
 * the rays of the other points stay parked and join the next round's queues (when no point qualifies, the points with the
 * most rays per byte are taken, one cache-full of them). Results do not depend on any of this; pgpu_counters.rounds does.
 * pgpu_set_leaf_cache sets the HBM budget of the cache in bytes (0 = room for every staged BVH); it must
 * hold the largest image. Staged ids share the id space of uploaded ones but
 * are valid only as pgpu_leaf.bvh_id for pgpu_trace and pgpu_trace_sharded (where every point that has work in a round is loaded
 * in that round -- rays are not held back there, a record's result goes home in the round it arrived): pgpu_trace_bvh8,
 * pgpu_trace_leaves and pgpu_trace_leaf_records return PGPU_ERR_NO_SCENE on a context that holds staged BVHs. Staging another
 * BVH (or pgpu_set_leaf_cache) empties the cache; it is rebuilt by the next trace. */
int pgpu_stage_bvh8(pgpu_ctx*, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_leaf_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, uint32_t* bvh_id);
int pgpu_set_leaf_cache(pgpu_ctx*, size_t bytes);
int pgpu_set_leaf_copy(pgpu_ctx*, int mode);
int pgpu_set_leaf_hold(pgpu_ctx*, uint32_t image_bytes_per_ray);
/* Queues that will not fill: once as many rays wait as still move (or nothing moves at all) the waiting queues are flushed, and a
 * flushed queue with fewer rays than one per `image_bytes_per_ray` bytes of its image is not loaded at all -- its rays traverse
 * the image where it lies, in pinned host memory, through the same kernel: a visit pulls the nodes and records it touches over
 * PCIe (through the L2, so rays of one launch share them), loading the image for a handful of rays moves all of it and orders
 * copies and launches one behind the other. Default 128 = the default of pgpu_set_leaf_hold: every flushed queue is traced in
 * place, images are loaded only for queues that reach the hold threshold. 0 = every flushed queue is loaded, fullest first, one
 * cache-full per round. The reference has no counterpart (its leaves are loaded whole from disk, offline_bvh_cache.cpp:127-148).
 * Same results. */
int pgpu_set_leaf_in_place(pgpu_ctx*, uint32_t image_bytes_per_ray);

/* Shared SVDAG array + one descriptor per DAG (SparseVoxelDAG members, sparse_voxel_dag.h:70-78). */
int pgpu_upload_svdags(pgpu_ctx*, const uint32_t* words, size_t n_words, const pgpu_svdag_desc* dags, size_t n_dags);
/* Which batching points get the SVDAG early-out. 0 (default) = every point that has an SVDAG, as the reference does
 * (offline_batching_acceleration_structure.h:323-386). 1 = only the points whose geometry is not in this GPU's HBM -- staged in
 * host memory (pgpu_stage_bvh8) or owned by another rank (PGPU_NO_BVH): the proxy test exists to save a load or a transfer, and
 * a resident point is cheaper to visit than to test (11.4 M-triangle city, resident: 5.2 ms per step without the tests against
 * 6.6 ms with them). Hits do not depend on the policy; pgpu_hits.visits and the SVDAG counters do. No counterpart in the
 * reference, where every leaf visit may load from disk. */
int pgpu_set_svdag_policy(pgpu_ctx*, int policy);

/* Where pgpu_trace runs the SVDAG casts (same results either way, bit for bit; no counterpart in the reference, whose cast is
 * a call inside PauseableBVH4::intersectT, pauseable_bvh4.h:376-396). min_rays = 0 (the default): inside the top-level
 * episode kernel. min_rays > 0: passes of at least that many rays stop at every batching point that has a DAG, the cast
 * requests are sorted by point and cast by a kernel that stages the point's DAG in LDS, `passes` (1..8) times per round
 * before the episode kernel finishes the remainder. Measured slower on the benchmark scene; kept
 * selectable for scenes with deeper DAGs or higher cull rates. */
int pgpu_set_svdag_defer(pgpu_ctx*, uint32_t min_rays, uint32_t passes);

/* Top-level PauseableBVH4 (m_bvhNodes, m_leafs, m_rootHandle; pauseable_bvh4.h:100-103). Leaf k is a
 * batching point: its bottom-level BVH and its SVDAG (PGPU_NO_SVDAG = none, i.e. svdagRes 0). */
typedef struct {
    uint32_t bvh_id;
    uint32_t svdag;
} pgpu_leaf;
int pgpu_upload_top_bvh4(pgpu_ctx*, const void* nodes112, size_t n_nodes, uint32_t root, const pgpu_leaf* leaves, size_t n_leaves);

/* ------------------------------------------------------------------ on-disk BVH8 (host only)
 * The flatbuffer the reference writes for a cached bottom-level BVH (pandora/flatbuffers/wive_bvh8.fbs +
 * contiguous_allocator.fbs; WiVeBVH8::serialize, wive_bvh8_impl.h:37-49; read back by GetWiVeBVH8,
 * offline_bvh_cache.cpp:138-143). serialize returns a malloc'ed buffer (pgpu_bytes_free); deserialize returns
 * pointers INTO `bytes` (4-byte aligned; pgpu_upload_bvh8 copies), every offset bounds-checked. */
int pgpu_bvh8_serialize(const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_leaf_indices, uint32_t root_handle,
    uint64_t num_leaf_objects, void** out_bytes, size_t* out_size);
int pgpu_bvh8_deserialize(const void* bytes, size_t size, const void** nodes256, size_t* n_nodes, const uint32_t** leaf_indices,
    size_t* n_leaf_indices, uint32_t* root_handle, uint64_t* num_leaf_objects);
void pgpu_bytes_free(void* bytes);
/* The triangle mesh chunk of the reference (pandora/flatbuffers/triangle_mesh.fbs: indices [Vec3u], positions [Vec3], normals
 * [Vec3], texCoords [Vec2], bounds), as TriangleShape::serialize writes it (shapes/triangle_shape.cpp:520-553) and
 * TriangleShape::loadSerialized reads it (:557-611). normals / texcoords may be NULL (left out, as the reference does for empty
 * ones); bounds are those of the positions. The reader returns pointers into `bytes` (every offset bounds-checked; absent vectors
 * read as empty) and does not validate the index values -- pgpu_upload_mesh does. Free the writer's buffer with pgpu_bytes_free. */
int pgpu_mesh_serialize(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles, const float* normals,
    const float* texcoords, void** out_bytes, size_t* out_size);
int pgpu_mesh_deserialize(const void* bytes, size_t size, const float** positions, size_t* n_vertices, const uint32_t** indices,
    size_t* n_triangles, const float** normals, size_t* n_normals, const float** texcoords, size_t* n_texcoords, float* bounds6);

/* ------------------------------------------------------------------ tracing
 * SoA ray batch: Ray{origin, direction, tnear, tfar} of graphics_core/ray.h:14-40. */
typedef struct {
    const float *ox, *oy, *oz, *dx, *dy, *dz, *tnear, *tfar;
} pgpu_rays;
/* Closest-hit result per ray: t = final ray.tfar (the input tfar on a miss), obj = leaf-object index
 * inside the BVH8 that was hit or PGPU_MISS, leaf = top-level leaf (0 for single level), u,v =
 * RayHit::geometricUV. visits (may be NULL) = Ray::numTopLevelIntersections. */
typedef struct {
    float* t;
    uint32_t* obj;
    uint32_t* leaf;
    float *u, *v;
    uint32_t* visits;
    uint32_t* inst; /* may be NULL. The instance a hit went through (leaf-object index in the traced BVH), PGPU_MISS = none */
} pgpu_hits;
typedef struct {
    uint64_t svdag_tests, svdag_culled; /* g_stats.svdag.numIntersectionTests / numRaysCulled */
    uint64_t leaf_visits;               /* rays parked at batching points, summed */
    uint64_t rounds;                    /* wavefront rounds the two-level loop needed */
    uint64_t inner_visits, leaf_pops, tri_tests; /* filled only when stats are enabled */
    uint64_t leaf_loads, streamed_bytes; /* staged BVHs copied into the HBM leaf cache by the last pgpu_trace / pgpu_trace_sharded */
    uint64_t wave_steps, leaf_passes, refill_passes; /* kernel-loop diagnostics (stats enabled): wavefront iterations,
                                                       * those that ran the triangle code, those that fetched new rays */
    uint64_t records_routed, exchange_bytes; /* pgpu_trace_sharded: records sent to other ranks; bytes sent + received */
} pgpu_counters;

/* WiVeBVH8::intersect / intersectAny over a batch (wive_bvh8_impl.h:57-201): single-level, in-core.
 * kind PGPU_CLOSEST fills `hits`; PGPU_ANY fills any_hit[i] = 0/1. mem says where the ray and result
 * arrays live; stream is a hipStream_t (NULL = the context's own stream; device arrays only). */
int pgpu_trace_bvh8(pgpu_ctx*, uint32_t bvh_id, int kind, int tri_mode, const pgpu_rays* rays, size_t n, int mem,
    const pgpu_hits* hits, uint32_t* any_hit, void* stream);

/* pgpu_trace_bvh8 for callers that cannot allocate. No counterpart in the reference: it exists for callers that cannot call
 * hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid allocation. Same kernels and
 * bit-identical results as pgpu_trace_bvh8 with PGPU_MEM_DEVICE (ray and result arrays are device memory; stream as above),
 * but the kernels' scratch (work cursors + the overflow of the per-ray traversal stacks) is a static device array of
 * PGPU_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends. The call makes no hipMalloc /
 * hipFree. Limits: n <= PGPU_NOALLOC_MAX_RAYS (larger batches are rejected with PGPU_ERR_INVALID); on trees whose stack
 * overflow does not fit the static array at full occupancy the grid is made smaller (never wrong, only slower); the bvh
 * must have been uploaded with pgpu_upload_bvh8 (which also brings the device BVH table up to date -- if a BVH was staged
 * since, the call returns PGPU_ERR_NO_SCENE rather than allocate). All calls on one GPU share that GPU's array: calls that
 * overlap in time on one GPU race, whatever their contexts or streams. */
#define PGPU_NOALLOC_MAX_RAYS 0x3FFFFFFFu
#define PGPU_NOALLOC_SCRATCH_BYTES (256u + (64u << 20))
int pgpu_trace_bvh8_noalloc(pgpu_ctx*, uint32_t bvh_id, int kind, int tri_mode, const pgpu_rays* rays, size_t n,
    const pgpu_hits* hits, uint32_t* any_hit, void* stream);

/* OfflineBatchingAccelerationStructure::intersect / intersectAny (offline_batching...h:406-437)
 * driven to completion for the whole batch: top-level traversal, SVDAG early-out, parking at batching
 * points, per-leaf bottom-level traversal, resume -- until every ray has exited. */
int pgpu_trace(pgpu_ctx*, int kind, int tri_mode, const pgpu_rays* rays, size_t n, int mem,
    const pgpu_hits* hits, uint32_t* any_hit, pgpu_counters* counters, void* stream);

/* ---- building blocks of pgpu_trace, for loops that span several GPUs (leaves sharded over HBM).
 * All arrays are device memory; the calls queue work on `stream` (NULL = the context's stream) and return.
 *
 * pgpu_ray_state_init: PauseableBVH4 start token {root, ~0} (pauseable_bvh4.h:202) for n rays, t = tfar, miss records.
 * pgpu_top_episode:    PauseableBVH4::intersectT (pauseable_bvh4.h:225-355) from each listed ray's token to its next
 *                      batching point (SVDAG test included, offline_batching...h:269-305) or to the exit.
 *                      active = ray ids to advance (NULL: 0..n_active-1); parked_leaf[i] = batching point work item i
 *                      parked at, or PGPU_MISS. t_state = current ray.tfar per ray (closest hit); any_out = occlusion
 *                      flags (any hit; occluded rays are skipped). SVDAG counters accumulate in the context.
 *                      may_park = 0 makes a ray that reaches a batching point leave instead of pausing: the reference's
 *                      traversal does not terminate on every ray (deviation 4), so a loop passes 0 once
 *                      it has run n_leaves rounds.
 * pgpu_trace_leaves:   the leaf tasks (offline_batching...h:179-205, :226-259) without the resume: work item i
 *                      sends ray work_ray[i] through the bottom-level BVH of batching point work_leaf[i], which must be
 *                      resident here. Closest hit: hits->t is read as the ray's tfar and, like obj/leaf/u/v, written
 *                      only when the leaf yields a closer hit. Any hit: any_hit[ray] = 1 on a hit. */
int pgpu_ray_state_init(pgpu_ctx*, size_t n, const float* tfar, uint32_t* node, uint64_t* stack, const pgpu_hits* hits, uint32_t* any_hit, void* stream);
int pgpu_top_episode(pgpu_ctx*, int kind, const pgpu_rays* rays, const uint32_t* active, size_t n_active, uint32_t* node, uint64_t* stack,
    const float* t_state, uint32_t* visits, uint32_t* any_hit, uint32_t* parked_leaf, int may_park, void* stream);
int pgpu_trace_leaves(pgpu_ctx*, int kind, int tri_mode, const pgpu_rays* rays, const uint32_t* work_ray, const uint32_t* work_leaf,
    size_t n_work, const pgpu_hits* hits, uint32_t* any_hit, void* stream);

/* Device-side record handling of the same loop (everything between two all-to-alls stays on the GPU). A record is 9
 * dwords {ox oy oz dx dy dz tnear tfar leaf} with tfar = the ray's current t for closest hit; a result is 4 dwords
 * {t, obj, u, v} (obj = PGPU_MISS: nothing closer) or 1 dword {hit} for any hit.
 * pgpu_route_pack:         groups the work items that parked (parked_leaf != PGPU_MISS) by owner_of_leaf[leaf] (device
 *                          array, one rank < world <= 64 per batching point): dest_counts_host[world] (host, written
 *                          before the call returns - the one synchronisation), records in destination order, and per
 *                          record the home ray id and the batching point (sent_slot doubles as next round's active list).
 * pgpu_trace_leaf_records: owner side: k received records through the named (resident) batching points -> k results,
 *                          same order.
 * pgpu_route_merge:        home side: returned results replace the hit record of rays that found something closer. */
int pgpu_route_pack(pgpu_ctx*, const pgpu_rays* rays, const float* t_state, const uint32_t* active, const uint32_t* parked_leaf, size_t n_active,
    const uint32_t* owner_of_leaf, uint32_t world, uint32_t* dest_counts_host, float* records, uint32_t* sent_slot, uint32_t* sent_leaf, void* stream);
int pgpu_trace_leaf_records(pgpu_ctx*, int kind, int tri_mode, const float* records, size_t k, float* results, void* stream);
int pgpu_route_merge(pgpu_ctx*, int kind, const float* results, const uint32_t* sent_slot, const uint32_t* sent_leaf, size_t k,
    const pgpu_hits* hits, uint32_t* any_hit, void* stream);

/* ------------------------------------------------------------------ several GPUs: batching points sharded over HBM
 * (BASELINE configs[3]). One process and one context per GPU. Every rank uploads the top level and the SVDAGs
 * whole and the bottom-level BVHs of the batching points it owns (the others are PGPU_NO_BVH); rays stay on their home rank.
 * The round loop of pgpu_trace then runs on every rank in lockstep: a ray parked at a batching point another rank owns
 * travels there as a 36-byte record and its result ({t, obj, u, v} or {hit}) travels back -- an all-to-all of records in
 * each direction per round, which replaces the reference's per-leaf MPMC queues and disk cache
 * (stream/task_graph.h:130-156, task_graph.cpp:19-56; offline_batching...h:158-260). Rays parked at a batching point
 * of their own rank never leave the SoA arrays. Results equal pgpu_trace's on one GPU bit for bit.
 *
 * pgpu_comm_unique_id: 128 bytes (an ncclUniqueId) that rank 0 makes and hands to the other ranks by any means.
 * pgpu_comm_init:      joins the RCCL communicator of `world` ranks (collective: every rank calls it); world <= 64.
 *                      RCCL is loaded at this call (dlopen librccl.so.1), not when the library is loaded.
 * pgpu_comm_init_transport: the same with the three collectives supplied by the caller (MPI, a test harness, ...):
 *     allgather_u32: every rank contributes n values, recv[r * n + i] = rank r's send[i];
 *     exchange:      rows of row_bytes grouped by destination rank, send_counts[d] rows to rank d (in rank order), recv_counts[s]
 *                    rows from rank s (in rank order); counts are host arrays of `world` entries, buffers device memory;
 *     allreduce_u64: element-wise sum over ranks, in place.
 *   All three work on device pointers, are ordered on `stream`, and return 0 or non-zero for failure.
 * pgpu_set_leaf_owners: owner rank of every batching point (leaf slot order, host array). Needed before pgpu_trace_sharded.
 * pgpu_trace_sharded: collective. Device arrays only (rays, hits, any_hit as in pgpu_trace with PGPU_MEM_DEVICE); n may
 *                      differ between ranks (0 included). One host synchronisation per round (the matrix of record counts).
 *                      The rays parked at this rank's own batching points are traced on a second stream of the context, next to
 *                      the record exchange and the trace of the received records; the round's merge waits for both.
 *                      counters additionally reports records_routed (records this rank sent to other ranks) and
 *                      exchange_bytes (bytes this rank sent and received in all-to-alls, both directions).
 *                      Failures: every buffer a round needs is reserved before the first round (the ranks tell each other their
 *                      ray counts up front: no round can deliver more records than the other ranks have rays). A rank that fails
 *                      -- there, or in a launch later -- keeps taking part in its round's collectives and raises a flag that
 *                      travels with the next count rows; every rank then leaves the loop in that same round, the failing rank
 *                      with its own error, the others with PGPU_ERR_PEER. Only a failing transport call returns at once.
 * pgpu_film_allreduce: sums the u64 fixed-point films of all ranks in place (the integer all-reduce of SURVEY.md K8). */
typedef struct {
    void* user;
    int (*allgather_u32)(void* user, const uint32_t* send, uint32_t* recv, size_t n, void* stream);
    int (*exchange)(void* user, const void* send, const uint32_t* send_counts, void* recv, const uint32_t* recv_counts, size_t row_bytes, void* stream);
    int (*allreduce_u64)(void* user, unsigned long long* buf, size_t n, void* stream);
} pgpu_transport;
int pgpu_comm_unique_id(void* out128);
int pgpu_comm_init(pgpu_ctx*, int rank, int world, const void* unique_id128);
int pgpu_comm_init_transport(pgpu_ctx*, int rank, int world, const pgpu_transport* transport);
int pgpu_comm_destroy(pgpu_ctx*);
/* Collective: runs every collective of the transport once on small known buffers and checks what comes back (with RCCL also a
 * point-to-point message from the rank to itself). PGPU_OK, or PGPU_ERR_HIP with the failing step in pgpu_last_error. */
int pgpu_comm_selftest(pgpu_ctx*);
int pgpu_set_leaf_owners(pgpu_ctx*, const uint32_t* owner_of_leaf, size_t n_leaves);
int pgpu_trace_sharded(pgpu_ctx*, int kind, int tri_mode, const pgpu_rays* rays, size_t n, const pgpu_hits* hits, uint32_t* any_hit,
    pgpu_counters* counters, void* stream);
int pgpu_film_allreduce(pgpu_ctx*, unsigned long long* film, size_t n_pixels, void* stream);
/* Test hook: the next pgpu_trace_sharded calls of this rank fail once they have completed `after_rounds` rounds (-1: off), the
 * way a failed launch would -- to check that every rank leaves the loop together (tests/test_gpu_sharded.py). */
int pgpu_debug_fail_sharded(pgpu_ctx*, int after_rounds);

/* Film. Sensor::addPixelContribution (graphics_core/sensor.cpp:36-49) over a batch of contributions:
 * `film` holds three u64 per pixel in 40.24 fixed point (sensor.h:29-35), film[3 * pixel[i] + c] += fixed(rgb[i][c]).
 * A contribution with a NaN or inf channel is dropped whole (sensor.cpp:41-42); so is one with |channel| >= 2^39 (outside
 * the format, undefined in the reference) or pixel[i] >= n_pixels. Negative channels wrap (adding them subtracts).
 * Integer adds commute: the film does not depend on the order of the contributions, and the films of several GPUs
 * sum with an integer all-reduce. All pointers are device memory; the call only enqueues work on `stream` (NULL =
 * the context's stream). n <= 2^30 contributions per call, like the trace entry points. pgpu_film_resolve is Sensor::Pixel -> vec3 (sensor.cpp:96-99): rgb_out[3 i + c] =
 * float(film[3 i + c]) * 2^-24. */
int pgpu_film_add(pgpu_ctx*, unsigned long long* film, size_t n_pixels, const uint32_t* pixel, const float* r, const float* g,
    const float* b, size_t n, void* stream);
int pgpu_film_resolve(pgpu_ctx*, const unsigned long long* film, size_t n_pixels, float* rgb_out, void* stream);

/* ------------------------------------------------------------------ the benchmark integrator's ray sources on the device
 * (SURVEY.md 8f N4). The integrators stay host code in Pandora; once traversal runs at Grays/s, generating camera rays and
 * bounce rays on the host and shipping them over PCIe dominates, so the three steps the benchmark integrator needs around
 * the trace calls exist as kernels that read and write the same SoA arrays (all pointers device memory, work queued on `stream`):
 *   pgpu_camera_rays    SamplerIntegrator::spawnNewPaths (integrators/sampler_integrator.cpp:131-163): sample index i = first + k,
 *                       pixel = i / spp, rng = PcgRng(seed + i) (samplers/rng/pcg.cpp:23-39), camera sample = (pixel + rng.uniformFloat2())
 *                       / resolution, PerspectiveCamera::generateRay (graphics_core/perspective_camera.cpp:37-47). `cam` holds the six
 *                       member vectors PerspectiveCamera::updatePosition computes (:50-65). Fills rays and the path state.
 *   pgpu_lambert_bounce PathIntegrator::randomBounce (integrators/path_integrator.cpp:139-158) for a Lambertian surface of constant
 *                       albedo: hit record of the triangle test (geometric normal facing the ray, shapes/triangle_intersect.cpp:455-460),
 *                       TriangleShape::fillSurfaceInteraction (hit point from the barycentrics, shapes/triangle_shading.cpp:11-21),
 *                       BSDF frame (graphics_core/material.cpp:5-20), cosineSampleHemisphere (graphics_core/sampling.h:8-49),
 *                       spawnRay with RAY_EPSILON offset (graphics_core/interaction.cpp:24-55), new PcgRng(rng.uniformU64()), weight update.
 *                       Rays that missed (obj = PGPU_MISS) or whose sample has zero pdf become dead rays (tnear 1 > tfar 0: they miss
 *                       everything). bvh_id = the single-level BVH the hits refer to, or PGPU_NO_BVH: hits->leaf names the batching
 *                       point of the uploaded top level. Needs the vertices per leaf object on the device: pgpu_set_keep_shading(ctx, 1)
 *                       BEFORE the BVHs are uploaded (36 bytes per leaf object; resident, non-instanced BVHs only).
 *   pgpu_splat_misses   PathIntegrator::rayMiss with one constant infinite light (:113-125) + Sensor::addPixelContribution: every
 *                       ray with obj = PGPU_MISS adds weight * sky to its pixel of the fixed-point film (pgpu_film_add's format).
 *   pgpu_compact_paths  the end of a path (path_integrator.cpp:109-127: a path that missed or whose bounce failed spawns nothing):
 *                       the rays with tnear <= tfar of a batch and their path states, in their order, at the front of another set of
 *                       arrays (of at least n elements; not the input arrays); *n_live = how many. The next depth then traces n_live rays
 *                       instead of n slots of which most are dead (a 4-bounce frame of the benchmark city: 15.6 M rays instead of
 *                       33.5 M slots). Synchronises the stream (the count comes back to the host).
 * Arithmetic is binary32 in the reference's operation order; cos / sin of the disk mapping are the device's (<= 2 ulp from
 * glibc's cosf / sinf, which the reference calls): bounce directions agree with the CPU restatement to 1e-6, everything else
 * (camera rays, origins, rng streams, pixels, film) bit for bit. */
typedef struct {
    float lens_center[3], lens_du[3], lens_dv[3], screen_center[3], screen_du[3], screen_dv[3];
} pgpu_camera;
typedef struct {
    float *ox, *oy, *oz, *dx, *dy, *dz, *tnear, *tfar;
} pgpu_rays_out;
typedef struct {
    unsigned long long *rng_state, *rng_inc; /* PcgRng m_state, m_inc */
    uint32_t* pixel;                         /* y * width + x */
    float *wr, *wg, *wb;                     /* BounceRayState::weight */
} pgpu_paths;
int pgpu_set_keep_shading(pgpu_ctx*, int on);
int pgpu_camera_rays(pgpu_ctx*, const pgpu_camera* cam, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint64_t first, size_t n,
    const pgpu_rays_out* rays, const pgpu_paths* paths, void* stream);
int pgpu_lambert_bounce(pgpu_ctx*, uint32_t bvh_id, const pgpu_rays* rays_in, const pgpu_hits* hits, size_t n, const float albedo[3],
    const pgpu_rays_out* rays, const pgpu_paths* paths, void* stream);
int pgpu_splat_misses(pgpu_ctx*, unsigned long long* film, size_t n_pixels, const uint32_t* obj, const pgpu_paths* paths, size_t n, const float sky[3],
    void* stream);
int pgpu_compact_paths(pgpu_ctx*, const pgpu_rays* rays_in, const pgpu_paths* paths_in, size_t n, const pgpu_rays_out* rays_out, const pgpu_paths* paths_out,
    size_t* n_live, void* stream);

/* Memory, copies and streams for hosts without HIP of their own (no counterpart in the reference, which never leaves host
 * memory; used by include/pandora_amd/gpu_acceleration_structure.hpp). pgpu_host_alloc gives pinned host memory: copies from
 * and to it run at PCIe rate and asynchronously. pgpu_copy is hipMemcpyAsync(..., hipMemcpyDefault) on a stream made by
 * pgpu_stream_create (never NULL); pgpu_stream_sync waits for it. A context is otherwise used by one host thread at a time;
 * these eight keep no state in it and may run on other threads while a trace call is in flight on the same context. */
int pgpu_host_alloc(pgpu_ctx*, size_t bytes, void** out);
int pgpu_host_free(pgpu_ctx*, void* p);
int pgpu_device_alloc(pgpu_ctx*, size_t bytes, void** out);
int pgpu_device_free(pgpu_ctx*, void* p);
int pgpu_stream_create(pgpu_ctx*, void** stream);
int pgpu_stream_destroy(pgpu_ctx*, void* stream);
int pgpu_copy(pgpu_ctx*, void* dst, const void* src, size_t bytes, void* stream);
int pgpu_stream_sync(pgpu_ctx*, void* stream);

/* Diagnostics: device time of the traversal kernels of the last trace call on this context, measured
 * with HIP events on the stream they ran on (ms); enables/disables visit counters. */
float pgpu_last_kernel_ms(const pgpu_ctx*);
/* Waits for the traces issued with PGPU_MEM_DEVICE on `stream` (NULL = the context's stream) and
 * refreshes pgpu_last_kernel_ms / pgpu_get_counters. Host-memory traces are synchronous already. */
int pgpu_sync(pgpu_ctx*, void* stream);
int pgpu_set_stats(pgpu_ctx*, int enabled);
int pgpu_get_counters(pgpu_ctx*, pgpu_counters* out);
int pgpu_set_tuning(pgpu_ctx*, int blocks_per_cu, int chunk); /* 0 = automatic (both) */
/* Bottom-level kernel scheduling knobs: how many of a wave's rays must have finished before new rays
 * are fetched (0 = automatic), how many entries of each ray's traversal stack stay in LDS (0 = automatic; deeper entries
 * spill to HBM), and how many rays of a wave must wait at a leaf before the wave runs its triangle code
 * (0 = automatic). Results do not depend on any of them. */
int pgpu_set_kernel_variant(pgpu_ctx*, int refill_min, int stack_window, int leaf_min);
/* Lanes of a wave64 that share one ray in the bottom-level kernel: 4 (16 rays per wave, node staged in LDS),
 * 2 (32 rays per wave, node halves in registers, two triangles per lane) or 0 (the default: 4 for a batch through one BVH,
 * 2 for the per-batching-point work lists of the two-level loops). Results do not depend on it. */
int pgpu_set_ray_layout(pgpu_ctx*, int lanes_per_ray);
/* pgpu_trace on resident scenes queues each round's kernels before the host has read the previous round's parked count (the
 * kernels take their list lengths from device memory), so the GPU never waits for that read; 0 restores the round-by-round
 * loop, which streamed scenes and pgpu_set_svdag_defer always use. On by default. Results do not depend on it. */
int pgpu_set_round_lookahead(pgpu_ctx*, int on);
/* pgpu_trace on resident scenes runs a large batch as `parts` batches side by side, each on a stream of its own (2..4; 1 = never;
 * 0 = the default: two parts from 2^21 rays on). A round is a chain of kernels of opposite character -- the top-level episode waits
 * on memory, the bottom level on instruction issue -- and every one of them ends in a tail that leaves CUs idle: two chains fill each
 * other's gaps. The parts have work buffers of their own and share the scene; pgpu_counters adds up
 * (rounds: the largest), pgpu_last_kernel_ms is the longest part. Applies when the look-ahead loop does (pgpu_set_round_lookahead),
 * to host and device batches alike. Results do not depend on it: rays do not interact.
 * (No counterpart in the reference, whose CPU tasks already run side by side, task_graph.h:259-292.) */
int pgpu_set_batch_parts(pgpu_ctx*, int parts);
/* The top-level kernel of pgpu_trace: 1 = waves in lockstep (64 rays start an episode together and the wave waits for the
 * slowest), a ray's SVDAG cast running inside the step that reached the DAG; 4 = the same waves with the casts gathered: a wave
 * walks until every ray has parked, left or stands in front of a DAG, then runs all those casts together, then the rays whose
 * cast missed walk on (a few cast phases per episode instead of one per step); 2 = lanes take the next ray as soon as theirs has
 * parked or left, and a wave runs BVH4 steps and SVDAG cast trips as separate phases; 3 = variant 1 for a trace's first episode
 * (every ray from the root) and variant 2 for the resumed rays of the later rounds; 0 = automatic (4). refill_min = idle lanes of
 * a wave before it fetches new rays, cast_trips = cast trips per cast phase (0 = defaults; variants 2 and 3). Results do not
 * depend on any of them. */
int pgpu_set_top_variant(pgpu_ctx*, int variant, int refill_min, int cast_trips);

#ifdef __cplusplus
}
#endif
#endif
