/*
 * pandora_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference's ray-traversal hot path (mathijs727/pandora).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (pandora_amd/, include/) never links, loads or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - SIMD primitive semantics (x86 min/max NaN rule, permute, compress, horizontalMinIndex,
 *     bitScan4, bitScanReverse32): pinned against the reference's own simd headers compiled
 *     from /root/reference into oracle/_ref (oracle/ref_simd_harness.cpp) and against the
 *     known-answer values of projects/simd/tests/src/test_simd{4,8}.cpp (tests/golden/simd_kat.json).
 *   - Moller-Trumbore leaf test: pinned by the known-answer rays of
 *     projects/pandora/tests/test_triangle.cpp:12-64 (tests/golden/triangle_kat.json).
 *   - Traversal control flow (BVH8 / BVH4 / SVDAG loops): PARITY UNPINNED by reference tests
 *     (the reference has none) and the reference traversal headers cannot be built
 *     here without writing stand-ins for EASTL/flatbuffers/Embree/mio/glm, so they are treated as
 *     unbuildable. The loops are restated from source and cross-checked against a second,
 *     independently written restatement that runs on the reference's simd library (oracle/_ref).
 *
 * All arithmetic is IEEE binary32 with contraction OFF (-ffp-contract=off), matching the kernels.
 */
#ifndef PANDORA_ORACLE_H
#define PANDORA_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MISS 0xFFFFFFFFu

/* wive_bvh8.h:76-86 -- 256-byte node, on-disk layout verbatim */
typedef struct {
    float minX[8], maxX[8], minY[8], maxY[8], minZ[8], maxZ[8];
    uint32_t children[8];
    uint32_t perm[8];
} orc_bvh8_node;

/* pauseable_bvh4.h:74-98 -- 112-byte node; bits = validMask:4 | leafMask:4<<4 | depth:8<<8 */
typedef struct {
    float minX[4], minY[4], minZ[4], maxX[4], maxY[4], maxZ[4];
    uint32_t firstChild, firstLeaf, parent, bits;
} orc_bvh4_node;

typedef struct {
    float o[3], d[3], tnear, tfar;
} orc_ray;

typedef struct {
    float t;       /* final ray.tfar (input tfar when missed)            */
    uint32_t obj;  /* leaf-object index inside the BVH8, ORC_MISS = miss */
    uint32_t leaf; /* top-level leaf (batching point) index, 0 if single level */
    float u, v;
    uint32_t inst; /* leaf-object index of the instance the hit went through, ORC_MISS = none (obj then indexes the instance's BVH) */
} orc_hit;

/* indexed triangle mesh, triangle.h:78-87 */
typedef struct {
    const float* pos;    /* nv*3 */
    const uint32_t* idx; /* nt*3 */
} orc_mesh;

/* OfflineBVHLeaf::Instance (offline_bvh_cache.h:18-39): a leaf object that is a whole other BVH under a transform.
 * child == NULL marks a leaf object that is a primitive. world_to_local = rows 0..2 of Transform::m_inverseMatrix (row-major 3x4). */
struct orc_bvh8_s;
typedef struct {
    const struct orc_bvh8_s* child;
    float world_to_local[12];
} orc_instance;

/* one bottom-level BVH8 over leaf objects; leaf object j = triangle obj_prim[j] of mesh, or inst[j] when that is an instance */
typedef struct orc_bvh8_s {
    const orc_bvh8_node* nodes;
    const uint32_t* leaf_idx;
    uint32_t root;
    const uint32_t* obj_prim; /* n_objs, may be NULL (identity) */
    orc_mesh mesh;
    const orc_instance* inst; /* n_objs or NULL (no instances) */
} orc_bvh8;

/* sparse_voxel_dag.h:70-78 */
typedef struct {
    const uint32_t* data; /* shared node array */
    uint32_t root;        /* m_rootNodeOffset */
    float bounds_min[3];
    float inv_extent; /* 1 / max extent, splat */
    int enabled;
} orc_svdag;

typedef struct {
    orc_bvh8 bvh;
    orc_svdag dag;
} orc_leaf;

typedef struct {
    const orc_bvh4_node* nodes;
    uint32_t root;
    const orc_leaf* leaves;
    uint32_t n_leaves;
    uint32_t n_nodes; /* size of `nodes` (bounds the work of a terminating traversal, see orc_bvh4_step_ex) */
} orc_scene;

typedef struct {
    uint64_t svdag_tests, svdag_culled; /* g_stats.svdag.* (offline_batching...h:275-279) */
    uint64_t leaf_visits;               /* sum of ray.numTopLevelIntersections */
    uint64_t inner_visits, leaf_pops, tri_tests; /* traffic model */
} orc_counters;

enum { ORC_TRI_MT = 0, ORC_TRI_WATERTIGHT = 1 };

/* triangle_intersect.cpp:425-462 (MT, active) / :340-423 (watertight, #if 0). Returns 1 on hit and
 * shortens *tfar; u,v = geometricUV. */
int orc_tri_intersect(int mode, const float o[3], const float d[3], float tnear, float* tfar,
    const float p0[3], const float p1[3], const float p2[3], float* u, float* v);

/* simd primitive restatements (S4) */
void orc_compress_perm8(unsigned mask, uint32_t out[8]);   /* simd8_avx2.ipp:1-17,103-118 */
unsigned orc_hmin_index4(const float v[4]);                /* simd4_avx2.ipp:412-430 */
int orc_bitscan4(unsigned mask);                           /* intrinsics.h:22-34 */
int orc_bsr32(uint32_t mask);                              /* intrinsics.h:93-126 */
uint32_t orc_sign_shift(float dx, float dy, float dz);     /* wive_bvh8_impl.h:384-388 */
/* wive_bvh8_build8_impl.h:172-203: octant permutation words for one node's child boxes */
void orc_bvh8_perm(const float lo[][3], const float hi[][3], unsigned n_children, uint32_t perm[8]);

/* wive_bvh8_impl.h:209-238 */
unsigned orc_bvh8_inner(const orc_bvh8_node* n, const orc_ray* r, uint32_t out_children[8], float out_dist[8]);

/* wive_bvh8_impl.h:57-140 / :142-201. Ray tfar is updated like the reference (closest: hit t; any: -inf). */
int orc_bvh8_intersect(const orc_bvh8* b, int tri_mode, orc_ray* r, orc_hit* h, orc_counters* c);
int orc_bvh8_intersect_any(const orc_bvh8* b, int tri_mode, orc_ray* r, orc_counters* c);

/* sparse_voxel_dag.cpp:350-567; lanes = 4 (faithful, incl. the lane-3 artefact) or 3. Guard added on both sides (oracle
 * and GPU): with NaN operands the reference's ADVANCE makes no progress and loops forever; after (16 levels + 16) << levels
 * loop trips (levels = height of the DAG) the cast gives up and reports a hit (conservative: the ray is not culled). */
int orc_svdag_intersect(const orc_svdag* d, const orc_ray* r, int lanes);

/* pauseable_bvh4.h:225-355 driven to completion per ray: every pause is followed immediately by the
 * leaf's bottom-level traversal and a resume (offline_batching...h:179-205, :226-259). */
int orc_scene_intersect(const orc_scene* s, int tri_mode, int svdag_lanes, orc_ray* r, orc_hit* h, orc_counters* c);
int orc_scene_intersect_any(const orc_scene* s, int tri_mode, int svdag_lanes, orc_ray* r, orc_counters* c);

/* one top-level episode: runs until the ray pauses at a leaf (returns 1, leaf, node and stack are set)
 * or exits (returns 0). Used to check the per-visit tokens the GPU emits. */
int orc_bvh4_step(const orc_scene* s, int any_hit, int svdag_lanes, const orc_ray* r,
    uint32_t* node, uint64_t* stack, uint32_t* leaf, orc_counters* c);
/* The same with the two guards this repository adds on BOTH sides (oracle and GPU) because the reference's loop does not
 * terminate on some rays: when every candidate child has tmin = +inf, blend(FLT_MAX, tmin) + horizontalMinIndex
 * (pauseable_bvh4.h:297-299) picks a child that is NOT a candidate and the XOR at :304 then SETS its stack bit. A
 * terminating traversal enters a node at most 5 times, so an episode is cut after 8 (n_nodes + n_leaves) + 64 loop trips
 * (the ray leaves the top level), and a ray pauses at most n_leaves times: with may_park = 0 a ray that reaches a
 * batching point leaves instead of pausing (orc_scene_intersect* pass that once a ray has paused n_leaves times). */
int orc_bvh4_step_ex(const orc_scene* s, int any_hit, int svdag_lanes, const orc_ray* r,
    uint32_t* node, uint64_t* stack, uint32_t* leaf, orc_counters* c, int may_park);

/* batch drivers (SoA rays [8][n]: ox,oy,oz,dx,dy,dz,tnear,tfar), n_threads >= 1 */
void orc_bvh8_trace(const orc_bvh8* b, int any_hit, int tri_mode, const float* rays, size_t n,
    float* t, uint32_t* obj, float* u, float* v, uint32_t* anyhit, orc_counters* c, int n_threads);
void orc_scene_trace(const orc_scene* s, int any_hit, int tri_mode, int svdag_lanes, const float* rays, size_t n,
    float* t, uint32_t* obj, uint32_t* leaf, float* u, float* v, uint32_t* anyhit, uint32_t* visits,
    orc_counters* c, int n_threads);

/* The same with the instance id of every hit (ORC_MISS where the hit is a primitive of the traced BVH itself). */
void orc_bvh8_trace_inst(const orc_bvh8* b, int any_hit, int tri_mode, const float* rays, size_t n,
    float* t, uint32_t* obj, float* u, float* v, uint32_t* anyhit, uint32_t* inst, orc_counters* c, int n_threads);
void orc_scene_trace_inst(const orc_scene* s, int any_hit, int tri_mode, int svdag_lanes, const float* rays, size_t n,
    float* t, uint32_t* obj, uint32_t* leaf, float* u, float* v, uint32_t* anyhit, uint32_t* visits, uint32_t* inst,
    orc_counters* c, int n_threads);
/* Transform::transformToLocal (graphics_core/transform.cpp:40-45) with glm's mat4 * vec4 association
 * ((m0 x + m1 y) + (m2 z + m3 w)); tnear / tfar are kept: the parameter t means the same in both spaces. */
void orc_ray_to_local(const float world_to_local[12], const orc_ray* in, orc_ray* out);

/* brute force over all triangles with the same leaf test; independent of any tree */
void orc_brute_trace(const orc_mesh* m, size_t n_tris, int tri_mode, const float* rays, size_t n,
    float* t, uint32_t* prim, int n_threads);

/* Batch form of orc_svdag_intersect over rays [8][n] (SoA, as orc_bvh8_trace): out[i] = 0 / 1. */
void orc_svdag_trace(const orc_svdag* d, int lanes, const float* rays, size_t n, unsigned char* out);

/* Film: Sensor::addPixelContribution over a batch (graphics_core/sensor.cpp:36-49, sensor.h:29-35; the fixed-point type
 * is cnl::fixed_point<uint64_t, -24>, a dependency that is not in the container: restated from its definition, parity
 * unpinned). film: 3 u64 per pixel. */
void orc_film_add(uint64_t* film, size_t n_pixels, const uint32_t* pixel, const float* r, const float* g, const float* b, size_t n);
void orc_film_resolve(const uint64_t* film, size_t n_pixels, float* rgb_out); /* sensor.cpp:96-99 */

/* ---- the benchmark integrator's ray sources and film splat, restated in binary32:
 * PcgRng (samplers/rng/pcg.cpp:23-39,58-61), SamplerIntegrator::spawnNewPaths (integrators/sampler_integrator.cpp:131-163),
 * PerspectiveCamera::generateRay (graphics_core/perspective_camera.cpp:37-47), the hit record of the triangle test
 * (shapes/triangle_intersect.cpp:455-460), TriangleShape::fillSurfaceInteraction (shapes/triangle_shading.cpp:11-42, meshes
 * without vertex normals), BSDF::BSDF / sampleF / localToWorld (graphics_core/material.cpp:5-47,85-100) with one
 * LambertianReflection (cosineSampleHemisphere, graphics_core/sampling.h:8-49), PathIntegrator::randomBounce
 * (integrators/path_integrator.cpp:139-158), Interaction::spawnRay (graphics_core/interaction.cpp:24-27,39-55),
 * PathIntegrator::rayMiss with one constant infinite light (:113-125). cos / sin are libm's cosf / sinf. */
typedef struct {
    float lens_center[3], lens_du[3], lens_dv[3], screen_center[3], screen_du[3], screen_dv[3]; /* members after updatePosition (:50-65) */
} orc_camera;
typedef struct {
    uint64_t *rng_state, *rng_inc; /* PcgRng m_state, m_inc */
    uint32_t* pixel;               /* y * width + x */
    float *wr, *wg, *wb;           /* BounceRayState::weight */
} orc_paths;
void orc_camera_rays(const orc_camera* cam, uint32_t width, uint32_t height, uint32_t spp, uint64_t seed, uint64_t first, size_t n,
    float* rays_out /* [8][n] */, const orc_paths* paths);
/* One Lambert bounce for every ray that hit (obj != ORC_MISS); the others (and failed samples) become dead rays (tnear 1, tfar 0).
 * Single level: b != NULL, leaf ignored; two level: s != NULL, leaf = batching point of the hit. */
void orc_lambert_bounce(const orc_bvh8* b, const orc_scene* s, const float* rays_in, const float* t, const uint32_t* obj, const uint32_t* leaf,
    const float* u, const float* v, size_t n, const float albedo[3], float* rays_out, const orc_paths* paths);
void orc_splat_misses(uint64_t* film, size_t n_pixels, const uint32_t* obj, const orc_paths* paths, size_t n, const float sky[3]);

#ifdef __cplusplus
}
#endif
#endif
