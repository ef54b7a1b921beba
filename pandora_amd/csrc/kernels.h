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
};

struct Bvh8TraceArgs {
    const Bvh8Dev* bvhs;
    const uint32_t* work_ray; // work item -> ray id (nullptr: identity)
    const uint32_t* work_bvh; // work item -> bvh id (nullptr: fixed_bvh)
    uint32_t fixed_bvh;
    uint32_t n_work;
    const float *ox, *oy, *oz, *dx, *dy, *dz, *tnear, *tfar_in;
    float* t_out;
    uint32_t* obj_out;
    uint32_t* leaf_out;
    float *u_out, *v_out;
    uint32_t* any_out;
    uint32_t* cursor; // zeroed before launch
    uint32_t chunk;
    uint32_t init_miss; // 1: write miss records too (single level); 0: only touch rays that hit
    unsigned long long* stats;
};

// Top level: PauseableBVH4 nodes (112 bytes, pauseable_bvh4.h:74-98), its leaves (batching points)
// and the shared SVDAG array (sparse_voxel_dag.h:40-78).
struct SvdagDev {
    uint32_t root;
    float bmin[3];
    float invExtent;
};
struct TopLeafDev {
    uint32_t bvh;
    uint32_t svdag; // PGPU_MISS_ID: no proxy geometry
};
struct TopDev {
    const unsigned char* nodes;
    const TopLeafDev* leaves;
    const uint32_t* dagWords;
    const SvdagDev* dags;
    uint32_t root, n_nodes, n_leaves, n_dags;
};

size_t bvh8TraceLdsBytes();
hipError_t launchBvh8Trace(const Bvh8TraceArgs& args, int kind, int triMode, bool stats, uint32_t blocks, hipStream_t stream);

} // namespace pandora_hip
