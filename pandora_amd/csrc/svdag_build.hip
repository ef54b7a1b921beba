// Proxy geometry built on the device: voxelisation of every batching point's triangles, the sparse voxel octrees, and
// their merge into one hash-consed DAG array -- what pandora_amd/host/svdag_builder.cpp does on the CPU
//   voxelise  : TriangleShape::voxelize            projects/pandora/src/shapes/triangle_intersect.cpp:224-329
//   SVO       : constructSVOBreadthFirst           projects/pandora/src/svo/sparse_voxel_dag.cpp:39-133
//   DAG merge : SparseVoxelDAG::compressDAGs       projects/pandora/src/svo/sparse_voxel_dag.cpp:135-226
// re-designed for the GPU. The reference builds one octree per batching point breadth-first from a Morton-sorted voxel
// list and then interns every subtree in a hash map, one after the other. Here all batching points go through together:
//   1. bounds per batching point (ordered-integer atomics), 2. one thread per triangle sets voxel bits (atomicOr),
//   3. an occupancy pyramid in Morton order (a node's eight children are adjacent), one level per launch, all points at once,
//   4. hash-consing per level, bottom-up: every node of a level becomes a key (descriptor + its children's canonical
//      offsets); equal keys meet in one slot of an open-addressing table (64-bit hash, full key verified afterwards), the
//      lowest node index becomes the representative, an exclusive scan over the representatives' sizes gives their
//      offsets, and the representatives write their words.
// The result has the reference's format (descriptor word + one absolute offset per inner child, sparse_voxel_dag.h:40-78).
// The ORDER of the nodes in the array is this builder's own (by level, then by first occurrence); what must equal the host
// builder's result -- and is tested to -- is each DAG's expanded voxel set, its bounds, the leaf promotions (they decide the
// trip count of a cast) and the number of words left after merging.
#include "context.h"
#include <hipcub/hipcub.hpp>
#include <cfloat>
#include <cstring>
#include <vector>

#pragma clang fp contract(off)

using namespace pandora_hip;

namespace {

// float <-> unsigned with the same order (for atomicMin / atomicMax)
__device__ __forceinline__ uint32_t orderKey(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ inline float orderValue(uint32_t k)
{
    const uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}

__device__ __forceinline__ float fmax0(float x) { return 0.0f < x ? x : 0.0f; }            // std::max(0.0f, x)
__device__ __forceinline__ float minf2(float a, float b) { return b < a ? b : a; }          // std::min(a, b)
__device__ __forceinline__ float maxf2(float a, float b) { return a < b ? b : a; }          // std::max(a, b)
// float -> int as x86's cvttss2si does it (what the reference's int(...) compiles to): NaN and out-of-range give INT_MIN, where
// v_cvt_i32_f32 gives 0 or saturates -- a point-sized batching point (zero extent: 0 * inf) depends on it
__device__ __forceinline__ int x86int(float f) { return (f >= -2147483648.0f && f < 2147483648.0f) ? int(f) : int(0x80000000u); }
__device__ __forceinline__ float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }

// ---- 1. bounds: keys[6 * g + a] = min over the point's vertices of component a (a < 3), max for a >= 3
__global__ __launch_bounds__(256) void boundsKernel(const float* __restrict__ pos, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ prims,
    const unsigned long long* __restrict__ offsets, uint32_t* __restrict__ keys)
{
    const uint32_t g = blockIdx.y;
    const unsigned long long first = offsets[g], n = offsets[g + 1] - first;
    uint32_t lo[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, hi[3] = { 0u, 0u, 0u };
    for (unsigned long long j = blockIdx.x * 256ull + threadIdx.x; j < n; j += gridDim.x * 256ull) {
        const uint32_t* t = idx + 3ull * prims[first + j];
        for (int k = 0; k < 3; ++k) {
            const float* p = pos + 3ull * t[k];
            for (int a = 0; a < 3; ++a) {
                const uint32_t key = orderKey(p[a]);
                lo[a] = min(lo[a], key);
                hi[a] = max(hi[a], key);
            }
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = min(lo[a], uint32_t(__shfl_xor(int(lo[a]), o)));
            hi[a] = max(hi[a], uint32_t(__shfl_xor(int(hi[a]), o)));
        }
        if ((threadIdx.x & 63u) == 0u) {
            atomicMin(&keys[6u * g + uint32_t(a)], lo[a]);
            atomicMax(&keys[6u * g + 3u + uint32_t(a)], hi[a]);
        }
    }
}

// ---- 2. voxelisation (triangle_intersect.cpp:224-329; Schwarz & Seidel's conservative triangle / box overlap per voxel of
// the triangle's bounding box). One thread per triangle; expressions and their order as on the host, so the grids are equal.
__global__ __launch_bounds__(256) void voxelizeKernel(const float* __restrict__ pos, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ prims,
    const unsigned long long* __restrict__ offsets, const uint32_t* __restrict__ keys, int res, uint32_t wordsPerGrid, uint32_t* __restrict__ grids)
{
    const uint32_t g = blockIdx.y;
    const unsigned long long first = offsets[g], n = offsets[g + 1] - first;
    float glo[3], ghi[3];
    for (int a = 0; a < 3; ++a) {
        glo[a] = orderValue(keys[6u * g + uint32_t(a)]);
        ghi[a] = orderValue(keys[6u * g + 3u + uint32_t(a)]);
    }
    float scale = maxf2(ghi[0] - glo[0], maxf2(ghi[1] - glo[1], ghi[2] - glo[2]));
    if (!(scale > 0.0f))
        scale = FLT_MIN;
    const float w2v = float(res) / scale, v2w = scale / float(res);
    const float dp = scale / float(res); // world-space voxel edge
    uint32_t* const grid = grids + size_t(g) * wordsPerGrid;
    for (unsigned long long j = blockIdx.x * 256ull + threadIdx.x; j < n; j += gridDim.x * 256ull) {
        const uint32_t* t = idx + 3ull * prims[first + j];
        const float* v[3] = { pos + 3ull * t[0], pos + 3ull * t[1], pos + 3ull * t[2] };
        float e[3][3], nrm[3], tlo[3], thi[3];
        for (int a = 0; a < 3; ++a) {
            e[0][a] = v[1][a] - v[0][a];
            e[1][a] = v[2][a] - v[1][a];
            e[2][a] = v[0][a] - v[2][a];
            tlo[a] = minf2(v[0][a], minf2(v[1][a], v[2][a]));
            thi[a] = maxf2(v[0][a], maxf2(v[1][a], v[2][a]));
        }
        nrm[0] = e[0][1] * e[1][2] - e[1][1] * e[0][2];
        nrm[1] = e[0][2] * e[1][0] - e[1][2] * e[0][0];
        nrm[2] = e[0][0] * e[1][1] - e[1][0] * e[0][1];
        int lo[3], hi[3];
        bool single = true;
        for (int a = 0; a < 3; ++a) {
            lo[a] = min(x86int((tlo[a] - glo[a]) * w2v), res - 1);
            hi[a] = x86int(((tlo[a] + (thi[a] - tlo[a])) - glo[a]) * w2v) + 1;
            if (lo[a] < 0)
                lo[a] = 0;
            if (hi[a] - lo[a] != 1)
                single = false;
        }
        auto setVoxel = [&](int x, int y, int z) {
            const uint32_t p = (uint32_t(z) * uint32_t(res) + uint32_t(y)) * uint32_t(res) + uint32_t(x);
            atomicOr(&grid[p >> 5], 1u << (p & 31u));
        };
        if (single) {
            setVoxel(lo[0], lo[1], lo[2]);
            continue;
        }
        const float c[3] = { nrm[0] > 0 ? dp : 0.0f, nrm[1] > 0 ? dp : 0.0f, nrm[2] > 0 ? dp : 0.0f };
        const float cm[3] = { c[0] - v[0][0], c[1] - v[0][1], c[2] - v[0][2] };
        const float dm[3] = { (dp - c[0]) - v[0][0], (dp - c[1]) - v[0][1], (dp - c[2]) - v[0][2] };
        const float d1 = dot3(nrm, cm), d2 = dot3(nrm, dm);
        const bool useXY = fabsf(nrm[2]) > 0, useZX = fabsf(nrm[1]) > 0, useYZ = fabsf(nrm[0]) > 0;
        const float sXY = nrm[2] >= 0 ? 1.0f : -1.0f, sZX = nrm[1] >= 0 ? -1.0f : 1.0f, sYZ = nrm[0] >= 0 ? 1.0f : -1.0f;
        for (int z = lo[2]; z < min(hi[2], res); ++z)
            for (int y = lo[1]; y < min(hi[1], res); ++y)
                for (int x = lo[0]; x < min(hi[0], res); ++x) {
                    const float p[3] = { float(x) * v2w + glo[0], float(y) * v2w + glo[1], float(z) * v2w + glo[2] };
                    const float np = dot3(nrm, p);
                    if (!(((np + d1) * (np + d2)) <= 0))
                        continue;
                    bool inside = true;
                    for (int i = 0; i < 3 && inside; ++i) {
                        if (useXY) {
                            const float nx = -e[i][1] * sXY, ny = e[i][0] * sXY;
                            const float d = dot2(p[0], p[1], nx, ny) + fmax0(dp * nx) + fmax0(dp * ny) - dot2(nx, ny, v[i][0], v[i][1]);
                            inside &= d >= 0;
                        }
                        if (useZX) {
                            const float nx = -e[i][2] * sZX, nz = e[i][0] * sZX;
                            const float d = dot2(p[0], p[2], nx, nz) + fmax0(dp * nx) + fmax0(dp * nz) - dot2(nx, nz, v[i][0], v[i][2]);
                            inside &= d >= 0;
                        }
                        if (useYZ) {
                            const float ny = -e[i][2] * sYZ, nz = e[i][1] * sYZ;
                            const float d = dot2(p[1], p[2], ny, nz) + fmax0(dp * ny) + fmax0(dp * nz) - dot2(ny, nz, v[i][1], v[i][2]);
                            inside &= d >= 0;
                        }
                    }
                    for (int a = 0; a < 3 && inside; ++a) // Bounds::overlaps of the triangle's box and the voxel's box
                        inside &= !(tlo[a] > p[a] + dp || thi[a] < p[a]);
                    if (inside)
                        setVoxel(x, y, z);
                }
    }
}

// ---- 3. pyramid. Nodes of level l (cells of 2^l voxels per side) in Morton order per batching point: the children of node m
// are the nodes 8 m .. 8 m + 7 of the level below. masks[node] = valid | leaf << 8. A child is a LEAF of its parent when it is
// a voxel (level 1) or a completely filled cell -- the reference promotes a node whose eight children are all leaves to a leaf
// of its own parent, except the root (sparse_voxel_dag.cpp:92-95, 117-119).
__device__ __forceinline__ uint32_t compact3(uint32_t m) // every third bit of m
{
    m &= 0x09249249u;
    m = (m ^ (m >> 2)) & 0x030C30C3u;
    m = (m ^ (m >> 4)) & 0x0300F00Fu;
    m = (m ^ (m >> 8)) & 0x030000FFu;
    m = (m ^ (m >> 16)) & 0x000003FFu;
    return m;
}

__global__ __launch_bounds__(256) void pyramidBaseKernel(const uint32_t* __restrict__ grids, uint32_t wordsPerGrid, uint32_t res, uint32_t nodesPerPoint,
    uint16_t* __restrict__ masks)
{
    const uint32_t g = blockIdx.y;
    const uint32_t* const grid = grids + size_t(g) * wordsPerGrid;
    for (uint32_t m = blockIdx.x * 256u + threadIdx.x; m < nodesPerPoint; m += gridDim.x * 256u) {
        const uint32_t x0 = compact3(m) << 1, y0 = compact3(m >> 1) << 1, z0 = compact3(m >> 2) << 1;
        uint32_t valid = 0;
        for (uint32_t c = 0; c < 8; ++c) {
            const uint32_t p = ((z0 + (c >> 2)) * res + (y0 + ((c >> 1) & 1u))) * res + (x0 + (c & 1u));
            valid |= (grid[p >> 5] >> (p & 31u) & 1u) << c;
        }
        masks[size_t(g) * nodesPerPoint + m] = uint16_t(valid | (valid << 8)); // voxels are leaves
    }
}

__global__ __launch_bounds__(256) void pyramidKernel(const uint16_t* __restrict__ below, uint32_t nodesPerPoint, uint16_t* __restrict__ masks)
{
    const uint32_t g = blockIdx.y;
    const uint16_t* const kids = below + size_t(g) * nodesPerPoint * 8u;
    for (uint32_t m = blockIdx.x * 256u + threadIdx.x; m < nodesPerPoint; m += gridDim.x * 256u) {
        uint32_t valid = 0, leaf = 0;
        for (uint32_t c = 0; c < 8; ++c) {
            const uint32_t k = kids[8u * m + c];
            valid |= uint32_t((k & 0xFFu) != 0u) << c;
            leaf |= uint32_t(k == 0xFFFFu) << c;
        }
        masks[size_t(g) * nodesPerPoint + m] = uint16_t(valid | (leaf << 8));
    }
}

// ---- 4. hash-consing of one level. A node is KEPT (gets words of its own) when it has a valid child and is not promoted.
struct LevelView {
    const uint16_t* masks;     // this level
    const uint32_t* kidCanon;  // canonical offsets of the level below (nullptr at level 1: no inner children there)
    uint32_t nNodes;           // nodes of this level over all batching points
    uint32_t isRoot;           // the top level: never promoted
};

__device__ __forceinline__ bool keptNode(const LevelView& v, uint32_t node)
{
    const uint32_t m = v.masks[node];
    return (m & 0xFFu) != 0u && (v.isRoot || m != 0xFFFFu);
}

struct NodeKey {
    uint32_t desc, n;
    uint32_t kid[8];
};

__device__ __forceinline__ NodeKey nodeKey(const LevelView& v, uint32_t node)
{
    NodeKey k;
    const uint32_t m = v.masks[node], valid = m & 0xFFu, leaf = m >> 8;
    k.desc = leaf | (valid << 8); // descriptor word: leafMask = byte 0, validMask = byte 1
    k.n = 0;
    const uint32_t inner = valid & ~leaf;
    for (uint32_t c = 0; c < 8; ++c)
        if (inner >> c & 1u)
            k.kid[k.n++] = v.kidCanon[8ull * node + c];
    return k;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) // splitmix64's finaliser: every input bit reaches every output bit
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ unsigned long long hashKey(const NodeKey& k)
{
    // (descriptor and offsets are small numbers: each is mixed before the next joins, a plain xor chain would let them cancel)
    unsigned long long h = mix64(k.desc | (unsigned long long)k.n << 32);
    for (uint32_t i = 0; i < k.n; ++i)
        h = mix64(h ^ k.kid[i]);
    return h ? h : 1ull; // 0 marks an empty slot
}

// One table for all levels: equal subtrees also meet across levels (a cell whose children are all leaves has the same key at
// any size). A slot claimed at an earlier level has its words written already (offset); a slot claimed at this level collects
// the lowest index among this level's nodes that share the key (rep) and gets its offset when that node writes its words.
constexpr uint32_t kNoOffset = 0xFFFFFFFFu;
struct Slot {
    unsigned long long hash;
    uint32_t rep, offset;
};

__global__ __launch_bounds__(256) void tableInitKernel(Slot* __restrict__ table, uint32_t n)
{
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u)
        table[i] = Slot { 0ull, 0xFFFFFFFFu, kNoOffset }; // (rep starts at the largest value: atomicMin finds the lowest node index)
}

__global__ __launch_bounds__(256) void internKernel(LevelView v, Slot* __restrict__ table, uint32_t tableMask, uint32_t* __restrict__ slotOf)
{
    for (uint32_t node = blockIdx.x * 256u + threadIdx.x; node < v.nNodes; node += gridDim.x * 256u) {
        if (!keptNode(v, node)) {
            slotOf[node] = 0xFFFFFFFFu;
            continue;
        }
        const unsigned long long h = hashKey(nodeKey(v, node));
        uint32_t at = uint32_t(h) & tableMask;
        // (look before the atomic: the lowest level has millions of nodes and at most 255 different keys -- once a key is in its
        // slot and a low index has been seen there, later arrivals only read)
        for (;;) {
            unsigned long long prev = __hip_atomic_load(&table[at].hash, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == 0ull)
                prev = atomicCAS(&table[at].hash, 0ull, h);
            if (prev == 0ull || prev == h)
                break;
            at = (at + 1u) & tableMask; // (the table is at most half full: the probe ends)
        }
        if (table[at].offset == kNoOffset // (offsets are written by sealKernel only: stable during this launch)
            && __hip_atomic_load(&table[at].rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > node)
            atomicMin(&table[at].rep, node);
        slotOf[node] = at;
    }
}

// sizes[node] = words of a representative, 0 for everybody else; keptWords counts the words without merging; a key that
// differs from its representative's (a 64-bit hash collision) raises `collision`
__global__ __launch_bounds__(256) void sizeKernel(LevelView v, const Slot* __restrict__ table, const uint32_t* __restrict__ slotOf, const uint32_t* __restrict__ words,
    uint32_t* __restrict__ sizes, unsigned long long* __restrict__ keptWords, uint32_t* __restrict__ collision)
{
    unsigned long long mine = 0;
    for (uint32_t node = blockIdx.x * 256u + threadIdx.x; node < v.nNodes; node += gridDim.x * 256u) {
        uint32_t size = 0;
        if (slotOf[node] != 0xFFFFFFFFu) {
            const Slot s = table[slotOf[node]];
            const NodeKey k = nodeKey(v, node);
            mine += 1u + k.n;
            bool same = true;
            if (s.offset != kNoOffset) { // an earlier level wrote these words: they must spell this node's key
                same = words[s.offset] == k.desc;
                for (uint32_t i = 0; same && i < k.n; ++i)
                    same = words[s.offset + 1u + i] == k.kid[i];
            } else if (s.rep == node) {
                size = 1u + k.n;
            } else {
                const NodeKey r = nodeKey(v, s.rep);
                same = r.desc == k.desc && r.n == k.n;
                for (uint32_t i = 0; same && i < k.n; ++i)
                    same = r.kid[i] == k.kid[i];
            }
            if (!same)
                atomicOr(collision, 1u);
        }
        sizes[node] = size;
    }
    for (int o = 32; o > 0; o >>= 1)
        mine += (unsigned long long)__shfl_xor((long long)mine, o);
    if ((threadIdx.x & 63u) == 0u && mine)
        atomicAdd(keptWords, mine);
}

// representatives write their words; every kept node learns its canonical offset
__global__ __launch_bounds__(256) void emitKernel(LevelView v, const Slot* __restrict__ table, const uint32_t* __restrict__ slotOf, const uint32_t* __restrict__ offsets,
    uint32_t base, uint32_t* __restrict__ words, uint32_t* __restrict__ canon)
{
    for (uint32_t node = blockIdx.x * 256u + threadIdx.x; node < v.nNodes; node += gridDim.x * 256u) {
        if (slotOf[node] == 0xFFFFFFFFu) {
            canon[node] = 0u; // (never referenced: its parent lists it as absent or as a leaf)
            continue;
        }
        const Slot s = table[slotOf[node]];
        if (s.offset != kNoOffset && s.rep == 0xFFFFFFFFu) { // (a slot of an earlier level)
            canon[node] = s.offset;
            continue;
        }
        const uint32_t at = base + offsets[s.rep];
        canon[node] = at;
        if (s.rep == node) {
            const NodeKey k = nodeKey(v, node);
            words[at] = k.desc;
            for (uint32_t i = 0; i < k.n; ++i)
                words[at + 1u + i] = k.kid[i];
        }
    }
}

// the slots this level claimed get their offsets and forget their representative (node indices are per level)
__global__ __launch_bounds__(256) void sealKernel(LevelView v, Slot* __restrict__ table, const uint32_t* __restrict__ slotOf, const uint32_t* __restrict__ canon)
{
    for (uint32_t node = blockIdx.x * 256u + threadIdx.x; node < v.nNodes; node += gridDim.x * 256u) {
        if (slotOf[node] == 0xFFFFFFFFu)
            continue;
        Slot& s = table[slotOf[node]];
        if (s.rep == node) {
            s.offset = canon[node];
            s.rep = 0xFFFFFFFFu;
        }
    }
}

struct Scratch { // device allocations of one build, freed on every way out
    std::vector<void*> ptrs;
    ~Scratch()
    {
        for (void* p : ptrs)
            (void)hipFree(p);
    }
    template <typename T>
    hipError_t alloc(T** out, size_t count)
    {
        void* p = nullptr;
        const hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess)
            ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
};

bool powerOfTwo(uint32_t v) { return v && !(v & (v - 1)); }

} // namespace

extern "C" int pgpu_build_svdags_device(pgpu_ctx* c, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* group_prims, const uint64_t* group_offsets, size_t n_groups, uint32_t resolution, pgpu_host_svdags** out)
{
    if (!c || !positions || !indices || !group_prims || !group_offsets || !out || n_groups == 0)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_svdags_device: null or empty argument");
    if (!powerOfTwo(resolution) || resolution < 4 || resolution > 256)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_svdags_device: resolution must be a power of two in 4..256 (pgpu_build_svdags builds larger ones on the host)");
    if (n_groups > 65535 || n_vertices > 0xFFFFFFFFull / 3 || n_triangles > 0xFFFFFFFFull / 3)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_svdags_device: more than 65535 batching points or 2^32 / 3 vertices / triangles");
    const uint64_t nEntries = group_offsets[n_groups];
    for (size_t g = 0; g < n_groups; ++g)
        if (group_offsets[g] > group_offsets[g + 1])
            return fail(c, PGPU_ERR_INVALID, "pgpu_build_svdags_device: group offsets must not decrease");
    for (uint64_t i = 0; i < nEntries; ++i)
        if (group_prims[i] >= n_triangles)
            return fail(c, PGPU_ERR_INVALID, "pgpu_build_svdags_device: primitive index out of range");
    for (size_t i = 0; i < 3 * n_triangles; ++i)
        if (indices[i] >= n_vertices)
            return fail(c, PGPU_ERR_INVALID, "pgpu_build_svdags_device: vertex index out of range");
    int depth = 0;
    while ((1u << depth) < resolution)
        ++depth;
    // nodes per batching point and level (level l: cells of 2^l voxels per side), level 1 .. depth
    std::vector<uint32_t> perPoint(size_t(depth) + 1, 0);
    uint64_t totalNodes = 0;
    for (int l = 1; l <= depth; ++l) {
        const uint64_t side = resolution >> l;
        perPoint[size_t(l)] = uint32_t(side * side * side);
        totalNodes += uint64_t(perPoint[size_t(l)]) * n_groups;
    }
    if (totalNodes > 0x7FFFFFFFull || uint64_t(perPoint[1]) * n_groups > 0x7FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_svdags_device: resolution x batching points too large for one build (2^31 nodes)");

    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = c->stream;
    Scratch mem;
    const uint32_t G = uint32_t(n_groups);
    const uint32_t wordsPerGrid = uint32_t((uint64_t(resolution) * resolution * resolution + 31) / 32);
    float* dPos = nullptr;
    uint32_t *dIdx = nullptr, *dPrims = nullptr, *dKeys = nullptr, *dGrids = nullptr;
    unsigned long long* dOffsets = nullptr;
    HIP_TRY(c, mem.alloc(&dPos, 3 * n_vertices));
    HIP_TRY(c, mem.alloc(&dIdx, 3 * n_triangles));
    HIP_TRY(c, mem.alloc(&dPrims, size_t(nEntries)));
    HIP_TRY(c, mem.alloc(&dOffsets, n_groups + 1));
    HIP_TRY(c, mem.alloc(&dKeys, 6 * n_groups));
    HIP_TRY(c, mem.alloc(&dGrids, size_t(G) * wordsPerGrid));
    HIP_TRY(c, hipMemcpyAsync(dPos, positions, 3 * n_vertices * sizeof(float), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dIdx, indices, 3 * n_triangles * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dPrims, group_prims, size_t(nEntries) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dOffsets, group_offsets, (n_groups + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    std::vector<uint32_t> keyInit(6 * n_groups);
    for (size_t g = 0; g < n_groups; ++g)
        for (int a = 0; a < 3; ++a) {
            keyInit[6 * g + size_t(a)] = 0xFFFFFFFFu;
            keyInit[6 * g + 3 + size_t(a)] = 0u;
        }
    HIP_TRY(c, hipMemcpyAsync(dKeys, keyInit.data(), keyInit.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemsetAsync(dGrids, 0, size_t(G) * wordsPerGrid * 4, stream));

    // blocks per batching point: enough to spread the largest point over the chip without starving the small ones
    const uint64_t avg = std::max<uint64_t>(1, nEntries / n_groups);
    const uint32_t bx = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(64, (avg + 1023) / 1024)));
    hipLaunchKernelGGL(boundsKernel, dim3(bx, G), dim3(256), 0, stream, dPos, dIdx, dPrims, dOffsets, dKeys);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(voxelizeKernel, dim3(bx, G), dim3(256), 0, stream, dPos, dIdx, dPrims, dOffsets, dKeys, int(resolution), wordsPerGrid, dGrids);
    HIP_TRY(c, hipGetLastError());

    // pyramid: masks of every level, canonical offsets of every level
    std::vector<uint16_t*> dMasks(size_t(depth) + 1, nullptr);
    std::vector<uint32_t*> dCanon(size_t(depth) + 1, nullptr);
    for (int l = 1; l <= depth; ++l) {
        HIP_TRY(c, mem.alloc(&dMasks[size_t(l)], size_t(perPoint[size_t(l)]) * G));
        HIP_TRY(c, mem.alloc(&dCanon[size_t(l)], size_t(perPoint[size_t(l)]) * G));
        const uint32_t blocks = std::max(1u, std::min(256u, (perPoint[size_t(l)] + 255u) / 256u));
        if (l == 1)
            hipLaunchKernelGGL(pyramidBaseKernel, dim3(blocks, G), dim3(256), 0, stream, dGrids, wordsPerGrid, resolution, perPoint[1], dMasks[1]);
        else
            hipLaunchKernelGGL(pyramidKernel, dim3(blocks, G), dim3(256), 0, stream, dMasks[size_t(l) - 1], perPoint[size_t(l)], dMasks[size_t(l)]);
        HIP_TRY(c, hipGetLastError());
    }

    // hash-consing, bottom-up. The largest level decides the table and the scan buffers.
    const uint32_t maxLevelNodes = perPoint[1] * G;
    uint32_t tableSize = 1024;
    while (uint64_t(tableSize) < 2u * totalNodes && tableSize < 0x80000000u)
        tableSize <<= 1;
    Slot* dTable = nullptr;
    uint32_t *dSlotOf = nullptr, *dSizes = nullptr, *dOffs = nullptr, *dCollision = nullptr;
    unsigned long long* dKeptWords = nullptr;
    HIP_TRY(c, mem.alloc(&dTable, tableSize));
    HIP_TRY(c, mem.alloc(&dSlotOf, maxLevelNodes));
    HIP_TRY(c, mem.alloc(&dSizes, size_t(maxLevelNodes) + 1));
    HIP_TRY(c, mem.alloc(&dOffs, size_t(maxLevelNodes) + 1));
    HIP_TRY(c, mem.alloc(&dCollision, 1));
    HIP_TRY(c, mem.alloc(&dKeptWords, 1));
    HIP_TRY(c, hipMemsetAsync(dCollision, 0, 4, stream));
    HIP_TRY(c, hipMemsetAsync(dKeptWords, 0, 8, stream));
    size_t scanBytes = 0;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, dSizes, dOffs, int(maxLevelNodes) + 1, stream));
    unsigned char* dScanTemp = nullptr;
    HIP_TRY(c, mem.alloc(&dScanTemp, scanBytes));
    hipLaunchKernelGGL(tableInitKernel, dim3(std::min(tableSize / 256u, uint32_t(c->numCUs) * 16u)), dim3(256), 0, stream, dTable, tableSize);
    HIP_TRY(c, hipGetLastError());

    uint32_t* dWords = nullptr;
    size_t wordsCap = 0, wordsUsed = 0;
    auto growWords = [&](size_t need) -> hipError_t {
        if (need <= wordsCap)
            return hipSuccess;
        size_t cap = std::max<size_t>(need, std::max<size_t>(2 * wordsCap, size_t(1) << 20));
        uint32_t* bigger = nullptr;
        hipError_t e = mem.alloc(&bigger, cap);
        if (e != hipSuccess)
            return e;
        if (wordsUsed)
            e = hipMemcpyAsync(bigger, dWords, wordsUsed * 4, hipMemcpyDeviceToDevice, stream);
        dWords = bigger; // (the old block is freed with the rest)
        wordsCap = cap;
        return e;
    };
    for (int l = 1; l <= depth; ++l) {
        LevelView v {};
        v.masks = dMasks[size_t(l)];
        v.kidCanon = l > 1 ? dCanon[size_t(l) - 1] : nullptr;
        v.nNodes = perPoint[size_t(l)] * G;
        v.isRoot = l == depth ? 1u : 0u;
        const uint32_t blocks = std::max(1u, std::min(uint32_t(c->numCUs) * 8u, (v.nNodes + 255u) / 256u));
        hipLaunchKernelGGL(internKernel, dim3(blocks), dim3(256), 0, stream, v, dTable, tableSize - 1u, dSlotOf);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(sizeKernel, dim3(blocks), dim3(256), 0, stream, v, dTable, dSlotOf, dWords, dSizes, dKeptWords, dCollision);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemsetAsync(dSizes + v.nNodes, 0, 4, stream));
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(dScanTemp, scanBytes, dSizes, dOffs, int(v.nNodes) + 1, stream));
        uint32_t levelWords = 0;
        HIP_TRY(c, hipMemcpyAsync(&levelWords, dOffs + v.nNodes, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        if (wordsUsed + levelWords > 0xFFFFFFFFull)
            return fail(c, PGPU_ERR_INVALID, "pgpu_build_svdags_device: the DAG array would exceed 2^32 words");
        HIP_TRY(c, growWords(wordsUsed + levelWords));
        hipLaunchKernelGGL(emitKernel, dim3(blocks), dim3(256), 0, stream, v, dTable, dSlotOf, dOffs, uint32_t(wordsUsed), dWords, dCanon[size_t(l)]);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(sealKernel, dim3(blocks), dim3(256), 0, stream, v, dTable, dSlotOf, dCanon[size_t(l)]);
        HIP_TRY(c, hipGetLastError());
        wordsUsed += levelWords;
    }

    // roots, bounds, and the array itself
    std::vector<uint32_t> rootCanon(n_groups), keys(6 * n_groups);
    uint32_t collision = 0;
    unsigned long long keptWords = 0;
    HIP_TRY(c, hipMemcpyAsync(rootCanon.data(), dCanon[size_t(depth)], n_groups * 4, hipMemcpyDeviceToHost, stream));
    std::vector<uint16_t> rootMasks(n_groups);
    HIP_TRY(c, hipMemcpyAsync(rootMasks.data(), dMasks[size_t(depth)], n_groups * 2, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipMemcpyAsync(keys.data(), dKeys, keys.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipMemcpyAsync(&collision, dCollision, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipMemcpyAsync(&keptWords, dKeptWords, 8, hipMemcpyDeviceToHost, stream));
    std::vector<uint32_t> words(wordsUsed + 1);
    if (wordsUsed)
        HIP_TRY(c, hipMemcpyAsync(words.data(), dWords, wordsUsed * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    if (collision)
        return fail(c, PGPU_ERR_HIP, "pgpu_build_svdags_device: two different nodes met under one 64-bit hash (build on the host with pgpu_build_svdags)");
    // an empty batching point (no triangles, or none voxelised) gets the empty root the host builder gives it
    uint32_t emptyRoot = 0xFFFFFFFFu;
    std::vector<pgpu_svdag_desc> descs(n_groups);
    for (size_t g = 0; g < n_groups; ++g) {
        pgpu_svdag_desc& d = descs[g];
        if ((rootMasks[g] & 0xFFu) == 0u) {
            if (emptyRoot == 0xFFFFFFFFu) {
                emptyRoot = uint32_t(wordsUsed);
                words[wordsUsed++] = 0u;
                keptWords += 1;
            }
            d.root = emptyRoot;
        } else {
            d.root = rootCanon[g];
        }
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = group_offsets[g] == group_offsets[g + 1] ? FLT_MAX : orderValue(keys[6 * g + size_t(a)]);
            hi[a] = group_offsets[g] == group_offsets[g + 1] ? -FLT_MAX : orderValue(keys[6 * g + 3 + size_t(a)]);
            d.bounds_min[a] = lo[a];
        }
        const float ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
        d.inv_extent = 1.0f / ext; // sparse_voxel_dag.cpp:25-26
        d.resolution = resolution;
    }
    words.resize(wordsUsed);
    *out = makeHostSvdags(std::move(words), std::move(descs), size_t(keptWords));
    return PGPU_OK;
}
