// Host-side construction of the sparse-voxel-DAG proxy geometry used for the early-out test at
// batching points. Follows the reference's pipeline
//   voxelise  : TriangleShape::voxelize            projects/pandora/src/shapes/triangle_intersect.cpp:224-329
//   grid      : VoxelGrid                          projects/pandora/src/svo/voxel_grid.cpp:10-179
//   SVO       : constructSVOBreadthFirst           projects/pandora/src/svo/sparse_voxel_dag.cpp:39-133
//   DAG merge : SparseVoxelDAG::compressDAGs       projects/pandora/src/svo/sparse_voxel_dag.cpp:135-226
// and emits the reference's node format (sparse_voxel_dag.h:40-78): one u32 array; a node is a
// descriptor word (leafMask = byte 0, validMask = byte 1) followed by one ABSOLUTE u32 offset per
// inner child (valid & ~leaf) in child-index order; child index bit0 = x, bit1 = y, bit2 = z.
#include "pandora_hip.h"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace {

struct Grid {
    uint32_t res;
    float lo[3], hi[3];
    std::vector<uint32_t> bits;
    Grid(uint32_t r)
        : res(r)
        , bits((size_t(r) * r * r + 31) / 32, 0u)
    {
    }
    void set(int x, int y, int z)
    {
        const size_t p = (size_t(z) * res + size_t(y)) * res + size_t(x);
        bits[p >> 5] |= 1u << (p & 31);
    }
    bool get(uint32_t x, uint32_t y, uint32_t z) const
    {
        const size_t p = (size_t(z) * res + size_t(y)) * res + size_t(x);
        return bits[p >> 5] >> (p & 31) & 1u;
    }
};

inline float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
inline float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }

// Schwarz & Seidel conservative triangle/box overlap per voxel, as the reference applies it.
void voxelizeTriangle(Grid& g, const float* v0, const float* v1, const float* v2)
{
    const int res = int(g.res);
    float scale = std::max(g.hi[0] - g.lo[0], std::max(g.hi[1] - g.lo[1], g.hi[2] - g.lo[2]));
    if (!(scale > 0.0f))
        scale = FLT_MIN;
    const float w2v = float(res) / scale, v2w = scale / float(res);
    const float dp = scale / float(res); // world-space voxel edge
    const float* v[3] = { v0, v1, v2 };
    float e[3][3], n[3], tlo[3], thi[3];
    for (int a = 0; a < 3; ++a) {
        e[0][a] = v1[a] - v0[a];
        e[1][a] = v2[a] - v1[a];
        e[2][a] = v0[a] - v2[a];
        tlo[a] = std::min(v0[a], std::min(v1[a], v2[a]));
        thi[a] = std::max(v0[a], std::max(v1[a], v2[a]));
    }
    n[0] = e[0][1] * e[1][2] - e[1][1] * e[0][2];
    n[1] = e[0][2] * e[1][0] - e[1][2] * e[0][0];
    n[2] = e[0][0] * e[1][1] - e[1][0] * e[0][1];
    int lo[3], hi[3];
    bool single = true;
    for (int a = 0; a < 3; ++a) {
        lo[a] = std::min(int((tlo[a] - g.lo[a]) * w2v), res - 1);
        hi[a] = int(((tlo[a] + (thi[a] - tlo[a])) - g.lo[a]) * w2v) + 1;
        if (lo[a] < 0)
            lo[a] = 0;
        if (hi[a] - lo[a] != 1)
            single = false;
    }
    if (single) {
        g.set(lo[0], lo[1], lo[2]);
        return;
    }
    const float c[3] = { n[0] > 0 ? dp : 0.0f, n[1] > 0 ? dp : 0.0f, n[2] > 0 ? dp : 0.0f };
    const float cm[3] = { c[0] - v0[0], c[1] - v0[1], c[2] - v0[2] };
    const float dm[3] = { (dp - c[0]) - v0[0], (dp - c[1]) - v0[1], (dp - c[2]) - v0[2] };
    const float d1 = dot3(n, cm), d2 = dot3(n, dm);
    for (int z = lo[2]; z < std::min(hi[2], res); ++z)
        for (int y = lo[1]; y < std::min(hi[1], res); ++y)
            for (int x = lo[0]; x < std::min(hi[0], res); ++x) {
                const float p[3] = { float(x) * v2w + g.lo[0], float(y) * v2w + g.lo[1], float(z) * v2w + g.lo[2] };
                const float np = dot3(n, p);
                if (!(((np + d1) * (np + d2)) <= 0))
                    continue;
                bool inside = true;
                for (int i = 0; i < 3 && inside; ++i) {
                    if (std::fabs(n[2]) > 0) { // XY projection
                        const float s = n[2] >= 0 ? 1.0f : -1.0f;
                        const float nx = -e[i][1] * s, ny = e[i][0] * s;
                        const float d = dot2(p[0], p[1], nx, ny) + std::max(0.0f, dp * nx) + std::max(0.0f, dp * ny) - dot2(nx, ny, v[i][0], v[i][1]);
                        inside &= d >= 0;
                    }
                    if (std::fabs(n[1]) > 0) { // ZX projection
                        const float s = n[1] >= 0 ? -1.0f : 1.0f;
                        const float nx = -e[i][2] * s, nz = e[i][0] * s;
                        const float d = dot2(p[0], p[2], nx, nz) + std::max(0.0f, dp * nx) + std::max(0.0f, dp * nz) - dot2(nx, nz, v[i][0], v[i][2]);
                        inside &= d >= 0;
                    }
                    if (std::fabs(n[0]) > 0) { // YZ projection
                        const float s = n[0] >= 0 ? 1.0f : -1.0f;
                        const float ny = -e[i][2] * s, nz = e[i][1] * s;
                        const float d = dot2(p[1], p[2], ny, nz) + std::max(0.0f, dp * ny) + std::max(0.0f, dp * nz) - dot2(ny, nz, v[i][1], v[i][2]);
                        inside &= d >= 0;
                    }
                }
                // Bounds::overlaps of triangle box and voxel box
                for (int a = 0; a < 3 && inside; ++a)
                    inside &= !(tlo[a] > p[a] + dp || thi[a] < p[a]);
                if (inside)
                    g.set(x, y, z);
            }
}

inline void mortonDecode(uint32_t m, uint32_t& x, uint32_t& y, uint32_t& z)
{
    x = y = z = 0;
    for (uint32_t b = 0; b < 10; ++b) {
        x |= ((m >> (3 * b)) & 1u) << b;
        y |= ((m >> (3 * b + 1)) & 1u) << b;
        z |= ((m >> (3 * b + 2)) & 1u) << b;
    }
}

inline uint32_t descWord(uint32_t valid, uint32_t leaf) { return (leaf & 0xFFu) | ((valid & 0xFFu) << 8); }

// Bottom-up, Morton-ordered octree construction; a fully occupied node is promoted to a leaf of its
// parent (except at the root), sparse_voxel_dag.cpp:92-95,117-119.
uint32_t buildSvo(const Grid& g, std::vector<uint32_t>& out)
{
    struct Lvl {
        uint32_t morton;
        bool leaf;
        uint32_t offset;
    };
    int depth = 0;
    while ((1u << depth) < g.res)
        ++depth;
    std::vector<Lvl> prev, cur;
    const uint32_t total = g.res * g.res * g.res;
    for (uint32_t m = 0; m < total; ++m) {
        uint32_t x, y, z;
        mortonDecode(m, x, y, z);
        if (g.get(x, y, z))
            cur.push_back({ m, true, 0 });
    }
    auto store = [&](uint32_t valid, uint32_t leaf, const std::vector<uint32_t>& kids) {
        const uint32_t at = uint32_t(out.size());
        out.push_back(descWord(valid, leaf));
        out.insert(out.end(), kids.begin(), kids.end());
        return at;
    };
    if (cur.empty())
        return store(0, 0, {}); // nothing voxelised: an empty root never reports a hit
    uint32_t root = 0;
    for (int level = 0; level < depth; ++level) {
        std::swap(prev, cur);
        cur.clear();
        uint32_t valid = 0, leaf = 0;
        std::vector<uint32_t> kids;
        uint32_t open = prev[0].morton >> 3;
        for (const Lvl& c : prev) {
            const uint32_t parentCode = c.morton >> 3;
            if (parentCode != open) {
                if ((valid & leaf) == 0xFFu)
                    cur.push_back({ open, true, 0 });
                else
                    cur.push_back({ open, false, store(valid, leaf, kids) });
                valid = leaf = 0;
                kids.clear();
                open = parentCode;
            }
            const uint32_t slot = c.morton & 7u;
            valid |= 1u << slot;
            if (c.leaf)
                leaf |= 1u << slot;
            else
                kids.push_back(c.offset);
        }
        if ((valid & leaf) == 0xFFu && level != depth - 1) {
            cur.push_back({ open, true, 0 });
        } else {
            root = store(valid, leaf, kids);
            cur.push_back({ open, false, root });
        }
    }
    return root;
}

struct NodeKey {
    uint32_t desc;
    uint32_t n;
    uint32_t kid[8];
    bool operator==(const NodeKey& o) const { return desc == o.desc && n == o.n && std::memcmp(kid, o.kid, n * 4) == 0; }
};
struct NodeKeyHash {
    size_t operator()(const NodeKey& k) const
    {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ k.desc;
        for (uint32_t i = 0; i < k.n; ++i)
            h = (h ^ k.kid[i]) * 0xff51afd7ed558ccdull + 0x632BE59BD9B4E019ull;
        return size_t(h ^ (h >> 29));
    }
};

// Hash-consing of all sub-trees of all SVOs into one array (compressDAGs).
struct DagMerger {
    std::vector<uint32_t> words;
    std::unordered_map<NodeKey, uint32_t, NodeKeyHash> seen;
    uint32_t intern(const std::vector<uint32_t>& svo, uint32_t node)
    {
        const uint32_t d = svo[node];
        const uint32_t inner = ((d >> 8) & 0xFFu) & ~(d & 0xFFu);
        NodeKey key;
        std::memset(&key, 0, sizeof(key));
        key.desc = d;
        uint32_t slot = 0;
        for (uint32_t c = 0; c < 8; ++c)
            if (inner >> c & 1u)
                key.kid[key.n++] = intern(svo, svo[node + 1 + slot++]);
        auto it = seen.find(key);
        if (it != seen.end())
            return it->second;
        const uint32_t at = uint32_t(words.size());
        words.push_back(d);
        for (uint32_t i = 0; i < key.n; ++i)
            words.push_back(key.kid[i]);
        seen.emplace(key, at);
        return at;
    }
};

void groupBounds(const float* pos, const uint32_t* idx, const uint32_t* prims, size_t n, float lo[3], float hi[3])
{
    for (int a = 0; a < 3; ++a) {
        lo[a] = FLT_MAX;
        hi[a] = -FLT_MAX;
    }
    for (size_t j = 0; j < n; ++j)
        for (int k = 0; k < 3; ++k) {
            const float* p = pos + 3u * size_t(idx[3u * size_t(prims[j]) + size_t(k)]);
            for (int a = 0; a < 3; ++a) {
                lo[a] = std::min(lo[a], p[a]);
                hi[a] = std::max(hi[a], p[a]);
            }
        }
}

void voxelizeGroup(Grid& g, const float* pos, const uint32_t* idx, const uint32_t* prims, size_t n)
{
    groupBounds(pos, idx, prims, n, g.lo, g.hi);
    for (size_t j = 0; j < n; ++j) {
        const uint32_t* t = idx + 3u * size_t(prims[j]);
        voxelizeTriangle(g, pos + 3u * size_t(t[0]), pos + 3u * size_t(t[1]), pos + 3u * size_t(t[2]));
    }
}

bool powerOfTwo(uint32_t v) { return v && !(v & (v - 1)); }

} // namespace

extern "C" {

struct pgpu_host_svdags {
    std::vector<uint32_t> words;
    std::vector<pgpu_svdag_desc> descs;
    size_t wordsBefore = 0;
};

} // extern "C"

namespace pandora_hip {
pgpu_host_svdags* makeHostSvdags(std::vector<uint32_t>&& words, std::vector<pgpu_svdag_desc>&& descs, size_t wordsBeforeCompression)
{
    auto* r = new pgpu_host_svdags();
    r->words = std::move(words);
    r->descs = std::move(descs);
    r->wordsBefore = wordsBeforeCompression;
    return r;
}
}

extern "C" {

int pgpu_build_svdags(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* group_prims, const uint64_t* group_offsets, size_t n_groups, uint32_t resolution,
    pgpu_host_svdags** out)
{
    if (!positions || !indices || !group_prims || !group_offsets || !out || n_groups == 0)
        return PGPU_ERR_INVALID;
    if (!powerOfTwo(resolution) || resolution < 4 || resolution > 1024)
        return PGPU_ERR_INVALID;
    for (uint64_t i = 0; i < group_offsets[n_groups]; ++i)
        if (group_prims[i] >= n_triangles)
            return PGPU_ERR_INVALID;
    for (size_t i = 0; i < 3 * n_triangles; ++i)
        if (indices[i] >= n_vertices)
            return PGPU_ERR_INVALID;
    auto* r = new pgpu_host_svdags();
    DagMerger merger;
    for (size_t gi = 0; gi < n_groups; ++gi) {
        Grid g(resolution);
        const size_t n = size_t(group_offsets[gi + 1] - group_offsets[gi]);
        voxelizeGroup(g, positions, indices, group_prims + group_offsets[gi], n);
        std::vector<uint32_t> svo;
        const uint32_t root = buildSvo(g, svo);
        r->wordsBefore += svo.size();
        pgpu_svdag_desc d;
        d.root = merger.intern(svo, root);
        const float ext = std::max(g.hi[0] - g.lo[0], std::max(g.hi[1] - g.lo[1], g.hi[2] - g.lo[2]));
        for (int a = 0; a < 3; ++a)
            d.bounds_min[a] = g.lo[a];
        d.inv_extent = 1.0f / ext; // sparse_voxel_dag.cpp:25-26
        d.resolution = resolution;
        r->descs.push_back(d);
    }
    r->words = std::move(merger.words);
    *out = r;
    return PGPU_OK;
}
const uint32_t* pgpu_host_svdags_words(const pgpu_host_svdags* s, size_t* n)
{
    if (n)
        *n = s->words.size();
    return s->words.data();
}
const pgpu_svdag_desc* pgpu_host_svdags_descs(const pgpu_host_svdags* s, size_t* n)
{
    if (n)
        *n = s->descs.size();
    return s->descs.data();
}
size_t pgpu_host_svdags_words_before_compression(const pgpu_host_svdags* s) { return s->wordsBefore; }
void pgpu_host_svdags_free(pgpu_host_svdags* s) { delete s; }

int pgpu_voxelize(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* prims, size_t n_prims, uint32_t resolution, float bounds_out[6], uint32_t* grid_bits)
{
    if (!positions || !indices || !prims || !grid_bits || !powerOfTwo(resolution) || resolution > 1024)
        return PGPU_ERR_INVALID;
    for (size_t j = 0; j < n_prims; ++j) {
        if (prims[j] >= n_triangles)
            return PGPU_ERR_INVALID;
        for (int k = 0; k < 3; ++k)
            if (indices[3u * size_t(prims[j]) + size_t(k)] >= n_vertices)
                return PGPU_ERR_INVALID;
    }
    Grid g(resolution);
    voxelizeGroup(g, positions, indices, prims, n_prims);
    std::memcpy(grid_bits, g.bits.data(), g.bits.size() * 4);
    if (bounds_out)
        for (int a = 0; a < 3; ++a) {
            bounds_out[a] = g.lo[a];
            bounds_out[3 + a] = g.hi[a];
        }
    return PGPU_OK;
}

int pgpu_svdag_rebase(uint32_t* words, size_t n_words, uint32_t base)
{
    if (!words && n_words)
        return PGPU_ERR_INVALID;
    if (uint64_t(n_words) + base > 0xFFFFFFFFull)
        return PGPU_ERR_INVALID;
    // first pass: the array must be a whole number of nodes whose offsets stay inside it (nothing is changed on failure)
    for (int pass = 0; pass < 2; ++pass) {
        size_t at = 0;
        while (at < n_words) {
            const uint32_t d = words[at];
            if (d >> 16)
                return PGPU_ERR_INVALID;
            const uint32_t inner = ((d >> 8) & 0xFFu) & ~(d & 0xFFu);
            const size_t k = size_t(__builtin_popcount(inner));
            if (at + 1 + k > n_words)
                return PGPU_ERR_INVALID;
            for (size_t i = 1; i <= k; ++i) {
                if (pass == 0) {
                    if (words[at + i] >= n_words)
                        return PGPU_ERR_INVALID;
                } else {
                    words[at + i] += base;
                }
            }
            at += 1 + k;
        }
    }
    return PGPU_OK;
}

int pgpu_svdag_expand(const uint32_t* words, size_t n_words, const pgpu_svdag_desc* dag, uint32_t* grid_bits)
{
    if (!words || !dag || !grid_bits || dag->root >= n_words || !powerOfTwo(dag->resolution))
        return PGPU_ERR_INVALID;
    const uint32_t res = dag->resolution;
    std::memset(grid_bits, 0, ((size_t(res) * res * res + 31) / 32) * 4);
    struct Todo {
        uint32_t node, x, y, z, size;
    };
    std::vector<Todo> todo { { dag->root, 0, 0, 0, res } };
    while (!todo.empty()) {
        const Todo t = todo.back();
        todo.pop_back();
        if (t.node >= n_words)
            return PGPU_ERR_INVALID;
        const uint32_t d = words[t.node], valid = (d >> 8) & 0xFFu, leaf = d & 0xFFu;
        const uint32_t half = t.size / 2;
        uint32_t slot = 0;
        for (uint32_t c = 0; c < 8; ++c) {
            if (!(valid >> c & 1u))
                continue;
            const uint32_t cx = t.x + (c & 1u) * half, cy = t.y + ((c >> 1) & 1u) * half, cz = t.z + ((c >> 2) & 1u) * half;
            if (leaf >> c & 1u) {
                for (uint32_t z = cz; z < cz + half; ++z)
                    for (uint32_t y = cy; y < cy + half; ++y)
                        for (uint32_t x = cx; x < cx + half; ++x) {
                            const size_t p = (size_t(z) * res + y) * res + x;
                            grid_bits[p >> 5] |= 1u << (p & 31);
                        }
            } else {
                if (half == 0)
                    return PGPU_ERR_INVALID;
                todo.push_back({ words[t.node + 1 + slot], cx, cy, cz, half });
                ++slot;
            }
        }
    }
    return PGPU_OK;
}
}
