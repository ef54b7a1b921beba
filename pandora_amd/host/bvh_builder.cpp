// Host-side acceleration-structure construction for the MI355X traversal path.
//
// The reference delegates every BVH build to Embree 3's rtcBuildBVH (bottom level:
// wive_bvh8_build8_impl.h:87-104, branching 8, leaves 1..4; top level: pauseable_bvh4.h:160-176,
// branching 4, one batching point per leaf; scene cut: batching.cpp:106-159). Embree is not part of
// this system, so this file is an own top-down binned-SAH wide builder that emits the reference's
// node layouts bit for bit (256-byte BVH8 node with the 24-bit octant permutation words of
// wive_bvh8_build8_impl.h:172-203; 112-byte BVH4 node in the DFS order of pauseable_bvh4.h:398-483).
// Tree topology therefore differs from an Embree-built tree; hit parity is defined per tree.
//
// Depth is bounded by construction: the bottom level never nests more than kMaxInnerDepth8 inner
// nodes so that the traversal stack (7 * depth + 8 entries, wive_bvh8_impl.h:74,97-102) fits the
// reference's 96 entries, and the top level stays within 16 levels (4 stack bits per level in a
// 64-bit word, pauseable_bvh4.h:262-263).
#include "pandora_hip.h"
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <condition_variable>
#include <mutex>
#include <new>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

// Host worker threads this process may use: PANDORA_HOST_THREADS when set, else the node's CPUs divided by the ranks that share it
// (LOCAL_WORLD_SIZE, set by torch.distributed.run and bench.py's launcher), at most 16.
static unsigned hostThreads()
{
    if (const char* e = std::getenv("PANDORA_HOST_THREADS"))
        if (const long v = std::strtol(e, nullptr, 10); v > 0)
            return unsigned(std::min<long>(v, 256));
    long local = 1;
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE"))
        local = std::max<long>(1, std::strtol(e, nullptr, 10));
    const long hw = long(std::thread::hardware_concurrency());
    return unsigned(std::max<long>(1, std::min<long>(16, hw / local)));
}

namespace {

constexpr int kBins = 16;
constexpr int kMaxInnerDepth8 = 12; // 7*12 + 8 <= 96
constexpr int kMaxDepth4 = 15;      // depth field 0..15

struct Box {
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX };
    float hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    void grow(const float* p)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box& b)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float halfArea() const
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};

struct Item {
    Box box;
    float c[3];
    uint32_t id;
    bool alone = false; // must sit in a leaf of its own (an instance: its leaf is a nested traversal, not a triangle test)
};

struct Part {
    size_t begin, end;
    Box box;
    size_t count() const { return end - begin; }
};

// a part that may become one leaf: at most maxLeaf objects, and an object that must be alone is alone
bool leafable(const std::vector<Item>& items, size_t b, size_t e, size_t maxLeaf)
{
    if (e - b <= 1)
        return true;
    if (e - b > maxLeaf)
        return false;
    for (size_t i = b; i < e; ++i)
        if (items[i].alone)
            return false;
    return true;
}

Box boundsOf(const std::vector<Item>& items, size_t b, size_t e)
{
    Box r;
    for (size_t i = b; i < e; ++i)
        r.grow(items[i].box);
    return r;
}

// Binned SAH split of items[begin,end); falls back to an index median when every centroid coincides.
// forceMedian gives the balanced split used when a subtree would otherwise exceed the depth budget.
// quantum != 0: a side weighs ceil(objects / quantum) instead of its objects (the scene cut, pgpu_partition_prims).
size_t splitPart(std::vector<Item>& items, size_t begin, size_t end, bool forceMedian, size_t quantum = 0, float mix = 0.0f)
{
    const size_t n = end - begin;
    Box cb;
    for (size_t i = begin; i < end; ++i)
        cb.grow(items[i].c);
    int bestAxis = -1, bestBin = -1;
    float bestCost = FLT_MAX;
    if (!forceMedian) {
        for (int a = 0; a < 3; ++a) {
            const float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.0f))
                continue;
            const float scale = float(kBins) / ext;
            Box bb[kBins];
            size_t bc[kBins] = {};
            for (size_t i = begin; i < end; ++i) {
                int b = std::min(kBins - 1, std::max(0, int((items[i].c[a] - cb.lo[a]) * scale)));
                bb[b].grow(items[i].box);
                bc[b]++;
            }
            float rightArea[kBins];
            size_t rightCount[kBins];
            Box acc;
            size_t cnt = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(bb[b]);
                cnt += bc[b];
                rightArea[b] = cnt ? acc.halfArea() : 0.0f;
                rightCount[b] = cnt;
            }
            acc = Box();
            cnt = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                acc.grow(bb[b]);
                cnt += bc[b];
                if (cnt == 0 || rightCount[b + 1] == 0)
                    continue;
                auto weight = [&](size_t c) { return quantum ? float((c + quantum - 1) / quantum) + mix * float(c) / float(quantum) : float(c); };
                const float cost = acc.halfArea() * weight(cnt) + rightArea[b + 1] * weight(rightCount[b + 1]);
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = a;
                    bestBin = b;
                }
            }
        }
    }
    if (bestAxis >= 0) {
        const float ext = cb.hi[bestAxis] - cb.lo[bestAxis];
        const float scale = float(kBins) / ext;
        const float lo = cb.lo[bestAxis];
        auto mid = std::partition(items.begin() + begin, items.begin() + end, [&](const Item& it) {
            int b = std::min(kBins - 1, std::max(0, int((it.c[bestAxis] - lo) * scale)));
            return b <= bestBin;
        });
        size_t m = size_t(mid - items.begin());
        if (m > begin && m < end)
            return m;
    }
    // median along the widest centroid axis (or plain index median for identical centroids)
    int axis = 0;
    for (int a = 1; a < 3; ++a)
        if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis])
            axis = a;
    const size_t m = begin + n / 2;
    std::nth_element(items.begin() + begin, items.begin() + m, items.begin() + end,
        [axis](const Item& x, const Item& y) { return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.id < y.id); });
    return m;
}

// Splits [begin,end) into at most `width` parts, largest surface first; parts of <= maxLeaf items stay whole.
// capacity = the most items a child subtree may hold given the remaining depth budget.
std::vector<Part> makeParts(std::vector<Item>& items, size_t begin, size_t end, int width, size_t maxLeaf, double capacity)
{
    std::vector<Part> parts;
    parts.push_back({ begin, end, boundsOf(items, begin, end) });
    bool balanced = double(end - begin) > capacity * width * 0.5; // close to the budget: go balanced straight away
    for (;;) {
        // every part above capacity must be split, whatever its area
        int pick = -1;
        float bestArea = -1.0f;
        bool over = false;
        for (size_t i = 0; i < parts.size(); ++i) {
            if (leafable(items, parts[i].begin, parts[i].end, maxLeaf))
                continue;
            const bool isOver = double(parts[i].count()) > capacity;
            const float a = parts[i].box.halfArea();
            if ((isOver && !over) || (isOver == over && a > bestArea)) {
                pick = int(i);
                bestArea = a;
                over = isOver;
            }
        }
        if (pick < 0 || int(parts.size()) >= width)
            break;
        Part p = parts[size_t(pick)];
        const size_t m = splitPart(items, p.begin, p.end, balanced);
        parts[size_t(pick)] = { p.begin, m, boundsOf(items, p.begin, m) };
        parts.insert(parts.begin() + pick + 1, { m, p.end, boundsOf(items, m, p.end) });
    }
    bool ok = true;
    for (const Part& p : parts)
        if (double(p.count()) > capacity)
            ok = false;
    if (!ok && !balanced) {
        // SAH produced a lopsided cut that the depth budget cannot absorb: redo with median cuts
        parts.clear();
        parts.push_back({ begin, end, boundsOf(items, begin, end) });
        for (;;) {
            int pick = -1;
            size_t most = 0;
            for (size_t i = 0; i < parts.size(); ++i)
                if (parts[i].count() > most && !leafable(items, parts[i].begin, parts[i].end, maxLeaf)) {
                    most = parts[i].count();
                    pick = int(i);
                }
            if (pick < 0 || int(parts.size()) >= width)
                break;
            Part p = parts[size_t(pick)];
            const size_t m = splitPart(items, p.begin, p.end, true);
            parts[size_t(pick)] = { p.begin, m, boundsOf(items, p.begin, m) };
            parts.insert(parts.begin() + pick + 1, { m, p.end, boundsOf(items, m, p.end) });
        }
    }
    return parts;
}

// ---------------------------------------------------------------------------------------------- BVH8

struct Node8 { // wive_bvh8.h:76-86
    float minX[8], maxX[8], minY[8], maxY[8], minZ[8], maxZ[8];
    uint32_t children[8];
    uint32_t perm[8];
};
static_assert(sizeof(Node8) == 256, "BVH8 node must be 256 bytes");

inline uint32_t handleInner(uint32_t i) { return (i & 0x1FFFFFFFu) | (2u << 29); }                   // wive_bvh8_impl.h:333-338
inline uint32_t handleLeaf(uint32_t i, uint32_t n) { return (i & 0x1FFFFFFFu) | ((4u | (n - 1u)) << 29); } // :340-346

// wive_bvh8_build8_impl.h:172-203: for each of the 8 ray-direction octants, child indices ordered
// back to front; slot i of the word holds the i-th child of that order in bits [shift, shift+3).
void octantPermutation(const Box* kids, unsigned n, uint32_t perm[8])
{
    std::fill(perm, perm + 8, 0u);
    for (int x = -1; x <= 1; x += 2)
        for (int y = -1; y <= 1; y += 2)
            for (int z = -1; z <= 1; z += 2) {
                float dist[8];
                uint32_t order[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
                for (unsigned i = 0; i < 8; ++i) {
                    if (i >= n) {
                        dist[i] = FLT_MAX;
                        continue;
                    }
                    const float ex = x < 0 ? kids[i].lo[0] : kids[i].hi[0];
                    const float ey = y < 0 ? kids[i].lo[1] : kids[i].hi[1];
                    const float ez = z < 0 ? kids[i].lo[2] : kids[i].hi[2];
                    dist[i] = (ex * float(x) + ey * float(y)) + ez * float(z);
                }
                std::stable_sort(order, order + 8, [&](uint32_t a, uint32_t b) { return dist[a] > dist[b]; });
                const uint32_t shift = ((x > 0 ? 1u : 0u) | (y > 0 ? 2u : 0u) | (z > 0 ? 4u : 0u)) * 3u;
                for (int i = 0; i < 8; ++i)
                    perm[i] |= order[i] << shift;
            }
}

struct Bvh8Builder {
    std::vector<Item> items;
    std::vector<Node8> nodes;
    std::vector<uint32_t> leafIdx;
    uint32_t maxLeaf = 4;
    int maxDepthSeen = 0;

    uint32_t emitLeaf(const Part& p)
    {
        const uint32_t at = uint32_t(leafIdx.size());
        for (size_t i = p.begin; i < p.end; ++i)
            leafIdx.push_back(items[i].id);
        return handleLeaf(at, uint32_t(p.count()));
    }

    // ---- second construction: a binary SAH tree down to single objects, collapsed into 8-wide nodes by dynamic programming
    // (Ylitie, Karras, Laine 2017, section 3: the cheapest wide tree that a given binary tree contains). The greedy
    // construction below splits whichever part is largest until a node has 8 children and turns every part of at most maxLeaf
    // objects into a leaf on sight: near the bottom that leaves nodes with two or three leaves under parents whose slots are
    // taken (3.96 children per node on the benchmark atrium). The collapse decides per binary subtree whether it becomes a
    // leaf, one wide node, or several children of its ancestor's wide node, by what the traversal pays: a node visit per unit of
    // surface area, a leaf visit likewise (a leaf of up to four triangles is one step of the traversal kernels).
    struct Bin {
        size_t begin, end;
        Box box;
        int left = -1, right = -1;
        float cost[9];   // cost[j]: cheapest forest of at most j wide-tree roots (leaves or nodes) that covers this subtree; [1] = one root
        uint8_t cut[9];  // cut[j] (j >= 2): roots given to the left subtree; 0 = the subtree stays one root
        uint8_t inner1;  // the one-root form is a wide node (1) or a leaf (0)
        uint8_t kLeft;   // one-root wide node: roots taken from the left subtree (the rest of the 8 from the right)
        uint8_t height;  // inner nodes on the longest path of the one-root form
    };
    std::vector<Bin> bins;
    float costInner = 1.0f, costLeaf = 1.0f, costObject = 0.0f;
    int reinsertPasses = 10;
    float reinsertFraction = 0.01f;

    // exact SAH over the three axes for small ranges (sorted sweep), binned above that
    size_t splitBinary(size_t begin, size_t end, bool forceMedian)
    {
        const size_t n = end - begin;
        if (n > 64 || forceMedian)
            return splitPart(items, begin, end, forceMedian);
        int bestAxis = -1;
        size_t bestAt = 0;
        float bestCost = FLT_MAX;
        std::vector<float> rightArea(n);
        for (int a = 0; a < 3; ++a) {
            std::sort(items.begin() + begin, items.begin() + end,
                [a](const Item& x, const Item& y) { return x.c[a] < y.c[a] || (x.c[a] == y.c[a] && x.id < y.id); });
            Box acc;
            for (size_t i = n - 1; i > 0; --i) {
                acc.grow(items[begin + i].box);
                rightArea[i] = acc.halfArea();
            }
            acc = Box();
            for (size_t i = 0; i + 1 < n; ++i) {
                acc.grow(items[begin + i].box);
                const float cost = acc.halfArea() * float(i + 1) + rightArea[i + 1] * float(n - i - 1);
                if (cost < bestCost) {
                    bestCost = cost;
                    bestAxis = a;
                    bestAt = i + 1;
                }
            }
        }
        if (bestAxis < 0) // (no split has a cost that compares: boxes that are not numbers -- a balanced split, as for coincident centroids)
            return splitPart(items, begin, end, true);
        if (bestAxis != 2)
            std::sort(items.begin() + begin, items.begin() + end,
                [bestAxis](const Item& x, const Item& y) { return x.c[bestAxis] < y.c[bestAxis] || (x.c[bestAxis] == y.c[bestAxis] && x.id < y.id); });
        return begin + bestAt;
    }

    int buildBinary(size_t begin, size_t end, int depth = 0)
    {
        const int self = int(bins.size());
        bins.emplace_back();
        bins[size_t(self)].begin = begin;
        bins[size_t(self)].end = end;
        bins[size_t(self)].box = boundsOf(items, begin, end);
        if (end - begin > 1) {
            const size_t m = splitBinary(begin, end, depth > 48); // (a chain of lopsided splits ends in balanced ones)
            const int l = buildBinary(begin, m, depth + 1);
            const int r = buildBinary(m, end, depth + 1);
            bins[size_t(self)].left = l;
            bins[size_t(self)].right = r;
        }
        return self;
    }

    // ---- the binary tree restructured before the collapse: subtrees taken out and put back where they add the least surface area
    // (Bittner, Hapala, Havran 2013, "Fast insertion-based optimization of bounding volume hierarchies"). Per pass the inner
    // nodes that look worst (area x area / smallest child x area / mean child) lose their place: the node N and its parent P are
    // freed, N's sibling moves up, and N's two subtrees are reinserted, the larger first, each under one of the freed nodes at the
    // position a branch-and-bound search finds. The tree keeps its node count and every object stays one leaf; afterwards the
    // nodes are renumbered parent first and the objects reordered so that every subtree is a range again (what solve and the
    // emission expect).
    struct Link {
        int parent, left, right;
    };

    static float unionArea(const Box& a, const Box& b)
    {
        Box u = a;
        u.grow(b);
        return u.halfArea();
    }

    void refit(std::vector<Link>& links, int i)
    {
        for (; i >= 0; i = links[size_t(i)].parent) {
            Box u = bins[size_t(links[size_t(i)].left)].box;
            u.grow(bins[size_t(links[size_t(i)].right)].box);
            Box& was = bins[size_t(i)].box;
            if (std::memcmp(&was, &u, sizeof(Box)) == 0)
                break;
            was = u;
        }
    }

    void optimiseBinary(int passes, float fraction)
    {
        const size_t n = bins.size();
        if (n < 64)
            return;
        std::vector<Link> links(n);
        links[0].parent = -1;
        for (size_t i = 0; i < n; ++i) {
            links[i].left = bins[i].left;
            links[i].right = bins[i].right;
            if (bins[i].left >= 0) {
                links[size_t(bins[i].left)].parent = int(i);
                links[size_t(bins[i].right)].parent = int(i);
            }
        }
        int root = 0;
        const bool debug = std::getenv("PANDORA_BVH8_REINSERT_DEBUG") != nullptr;
        std::vector<std::pair<float, int>> worst;
        std::vector<std::pair<float, int>> heap; // (-induced cost, node): std::push_heap keeps the largest = least induced cost on top
        auto place = [&](int x, int freeNode) {
            const Box xb = bins[size_t(x)].box;
            const float xa = xb.halfArea();
            float best = FLT_MAX;
            int at = root;
            heap.clear();
            heap.push_back({ -0.0f, root });
            while (!heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const float induced = -heap.back().first;
                const int y = heap.back().second;
                heap.pop_back();
                if (!(induced + xa < best))
                    break;
                const float total = induced + unionArea(bins[size_t(y)].box, xb);
                if (total < best) {
                    best = total;
                    at = y;
                }
                const float below = total - bins[size_t(y)].box.halfArea();
                if (links[size_t(y)].left >= 0 && below + xa < best) {
                    heap.push_back({ -below, links[size_t(y)].left });
                    std::push_heap(heap.begin(), heap.end());
                    heap.push_back({ -below, links[size_t(y)].right });
                    std::push_heap(heap.begin(), heap.end());
                }
            }
            // freeNode takes at's place, with at and x below it
            const int up = links[size_t(at)].parent;
            links[size_t(freeNode)] = { up, at, x };
            if (up < 0)
                root = freeNode;
            else if (links[size_t(up)].left == at)
                links[size_t(up)].left = freeNode;
            else
                links[size_t(up)].right = freeNode;
            links[size_t(at)].parent = freeNode;
            links[size_t(x)].parent = freeNode;
            bins[size_t(freeNode)].box = Box(); // (so that the refit below sees a change)
            refit(links, freeNode);
        };
        for (int pass = 0; pass < passes; ++pass) {
            worst.clear();
            for (size_t i = 0; i < n; ++i) {
                const Link& k = links[i];
                if (k.left < 0 || k.parent < 0 || links[size_t(k.parent)].parent < 0)
                    continue;
                const float a = bins[i].box.halfArea();
                const float al = bins[size_t(k.left)].box.halfArea(), ar = bins[size_t(k.right)].box.halfArea();
                const float smallest = std::min(al, ar), mean = 0.5f * (al + ar);
                if (!(a > 0.0f) || !(smallest > 0.0f))
                    continue;
                worst.push_back({ a * (a / smallest) * (a / mean), int(i) });
            }
            const size_t take = std::min(worst.size(), std::max<size_t>(1, size_t(double(worst.size()) * fraction)));
            if (take == 0)
                break;
            std::partial_sort(worst.begin(), worst.begin() + take, worst.end(),
                [](const std::pair<float, int>& x, const std::pair<float, int>& y) { return x.first > y.first || (x.first == y.first && x.second < y.second); });
            for (size_t w = 0; w < take; ++w) {
                const int N = worst[w].second;
                const int P = links[size_t(N)].parent;
                if (P < 0 || links[size_t(P)].parent < 0) // (an earlier reinsertion of this pass moved it next to the root)
                    continue;
                const int G = links[size_t(P)].parent;
                const int S = links[size_t(P)].left == N ? links[size_t(P)].right : links[size_t(P)].left;
                int L = links[size_t(N)].left, R = links[size_t(N)].right;
                if (links[size_t(G)].left == P)
                    links[size_t(G)].left = S;
                else
                    links[size_t(G)].right = S;
                links[size_t(S)].parent = G;
                refit(links, G);
                if (bins[size_t(L)].box.halfArea() < bins[size_t(R)].box.halfArea())
                    std::swap(L, R);
                place(L, N);
                place(R, P);
            }
            if (debug && (pass % 10 == 9 || pass + 1 == passes)) {
                double sum = 0.0;
                for (size_t i = 0; i < n; ++i)
                    if (links[i].left >= 0)
                        sum += bins[i].box.halfArea();
                std::fprintf(stderr, "reinsertion pass %d: inner area / root area = %.3f\n", pass + 1, sum / bins[size_t(root)].box.halfArea());
            }
        }
        relinearise(links, root);
    }

    // renumber parent first, reorder the objects by the tree's leaf order
    void relinearise(const std::vector<Link>& links, int root)
    {
        const size_t n = bins.size();
        std::vector<Bin> out;
        out.reserve(n);
        std::vector<Item> ordered;
        ordered.reserve(items.size());
        size_t base = items.size();
        for (const Bin& b : bins)
            base = std::min(base, b.begin);
        for (size_t i = 0; i < base; ++i)
            ordered.push_back(items[i]);
        std::vector<int> newIndex(n, -1);
        struct Todo {
            int node, stage;
        };
        std::vector<Todo> stack;
        stack.push_back({ root, 0 });
        while (!stack.empty()) {
            Todo& t = stack.back();
            const Link& k = links[size_t(t.node)];
            if (t.stage == 0) {
                newIndex[size_t(t.node)] = int(out.size());
                out.emplace_back();
                Bin& b = out.back();
                b.box = bins[size_t(t.node)].box;
                b.begin = ordered.size();
                if (k.left < 0) {
                    ordered.push_back(items[bins[size_t(t.node)].begin]);
                    b.end = ordered.size();
                    stack.pop_back();
                    continue;
                }
                t.stage = 1;
                stack.push_back({ k.left, 0 });
            } else if (t.stage == 1) {
                t.stage = 2;
                stack.push_back({ k.right, 0 });
            } else {
                Bin& b = out[size_t(newIndex[size_t(t.node)])];
                b.left = newIndex[size_t(k.left)];
                b.right = newIndex[size_t(k.right)];
                b.end = ordered.size();
                stack.pop_back();
            }
        }
        for (size_t i = ordered.size(); i < items.size(); ++i)
            ordered.push_back(items[i]);
        items.swap(ordered);
        bins.swap(out);
    }

    void solve(int at)
    {
        // (children first; the binary tree is as deep as the splits make it, so no recursion here: nodes were created parent first)
        for (int i = int(bins.size()) - 1; i >= at; --i) {
            Bin& b = bins[size_t(i)];
            const size_t n = b.end - b.begin;
            const float area = b.box.halfArea();
            const float asLeaf = leafable(items, b.begin, b.end, maxLeaf) ? area * (costLeaf + costObject * float(n)) : FLT_MAX;
            if (b.left < 0) {
                for (int j = 1; j <= 8; ++j) {
                    b.cost[j] = asLeaf;
                    b.cut[j] = 0;
                }
                b.inner1 = 0;
                b.kLeft = 0;
                b.height = 0;
                continue;
            }
            const Bin &l = bins[size_t(b.left)], &r = bins[size_t(b.right)];
            float bestKids = FLT_MAX;
            uint8_t bestK = 1;
            for (int k = 1; k <= 7; ++k) {
                const float c = l.cost[k] + r.cost[8 - k];
                if (c < bestKids) {
                    bestKids = c;
                    bestK = uint8_t(k);
                }
            }
            const float asInner = area * costInner + bestKids;
            // (more than a leaf holds is a node whatever the numbers say: surfaces that overflow or are not numbers compare false)
            b.inner1 = (asLeaf == FLT_MAX || asInner < asLeaf) ? 1 : 0;
            b.kLeft = bestK;
            b.cost[1] = b.inner1 ? asInner : asLeaf;
            b.cut[1] = 0;
            for (int j = 2; j <= 8; ++j) {
                float best = b.cost[1];
                uint8_t cut = 0;
                for (int k = 1; k < j; ++k) {
                    const float c = l.cost[k] + r.cost[j - k];
                    if (c < best) {
                        best = c;
                        cut = uint8_t(k);
                    }
                }
                b.cost[j] = best;
                b.cut[j] = cut;
            }
            b.height = 0;
            if (b.inner1) {
                int roots[8];
                const int nr = gather(b.left, b.kLeft, roots, 0);
                const int nn = gather(b.right, 8 - b.kLeft, roots, nr);
                uint8_t h = 0;
                for (int q = 0; q < nn; ++q)
                    h = std::max(h, bins[size_t(roots[q])].height);
                b.height = uint8_t(std::min(255, int(h) + 1));
            }
        }
    }

    // the roots of the cheapest forest of at most j roots under binary node i
    int gather(int i, int j, int* roots, int n) const
    {
        const Bin& b = bins[size_t(i)];
        if (j <= 1 || b.cut[j] == 0) {
            roots[n] = i;
            return n + 1;
        }
        n = gather(b.left, b.cut[j], roots, n);
        return gather(b.right, j - b.cut[j], roots, n);
    }

    uint32_t emitCollapsed(int i, int depth)
    {
        const Bin b = bins[size_t(i)];
        if (!b.inner1)
            return emitLeaf(Part { b.begin, b.end, b.box });
        maxDepthSeen = std::max(maxDepthSeen, depth + 1);
        int roots[8];
        const int nl = gather(b.left, b.kLeft, roots, 0);
        const unsigned n = unsigned(gather(b.right, 8 - b.kLeft, roots, nl));
        const uint32_t self = uint32_t(nodes.size());
        nodes.emplace_back();
        Box kids[8];
        uint32_t handles[8];
        for (unsigned q = 0; q < n; ++q) {
            kids[q] = bins[size_t(roots[q])].box;
            handles[q] = emitCollapsed(roots[q], depth + 1);
        }
        fillNode(self, kids, handles, n);
        return handleInner(self);
    }

    // false: the collapsed tree would nest deeper than the traversal stack allows (the caller falls back to the greedy construction,
    // which balances its splits against that budget)
    bool buildCollapsed(const Part& whole, uint32_t* root)
    {
        bins.clear();
        bins.reserve(2 * whole.count());
        buildBinary(whole.begin, whole.end);
        if (reinsertPasses > 0)
            optimiseBinary(reinsertPasses, reinsertFraction);
        solve(0);
        if (int(bins[0].height) > kMaxInnerDepth8) {
            bins.clear();
            return false;
        }
        *root = emitCollapsed(0, 0);
        bins.clear();
        bins.shrink_to_fit();
        return true;
    }

    void fillNode(uint32_t self, const Box* kids, const uint32_t* handles, unsigned n)
    {
        Node8& nd = nodes[self];
        std::memset(&nd, 0, sizeof(nd)); // absent children: zero boxes, handle 0 (wive_bvh8_build8_impl.h:142-144,168-170)
        for (unsigned i = 0; i < n; ++i) {
            nd.minX[i] = kids[i].lo[0];
            nd.maxX[i] = kids[i].hi[0];
            nd.minY[i] = kids[i].lo[1];
            nd.maxY[i] = kids[i].hi[1];
            nd.minZ[i] = kids[i].lo[2];
            nd.maxZ[i] = kids[i].hi[2];
            nd.children[i] = handles[i];
        }
        octantPermutation(kids, n, nd.perm);
    }

    uint32_t buildTree(const Part& whole)
    {
        const char* e = std::getenv("PANDORA_BVH8_BUILDER");
        if (const char* v = std::getenv("PANDORA_SAH_LEAF"))
            costLeaf = float(std::atof(v));
        if (const char* v = std::getenv("PANDORA_SAH_OBJECT"))
            costObject = float(std::atof(v));
        if (const char* v = std::getenv("PANDORA_BVH8_REINSERT"))
            reinsertPasses = std::atoi(v);
        if (const char* v = std::getenv("PANDORA_BVH8_REINSERT_FRACTION"))
            reinsertFraction = float(std::atof(v));
        uint32_t root = 0;
        if (!(e && std::strcmp(e, "greedy") == 0) && buildCollapsed(whole, &root))
            return root;
        return build(whole, 0);
    }

    // depth = number of inner nodes above this one
    uint32_t build(const Part& whole, int depth)
    {
        if (leafable(items, whole.begin, whole.end, maxLeaf))
            return emitLeaf(whole);
        maxDepthSeen = std::max(maxDepthSeen, depth + 1);
        // a child at inner depth d+1 may hold at most maxLeaf * 8^(kMax - (d+1)) items
        const double capacity = double(maxLeaf) * std::pow(8.0, double(kMaxInnerDepth8 - (depth + 1)));
        std::vector<Part> parts = makeParts(items, whole.begin, whole.end, 8, maxLeaf, capacity);
        const uint32_t self = uint32_t(nodes.size());
        nodes.emplace_back();
        Box kids[8];
        uint32_t handles[8];
        const unsigned n = unsigned(parts.size());
        for (unsigned i = 0; i < n; ++i) {
            kids[i] = parts[i].box;
            handles[i] = build(parts[i], depth + 1);
        }
        fillNode(self, kids, handles, n);
        return handleInner(self);
    }
};

// ---------------------------------------------------------------------------------------------- BVH4

struct Node4 { // pauseable_bvh4.h:74-98
    float minX[4], minY[4], minZ[4], maxX[4], maxY[4], maxZ[4];
    uint32_t firstChild, firstLeaf, parent, bits;
};
static_assert(sizeof(Node4) == 112, "BVH4 node must be 112 bytes");

struct Bvh4Builder {
    std::vector<Item> items;
    std::vector<Node4> nodes;
    std::vector<uint32_t> leafOrder; // final leaf slot -> input leaf id
    int maxDepthSeen = 0;

    // Fills nodes[self] (already allocated) from items[part); mirrors generateFinalBVHRecurse :411-483:
    // leaf children are appended to the leaf list in child order, inner children get consecutive
    // node slots allocated before any of them is filled.
    void fill(uint32_t self, const Part& whole, uint32_t parent, int depth)
    {
        maxDepthSeen = std::max(maxDepthSeen, depth);
        const double capacity = std::pow(4.0, double(kMaxDepth4 - depth));
        std::vector<Part> parts = makeParts(items, whole.begin, whole.end, 4, 1, capacity);
        Node4 nd;
        std::memset(&nd, 0, sizeof(nd));
        uint32_t valid = 0, leaf = 0;
        for (size_t i = 0; i < parts.size(); ++i) {
            nd.minX[i] = parts[i].box.lo[0];
            nd.minY[i] = parts[i].box.lo[1];
            nd.minZ[i] = parts[i].box.lo[2];
            nd.maxX[i] = parts[i].box.hi[0];
            nd.maxY[i] = parts[i].box.hi[1];
            nd.maxZ[i] = parts[i].box.hi[2];
            valid |= 1u << i;
            if (parts[i].count() == 1)
                leaf |= 1u << i;
        }
        nd.parent = parent;
        nd.bits = valid | (leaf << 4) | (uint32_t(depth) << 8);
        if (leaf) {
            nd.firstLeaf = uint32_t(leafOrder.size());
            for (size_t i = 0; i < parts.size(); ++i)
                if (parts[i].count() == 1)
                    leafOrder.push_back(items[parts[i].begin].id);
        }
        const uint32_t nInner = uint32_t(__builtin_popcount(valid & ~leaf));
        if (nInner) {
            nd.firstChild = uint32_t(nodes.size());
            nodes.resize(nodes.size() + nInner);
        }
        nodes[self] = nd;
        uint32_t next = nd.firstChild;
        for (size_t i = 0; i < parts.size(); ++i)
            if (parts[i].count() > 1)
                fill(next++, parts[i], self, depth + 1);
    }
};

struct HostBvh8 {
    std::vector<Node8> nodes;
    std::vector<uint32_t> leafIdx;
    uint32_t root = 0;
    int depth = 0;
};
struct HostBvh4 {
    std::vector<Node4> nodes;
    std::vector<uint32_t> leafOrder;
    uint32_t root = 0;
    int depth = 0;
};

Item triangleItem(const float* pos, const uint32_t* idx, uint32_t prim, uint32_t id)
{
    Item it;
    for (int k = 0; k < 3; ++k)
        it.box.grow(pos + 3u * size_t(idx[3u * size_t(prim) + k]));
    for (int a = 0; a < 3; ++a)
        it.c[a] = 0.5f * (it.box.lo[a] + it.box.hi[a]);
    it.id = id;
    return it;
}

} // namespace

extern "C" {

struct pgpu_host_bvh8 {
    HostBvh8 v;
};
struct pgpu_host_bvh4 {
    HostBvh4 v;
};

int pgpu_build_bvh8(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* prim_ids, size_t n_prims, uint32_t max_leaf, pgpu_host_bvh8** out)
{
    if (!positions || !indices || !out || n_prims == 0 || max_leaf < 1 || max_leaf > 4)
        return PGPU_ERR_INVALID;
    if (n_prims >= (1u << 29) - 1u)
        return PGPU_ERR_INVALID;
    for (size_t j = 0; j < n_prims; ++j) {
        const size_t p = prim_ids ? prim_ids[j] : j;
        if (p >= n_triangles)
            return PGPU_ERR_INVALID;
        for (int k = 0; k < 3; ++k)
            if (indices[3 * p + size_t(k)] >= n_vertices)
                return PGPU_ERR_INVALID;
    }
    Bvh8Builder b;
    b.maxLeaf = max_leaf;
    b.items.reserve(n_prims);
    for (size_t j = 0; j < n_prims; ++j) // leaf object j = triangle prim_ids[j] (offline_bvh_cache.cpp:303-304)
        b.items.push_back(triangleItem(positions, indices, prim_ids ? prim_ids[j] : uint32_t(j), uint32_t(j)));
    Part whole { 0, n_prims, boundsOf(b.items, 0, n_prims) };
    auto* r = new pgpu_host_bvh8();
    r->v.root = b.buildTree(whole);
    r->v.nodes = std::move(b.nodes);
    r->v.leafIdx = std::move(b.leafIdx);
    r->v.depth = b.maxDepthSeen;
    *out = r;
    return PGPU_OK;
}

int pgpu_build_bvh8_boxes(const float* bounds6, size_t n_objects, const uint8_t* alone, uint32_t max_leaf, pgpu_host_bvh8** out)
{
    if (!bounds6 || !out || n_objects == 0 || max_leaf < 1 || max_leaf > 4 || n_objects >= (1u << 29) - 1u)
        return PGPU_ERR_INVALID;
    Bvh8Builder b;
    b.maxLeaf = max_leaf;
    b.items.resize(n_objects);
    for (size_t j = 0; j < n_objects; ++j) {
        Item& it = b.items[j];
        for (int a = 0; a < 3; ++a) {
            it.box.lo[a] = bounds6[6 * j + size_t(a)];
            it.box.hi[a] = bounds6[6 * j + 3 + size_t(a)];
            if (!(it.box.lo[a] <= it.box.hi[a]))
                return PGPU_ERR_INVALID;
            it.c[a] = 0.5f * (it.box.lo[a] + it.box.hi[a]);
        }
        it.id = uint32_t(j);
        it.alone = alone && alone[j];
    }
    Part whole { 0, n_objects, boundsOf(b.items, 0, n_objects) };
    auto* r = new pgpu_host_bvh8();
    r->v.root = b.buildTree(whole);
    r->v.nodes = std::move(b.nodes);
    r->v.leafIdx = std::move(b.leafIdx);
    r->v.depth = b.maxDepthSeen;
    *out = r;
    return PGPU_OK;
}

const void* pgpu_host_bvh8_nodes(const pgpu_host_bvh8* b, size_t* n_nodes)
{
    if (n_nodes)
        *n_nodes = b->v.nodes.size();
    return b->v.nodes.data();
}
const uint32_t* pgpu_host_bvh8_leaf_indices(const pgpu_host_bvh8* b, size_t* n)
{
    if (n)
        *n = b->v.leafIdx.size();
    return b->v.leafIdx.data();
}
uint32_t pgpu_host_bvh8_root(const pgpu_host_bvh8* b) { return b->v.root; }
int pgpu_host_bvh8_depth(const pgpu_host_bvh8* b) { return b->v.depth; }
void pgpu_host_bvh8_free(pgpu_host_bvh8* b) { delete b; }

int pgpu_build_top_bvh4(const float* leaf_bounds, size_t n_leaves, pgpu_host_bvh4** out)
{
    if (!leaf_bounds || !out)
        return PGPU_ERR_INVALID;
    if (n_leaves < 2) // pauseable_bvh4.h:180: a single batching point is rejected by the reference too
        return PGPU_ERR_SINGLE_LEAF;
    Bvh4Builder b;
    b.items.resize(n_leaves);
    for (size_t i = 0; i < n_leaves; ++i) {
        Item& it = b.items[i];
        for (int a = 0; a < 3; ++a) {
            it.box.lo[a] = leaf_bounds[6 * i + size_t(a)];
            it.box.hi[a] = leaf_bounds[6 * i + 3 + size_t(a)];
            it.c[a] = 0.5f * (it.box.lo[a] + it.box.hi[a]);
        }
        it.id = uint32_t(i);
    }
    b.nodes.emplace_back();
    b.fill(0, Part { 0, n_leaves, boundsOf(b.items, 0, n_leaves) }, 0, 0);
    if (b.maxDepthSeen > kMaxDepth4)
        return PGPU_ERR_DEPTH;
    auto* r = new pgpu_host_bvh4();
    r->v.nodes = std::move(b.nodes);
    r->v.leafOrder = std::move(b.leafOrder);
    r->v.root = 0;
    r->v.depth = b.maxDepthSeen;
    *out = r;
    return PGPU_OK;
}
const void* pgpu_host_bvh4_nodes(const pgpu_host_bvh4* b, size_t* n_nodes)
{
    if (n_nodes)
        *n_nodes = b->v.nodes.size();
    return b->v.nodes.data();
}
const uint32_t* pgpu_host_bvh4_leaf_order(const pgpu_host_bvh4* b, size_t* n)
{
    if (n)
        *n = b->v.leafOrder.size();
    return b->v.leafOrder.data();
}
uint32_t pgpu_host_bvh4_root(const pgpu_host_bvh4* b) { return b->v.root; }
int pgpu_host_bvh4_depth(const pgpu_host_bvh4* b) { return b->v.depth; }
void pgpu_host_bvh4_free(pgpu_host_bvh4* b) { delete b; }

// Spatial grouping of triangles into batching points of at most max_prims primitives. The reference
// cuts an Embree BVH2 over scene objects where a subtree holds <= primitivesPerBatchingPoint
// (batching.cpp:38-265); here the same rule is applied to a binned-SAH split over triangles.
int pgpu_partition_prims(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    size_t max_prims, uint32_t* group_of_prim, uint32_t* n_groups)
{
    if (!positions || !indices || !group_of_prim || !n_groups || max_prims == 0)
        return PGPU_ERR_INVALID;
    const unsigned nThreads = hostThreads();
    auto parallelFor = [&](size_t n, auto&& body) { // body(begin, end) on contiguous slices
        const size_t slices = n < (size_t(1) << 16) ? 1 : nThreads;
        std::vector<std::thread> pool;
        for (size_t t = 1; t < slices; ++t)
            pool.emplace_back([&, t] { body(n * t / slices, n * (t + 1) / slices); });
        body(0, n / slices);
        for (auto& th : pool)
            th.join();
    };
    std::atomic<bool> bad { false };
    parallelFor(3 * n_triangles, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i)
            if (indices[i] >= n_vertices)
                bad = true;
    });
    if (bad)
        return PGPU_ERR_INVALID;
    std::vector<Item> items(n_triangles);
    parallelFor(n_triangles, [&](size_t b, size_t e) {
        for (size_t p = b; p < e; ++p)
            items[p] = triangleItem(positions, indices, uint32_t(p), uint32_t(p));
    });
    // The cut is a depth-first recursion whose sub-ranges are independent: a range is split by one thread, serially and in
    // place, so the outcome does not depend on the schedule; the groups are numbered afterwards in depth-first order (=
    // ascending position in the item array). Large right halves go to a shared stack for the other threads.
    typedef std::pair<size_t, size_t> Range;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Range> shared { { 0, n_triangles } };
    std::vector<Range> leaves;
    size_t busy = 0;
    const size_t grain = std::max(max_prims, n_triangles / (8 * size_t(nThreads)));
    // PANDORA_CUT_OBJECTIVE=points: a side of a split weighs the batching points it will need at least, ceil(triangles / max_prims),
    // instead of its triangles (=mixed[f]: that count + f x triangles / max_prims). Fewer, fuller, longer points: on the benchmark
    // city 208 instead of 256, 10 % fewer points entered per ray of the bench's diffuse batch and 7 % more per ray of a frame seen
    // from the street, the same summed surface either way -- which is why surface x triangles stays the default.
    const char* objective = std::getenv("PANDORA_CUT_OBJECTIVE");
    const bool byPoints = objective && (std::strcmp(objective, "points") == 0 || std::strncmp(objective, "mixed", 5) == 0);
    const size_t quantum = byPoints ? max_prims : 0;
    const float mix = byPoints && objective[0] == 'm' ? (objective[5] ? float(std::atof(objective + 5)) : 1.0f) : 0.0f;
    auto worker = [&] {
        std::vector<Range> local, mine;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return !shared.empty() || busy == 0; });
            if (shared.empty())
                return;
            local.assign(1, shared.back());
            shared.pop_back();
            ++busy;
            lk.unlock();
            while (!local.empty()) {
                const auto [b, e] = local.back();
                local.pop_back();
                if (e - b <= max_prims) {
                    if (e > b)
                        mine.push_back({ b, e });
                    continue;
                }
                const size_t m = splitPart(items, b, e, false, quantum, mix);
                if (e - m > grain) {
                    {
                        std::lock_guard<std::mutex> g(mu);
                        shared.push_back({ m, e });
                    }
                    cv.notify_one();
                } else {
                    local.push_back({ m, e });
                }
                local.push_back({ b, m });
            }
            lk.lock();
            leaves.insert(leaves.end(), mine.begin(), mine.end());
            mine.clear();
            if (--busy == 0 && shared.empty())
                cv.notify_all();
        }
    };
    {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < (n_triangles > 4 * max_prims ? nThreads : 1u); ++t)
            pool.emplace_back(worker);
        worker();
        for (auto& th : pool)
            th.join();
    }
    std::sort(leaves.begin(), leaves.end());
    parallelFor(leaves.size(), [&](size_t lb, size_t le) {
        for (size_t g = lb; g < le; ++g)
            for (size_t i = leaves[g].first; i < leaves[g].second; ++i)
                group_of_prim[items[i].id] = uint32_t(g);
    });
    *n_groups = uint32_t(leaves.size());
    return PGPU_OK;
}

// The scene cut of the reference at the level it makes it: over scene OBJECTS (shapes and instanced sub-scenes), not over
// triangles. detail::createSubScenes (batching.cpp:38-265) builds an Embree BVH2 over the objects' bounds and cuts it where
// a subtree holds <= primitivesPerBatchingPoint primitives (:161-224); the same rule on a binned-SAH BVH2 here. An object
// heavier than the limit ends up alone (the reference splits such shapes beforehand, splitLargeSceneObjects,
// batching.cpp:358-422 -- pgpu_partition_prims over the shape's triangles does that here).
int pgpu_partition_objects(const float* bounds6, const uint64_t* weights, size_t n_objects, uint64_t max_weight, uint32_t* group_of_object,
    uint32_t* n_groups)
{
    if (!bounds6 || !group_of_object || !n_groups || max_weight == 0)
        return PGPU_ERR_INVALID;
    std::vector<Item> items(n_objects);
    for (size_t j = 0; j < n_objects; ++j) {
        Item& it = items[j];
        for (int a = 0; a < 3; ++a) {
            it.box.lo[a] = bounds6[6 * j + size_t(a)];
            it.box.hi[a] = bounds6[6 * j + 3 + size_t(a)];
            if (!(it.box.lo[a] <= it.box.hi[a]))
                return PGPU_ERR_INVALID;
            it.c[a] = 0.5f * (it.box.lo[a] + it.box.hi[a]);
        }
        it.id = uint32_t(j);
    }
    auto weightOf = [&](size_t b, size_t e) {
        uint64_t w = 0;
        for (size_t i = b; i < e; ++i)
            w += weights ? weights[items[i].id] : 1u;
        return w;
    };
    std::vector<std::pair<size_t, size_t>> todo { { 0, n_objects } };
    uint32_t g = 0;
    while (!todo.empty()) {
        auto [b, e] = todo.back();
        todo.pop_back();
        if (e - b <= 1 || weightOf(b, e) <= max_weight) {
            for (size_t i = b; i < e; ++i)
                group_of_object[items[i].id] = g;
            if (e > b)
                ++g;
            continue;
        }
        const size_t m = splitPart(items, b, e, false);
        todo.push_back({ m, e });
        todo.push_back({ b, m });
    }
    *n_groups = g;
    return PGPU_OK;
}
}

namespace pandora_hip {
// wraps finished arrays (the device builder's, pandora_amd/csrc/bvh_build.hip) as the object pgpu_build_bvh8 returns
pgpu_host_bvh8* makeHostBvh8(const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices, uint32_t root, int depth)
{
    auto* r = new (std::nothrow) pgpu_host_bvh8();
    if (!r)
        return nullptr;
    r->v.nodes.resize(n_nodes);
    if (n_nodes)
        std::memcpy(r->v.nodes.data(), nodes256, n_nodes * sizeof(Node8));
    r->v.leafIdx.assign(leaf_indices, leaf_indices + n_indices);
    r->v.root = root;
    r->v.depth = depth;
    return r;
}
}
