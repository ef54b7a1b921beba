// Bottom-level BVH8 construction on the GPU ("GPU LBVH build"): pgpu_build_bvh8_device.
//
// The reference builds its 8-wide trees with Embree on the CPU (wive_bvh8_build8_impl.h:87-104, one rtcBuildBVH per batching
// point); the host builder of this repository (pandora_amd/host/bvh_builder.cpp) replaces that with a binary SAH tree collapsed
// by dynamic programming -- the best trees, 2 us per triangle and thread. This file builds the trees of ALL batching points of a
// scene in one pass over the GPU, for scenes whose construction time matters more than the last 15 % of traversal speed:
//   1. one thread per triangle: bounds, centroid; per-group centroid bounds (ordered-integer atomics, one per wave when a wave
//      lies in one group);
//   2. 64-bit keys  group | Morton code of the centroid inside the group's bounds (as many bits per axis as the group id leaves,
//      at most 21), one radix sort over all groups (hipCUB): the groups' triangles stay contiguous, in group order;
//   3. one binary radix tree over the whole sorted array (Karras 2012: every internal node finds its range and split with two
//      binary searches over common prefix lengths; equal keys fall back to the index) -- the group id is the top of the key, so
//      every group's triangles are exactly one subtree; its root is the node whose range is the group's;
//   4. boxes bottom-up (the second thread to arrive at a node computes it), and with them the cost tables of the collapse: per node
//      the cheapest forest of at most 1..8 wide-tree roots below it (the host builder's dynamic programme, run on the radix tree);
//   3b. by default, instead of 3 and 4: the binary tree by locally-ordered clustering over the same sorted order (clusters merge with
//      the neighbour, a few places along the Morton curve, with which they make the smallest box; boxes and tables are made with
//      the nodes), then the triangles reordered so that every subtree is a range again -- the better tree: a diffuse ray of the
//      benchmark pays 12.7 steps in it instead of 14.2, at the same build time (PANDORA_BVH8_DEVICE_TREE=lbvh keeps 3 and 4);
//   5. collapse into 8-wide nodes, level by level from all group roots at once: a node takes the roots its tables name. A group
//      whose tree comes out deeper than the traversal stack allows goes through a second pass in which a node adopts the children
//      of the child with the largest surface until it has eight (subtrees of at most max_leaf triangles are leaves on sight, as in
//      the host's greedy construction) and children too large for what is left of the depth budget are opened first, largest
//      first. Node indices are handed out per group, so every handle is local to its group's node array from the start;
//   6. emit: the reference's 256-byte nodes (wive_bvh8.h:76-86) with the octant permutation words of
//      wive_bvh8_build8_impl.h:172-203, group after group in one array, and the leaf-index arrays (= the sorted order).
// A group whose tree would nest deeper than the traversal stack allows (12 inner nodes, wive_bvh8_impl.h:74-75) is rebuilt by the
// host builder, which balances against that budget; the caller learns how many were. Topology is this builder's own (as the host
// builder's is: Embree is not here); the output is in the reference's layout and goes through the same upload path.
#include "context.h"
#include "trace_ranges.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace pandora_hip;

namespace {

constexpr uint32_t kLeafRef = 0x80000000u; // a binary-tree reference to a single sorted triangle
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kMaxLevels = 12; // nested inner nodes: 7 * 12 + 8 <= 96 stack entries
constexpr int kPlocMaxRounds = 1024; // rounds of the clustering (and with them the depth of its binary tree)

struct Box6 {
    float lo[3], hi[3];
};
struct WorkItem {
    uint32_t node, local, group;
};
struct NodeRec { // one wide node before it is written: its children in the binary tree and their handles
    uint32_t ref[8];
    uint32_t handle[8];
};

__device__ __forceinline__ uint32_t orderedBits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fromOrderedBits(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// ---- 1. triangle bounds, group of every position, centroid bounds per group
__global__ void __launch_bounds__(256) primBoundsKernel(const float* __restrict__ pos, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ prims,
    const unsigned long long* __restrict__ offsets, uint32_t nGroups, uint32_t n, uint32_t* __restrict__ groupOf, Box6* __restrict__ primBox,
    uint32_t* __restrict__ cbounds)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    const bool live = p < n;
    uint32_t g = 0;
    uint32_t kmin[3] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu }, kmax[3] = { 0u, 0u, 0u };
    if (live) {
        uint32_t lo = 0, hi = nGroups; // offsets[lo] <= p < offsets[hi]
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (offsets[mid] <= p)
                lo = mid;
            else
                hi = mid;
        }
        g = lo;
        const uint32_t t = prims ? prims[p] : p; // (no list: the mesh's triangles in their order)
        Box6 b;
        for (int a = 0; a < 3; ++a) {
            const float v0 = pos[3u * size_t(idx[3u * size_t(t)]) + a], v1 = pos[3u * size_t(idx[3u * size_t(t) + 1u]) + a],
                        v2 = pos[3u * size_t(idx[3u * size_t(t) + 2u]) + a];
            b.lo[a] = fminf(v0, fminf(v1, v2));
            b.hi[a] = fmaxf(v0, fmaxf(v1, v2));
            const float c = 0.5f * (b.lo[a] + b.hi[a]);
            kmin[a] = kmax[a] = orderedBits(c);
        }
        primBox[p] = b;
        groupOf[p] = g;
    }
    // a wave that lies in one group (nearly all of them) reduces first and sends six atomics instead of 384
    const uint32_t g0 = uint32_t(__builtin_amdgcn_readfirstlane(int(g)));
    const bool uniform = __all(!live || g == g0);
    if (uniform) {
        for (int s = 32; s >= 1; s >>= 1)
            for (int a = 0; a < 3; ++a) {
                kmin[a] = min(kmin[a], uint32_t(__shfl_xor(int(kmin[a]), s)));
                kmax[a] = max(kmax[a], uint32_t(__shfl_xor(int(kmax[a]), s)));
            }
        if ((threadIdx.x & 63u) == 0u && live)
            for (int a = 0; a < 3; ++a) {
                atomicMin(&cbounds[6u * g0 + a], kmin[a]);
                atomicMax(&cbounds[6u * g0 + 3u + a], kmax[a]);
            }
    } else if (live) {
        for (int a = 0; a < 3; ++a) {
            atomicMin(&cbounds[6u * g + a], kmin[a]);
            atomicMax(&cbounds[6u * g + 3u + a], kmax[a]);
        }
    }
}

// ---- 2. keys: group | Morton code of the centroid in the group's centroid bounds
__device__ __forceinline__ unsigned long long spread3(uint32_t v) // bit i of v (i < 21) to bit 3 i
{
    unsigned long long x = v & 0x1FFFFFu;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__global__ void __launch_bounds__(256) mortonKernel(const Box6* __restrict__ primBox, const uint32_t* __restrict__ groupOf, const uint32_t* __restrict__ cbounds,
    uint32_t n, int bitsPerAxis, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n)
        return;
    const uint32_t g = groupOf[p];
    const Box6 b = primBox[p];
    const float cells = float(1u << bitsPerAxis);
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = fromOrderedBits(cbounds[6u * g + a]), hi = fromOrderedBits(cbounds[6u * g + 3u + a]);
        const float c = 0.5f * (b.lo[a] + b.hi[a]);
        const float ext = hi - lo;
        float x = ext > 0.0f ? (c - lo) / ext * cells : 0.0f;
        x = x == x ? x : 0.0f; // (a NaN or infinite vertex: any cell will do, the tree stays a tree)
        x = fminf(fmaxf(x, 0.0f), cells - 1.0f);
        q[a] = uint32_t(x);
    }
    const unsigned long long morton = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
    keys[p] = (static_cast<unsigned long long>(g) << (3 * bitsPerAxis)) | morton;
    vals[p] = p;
}

// ---- 3. binary radix tree (Karras 2012, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", section 4)
__device__ __forceinline__ int prefixLen(const unsigned long long* __restrict__ keys, uint32_t n, uint32_t i, long long j)
{
    if (j < 0 || j >= static_cast<long long>(n))
        return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b)
        return 64 + __clz(int(i ^ uint32_t(j))); // equal keys: the positions themselves are the tie-break
    return __clzll(static_cast<long long>(a ^ b));
}

__global__ void __launch_bounds__(256) hierarchyKernel(const unsigned long long* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ groupOf,
    const unsigned long long* __restrict__ offsets, uint32_t* __restrict__ left, uint32_t* __restrict__ right, uint32_t* __restrict__ rangeLo,
    uint32_t* __restrict__ rangeHi, uint32_t* __restrict__ parentInt, uint32_t* __restrict__ parentLeaf, uint32_t* __restrict__ groupRoot)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i + 1u >= n)
        return;
    const long long ii = i;
    const int d = prefixLen(keys, n, i, ii + 1) - prefixLen(keys, n, i, ii - 1) >= 0 ? 1 : -1;
    const int dMin = prefixLen(keys, n, i, ii - d);
    long long lMax = 2;
    while (prefixLen(keys, n, i, ii + lMax * d) > dMin)
        lMax <<= 1;
    long long l = 0;
    for (long long t = lMax >> 1; t >= 1; t >>= 1)
        if (prefixLen(keys, n, i, ii + (l + t) * d) > dMin)
            l += t;
    const long long j = ii + l * d;
    const int dNode = prefixLen(keys, n, i, j);
    long long s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (prefixLen(keys, n, i, ii + (s + t) * d) > dNode)
            s += t;
    } while (t > 1);
    const long long gamma = ii + s * d + (d < 0 ? -1 : 0);
    const uint32_t lo = uint32_t(ii < j ? ii : j), hi = uint32_t(ii < j ? j : ii);
    const uint32_t gl = uint32_t(gamma), gr = uint32_t(gamma) + 1u;
    if (gl < lo || gr > hi)
        return; // (cannot happen on sorted keys; a node left unlinked fails the build instead of writing out of range)
    const uint32_t leftRef = lo == gl ? (gl | kLeafRef) : gl, rightRef = hi == gr ? (gr | kLeafRef) : gr;
    left[i] = leftRef;
    right[i] = rightRef;
    rangeLo[i] = lo;
    rangeHi[i] = hi;
    if (leftRef & kLeafRef)
        parentLeaf[gl] = i;
    else
        parentInt[gl] = i;
    if (rightRef & kLeafRef)
        parentLeaf[gr] = i;
    else
        parentInt[gr] = i;
    const uint32_t g = groupOf[lo];
    if (offsets[g] == lo && offsets[g + 1u] == static_cast<unsigned long long>(hi) + 1ull)
        groupRoot[g] = i;
}

// ---- 4. boxes, bottom-up: the second thread to reach a node computes it and goes on
// The two threads that meet at a node run on different CUs, in general on different XCDs, each with an L2 of its own. A box is
// therefore written and read with agent-scope atomic accesses (stores that write through, loads that do not stop at a stale line),
// and the writer waits for its stores to be acknowledged before it announces itself at the parent: no fences, which on this
// part mean writing back and invalidating an XCD's whole L2 per call (the first version, with __threadfence(), spent 68 of the
// build's 75 ms here on 11.4 M triangles).
__device__ __forceinline__ Box6 loadNodeBoxCoherent(const Box6* p)
{
    Box6 b;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = __uint_as_float(__hip_atomic_load(w + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        b.hi[a] = __uint_as_float(__hip_atomic_load(w + 3 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    return b;
}
__device__ __forceinline__ void storeNodeBoxCoherent(Box6* p, const Box6& b)
{
    uint32_t* w = reinterpret_cast<uint32_t*>(p);
    for (int a = 0; a < 3; ++a) {
        __hip_atomic_store(w + a, __float_as_uint(b.lo[a]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(w + 3 + a, __float_as_uint(b.hi[a]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The same pass carries the tables of the collapse of step 5 (pandora_amd/host/bvh_builder.cpp `solve`, after Ylitie, Karras and Laine
// 2017): per node cost[j], j = 1..8, the cheapest forest of at most j wide-tree roots that covers the subtree -- one root is a leaf
// (at most maxLeaf triangles) or a wide node, each at its surface; more roots are the best split of j between the two children --
// and, packed in one word, where the j roots are cut (3 bits per j = 2..8), how a wide node here divides its eight slots
// (bits 21..23) and whether the one-root form is a node (bit 24).
__device__ __forceinline__ float halfArea(const Box6& b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// the tables of a node from its children's (cl, cr: [1..8]); returns the packed word
__device__ __forceinline__ uint32_t combineTables(const float* cl, const float* cr, float area, uint32_t cnt, uint32_t maxLeaf, float* cj)
{
    const float asLeaf = cnt <= maxLeaf ? area : __builtin_inff();
    float bestKids = cl[1] + cr[7];
    uint32_t kLeft = 1;
    for (uint32_t k = 2; k <= 7u; ++k) {
        const float v = cl[k] + cr[8u - k];
        if (v < bestKids) {
            bestKids = v;
            kLeft = k;
        }
    }
    const float asInner = area + bestKids;
    const bool inner1 = cnt > maxLeaf || asInner < asLeaf; // (more than a leaf holds: a node whatever the numbers say)
    cj[1] = inner1 ? asInner : asLeaf;
    uint32_t word = (kLeft << 21) | (inner1 ? 1u << 24 : 0u);
    for (uint32_t j = 2; j <= 8u; ++j) {
        float best = cj[1];
        uint32_t cut = 0;
        for (uint32_t k = 1; k < j; ++k) {
            const float v = cl[k] + cr[j - k];
            if (v < best) {
                best = v;
                cut = k;
            }
        }
        cj[j] = best;
        word |= cut << ((j - 2u) * 3u);
    }
    return word;
}

__global__ void __launch_bounds__(256) fitKernel(uint32_t n, const uint32_t* __restrict__ order, const Box6* __restrict__ primBox, const uint32_t* __restrict__ left,
    const uint32_t* __restrict__ right, const uint32_t* __restrict__ parentInt, const uint32_t* __restrict__ parentLeaf, uint32_t* __restrict__ visits,
    Box6* nodeBox, const uint32_t* __restrict__ rangeLo, const uint32_t* __restrict__ rangeHi, uint32_t maxLeaf, float* cost, uint32_t* cutWord)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n)
        return;
    uint32_t cur = parentLeaf[p];
    for (uint32_t guard = 0; cur != kNone && cur + 1u < n && guard < 192u; ++guard) { // (a path is at most 64 + 32 nodes long)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the box written in the previous trip has arrived where the other thread will read it
        if (__hip_atomic_fetch_add(&visits[cur], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
            return;
        asm volatile("" ::: "memory");
        const uint32_t l = left[cur], r = right[cur];
        const Box6 a = (l & kLeafRef) ? primBox[order[l & ~kLeafRef]] : loadNodeBoxCoherent(nodeBox + l);
        const Box6 b = (r & kLeafRef) ? primBox[order[r & ~kLeafRef]] : loadNodeBoxCoherent(nodeBox + r);
        Box6 u;
        for (int k = 0; k < 3; ++k) {
            u.lo[k] = fminf(a.lo[k], b.lo[k]);
            u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
        }
        storeNodeBoxCoherent(nodeBox + cur, u);
        if (cost) {
            float cl[9], cr[9]; // [1..8]
            for (int j = 1; j <= 8; ++j) {
                cl[j] = (l & kLeafRef) ? halfArea(a) : __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(cost) + size_t(l) * 8u + (j - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                cr[j] = (r & kLeafRef) ? halfArea(b) : __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(cost) + size_t(r) * 8u + (j - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            float cj[9];
            const uint32_t word = combineTables(cl, cr, halfArea(u), rangeHi[cur] - rangeLo[cur] + 1u, maxLeaf, cj);
            for (int j = 1; j <= 8; ++j)
                __hip_atomic_store(reinterpret_cast<uint32_t*>(cost) + size_t(cur) * 8u + (j - 1), __float_as_uint(cj[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cutWord[cur] = word; // (read by later kernels only)
        }
        cur = parentInt[cur];
    }
}

// ---- 3b + 4b. the binary tree by locally-ordered clustering instead of steps 3 and 4 (Meister and Bittner 2018, "Parallel
// locally-ordered clustering for bounding volume hierarchy construction"): every sorted triangle starts as a cluster; per round
// each cluster looks `radius` places to either side along the Morton order (inside its group) for the neighbour with which it
// makes the smallest box, pairs that chose each other merge into a node that takes the lower one's place, and the array is
// compacted (one scan per round hands out the places and the node numbers, so the tree is the same on every run). Ties go to the
// pair whose first triangles' positions in the group differ in the lowest bits (symmetric, and a group's tree does not depend on
// what is built next to it; equal boxes, e.g. coincident triangles, pair up (0,1), (2,3), ... instead of forming a chain). Boxes,
// triangle counts and the tables of the collapse are known the moment a node is made. The leaves of a subtree are no longer a
// range of the sorted order, which the collapse and the emission rely on: afterwards every triangle finds its place in the tree's
// own left-to-right order by walking up (plocPositionKernel), and the order, the ranges and the leaf references are rewritten.
struct PlocCluster {
    uint32_t ref;   // a sorted triangle (| kLeafRef) or a node
    uint32_t group;
    uint32_t first; // position of its first triangle within the group
    uint32_t count; // triangles below
};

__global__ void __launch_bounds__(256) plocInitKernel(uint32_t n, const uint32_t* __restrict__ order, const Box6* __restrict__ primBox, const uint32_t* __restrict__ groupOf,
    const unsigned long long* __restrict__ offsets, PlocCluster* __restrict__ cl, Box6* __restrict__ cBox)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n)
        return;
    const uint32_t g = groupOf[p];
    cl[p] = PlocCluster { p | kLeafRef, g, p - uint32_t(offsets[g]), 1u };
    cBox[p] = primBox[order[p]];
}

__device__ __forceinline__ float unionArea(const Box6& a, const Box6& b)
{
    Box6 u;
    for (int k = 0; k < 3; ++k) {
        u.lo[k] = fminf(a.lo[k], b.lo[k]);
        u.hi[k] = fmaxf(a.hi[k], b.hi[k]);
    }
    const float area = halfArea(u);
    return area == area ? area : FLT_MAX; // (boxes that are not numbers: any partner will do, the same from both sides)
}

__global__ void __launch_bounds__(256) plocNearestKernel(uint32_t nActive, int radius, const PlocCluster* __restrict__ cl, const Box6* __restrict__ cBox, uint32_t* __restrict__ nn)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nActive)
        return;
    const PlocCluster me = cl[i];
    const Box6 box = cBox[i];
    float best = 0.0f;
    uint32_t bestJ = kNone, bestTie = 0u;
    for (int d = 1; d <= radius; ++d)
        for (int s = 0; s < 2; ++s) {
            const long long j = s ? static_cast<long long>(i) + d : static_cast<long long>(i) - d;
            if (j < 0 || j >= static_cast<long long>(nActive))
                continue;
            const PlocCluster other = cl[j];
            if (other.group != me.group)
                continue;
            const float a = unionArea(box, cBox[j]);
            const uint32_t tie = me.first ^ other.first;
            if (bestJ == kNone || a < best || (a == best && tie < bestTie)) {
                best = a;
                bestJ = uint32_t(j);
                bestTie = tie;
            }
        }
    nn[i] = bestJ;
}

// what becomes of every cluster: bit 0 = it keeps a place in the next round's array, bit 32 = it is the lower one of a pair that
// merges (one element more than there are clusters, so that the exclusive scan also yields the totals)
__global__ void __launch_bounds__(256) plocFlagKernel(uint32_t nActive, const uint32_t* __restrict__ nn, unsigned long long* __restrict__ flags)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i > nActive)
        return;
    if (i == nActive) {
        flags[i] = 0ull;
        return;
    }
    const uint32_t j = nn[i];
    const bool merged = j < nActive && nn[j] == i;
    flags[i] = (merged && j < i) ? 0ull : (1ull | (merged ? 1ull << 32 : 0ull));
}

__global__ void __launch_bounds__(256) plocCommitKernel(uint32_t nActive, uint32_t n, uint32_t nodesMade, const uint32_t* __restrict__ nn, const unsigned long long* __restrict__ flags,
    const unsigned long long* __restrict__ places, const PlocCluster* __restrict__ clIn, const Box6* __restrict__ boxIn, PlocCluster* __restrict__ clOut, Box6* __restrict__ boxOut,
    uint32_t* __restrict__ left, uint32_t* __restrict__ right, Box6* __restrict__ nodeBox, uint32_t* __restrict__ count, uint32_t* __restrict__ parentInt,
    uint32_t* __restrict__ parentLeaf, uint32_t maxLeaf, float* __restrict__ cost, uint32_t* __restrict__ cutWord)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nActive)
        return;
    const unsigned long long f = flags[i];
    if ((f & 1ull) == 0ull)
        return;
    const uint32_t place = uint32_t(places[i] & 0xFFFFFFFFull);
    if (place >= nActive)
        return; // (cannot happen: places count the clusters that stay)
    PlocCluster me = clIn[i];
    Box6 box = boxIn[i];
    if (f >> 32) {
        const uint32_t j = nn[i];
        const uint32_t node = nodesMade + uint32_t(places[i] >> 32);
        if (j >= nActive || node + 1u >= n)
            return; // (cannot happen: a tree over n triangles in G groups has n - G nodes)
        const uint32_t l = me.ref, r = clIn[j].ref;
        const Box6 a = box, b = boxIn[j];
        for (int k = 0; k < 3; ++k) {
            box.lo[k] = fminf(a.lo[k], b.lo[k]);
            box.hi[k] = fmaxf(a.hi[k], b.hi[k]);
        }
        left[node] = l;
        right[node] = r;
        nodeBox[node] = box;
        const uint32_t cnt = me.count + clIn[j].count;
        count[node] = cnt;
        if (l & kLeafRef)
            parentLeaf[l & ~kLeafRef] = node;
        else
            parentInt[l] = node;
        if (r & kLeafRef)
            parentLeaf[r & ~kLeafRef] = node;
        else
            parentInt[r] = node;
        if (cost) {
            float cl[9], cr[9], cj[9];
            for (int q = 1; q <= 8; ++q) {
                cl[q] = (l & kLeafRef) ? halfArea(a) : cost[size_t(l) * 8u + (q - 1)];
                cr[q] = (r & kLeafRef) ? halfArea(b) : cost[size_t(r) * 8u + (q - 1)];
            }
            cutWord[node] = combineTables(cl, cr, halfArea(box), cnt, maxLeaf, cj);
            for (int q = 1; q <= 8; ++q)
                cost[size_t(node) * 8u + (q - 1)] = cj[q];
        }
        me.ref = node;
        me.count = cnt;
    }
    clOut[place] = me;
    boxOut[place] = box;
}

__global__ void __launch_bounds__(256) plocRootsKernel(uint32_t nActive, uint32_t nGroups, const PlocCluster* __restrict__ cl, uint32_t* __restrict__ groupRoot)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= nActive)
        return;
    const PlocCluster c = cl[i];
    if (c.group < nGroups && (c.ref & kLeafRef) == 0u)
        groupRoot[c.group] = c.ref;
}

// a triangle's place in the left-to-right order of its group's tree: the triangles of every left sibling on the way up come first
__global__ void __launch_bounds__(256) plocPositionKernel(uint32_t n, uint32_t maxDepth, const uint32_t* __restrict__ groupOf, const unsigned long long* __restrict__ offsets,
    const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, const uint32_t* __restrict__ parentInt, const uint32_t* __restrict__ parentLeaf,
    const uint32_t* __restrict__ count, uint32_t* __restrict__ newPos, uint32_t* __restrict__ error)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n)
        return;
    uint32_t ref = p | kLeafRef, cur = parentLeaf[p], before = 0u, steps = 0u;
    while (cur != kNone) {
        if (cur + 1u >= n || ++steps > maxDepth) {
            atomicOr(error, 1u);
            break;
        }
        if (right[cur] == ref) {
            const uint32_t l = left[cur];
            before += (l & kLeafRef) ? 1u : count[l];
        }
        ref = cur;
        cur = parentInt[cur];
    }
    const uint32_t g = groupOf[p];
    const unsigned long long at = offsets[g] + before;
    if (at >= offsets[g + 1u]) {
        atomicOr(error, 2u);
        newPos[p] = p;
        return;
    }
    newPos[p] = uint32_t(at);
}

// the ranges of the nodes in the new order and the leaf references renamed (into arrays of their own: the walk below reads the old ones)
__global__ void __launch_bounds__(256) plocRangesKernel(uint32_t nNodes, uint32_t n, uint32_t maxDepth, const uint32_t* __restrict__ left, const uint32_t* __restrict__ right,
    const uint32_t* __restrict__ count, const uint32_t* __restrict__ newPos, uint32_t* __restrict__ rangeLo, uint32_t* __restrict__ rangeHi, uint32_t* __restrict__ leftOut,
    uint32_t* __restrict__ rightOut, uint32_t* __restrict__ error)
{
    const uint32_t node = blockIdx.x * 256u + threadIdx.x;
    if (node >= nNodes)
        return;
    const uint32_t l = left[node], r = right[node];
    uint32_t ref = l;
    for (uint32_t steps = 0; (ref & kLeafRef) == 0u; ++steps) {
        if (ref >= nNodes || steps > maxDepth) {
            atomicOr(error, 4u);
            ref = kLeafRef;
            break;
        }
        ref = left[ref];
    }
    const uint32_t first = (ref & ~kLeafRef) < n ? newPos[ref & ~kLeafRef] : 0u;
    rangeLo[node] = first;
    rangeHi[node] = first + count[node] - 1u;
    leftOut[node] = (l & kLeafRef) ? (((l & ~kLeafRef) < n ? newPos[l & ~kLeafRef] : 0u) | kLeafRef) : l;
    rightOut[node] = (r & kLeafRef) ? (((r & ~kLeafRef) < n ? newPos[r & ~kLeafRef] : 0u) | kLeafRef) : r;
}

__global__ void __launch_bounds__(256) plocReorderKernel(uint32_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ newPos, uint32_t* __restrict__ orderOut)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p < n && newPos[p] < n)
        orderOut[newPos[p]] = order[p];
}

// ---- 5. collapse into 8-wide nodes
__device__ __forceinline__ uint32_t handleInner(uint32_t i) { return (i & 0x1FFFFFFFu) | (2u << 29); }                    // wive_bvh8_impl.h:333-338
__device__ __forceinline__ uint32_t handleLeaf(uint32_t i, uint32_t n) { return (i & 0x1FFFFFFFu) | ((4u | (n - 1u)) << 29); } // :340-346

// A level of the collapse in two kernels with a scan between them, so that the numbering of the nodes does not depend on the order
// in which threads happen to run: the level's queue holds its nodes group after group (the seeds are in group order, and the
// children of a queue in order are a queue in order), expandKernel opens each node's children and counts those that become nodes,
// an exclusive scan of the counts gives every node the positions of its children in the next queue, and assignKernel hands out the
// handles: a group's nodes are numbered in breadth-first order -- child c of the queue's k-th node gets
// (number of the group's first node of this level) + (the group's nodes of this level) + (its position among the group's children).
__global__ void __launch_bounds__(256) expandKernel(const WorkItem* __restrict__ in, uint32_t nIn, uint32_t level, uint32_t maxLevels, uint32_t maxLeaf, unsigned long long capacity,
    const uint32_t* __restrict__ left, const uint32_t* __restrict__ right, const uint32_t* __restrict__ rangeLo, const uint32_t* __restrict__ rangeHi,
    const Box6* __restrict__ nodeBox, const unsigned long long* __restrict__ offsets, NodeRec* __restrict__ recs, uint32_t* __restrict__ innerCount,
    uint32_t* __restrict__ groupDepth, uint32_t* __restrict__ groupFailed, const uint32_t* __restrict__ cutWord)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k > nIn)
        return;
    if (k == nIn) { // (one element more, so that the scan also yields the total)
        innerCount[k] = 0u;
        return;
    }
    const WorkItem w = in[k];
    uint32_t kids[8];
    int n = 2;
    kids[0] = left[w.node];
    kids[1] = right[w.node];
    for (int j = 2; j < 8; ++j)
        kids[j] = kNone;
    if (cutWord) {
        // the roots the tables of fitKernel name: the node's eight slots divided between its two subtrees, each share cut further
        // wherever more roots are cheaper than one
        uint32_t todoNode[8], todoRoots[8];
        int top = 0;
        n = 0;
        const uint32_t kLeft = (cutWord[w.node] >> 21) & 7u;
        todoNode[top] = kids[1];
        todoRoots[top++] = 8u - kLeft;
        todoNode[top] = kids[0];
        todoRoots[top++] = kLeft;
        while (top > 0 && n < 8) {
            const uint32_t i = todoNode[--top], j = todoRoots[top];
            const uint32_t cut = ((i & kLeafRef) || j < 2u) ? 0u : (cutWord[i] >> ((j - 2u) * 3u)) & 7u;
            if (cut == 0u || cut >= j || top + 2 > 8) {
                kids[n++] = i;
            } else {
                todoNode[top] = right[i];
                todoRoots[top++] = j - cut;
                todoNode[top] = left[i];
                todoRoots[top++] = cut;
            }
        }
        for (int j = n; j < 8; ++j)
            kids[j] = kNone;
    }
    while (!cutWord && n < 8) {
        // open the child the depth budget cannot hold, else the one with the largest surface; subtrees of <= maxLeaf triangles are leaves
        int pick = -1;
        float best = -1.0f;
        uint32_t most = 0;
        bool over = false;
        for (int j = 0; j < n; ++j) {
            const uint32_t ref = kids[j];
            if (ref & kLeafRef)
                continue;
            const uint32_t cnt = rangeHi[ref] - rangeLo[ref] + 1u;
            if (cnt <= maxLeaf)
                continue;
            if (cnt > capacity) { // (among those: the one with the most triangles, which balances what is left of the budget)
                if (!over || cnt > most) {
                    pick = j;
                    most = cnt;
                    over = true;
                }
                continue;
            }
            if (over)
                continue;
            const Box6 b = nodeBox[ref];
            const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
            const float area = dx * dy + dy * dz + dz * dx;
            if (pick < 0 || area > best || (area == best && cnt > most)) { // (equal surfaces, e.g. coincident triangles: the larger)
                pick = j;
                best = area;
                most = cnt;
            }
        }
        if (pick < 0)
            break;
        const uint32_t ref = kids[pick];
        for (int j = n; j > pick + 1; --j)
            kids[j] = kids[j - 1];
        kids[pick] = left[ref];
        kids[pick + 1] = right[ref];
        ++n;
    }
    const uint32_t base = uint32_t(offsets[w.group]);
    NodeRec r;
    uint32_t nInner = 0;
    for (int j = 0; j < 8; ++j) {
        r.ref[j] = kids[j];
        r.handle[j] = 0u; // (a child that becomes a node: assignKernel fills it in)
        if (j >= n)
            continue;
        const uint32_t ref = kids[j];
        if (ref & kLeafRef) {
            r.handle[j] = handleLeaf((ref & ~kLeafRef) - base, 1u);
        } else {
            const uint32_t cnt = rangeHi[ref] - rangeLo[ref] + 1u;
            const bool asLeaf = cutWord ? ((cutWord[ref] >> 24) & 1u) == 0u && cnt <= maxLeaf : cnt <= maxLeaf;
            if (asLeaf)
                r.handle[j] = handleLeaf(rangeLo[ref] - base, cnt);
            else
                ++nInner;
        }
    }
    if (nInner != 0u && level + 1u >= maxLevels) {
        groupFailed[w.group] = 1u; // deeper than the traversal stack allows: the host builder takes this group
        nInner = 0u;
    }
    innerCount[k] = nInner;
    recs[size_t(base) + w.local] = r;
    atomicMax(&groupDepth[w.group], level + 1u);
}

__global__ void __launch_bounds__(256) assignKernel(const WorkItem* __restrict__ in, uint32_t nIn, const uint32_t* __restrict__ innerCount, const uint32_t* __restrict__ slotStart,
    const unsigned long long* __restrict__ offsets, NodeRec* __restrict__ recs, WorkItem* __restrict__ out, uint32_t* __restrict__ groupNodeCount)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;
    if (k >= nIn || innerCount[k] == 0u)
        return;
    const WorkItem w = in[k];
    // the stretch of the queue that holds this group's nodes of the level
    uint32_t lo = 0, hi = k; // first position whose group is w.group
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (in[mid].group < w.group)
            lo = mid + 1u;
        else
            hi = mid;
    }
    const uint32_t first = lo;
    lo = k + 1u;
    hi = nIn; // first position behind the group
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (in[mid].group <= w.group)
            lo = mid + 1u;
        else
            hi = mid;
    }
    const uint32_t end = lo;
    uint32_t slot = slotStart[k];
    uint32_t local = in[first].local + (end - first) + (slot - slotStart[first]);
    NodeRec* const r = recs + size_t(offsets[w.group]) + w.local;
    for (int j = 0; j < 8; ++j)
        if (r->ref[j] != kNone && r->handle[j] == 0u) {
            r->handle[j] = handleInner(local);
            out[slot++] = WorkItem { r->ref[j], local++, w.group };
        }
    atomicMax(&groupNodeCount[w.group], local);
}

// ---- 6. emit the reference's nodes and the leaf-index arrays
__global__ void __launch_bounds__(256) emitKernel(uint32_t n, const uint32_t* __restrict__ groupOf, const unsigned long long* __restrict__ offsets,
    const uint32_t* __restrict__ groupNodeCount, const uint32_t* __restrict__ groupFailed, const uint32_t* __restrict__ nodeBase, const NodeRec* __restrict__ recs,
    const uint32_t* __restrict__ order, const Box6* __restrict__ primBox, const Box6* __restrict__ nodeBox, unsigned char* __restrict__ nodesOut,
    uint32_t* __restrict__ leafIdx)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n)
        return;
    const uint32_t g = groupOf[p];
    const uint32_t local = p - uint32_t(offsets[g]);
    leafIdx[p] = order[p] - uint32_t(offsets[g]); // leaf object j of a group = its j-th triangle (offline_bvh_cache.cpp:303-304)
    if (groupFailed[g] != 0u || local >= groupNodeCount[g])
        return;
    const NodeRec r = recs[p];
    float lo[8][3], hi[8][3];
    uint32_t nKids = 0;
    for (int j = 0; j < 8; ++j) {
        Box6 b {};
        if (r.ref[j] != kNone && r.handle[j] != 0u) {
            b = (r.ref[j] & kLeafRef) ? primBox[order[r.ref[j] & ~kLeafRef]] : nodeBox[r.ref[j]];
            nKids = uint32_t(j) + 1u;
        }
        for (int a = 0; a < 3; ++a) {
            lo[j][a] = b.lo[a];
            hi[j][a] = b.hi[a];
        }
    }
    // wive_bvh8_build8_impl.h:172-203: per ray-direction octant the children back to front (stable, absent children first)
    uint32_t perm[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    for (int oct = 0; oct < 8; ++oct) {
        const float sx = (oct & 1) ? 1.0f : -1.0f, sy = (oct & 2) ? 1.0f : -1.0f, sz = (oct & 4) ? 1.0f : -1.0f;
        float dist[8];
        uint32_t ord[8];
        for (uint32_t i = 0; i < 8u; ++i) {
            const float ex = sx < 0.0f ? lo[i][0] : hi[i][0], ey = sy < 0.0f ? lo[i][1] : hi[i][1], ez = sz < 0.0f ? lo[i][2] : hi[i][2];
            dist[i] = i < nKids ? (ex * sx + ey * sy) + ez * sz : FLT_MAX;
            ord[i] = i;
        }
        for (int i = 1; i < 8; ++i) { // stable insertion sort, descending
            const uint32_t key = ord[i];
            const float dk = dist[key];
            int j = i - 1;
            while (j >= 0 && dk > dist[ord[j]]) {
                ord[j + 1] = ord[j];
                --j;
            }
            ord[j + 1] = key;
        }
        for (int i = 0; i < 8; ++i)
            perm[i] |= ord[i] << (3 * oct);
    }
    float* nd = reinterpret_cast<float*>(nodesOut + (size_t(nodeBase[g]) + local) * 256u);
    for (int j = 0; j < 8; ++j) {
        nd[j] = lo[j][0];
        nd[8 + j] = hi[j][0];
        nd[16 + j] = lo[j][1];
        nd[24 + j] = hi[j][1];
        nd[32 + j] = lo[j][2];
        nd[40 + j] = hi[j][2];
        nd[48 + j] = __uint_as_float(r.handle[j]);
        nd[56 + j] = __uint_as_float(perm[j]);
    }
}

// ---- the scene cut on the same radix tree (pgpu_partition_prims_device): with the whole mesh as one group, the subtrees that hold
// at most maxPrims triangles under a parent that holds more are the batching points. Sorted positions are cut into runs; a run's
// number is the count of run starts up to it.
__global__ void __launch_bounds__(256) cutKernel(uint32_t n, uint32_t maxPrims, const uint32_t* __restrict__ rangeLo, const uint32_t* __restrict__ rangeHi,
    const uint32_t* __restrict__ parentInt, const uint32_t* __restrict__ parentLeaf, uint32_t* __restrict__ startFlag)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n)
        return;
    auto count = [&](uint32_t node) { return rangeHi[node] - rangeLo[node] + 1u; };
    // as a triangle: a run of its own when its parent is too large to be one (n == 1: no parent)
    const uint32_t pl = parentLeaf[i];
    if (pl == kNone || count(pl) > maxPrims)
        startFlag[i] = 1u;
    // as an internal node (i < n - 1)
    if (i + 1u < n && count(i) <= maxPrims) {
        const uint32_t pi = parentInt[i];
        if (pi == kNone || count(pi) > maxPrims)
            startFlag[rangeLo[i]] = 1u;
    }
}

__global__ void __launch_bounds__(256) groupOfKernel(uint32_t n, const uint32_t* __restrict__ order, const uint32_t* __restrict__ startScan, uint32_t* __restrict__ groupOfPrim)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p < n)
        groupOfPrim[order[p]] = startScan[p] - 1u; // (inclusive scan of the run starts)
}

// ---- 7. (pgpu_build_upload_bvh8_device) the 48-byte triangle records of the trees, slot by slot: what pgpu_upload_bvh8 gathers on
// the host (Bvh8Dev in kernels.h). Group g's records start g records further than its slots (one zero record of padding behind
// every group: the pair-layout kernel reads records two at a time). shade (optional): the vertices per leaf object.
__global__ void __launch_bounds__(256) triRecordsKernel(uint32_t n, const float* __restrict__ pos, const uint32_t* __restrict__ idx, const uint32_t* __restrict__ prims,
    const unsigned long long* __restrict__ offsets, const uint32_t* __restrict__ groupOf, const uint32_t* __restrict__ leafIdx, float* __restrict__ tris,
    float* __restrict__ shade, uint32_t* __restrict__ groupNonFinite)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n)
        return;
    const uint32_t g = groupOf[p], j = leafIdx[p];
    const size_t base = size_t(offsets[g]);
    const uint32_t prim = prims[base + j];
    float* const t = tris + (size_t(p) + g) * 12u;
    bool finite = true;
    for (int k = 0; k < 3; ++k) {
        const float* v = pos + 3u * size_t(idx[3u * size_t(prim) + k]);
        for (int a = 0; a < 3; ++a) {
            t[4 * k + a] = v[a];
            finite = finite && fabsf(v[a]) <= 3.4028234663852886e38f;
            if (shade)
                shade[(base + j) * 9u + 3u * k + a] = v[a];
        }
        t[4 * k + 3] = k == 0 ? __uint_as_float(j) : 0.0f;
    }
    if (!finite)
        groupNonFinite[g] = 1u; // (its boxes may be unordered or infinite: the x86-exact slab test, Bvh8Dev::needs_exact)
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

inline uint32_t blocksFor(size_t n) { return uint32_t(std::max<size_t>(1, (n + 255) / 256)); }

struct DeviceBuild { // what a build leaves on the device (freed with the object, but for dNodes when the caller asked to keep them)
    Scratch mem;
    uint32_t N = 0;
    uint64_t totalNodes = 0;
    float* dPos = nullptr;
    uint32_t *dIdx = nullptr, *dPrims = nullptr, *dGroupOf = nullptr, *dLeafIdx = nullptr;
    unsigned long long* dOffsets = nullptr;
    unsigned char* dNodes = nullptr; // every group's nodes, group after group (nodeBase[g] = a group's first)
    std::vector<uint32_t> nodeCount, depth, failed, nodeBase;
};

struct ClusterTreeArrays { // in: order, primBox, groupOf, offsets; out (N each): the tree, its boxes, ranges, tables; left / right / order are replaced
    uint32_t* order;
    const Box6* primBox;
    const uint32_t* groupOf;
    const unsigned long long* offsets;
    uint32_t *left, *right, *lo, *hi;
    Box6* nodeBox;
    uint32_t *parentInt, *parentLeaf, *groupRoot;
    float* cost;
    uint32_t* cutWord;
};

// Steps 3b + 4b on the context's stream. *done = false: not built (one triangle per group, switched off by the environment, or
// more than kPlocMaxRounds rounds) and the parent / root arrays are as they were before the call; the caller builds the radix tree.
int clusterTree(pgpu_ctx* c, Scratch& mem, uint32_t N, uint32_t G, uint32_t max_leaf, ClusterTreeArrays& t, bool* done)
{
    hipStream_t stream = c->stream;
    *done = false;
    int radius = 3;
    if (const char* e = std::getenv("PANDORA_BVH8_DEVICE_PLOC_RADIUS"))
        radius = std::max(1, std::min(64, std::atoi(e)));
    int maxRounds = kPlocMaxRounds;
    if (const char* e = std::getenv("PANDORA_BVH8_DEVICE_PLOC_ROUNDS")) // (tests: a budget the clustering cannot meet, so that the radix tree takes over)
        maxRounds = std::max(1, std::min(kPlocMaxRounds, std::atoi(e)));
    const char* treeEnv = std::getenv("PANDORA_BVH8_DEVICE_TREE");
    if (N <= 1u || N <= G || (treeEnv && std::strcmp(treeEnv, "lbvh") == 0))
        return PGPU_OK;
    PlocCluster* dCl[2] = { nullptr, nullptr };
    Box6* dClBox[2] = { nullptr, nullptr };
    uint32_t *dNn = nullptr, *dCount = nullptr, *dNewPos = nullptr, *dOrder2 = nullptr, *dLeft2 = nullptr, *dRight2 = nullptr, *dError = nullptr;
    unsigned long long *dFlags = nullptr, *dPlaces = nullptr;
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(c, mem.alloc(&dCl[k], N));
        HIP_TRY(c, mem.alloc(&dClBox[k], N));
    }
    HIP_TRY(c, mem.alloc(&dNn, N));
    HIP_TRY(c, mem.alloc(&dCount, N));
    HIP_TRY(c, mem.alloc(&dNewPos, N));
    HIP_TRY(c, mem.alloc(&dOrder2, N));
    HIP_TRY(c, mem.alloc(&dLeft2, N));
    HIP_TRY(c, mem.alloc(&dRight2, N));
    HIP_TRY(c, mem.alloc(&dError, 1));
    HIP_TRY(c, mem.alloc(&dFlags, size_t(N) + 1));
    HIP_TRY(c, mem.alloc(&dPlaces, size_t(N) + 1));
    size_t placeBytes = 0;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, placeBytes, dFlags, dPlaces, int(N) + 1, stream));
    unsigned char* dPlaceTemp = nullptr;
    HIP_TRY(c, mem.alloc(&dPlaceTemp, placeBytes));
    HIP_TRY(c, hipMemsetAsync(dError, 0, 4, stream));
    hipLaunchKernelGGL(plocInitKernel, dim3(blocksFor(N)), dim3(256), 0, stream, N, t.order, t.primBox, t.groupOf, t.offsets, dCl[0], dClBox[0]);
    HIP_TRY(c, hipGetLastError());
    uint32_t nActive = N, nodesMade = 0;
    int rounds = 0;
    for (; nActive > G && rounds < maxRounds; ++rounds) {
        const int in = rounds & 1, out = in ^ 1;
        hipLaunchKernelGGL(plocNearestKernel, dim3(blocksFor(nActive)), dim3(256), 0, stream, nActive, radius, dCl[in], dClBox[in], dNn);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(plocFlagKernel, dim3(blocksFor(size_t(nActive) + 1)), dim3(256), 0, stream, nActive, dNn, dFlags);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(dPlaceTemp, placeBytes, dFlags, dPlaces, int(nActive) + 1, stream));
        hipLaunchKernelGGL(plocCommitKernel, dim3(blocksFor(nActive)), dim3(256), 0, stream, nActive, N, nodesMade, dNn, dFlags, dPlaces, dCl[in], dClBox[in], dCl[out],
            dClBox[out], t.left, t.right, t.nodeBox, dCount, t.parentInt, t.parentLeaf, max_leaf, t.cost, t.cutWord);
        HIP_TRY(c, hipGetLastError());
        unsigned long long totals = 0;
        HIP_TRY(c, hipMemcpyAsync(&totals, dPlaces + nActive, 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        const uint32_t stay = uint32_t(totals & 0xFFFFFFFFull), merges = uint32_t(totals >> 32);
        if (merges == 0u || stay + merges != nActive || stay < G)
            return fail(c, PGPU_ERR_HIP, "pgpu_build_bvh8_device: a round of the clustering merged nothing (internal error)");
        nodesMade += merges;
        nActive = stay;
    }
    if (std::getenv("PANDORA_BVH8_DEVICE_DEBUG"))
        std::fprintf(stderr, "bvh_build: clustering, radius %d: %d rounds, %u nodes, %u clusters left for %u groups\n", radius, rounds, nodesMade, nActive, G);
    if (nActive != G) {
        // too many rounds (a chain of merges: nothing seen on real geometry): the radix tree instead
        HIP_TRY(c, hipMemsetAsync(t.parentInt, 0xFF, size_t(N) * 4, stream));
        HIP_TRY(c, hipMemsetAsync(t.parentLeaf, 0xFF, size_t(N) * 4, stream));
        HIP_TRY(c, hipMemsetAsync(t.groupRoot, 0xFF, size_t(G) * 4, stream));
        return PGPU_OK;
    }
    uint32_t error = 0;
    hipLaunchKernelGGL(plocRootsKernel, dim3(blocksFor(nActive)), dim3(256), 0, stream, nActive, G, dCl[rounds & 1], t.groupRoot);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(plocPositionKernel, dim3(blocksFor(N)), dim3(256), 0, stream, N, uint32_t(rounds) + 1u, t.groupOf, t.offsets, t.left, t.right, t.parentInt, t.parentLeaf,
        dCount, dNewPos, dError);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(plocRangesKernel, dim3(blocksFor(nodesMade)), dim3(256), 0, stream, nodesMade, N, uint32_t(rounds) + 1u, t.left, t.right, dCount, dNewPos, t.lo, t.hi,
        dLeft2, dRight2, dError);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(plocReorderKernel, dim3(blocksFor(N)), dim3(256), 0, stream, N, t.order, dNewPos, dOrder2);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(&error, dError, 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    if (error != 0u)
        return fail(c, PGPU_ERR_HIP, "pgpu_build_bvh8_device: the clustered tree is not a tree (internal error)");
    t.left = dLeft2;
    t.right = dRight2;
    t.order = dOrder2;
    *done = true;
    return PGPU_OK;
}

// Validates the arguments and runs steps 1-6; ev0 is recorded behind the uploads (the caller records ev1).
int buildOnDevice(pgpu_ctx* c, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles, const uint32_t* group_prims,
    const uint64_t* group_offsets, size_t n_groups, uint32_t max_leaf, bool persistentNodes, DeviceBuild& b)
{
    if (!c || !positions || !indices || !group_prims || !group_offsets || n_groups == 0)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: null or empty argument");
    if (max_leaf < 1 || max_leaf > 4)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: max_leaf must be 1..4");
    if (n_groups > (1u << 20) || n_vertices > 0xFFFFFFFFull / 3 || n_triangles > 0xFFFFFFFFull / 3)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: more than 2^20 groups or 2^32 / 3 vertices / triangles");
    const uint64_t nEntries = group_offsets[n_groups];
    if (group_offsets[0] != 0 || nEntries == 0 || nEntries >= 0x7FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: offsets must start at 0 and name 1 .. 2^31 - 2 triangles in all");
    for (size_t g = 0; g < n_groups; ++g) {
        if (group_offsets[g] >= group_offsets[g + 1])
            return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: empty group (offsets must increase)");
        if (group_offsets[g + 1] - group_offsets[g] >= (1u << 29) - 1u)
            return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: more than 2^29 triangles in one group");
    }
    for (uint64_t i = 0; i < nEntries; ++i)
        if (group_prims[i] >= n_triangles)
            return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: primitive index out of range");
    for (size_t i = 0; i < 3 * n_triangles; ++i)
        if (indices[i] >= n_vertices)
            return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: vertex index out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = c->stream;
    Scratch& mem = b.mem;
    const uint32_t N = uint32_t(nEntries), G = uint32_t(n_groups);
    // 14 bits per axis whatever the number of groups (2^20 at most: 62 bits in all), so that a group's tree is the same whether
    // it is built alone or with others
    const int bitsPerAxis = 14, groupBits = 20;

    float* dPos = nullptr;
    uint32_t *dIdx = nullptr, *dPrims = nullptr, *dGroupOf = nullptr, *dCBounds = nullptr, *dVals = nullptr, *dOrder = nullptr;
    unsigned long long *dOffsets = nullptr, *dKeys = nullptr, *dKeysSorted = nullptr;
    Box6 *dPrimBox = nullptr, *dNodeBox = nullptr;
    uint32_t *dLeft = nullptr, *dRight = nullptr, *dLo = nullptr, *dHi = nullptr, *dParentInt = nullptr, *dParentLeaf = nullptr, *dVisits = nullptr;
    uint32_t *dGroupRoot = nullptr, *dGroupNodes = nullptr, *dGroupDepth = nullptr, *dGroupFailed = nullptr, *dNodeBase = nullptr, *dInnerCount = nullptr, *dSlotStart = nullptr, *dLeafIdx = nullptr;
    WorkItem* dQueue[2] = { nullptr, nullptr };
    NodeRec* dRecs = nullptr;
    HIP_TRY(c, mem.alloc(&dPos, 3 * n_vertices));
    HIP_TRY(c, mem.alloc(&dIdx, 3 * n_triangles));
    HIP_TRY(c, mem.alloc(&dPrims, N));
    HIP_TRY(c, mem.alloc(&dOffsets, n_groups + 1));
    HIP_TRY(c, mem.alloc(&dGroupOf, N));
    HIP_TRY(c, mem.alloc(&dCBounds, 6 * n_groups));
    HIP_TRY(c, mem.alloc(&dPrimBox, N));
    HIP_TRY(c, mem.alloc(&dKeys, N));
    HIP_TRY(c, mem.alloc(&dKeysSorted, N));
    HIP_TRY(c, mem.alloc(&dVals, N));
    HIP_TRY(c, mem.alloc(&dOrder, N));
    HIP_TRY(c, mem.alloc(&dLeft, N));
    HIP_TRY(c, mem.alloc(&dRight, N));
    HIP_TRY(c, mem.alloc(&dLo, N));
    HIP_TRY(c, mem.alloc(&dHi, N));
    HIP_TRY(c, mem.alloc(&dParentInt, N));
    HIP_TRY(c, mem.alloc(&dParentLeaf, N));
    HIP_TRY(c, mem.alloc(&dVisits, N));
    HIP_TRY(c, mem.alloc(&dNodeBox, N));
    float* dCost = nullptr;
    uint32_t* dCutWord = nullptr;
    const char* collapseEnv = std::getenv("PANDORA_BVH8_DEVICE_COLLAPSE"); // "greedy": without the tables (A/B, tests)
    const bool tables = !(collapseEnv && std::strcmp(collapseEnv, "greedy") == 0);
    if (tables) {
        HIP_TRY(c, mem.alloc(&dCost, size_t(N) * 8u));
        HIP_TRY(c, mem.alloc(&dCutWord, N));
    }
    HIP_TRY(c, mem.alloc(&dGroupRoot, n_groups));
    HIP_TRY(c, mem.alloc(&dGroupNodes, n_groups));
    HIP_TRY(c, mem.alloc(&dGroupDepth, n_groups));
    HIP_TRY(c, mem.alloc(&dGroupFailed, n_groups));
    HIP_TRY(c, mem.alloc(&dNodeBase, n_groups));
    HIP_TRY(c, mem.alloc(&dInnerCount, size_t(N) + 1));
    HIP_TRY(c, mem.alloc(&dSlotStart, size_t(N) + 1));
    HIP_TRY(c, mem.alloc(&dQueue[0], N));
    HIP_TRY(c, mem.alloc(&dQueue[1], N));
    HIP_TRY(c, mem.alloc(&dRecs, N));
    HIP_TRY(c, mem.alloc(&dLeafIdx, N));
    size_t sortBytes = 0;
    const int endBit = std::min(64, 3 * bitsPerAxis + groupBits);
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, dKeys, dKeysSorted, dVals, dOrder, int(N), 0, endBit, stream));
    unsigned char* dSortTemp = nullptr;
    HIP_TRY(c, mem.alloc(&dSortTemp, sortBytes));
    size_t scanBytes = 0;
    HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, dInnerCount, dSlotStart, int(N) + 1, stream));
    unsigned char* dScanTemp = nullptr;
    HIP_TRY(c, mem.alloc(&dScanTemp, scanBytes));

    HIP_TRY(c, hipMemcpyAsync(dPos, positions, 3 * n_vertices * sizeof(float), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dIdx, indices, 3 * n_triangles * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dPrims, group_prims, size_t(N) * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dOffsets, group_offsets, (n_groups + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
    std::vector<uint32_t> boundsInit(6 * n_groups);
    for (size_t g = 0; g < n_groups; ++g)
        for (int a = 0; a < 3; ++a) {
            boundsInit[6 * g + size_t(a)] = 0xFFFFFFFFu;
            boundsInit[6 * g + 3 + size_t(a)] = 0u;
        }
    HIP_TRY(c, hipMemcpyAsync(dCBounds, boundsInit.data(), boundsInit.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemsetAsync(dParentInt, 0xFF, size_t(N) * 4, stream));
    HIP_TRY(c, hipMemsetAsync(dParentLeaf, 0xFF, size_t(N) * 4, stream));
    HIP_TRY(c, hipMemsetAsync(dVisits, 0, size_t(N) * 4, stream));
    HIP_TRY(c, hipMemsetAsync(dGroupRoot, 0xFF, n_groups * 4, stream));
    HIP_TRY(c, hipMemsetAsync(dGroupDepth, 0, n_groups * 4, stream));
    HIP_TRY(c, hipMemsetAsync(dGroupFailed, 0, n_groups * 4, stream));
    HIP_TRY(c, hipEventRecord(c->ev0, stream));

    hipLaunchKernelGGL(primBoundsKernel, dim3(blocksFor(N)), dim3(256), 0, stream, dPos, dIdx, dPrims, dOffsets, G, N, dGroupOf, dPrimBox, dCBounds);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(mortonKernel, dim3(blocksFor(N)), dim3(256), 0, stream, dPrimBox, dGroupOf, dCBounds, N, bitsPerAxis, dKeys, dVals);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(dSortTemp, sortBytes, dKeys, dKeysSorted, dVals, dOrder, int(N), 0, endBit, stream));
    // the binary tree: locally-ordered clustering (PANDORA_BVH8_DEVICE_TREE=ploc, the default; PANDORA_BVH8_DEVICE_PLOC_RADIUS places to
    // either side, default 3) or the radix tree over the keys (=lbvh; also what is built if the clustering does not finish in
    // kPlocMaxRounds rounds, which bounds the depth of its tree)
    bool clustered = false;
    {
        ClusterTreeArrays t { dOrder, dPrimBox, dGroupOf, dOffsets, dLeft, dRight, dLo, dHi, dNodeBox, dParentInt, dParentLeaf, dGroupRoot, dCost, dCutWord };
        const int rt = clusterTree(c, mem, N, G, max_leaf, t, &clustered);
        if (rt != PGPU_OK)
            return rt;
        dLeft = t.left;
        dRight = t.right;
        dOrder = t.order;
    }
    if (N > 1u && !clustered) {
        hipLaunchKernelGGL(hierarchyKernel, dim3(blocksFor(N - 1u)), dim3(256), 0, stream, dKeysSorted, N, dGroupOf, dOffsets, dLeft, dRight, dLo, dHi, dParentInt,
            dParentLeaf, dGroupRoot);
        HIP_TRY(c, hipGetLastError());
        hipLaunchKernelGGL(fitKernel, dim3(blocksFor(N)), dim3(256), 0, stream, N, dOrder, dPrimBox, dLeft, dRight, dParentInt, dParentLeaf, dVisits, dNodeBox, dLo, dHi,
            max_leaf, dCost, dCutWord);
        HIP_TRY(c, hipGetLastError());
    }
    // seeds, in group order: the root of every group that is more than one leaf
    std::vector<uint32_t> roots(n_groups);
    std::vector<uint32_t>& nodeCount = b.nodeCount;
    nodeCount.assign(n_groups, 0u);
    HIP_TRY(c, hipMemcpyAsync(roots.data(), dGroupRoot, n_groups * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    std::vector<WorkItem> seeds;
    for (size_t g = 0; g < n_groups; ++g)
        if (group_offsets[g + 1] - group_offsets[g] > max_leaf) {
            if (roots[g] + 1u >= N)
                return fail(c, PGPU_ERR_HIP, "pgpu_build_bvh8_device: a group is not a subtree of the radix tree (internal error)");
            seeds.push_back(WorkItem { roots[g], 0u, uint32_t(g) });
            nodeCount[g] = 1u;
        }
    HIP_TRY(c, hipMemcpyAsync(dGroupNodes, nodeCount.data(), n_groups * 4, hipMemcpyHostToDevice, stream));
    if (!seeds.empty())
        HIP_TRY(c, hipMemcpyAsync(dQueue[0], seeds.data(), seeds.size() * sizeof(WorkItem), hipMemcpyHostToDevice, stream));
    int maxLevels = kMaxLevels;
    if (const char* e = std::getenv("PANDORA_BVH8_DEVICE_LEVELS")) // (tests: a smaller budget sends groups to the host builder)
        maxLevels = std::max(1, std::min(kMaxLevels, std::atoi(e)));
    int margin = 2;
    if (const char* e = std::getenv("PANDORA_BVH8_DEVICE_MARGIN"))
        margin = std::max(0, std::min(kMaxLevels, std::atoi(e)));
    // The collapse runs level by level over whatever queue it is given. First pass: every group, by the tables. The tables know
    // nothing of the depth budget; a group whose tree comes out deeper than the traversal stack allows goes through a second pass
    // that opens children by surface and, near the budget, by size (what then still does not fit is the host builder's).
    auto collapse = [&](uint32_t count, bool useTables) -> int {
        for (int level = 0; level < maxLevels && count != 0u; ++level) {
            if (std::getenv("PANDORA_BVH8_DEVICE_DEBUG"))
                std::fprintf(stderr, "bvh_build: level %d, %u nodes\n", level, count);
            // A child of a node at this level may root (maxLevels - 1 - level) levels of inner nodes, which hold max_leaf * 8^that
            // triangles if every node below is full. A radix tree is not that regular: children are opened by size rather than by
            // surface once they hold what `margin` levels fewer could (a sparse scene's largest boxes are single triangles near the
            // root; opening by surface alone peels them off one level at a time and runs out of depth on the dense parts).
            unsigned long long capacity = max_leaf;
            for (int l = level + 1 + margin; l < maxLevels; ++l)
                capacity *= 8ull;
            hipLaunchKernelGGL(expandKernel, dim3(blocksFor(size_t(count) + 1)), dim3(256), 0, stream, dQueue[level & 1], count, uint32_t(level), uint32_t(maxLevels), max_leaf,
                capacity, dLeft, dRight, dLo, dHi, dNodeBox, dOffsets, dRecs, dInnerCount, dGroupDepth, dGroupFailed, useTables ? dCutWord : nullptr);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipcub::DeviceScan::ExclusiveSum(dScanTemp, scanBytes, dInnerCount, dSlotStart, int(count) + 1, stream));
            hipLaunchKernelGGL(assignKernel, dim3(blocksFor(count)), dim3(256), 0, stream, dQueue[level & 1], count, dInnerCount, dSlotStart, dOffsets, dRecs,
                dQueue[(level + 1) & 1], dGroupNodes);
            HIP_TRY(c, hipGetLastError());
            uint32_t next = 0;
            HIP_TRY(c, hipMemcpyAsync(&next, dSlotStart + count, 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(c, hipStreamSynchronize(stream));
            if (next >= N)
                return fail(c, PGPU_ERR_HIP, "pgpu_build_bvh8_device: work queue overran (internal error)");
            count = next;
        }
        return PGPU_OK;
    };
    {
        const int rcA = collapse(uint32_t(seeds.size()), tables);
        if (rcA != PGPU_OK)
            return rcA;
    }
    if (tables) {
        std::vector<uint32_t> failedA(n_groups), zeros(n_groups, 0u);
        HIP_TRY(c, hipMemcpyAsync(failedA.data(), dGroupFailed, n_groups * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipMemcpyAsync(nodeCount.data(), dGroupNodes, n_groups * 4, hipMemcpyDeviceToHost, stream));
        std::vector<uint32_t> depthA(n_groups);
        HIP_TRY(c, hipMemcpyAsync(depthA.data(), dGroupDepth, n_groups * 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(c, hipStreamSynchronize(stream));
        std::vector<WorkItem> again;
        for (const WorkItem& w : seeds)
            if (failedA[w.group]) {
                again.push_back(w);
                nodeCount[w.group] = 1u;
                depthA[w.group] = 0u;
            }
        if (!again.empty()) {
            if (std::getenv("PANDORA_BVH8_DEVICE_DEBUG"))
                std::fprintf(stderr, "bvh_build: %zu groups too deep by the tables, second pass\n", again.size());
            HIP_TRY(c, hipMemcpyAsync(dGroupNodes, nodeCount.data(), n_groups * 4, hipMemcpyHostToDevice, stream));
            HIP_TRY(c, hipMemcpyAsync(dGroupDepth, depthA.data(), n_groups * 4, hipMemcpyHostToDevice, stream));
            HIP_TRY(c, hipMemsetAsync(dGroupFailed, 0, n_groups * 4, stream));
            HIP_TRY(c, hipMemcpyAsync(dQueue[0], again.data(), again.size() * sizeof(WorkItem), hipMemcpyHostToDevice, stream));
            HIP_TRY(c, hipStreamSynchronize(stream)); // (`again` and the vectors above go out of scope)
            const int rcB = collapse(uint32_t(again.size()), false);
            if (rcB != PGPU_OK)
                return rcB;
        }
    }
    std::vector<uint32_t>&depth = b.depth, &failed = b.failed, &nodeBase = b.nodeBase;
    depth.assign(n_groups, 0u);
    failed.assign(n_groups, 0u);
    nodeBase.assign(n_groups, 0u);
    HIP_TRY(c, hipMemcpyAsync(nodeCount.data(), dGroupNodes, n_groups * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipMemcpyAsync(depth.data(), dGroupDepth, n_groups * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipMemcpyAsync(failed.data(), dGroupFailed, n_groups * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    if (std::getenv("PANDORA_BVH8_DEVICE_DEBUG")) {
        for (size_t g = 0; g < std::min<size_t>(n_groups, 8); ++g)
            std::fprintf(stderr, "bvh_build: group %zu: %llu triangles, root %u, %u nodes, depth %u, failed %u\n", g,
                (unsigned long long)(group_offsets[g + 1] - group_offsets[g]), roots[g], nodeCount[g], depth[g], failed[g]);
    }
    uint64_t totalNodes = 0;
    for (size_t g = 0; g < n_groups; ++g) {
        if (nodeCount[g] >= group_offsets[g + 1] - group_offsets[g] && nodeCount[g] != 0)
            return fail(c, PGPU_ERR_HIP, "pgpu_build_bvh8_device: more nodes than triangles in a group (internal error)");
        if (failed[g])
            nodeCount[g] = 0;
        nodeBase[g] = uint32_t(totalNodes);
        totalNodes += nodeCount[g];
    }
    if (totalNodes >= 0xFFFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: more than 2^32 nodes in all");
    unsigned char* dNodes = nullptr;
    if (persistentNodes) { // (the caller's from here on: it becomes part of the scene)
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&dNodes), std::max<size_t>(size_t(totalNodes) * 256, 256)));
        b.dNodes = dNodes;
    } else {
        HIP_TRY(c, mem.alloc(&dNodes, size_t(totalNodes) * 256));
    }
    HIP_TRY(c, hipMemcpyAsync(dNodeBase, nodeBase.data(), n_groups * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dGroupNodes, nodeCount.data(), n_groups * 4, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(emitKernel, dim3(blocksFor(N)), dim3(256), 0, stream, N, dGroupOf, dOffsets, dGroupNodes, dGroupFailed, dNodeBase, dRecs, dOrder, dPrimBox, dNodeBox,
        dNodes, dLeafIdx);
    HIP_TRY(c, hipGetLastError());
    b.N = N;
    b.totalNodes = totalNodes;
    b.dPos = dPos;
    b.dIdx = dIdx;
    b.dPrims = dPrims;
    b.dOffsets = dOffsets;
    b.dGroupOf = dGroupOf;
    b.dLeafIdx = dLeafIdx;
    b.dNodes = dNodes;
    return PGPU_OK;
}

} // namespace

extern "C" int pgpu_build_bvh8_device(pgpu_ctx* c, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* group_prims, const uint64_t* group_offsets, size_t n_groups, uint32_t max_leaf, pgpu_host_bvh8** out, uint32_t* n_host_built)
{
    const TraceRange range_("pgpu_build_bvh8_device");
    if (!out)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: null argument");
    if (n_host_built)
        *n_host_built = 0;
    DeviceBuild b;
    {
        const int rb = buildOnDevice(c, positions, n_vertices, indices, n_triangles, group_prims, group_offsets, n_groups, max_leaf, false, b);
        if (rb != PGPU_OK)
            return rb;
    }
    for (size_t g = 0; g < n_groups; ++g)
        out[g] = nullptr;
    hipStream_t stream = c->stream;
    const uint32_t N = b.N;
    const uint64_t totalNodes = b.totalNodes;
    unsigned char* const dNodes = b.dNodes;
    uint32_t* const dLeafIdx = b.dLeafIdx;
    const std::vector<uint32_t>&nodeCount = b.nodeCount, &depth = b.depth, &failed = b.failed, &nodeBase = b.nodeBase;
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    std::vector<unsigned char> nodes(size_t(totalNodes) * 256);
    std::vector<uint32_t> leafIdx(N);
    if (totalNodes)
        HIP_TRY(c, hipMemcpyAsync(nodes.data(), dNodes, nodes.size(), hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipMemcpyAsync(leafIdx.data(), dLeafIdx, size_t(N) * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    HIP_TRY(c, hipEventElapsedTime(&c->lastKernelMs, c->ev0, c->ev1));
    c->timingStale = false;

    uint32_t hostBuilt = 0;
    int rc = PGPU_OK;
    for (size_t g = 0; g < n_groups && rc == PGPU_OK; ++g) {
        const size_t first = size_t(group_offsets[g]), count = size_t(group_offsets[g + 1] - group_offsets[g]);
        if (failed[g]) {
            ++hostBuilt;
            rc = pgpu_build_bvh8(positions, n_vertices, indices, n_triangles, group_prims + first, count, max_leaf, &out[g]);
            if (rc != PGPU_OK)
                c->err = "pgpu_build_bvh8_device: the host builder failed on a group the device tree was too deep for";
            continue;
        }
        const uint32_t root = count <= max_leaf ? ((uint32_t(count) - 1u) | 4u) << 29 : (2u << 29);
        out[g] = makeHostBvh8(nodes.data() + size_t(nodeBase[g]) * 256, nodeCount[g], leafIdx.data() + first, count, root, int(depth[g]));
        if (!out[g])
            rc = fail(c, PGPU_ERR_INVALID, "pgpu_build_bvh8_device: out of host memory");
    }
    if (rc != PGPU_OK) {
        for (size_t g = 0; g < n_groups; ++g)
            if (out[g]) {
                pgpu_host_bvh8_free(out[g]);
                out[g] = nullptr;
            }
        return rc;
    }
    if (n_host_built)
        *n_host_built = hostBuilt;
    return PGPU_OK;
}

// The same build with the trees left where they were made: nodes, triangle records (and, with pgpu_set_keep_shading, the vertices per
// leaf object) stay in HBM and join the context's BVH table as if pgpu_upload_bvh8 had been called per group with leaf objects
// (mesh, group_prims[j]) -- nothing of them crosses PCIe.
extern "C" int pgpu_build_upload_bvh8_device(pgpu_ctx* c, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    const uint32_t* group_prims, const uint64_t* group_offsets, size_t n_groups, uint32_t max_leaf, uint32_t* bvh_ids, uint32_t* n_host_built)
{
    const TraceRange range_("pgpu_build_upload_bvh8_device");
    if (!bvh_ids)
        return fail(c, PGPU_ERR_INVALID, "pgpu_build_upload_bvh8_device: null argument");
    if (n_host_built)
        *n_host_built = 0;
    DeviceBuild b;
    {
        const int rb = buildOnDevice(c, positions, n_vertices, indices, n_triangles, group_prims, group_offsets, n_groups, max_leaf, true, b);
        if (rb != PGPU_OK) {
            if (b.dNodes)
                (void)hipFree(b.dNodes);
            return rb;
        }
    }
    c->owned.push_back(b.dNodes); // (from here on the context frees it)
    hipStream_t stream = c->stream;
    const uint32_t N = b.N, G = uint32_t(n_groups);
    float *dTris = nullptr, *dShade = nullptr;
    uint32_t* dNonFinite = nullptr;
    const size_t triFloats = (size_t(N) + G) * 12u;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&dTris), triFloats * sizeof(float)));
    c->owned.push_back(dTris);
    if (c->keepShading) {
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&dShade), size_t(N) * 9u * sizeof(float)));
        c->owned.push_back(dShade);
    }
    HIP_TRY(c, b.mem.alloc(&dNonFinite, n_groups));
    HIP_TRY(c, hipMemsetAsync(dTris, 0, triFloats * sizeof(float), stream));
    HIP_TRY(c, hipMemsetAsync(dNonFinite, 0, n_groups * 4, stream));
    hipLaunchKernelGGL(triRecordsKernel, dim3(blocksFor(N)), dim3(256), 0, stream, N, b.dPos, b.dIdx, b.dPrims, b.dOffsets, b.dGroupOf, b.dLeafIdx, dTris, dShade, dNonFinite);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    std::vector<uint32_t> nonFinite(n_groups);
    HIP_TRY(c, hipMemcpyAsync(nonFinite.data(), dNonFinite, n_groups * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    HIP_TRY(c, hipEventElapsedTime(&c->lastKernelMs, c->ev0, c->ev1));
    c->timingStale = false;

    uint32_t hostBuilt = 0;
    const size_t firstId = c->bvhs.size();
    int rc = PGPU_OK;
    for (size_t g = 0; g < n_groups && rc == PGPU_OK; ++g) {
        const size_t first = size_t(group_offsets[g]), count = size_t(group_offsets[g + 1] - group_offsets[g]);
        Bvh8Dev d {};
        d.n_slots = uint32_t(count);
        d.needs_exact = nonFinite[g];
        d.shade = dShade ? dShade + first * 9u : nullptr;
        if (!b.failed[g]) {
            d.nodes = b.dNodes + size_t(b.nodeBase[g]) * 256u;
            d.tris = reinterpret_cast<const unsigned char*>(dTris + (first + g) * 12u);
            d.root = count <= max_leaf ? ((uint32_t(count) - 1u) | 4u) << 29 : (2u << 29);
            d.n_nodes = b.nodeCount[g];
            d.depth = b.depth[g];
        } else {
            // too deep for the traversal stack: the host builder's tree, its nodes and records uploaded apart (the shade array
            // is indexed by leaf object and already filled)
            ++hostBuilt;
            pgpu_host_bvh8* hb = nullptr;
            rc = pgpu_build_bvh8(positions, n_vertices, indices, n_triangles, group_prims + first, count, max_leaf, &hb);
            if (rc != PGPU_OK) {
                c->err = "pgpu_build_upload_bvh8_device: the host builder failed on a group the device tree was too deep for";
                break;
            }
            size_t nNodes = 0, nIdx = 0;
            const void* hNodes = pgpu_host_bvh8_nodes(hb, &nNodes);
            const uint32_t* hIdx = pgpu_host_bvh8_leaf_indices(hb, &nIdx);
            std::vector<float> tris((nIdx + 1) * 12, 0.0f);
            for (size_t sl = 0; sl < nIdx; ++sl) {
                const uint32_t j = hIdx[sl], prim = group_prims[first + j];
                for (int k = 0; k < 3; ++k)
                    std::memcpy(&tris[sl * 12 + size_t(k) * 4], positions + 3u * size_t(indices[3u * size_t(prim) + size_t(k)]), 12);
                std::memcpy(&tris[sl * 12 + 3], &j, 4);
            }
            unsigned char* devNodes = nullptr;
            float* devTris = nullptr;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&devNodes), std::max<size_t>(nNodes * 256, 256));
            if (e == hipSuccess) {
                c->owned.push_back(devNodes);
                e = hipMalloc(reinterpret_cast<void**>(&devTris), tris.size() * sizeof(float));
            }
            if (e == hipSuccess) {
                c->owned.push_back(devTris);
                if (nNodes)
                    e = hipMemcpy(devNodes, hNodes, nNodes * 256, hipMemcpyHostToDevice);
            }
            if (e == hipSuccess)
                e = hipMemcpy(devTris, tris.data(), tris.size() * sizeof(float), hipMemcpyHostToDevice);
            d.nodes = devNodes;
            d.tris = reinterpret_cast<const unsigned char*>(devTris);
            d.root = pgpu_host_bvh8_root(hb);
            d.n_nodes = uint32_t(nNodes);
            d.depth = uint32_t(pgpu_host_bvh8_depth(hb));
            pgpu_host_bvh8_free(hb);
            if (e != hipSuccess) {
                rc = fail(c, PGPU_ERR_HIP, "pgpu_build_upload_bvh8_device: upload of a host-built group", e);
                break;
            }
        }
        c->bvhs.push_back(d);
        bvh_ids[g] = uint32_t(c->bvhs.size() - 1);
    }
    if (rc != PGPU_OK) {
        c->bvhs.resize(firstId); // (the arenas stay with the context until it shuts down)
        return rc;
    }
    c->staged.resize(c->bvhs.size());
    c->bvhTableDirty = true;
    if (n_host_built)
        *n_host_built = hostBuilt;
    return syncBvhTable(c);
}

// A resident or staged BVH back in the reference's arrays (what pgpu_upload_bvh8 was given): for trees that were built on the device,
// checks, and the on-disk format (pgpu_bvh8_serialize).
extern "C" int pgpu_bvh8_info(pgpu_ctx* c, uint32_t bvh_id, size_t* n_nodes, size_t* n_indices, uint32_t* root_handle, int* depth)
{
    if (!c || bvh_id >= c->bvhs.size())
        return fail(c, PGPU_ERR_INVALID, "pgpu_bvh8_info: unknown bvh id");
    const Bvh8Dev& d = c->bvhs[bvh_id];
    if (n_nodes)
        *n_nodes = d.n_nodes;
    if (n_indices)
        *n_indices = d.n_slots;
    if (root_handle)
        *root_handle = d.root;
    if (depth)
        *depth = int(d.depth);
    return PGPU_OK;
}

extern "C" int pgpu_download_bvh8(pgpu_ctx* c, uint32_t bvh_id, void* nodes256, uint32_t* leaf_indices)
{
    if (!c || bvh_id >= c->bvhs.size() || !leaf_indices)
        return fail(c, PGPU_ERR_INVALID, "pgpu_download_bvh8: unknown bvh id or null argument");
    const Bvh8Dev& d = c->bvhs[bvh_id];
    if (d.n_nodes && !nodes256)
        return fail(c, PGPU_ERR_INVALID, "pgpu_download_bvh8: null node array");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<float> tris(size_t(d.n_slots) * 12u);
    if (bvh_id < c->staged.size() && c->staged[bvh_id].host) { // kept in pinned host memory: nodes, then (256-aligned) the records
        const StagedBvh& st = c->staged[bvh_id];
        if (d.n_nodes)
            std::memcpy(nodes256, st.host, size_t(d.n_nodes) * 256u);
        std::memcpy(tris.data(), st.host + alignUp(st.nodeBytes), tris.size() * sizeof(float));
    } else {
        if (!d.nodes || !d.tris)
            return fail(c, PGPU_ERR_NO_SCENE, "pgpu_download_bvh8: the BVH is not resident on this GPU");
        HIP_TRY(c, hipDeviceSynchronize());
        if (d.n_nodes)
            HIP_TRY(c, hipMemcpy(nodes256, d.nodes, size_t(d.n_nodes) * 256u, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(tris.data(), d.tris, tris.size() * sizeof(float), hipMemcpyDeviceToHost));
    }
    for (size_t sl = 0; sl < d.n_slots; ++sl)
        std::memcpy(&leaf_indices[sl], &tris[sl * 12u + 3u], 4); // (the leaf-object index rides in p0.w of every record)
    return PGPU_OK;
}

// The cut of a mesh into batching points (pgpu_partition_prims; the reference: detail::createSubScenes, batching.cpp:38-265) on the
// GPU: Morton order of the whole mesh (21 bits per axis), the radix tree, and the subtrees of at most max_prims triangles under a
// parent that holds more. Groups are numbered along the Morton curve. LBVH quality: the groups are boxes of the Morton grid rather
// than a SAH cut's, and some are small (a subtree next to a large sibling).
extern "C" int pgpu_partition_prims_device(pgpu_ctx* c, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    size_t max_prims, uint32_t* group_of_prim, uint32_t* n_groups)
{
    const TraceRange range_("pgpu_partition_prims_device");
    if (!c || !positions || !indices || !group_of_prim || !n_groups || max_prims == 0 || n_triangles == 0)
        return fail(c, PGPU_ERR_INVALID, "pgpu_partition_prims_device: null or empty argument");
    if (n_vertices > 0xFFFFFFFFull / 3 || n_triangles >= 0x7FFFFFFFull / 3)
        return fail(c, PGPU_ERR_INVALID, "pgpu_partition_prims_device: more than 2^31 / 3 triangles or 2^32 / 3 vertices");
    for (size_t i = 0; i < 3 * n_triangles; ++i)
        if (indices[i] >= n_vertices)
            return fail(c, PGPU_ERR_INVALID, "pgpu_partition_prims_device: vertex index out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = c->stream;
    Scratch mem;
    const uint32_t N = uint32_t(n_triangles);
    const uint32_t maxPrims = uint32_t(std::min<size_t>(max_prims, 0xFFFFFFFFu));
    float* dPos = nullptr;
    uint32_t *dIdx = nullptr, *dGroupOf = nullptr, *dCBounds = nullptr, *dVals = nullptr, *dOrder = nullptr, *dLeft = nullptr, *dRight = nullptr, *dLo = nullptr, *dHi = nullptr,
             *dParentInt = nullptr, *dParentLeaf = nullptr, *dGroupRoot = nullptr, *dFlag = nullptr, *dScan = nullptr, *dOut = nullptr;
    unsigned long long *dOffsets = nullptr, *dKeys = nullptr, *dKeysSorted = nullptr;
    Box6* dPrimBox = nullptr;
    HIP_TRY(c, mem.alloc(&dPos, 3 * n_vertices));
    HIP_TRY(c, mem.alloc(&dIdx, 3 * n_triangles));
    HIP_TRY(c, mem.alloc(&dOffsets, 2));
    HIP_TRY(c, mem.alloc(&dGroupOf, N));
    HIP_TRY(c, mem.alloc(&dCBounds, 6));
    HIP_TRY(c, mem.alloc(&dPrimBox, N));
    HIP_TRY(c, mem.alloc(&dKeys, N));
    HIP_TRY(c, mem.alloc(&dKeysSorted, N));
    HIP_TRY(c, mem.alloc(&dVals, N));
    HIP_TRY(c, mem.alloc(&dOrder, N));
    HIP_TRY(c, mem.alloc(&dLeft, N));
    HIP_TRY(c, mem.alloc(&dRight, N));
    HIP_TRY(c, mem.alloc(&dLo, N));
    HIP_TRY(c, mem.alloc(&dHi, N));
    HIP_TRY(c, mem.alloc(&dParentInt, N));
    HIP_TRY(c, mem.alloc(&dParentLeaf, N));
    HIP_TRY(c, mem.alloc(&dGroupRoot, 1));
    HIP_TRY(c, mem.alloc(&dFlag, N));
    HIP_TRY(c, mem.alloc(&dScan, N));
    HIP_TRY(c, mem.alloc(&dOut, N));
    const int bitsPerAxis = 21;
    size_t sortBytes = 0, scanBytes = 0;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, sortBytes, dKeys, dKeysSorted, dVals, dOrder, int(N), 0, 3 * bitsPerAxis, stream));
    HIP_TRY(c, hipcub::DeviceScan::InclusiveSum(nullptr, scanBytes, dFlag, dScan, int(N), stream));
    unsigned char* dTemp = nullptr;
    HIP_TRY(c, mem.alloc(&dTemp, std::max(sortBytes, scanBytes)));
    const unsigned long long offsets[2] = { 0ull, N };
    const uint32_t boundsInit[6] = { 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u };
    HIP_TRY(c, hipMemcpyAsync(dPos, positions, 3 * n_vertices * sizeof(float), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dIdx, indices, 3 * n_triangles * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dOffsets, offsets, sizeof(offsets), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipMemcpyAsync(dCBounds, boundsInit, sizeof(boundsInit), hipMemcpyHostToDevice, stream));
    HIP_TRY(c, hipStreamSynchronize(stream)); // (the two small arrays live on this frame)
    HIP_TRY(c, hipMemsetAsync(dParentInt, 0xFF, size_t(N) * 4, stream));
    HIP_TRY(c, hipMemsetAsync(dParentLeaf, 0xFF, size_t(N) * 4, stream));
    HIP_TRY(c, hipMemsetAsync(dFlag, 0, size_t(N) * 4, stream));
    HIP_TRY(c, hipEventRecord(c->ev0, stream));
    hipLaunchKernelGGL(primBoundsKernel, dim3(blocksFor(N)), dim3(256), 0, stream, dPos, dIdx, static_cast<const uint32_t*>(nullptr), dOffsets, 1u, N, dGroupOf, dPrimBox, dCBounds);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(mortonKernel, dim3(blocksFor(N)), dim3(256), 0, stream, dPrimBox, dGroupOf, dCBounds, N, bitsPerAxis, dKeys, dVals);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(dTemp, sortBytes, dKeys, dKeysSorted, dVals, dOrder, int(N), 0, 3 * bitsPerAxis, stream));
    if (N > 1u) {
        hipLaunchKernelGGL(hierarchyKernel, dim3(blocksFor(N - 1u)), dim3(256), 0, stream, dKeysSorted, N, dGroupOf, dOffsets, dLeft, dRight, dLo, dHi, dParentInt, dParentLeaf,
            dGroupRoot);
        HIP_TRY(c, hipGetLastError());
    }
    hipLaunchKernelGGL(cutKernel, dim3(blocksFor(N)), dim3(256), 0, stream, N, maxPrims, dLo, dHi, dParentInt, dParentLeaf, dFlag);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipcub::DeviceScan::InclusiveSum(dTemp, scanBytes, dFlag, dScan, int(N), stream));
    hipLaunchKernelGGL(groupOfKernel, dim3(blocksFor(N)), dim3(256), 0, stream, N, dOrder, dScan, dOut);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    uint32_t groups = 0;
    HIP_TRY(c, hipMemcpyAsync(group_of_prim, dOut, size_t(N) * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipMemcpyAsync(&groups, dScan + (N - 1u), 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(c, hipStreamSynchronize(stream));
    HIP_TRY(c, hipEventElapsedTime(&c->lastKernelMs, c->ev0, c->ev1));
    c->timingStale = false;
    if (groups == 0 || groups > N)
        return fail(c, PGPU_ERR_HIP, "pgpu_partition_prims_device: the cut is not a partition (internal error)");
    *n_groups = groups;
    return PGPU_OK;
}
