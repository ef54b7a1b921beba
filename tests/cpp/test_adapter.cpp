// Exercises the C++ host adapter the way an integrator uses the reference's interface:
// per-ray intersect()/intersectAny() from several threads, results through the four handlers, bounces through flush().
//   test_adapter                      two bounces + shadow rays on one GPU
//   test_adapter multi                one context vs two contexts (replicated) vs two contexts (batching points sharded,
//                                     exchange through host memory) on GPU 0: identical results per ray
//   test_adapter throughput [n] [t]   t threads (default 16) push n rays (default 1e7); prints rays/s at the boundary
// Exit code 0 = pass, 2 = no GPU (reported cleanly), 1 = failure.
#include "pandora_amd/gpu_acceleration_structure.hpp"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

using namespace pandora_amd;

struct BounceState {
    int pixel;
    int depth;
};
struct ShadowState {
    int pixel;
};

static TriangleShape quad(float z, float half, float cx = 0.0f, float cy = 0.0f)
{
    TriangleShape s;
    s.positions = { cx - half, cy - half, z, cx + half, cy - half, z, cx + half, cy + half, z, cx - half, cy + half, z };
    s.indices = { 0, 1, 2, 0, 2, 3 };
    return s;
}

// a bumpy height field over [-1, 1]^2 around z = -2: res x res cells, two triangles each
static TriangleShape terrain(int res)
{
    TriangleShape s;
    for (int y = 0; y <= res; ++y)
        for (int x = 0; x <= res; ++x) {
            const float fx = -1.0f + 2.0f * float(x) / float(res), fy = -1.0f + 2.0f * float(y) / float(res);
            s.positions.insert(s.positions.end(), { fx, fy, -2.0f + 0.15f * std::sin(9.0f * fx) * std::cos(7.0f * fy) });
        }
    for (int y = 0; y < res; ++y)
        for (int x = 0; x < res; ++x) {
            const uint32_t a = uint32_t(y * (res + 1) + x), b = a + 1, c = a + uint32_t(res + 1), d = c + 1;
            s.indices.insert(s.indices.end(), { a, b, d, a, d, c });
        }
    return s;
}

static uint32_t lcg(uint32_t& s)
{
    s = s * 1664525u + 1013904223u;
    return s >> 8;
}

static Ray cameraRay(int i)
{
    uint32_t s = uint32_t(i) * 2654435761u + 12345u;
    Ray r;
    r.origin[0] = -0.95f + 1.9f * float(lcg(s)) / 16777216.0f;
    r.origin[1] = -0.95f + 1.9f * float(lcg(s)) / 16777216.0f;
    r.origin[2] = 1.0f;
    r.direction[0] = 0.3f * (float(lcg(s)) / 16777216.0f - 0.5f);
    r.direction[1] = 0.3f * (float(lcg(s)) / 16777216.0f - 0.5f);
    r.direction[2] = (i % 10 == 9) ? 1.0f : -1.0f; // every tenth ray looks away: miss
    return r;
}

static int basic()
{
    std::vector<TriangleShape> shapes;
    for (int i = 0; i < 8; ++i)
        shapes.push_back(quad(-1.0f - float(i), 1.0f + 0.25f * float(i)));
    std::atomic<int> hits { 0 }, misses { 0 }, anyHits { 0 }, anyMisses { 0 }, bad { 0 };
    GpuAccelerationStructureBuilder builder(&shapes, 4, 16);
    using Accel = GpuAccelerationStructure<BounceState, ShadowState>;
    Accel* self = nullptr;
    Accel accel = builder.build<BounceState, ShadowState>(
        [&](const Ray& r, const RayHit& h, const BounceState& s) {
            ++hits;
            // rays start at z = +1 looking down -z: nearest quad is shape 0 at z = -1, t = 2
            if (s.depth == 0 && (h.shapeID != 0 || std::fabs(r.tfar - 2.0f) > 1e-6f || h.geometricNormal[2] <= 0))
                ++bad;
            if (s.depth == 0) { // spawn a shadow ray back towards the origin side and one bounce (from a pool thread)
                Ray sh;
                sh.origin[0] = r.origin[0];
                sh.origin[1] = r.origin[1];
                sh.origin[2] = -0.999f;
                sh.direction[2] = 1.0f;
                self->intersectAny(sh, ShadowState { s.pixel });
                Ray b = sh;
                b.origin[2] = -1.001f;
                b.direction[2] = -1.0f;
                self->intersect(b, BounceState { s.pixel, 1 });
            }
            if (s.depth == 1 && (h.shapeID != 1 || std::fabs(r.tfar - 0.999f) > 1e-4f))
                ++bad;
        },
        [&](const Ray&, const BounceState&) { ++misses; },
        [&](const Ray& r, const ShadowState&) { ++anyHits; if (!(r.tfar < 0)) ++bad; },
        [&](const Ray&, const ShadowState&) { ++anyMisses; });
    self = &accel;
    if (accel.numBatchingPoints() < 2) {
        std::printf("expected several batching points\n");
        return 1;
    }
    const int n = 1000;
    std::vector<std::thread> pool;
    for (int t = 0; t < 4; ++t)
        pool.emplace_back([&, t] {
            for (int i = t; i < n; i += 4) {
                Ray r;
                r.origin[0] = -0.9f + 1.8f * float(i % 32) / 31.0f;
                r.origin[1] = -0.9f + 1.8f * float(i / 32) / 31.0f;
                r.origin[2] = 1.0f;
                r.direction[2] = (i % 10 == 9) ? 1.0f : -1.0f; // every tenth ray looks away: miss
                accel.intersect(r, BounceState { i, 0 });
            }
        });
    for (auto& th : pool)
        th.join();
    size_t traced = 0;
    while (accel.pending())
        traced += accel.flush();
    std::printf("traced %zu rays: hits %d misses %d anyHits %d anyMisses %d bad %d, svdag tests %llu culled %llu\n", traced, hits.load(),
        misses.load(), anyHits.load(), anyMisses.load(), bad.load(), (unsigned long long)accel.counters().svdag_tests,
        (unsigned long long)accel.counters().svdag_culled);
    const int primaryHits = n - n / 10;
    if (hits != 2 * primaryHits || misses != n / 10 || anyMisses != primaryHits || anyHits != 0 || bad != 0 || traced != size_t(n + 2 * primaryHits))
        return 1;
    // The same rays again, as arrays in (intersectBatch / intersectAnyBatch) and as runs out (setBatchHandlers): per ray the batch
    // view must say what the per-ray handlers said -- hit or miss, shape 0 at t = 2 with the normal facing the ray, states intact.
    std::atomic<int> bHits { 0 }, bMisses { 0 }, bBad { 0 }, bAnyHits { 0 }, bAnyMisses { 0 }, runs { 0 };
    accel.setBatchHandlers(
        [&](const RayBatch<BounceState>& b) {
            ++runs;
            for (size_t i = 0; i < b.n; ++i) {
                const Ray r = b.ray(i);
                const bool expectMiss = b.states[i].pixel % 10 == 9;
                if (b.isHit(i) == expectMiss || b.rays != nullptr) {
                    ++bBad;
                    continue;
                }
                if (!b.isHit(i)) {
                    ++bMisses;
                    continue;
                }
                ++bHits;
                const RayHit h = b.hit(i);
                if (h.shapeID != 0 || std::fabs(r.tfar - 2.0f) > 1e-6f || std::fabs(b.t[i] - 2.0f) > 1e-6f || h.geometricNormal[2] <= 0 || r.numTopLevelIntersections == 0)
                    ++bBad;
            }
        },
        [&](const RayBatch<ShadowState>& b) {
            for (size_t i = 0; i < b.n; ++i) {
                const Ray r = b.ray(i);
                if (b.isHit(i)) {
                    ++bAnyHits;
                    if (!(r.tfar < 0))
                        ++bBad;
                } else {
                    ++bAnyMisses;
                }
            }
        });
    {
        std::vector<float> a(8 * size_t(n));
        std::vector<BounceState> st;
        std::vector<ShadowState> sst;
        for (int i = 0; i < n; ++i) {
            a[size_t(i)] = -0.9f + 1.8f * float(i % 32) / 31.0f;
            a[size_t(n + i)] = -0.9f + 1.8f * float(i / 32) / 31.0f;
            a[size_t(2 * n + i)] = 1.0f;
            a[size_t(3 * n + i)] = 0.0f;
            a[size_t(4 * n + i)] = 0.0f;
            a[size_t(5 * n + i)] = (i % 10 == 9) ? 1.0f : -1.0f;
            a[size_t(6 * n + i)] = 0.0f;
            a[size_t(7 * n + i)] = std::numeric_limits<float>::max();
            st.push_back(BounceState { i, 0 });
            sst.push_back(ShadowState { i });
        }
        const pgpu_rays soa { a.data(), a.data() + n, a.data() + 2 * n, a.data() + 3 * n, a.data() + 4 * n, a.data() + 5 * n, a.data() + 6 * n, a.data() + 7 * n };
        // in two pieces from two threads, plus the same rays as shadow rays (every ray that looks down is occluded by quad 0)
        std::thread t1([&] { accel.intersectBatch(soa, 600, st.data()); });
        t1.join();
        const pgpu_rays rest { soa.ox + 600, soa.oy + 600, soa.oz + 600, soa.dx + 600, soa.dy + 600, soa.dz + 600, soa.tnear + 600, soa.tfar + 600 };
        accel.intersectBatch(rest, size_t(n) - 600, st.data() + 600);
        accel.intersectAnyBatch(soa, size_t(n), sst.data());
        if (accel.pending() != size_t(2 * n) || accel.flush() != size_t(2 * n))
            return 1;
    }
    std::printf("batch interface: hits %d misses %d anyHits %d anyMisses %d bad %d in %d runs\n", bHits.load(), bMisses.load(), bAnyHits.load(), bAnyMisses.load(),
        bBad.load(), runs.load());
    if (bHits != primaryHits || bMisses != n / 10 || bAnyHits != primaryHits || bAnyMisses != n / 10 || bBad != 0 || runs < 2)
        return 1;
    std::printf("adapter ok\n");
    return 0;
}

// per-ray outcome of one run, indexed by the caller's pixel
struct Outcome {
    std::vector<uint32_t> shape, prim, tBits, visits, shadow;
    std::vector<float> nz;
    pgpu_counters counters {};
    size_t traced = 0;
};

static Outcome runScene(const std::vector<TriangleShape>& shapes, const AdapterOptions& opt, int n, unsigned pushThreads)
{
    Outcome out;
    out.shape.assign(size_t(n), 0xFFFFFFFFu);
    out.prim.assign(size_t(n), 0xFFFFFFFFu);
    out.tBits.assign(size_t(n), 0u);
    out.visits.assign(size_t(n), 0u);
    out.shadow.assign(size_t(n), 2u);
    out.nz.assign(size_t(n), 0.0f);
    GpuAccelerationStructureBuilder builder(&shapes, 1024, 16, opt);
    using Accel = GpuAccelerationStructure<BounceState, ShadowState>;
    Accel* self = nullptr;
    Accel accel = builder.build<BounceState, ShadowState>(
        [&](const Ray& r, const RayHit& h, const BounceState& s) {
            const size_t p = size_t(s.pixel);
            out.shape[p] = h.shapeID;
            out.prim[p] = h.primitiveID;
            std::memcpy(&out.tBits[p], &r.tfar, 4);
            out.visits[p] = uint32_t(r.numTopLevelIntersections);
            out.nz[p] = h.geometricNormal[2];
            Ray sh; // shadow ray from the hit point towards a light above the scene
            for (int a = 0; a < 3; ++a)
                sh.origin[a] = r.origin[a] + r.tfar * r.direction[a];
            sh.origin[2] += 1e-3f;
            sh.direction[0] = 0.2f;
            sh.direction[1] = 0.1f;
            sh.direction[2] = 1.0f;
            self->intersectAny(sh, ShadowState { s.pixel });
        },
        [&](const Ray& r, const BounceState& s) { out.visits[size_t(s.pixel)] = uint32_t(r.numTopLevelIntersections); },
        [&](const Ray&, const ShadowState& s) { out.shadow[size_t(s.pixel)] = 1u; },
        [&](const Ray&, const ShadowState& s) { out.shadow[size_t(s.pixel)] = 0u; });
    self = &accel;
    std::vector<std::thread> pushers;
    for (unsigned t = 0; t < pushThreads; ++t)
        pushers.emplace_back([&, t] {
            for (int i = int(t); i < n; i += int(pushThreads))
                accel.intersect(cameraRay(i), BounceState { i, 0 });
        });
    for (auto& th : pushers)
        th.join();
    while (accel.pending()) {
        out.traced += accel.flush();
        const pgpu_counters& c = accel.counters();
        out.counters.svdag_tests += c.svdag_tests;
        out.counters.svdag_culled += c.svdag_culled;
        out.counters.leaf_visits += c.leaf_visits;
        out.counters.records_routed += c.records_routed;
    }
    return out;
}

static int multi()
{
    std::vector<TriangleShape> shapes;
    shapes.push_back(terrain(96)); // 18432 triangles: 18+ batching points of <= 1024
    for (int i = 0; i < 6; ++i)
        shapes.push_back(quad(-0.5f - 0.2f * float(i), 0.15f, -0.6f + 0.25f * float(i), 0.3f - 0.1f * float(i)));
    const int n = 200000;
    AdapterOptions one;
    one.batchRays = 65536; // several batches per flush: the pipeline's stages get reused
    const Outcome a = runScene(shapes, one, n, 4);
    AdapterOptions two = one;
    two.devices = { 0, 0 };
    const Outcome b = runScene(shapes, two, n, 4);
    AdapterOptions sharded = two;
    sharded.multiGpu = MultiGpu::LeafSharded;
    sharded.hostStagedExchange = true;
    const Outcome c = runScene(shapes, sharded, n, 4);
    AdapterOptions rccl = one; // one rank, but through pgpu_comm_init (RCCL) + pgpu_trace_sharded
    rccl.multiGpu = MultiGpu::LeafSharded;
    const Outcome r = runScene(shapes, rccl, n, 4);
    AdapterOptions streamed = one; // batching points in pinned host memory, streamed through a small HBM cache (botLevelBVHCacheSize)
    streamed.botLevelBVHCacheBytes = size_t(1) << 20;
    const Outcome st = runScene(shapes, streamed, n, 4);
    AdapterOptions streamedTwo = two;
    streamedTwo.botLevelBVHCacheBytes = size_t(3) << 19;
    const Outcome st2 = runScene(shapes, streamedTwo, n, 4);
    size_t hits = 0, shadowed = 0;
    for (int i = 0; i < n; ++i) {
        hits += a.shape[size_t(i)] != 0xFFFFFFFFu;
        shadowed += a.shadow[size_t(i)] == 1u;
    }
    auto same = [&](const Outcome& x, const char* name) {
        const bool ok = x.shape == a.shape && x.prim == a.prim && x.tBits == a.tBits && x.visits == a.visits && x.shadow == a.shadow && x.nz == a.nz
            && x.traced == a.traced && x.counters.svdag_tests == a.counters.svdag_tests && x.counters.svdag_culled == a.counters.svdag_culled
            && x.counters.leaf_visits == a.counters.leaf_visits;
        std::printf("%s: %s (traced %zu, svdag tests %llu, leaf visits %llu, records routed %llu)\n", name, ok ? "identical" : "DIFFERENT", x.traced,
            (unsigned long long)x.counters.svdag_tests, (unsigned long long)x.counters.leaf_visits, (unsigned long long)x.counters.records_routed);
        return ok;
    };
    std::printf("one context: %zu of %d rays hit, %zu shadowed, traced %zu\n", hits, n, shadowed, a.traced);
    const bool okB = same(b, "two contexts, scene replicated");
    const bool okC = same(c, "two contexts, batching points sharded");
    const bool okR = same(r, "one context, sharded loop over an RCCL communicator of one rank");
    const bool okS = same(st, "one context, batching points streamed through a 1 MiB cache");
    const bool okS2 = same(st2, "two contexts, each streaming through a 1.5 MiB cache");
    bool refused = false;
    try {
        AdapterOptions bad = sharded;
        bad.botLevelBVHCacheBytes = size_t(1) << 20;
        GpuAccelerationStructureBuilder b2(&shapes, 1024, 16, bad);
    } catch (const std::exception&) {
        refused = true;
    }
    // SVDAG early-out only for non-resident points: with everything resident no test runs; the hits stay what they were
    AdapterOptions direct = one;
    direct.svdagOnlyForNonResident = true;
    const Outcome dr = runScene(shapes, direct, n, 4);
    const bool okD = dr.shape == a.shape && dr.prim == a.prim && dr.tBits == a.tBits && dr.shadow == a.shadow && dr.nz == a.nz && dr.traced == a.traced
        && dr.counters.svdag_tests == 0 && dr.counters.leaf_visits >= a.counters.leaf_visits;
    std::printf("one context, SVDAG test for non-resident points only: %s (svdag tests %llu, leaf visits %llu)\n", okD ? "same hits" : "DIFFERENT",
        (unsigned long long)dr.counters.svdag_tests, (unsigned long long)dr.counters.leaf_visits);
    AdapterOptions directStreamed = streamed; // ... and streamed points keep their test: everything as in the reference
    directStreamed.svdagOnlyForNonResident = true;
    const bool okDS = same(runScene(shapes, directStreamed, n, 4), "one context, streamed, SVDAG test for non-resident points only");
    // the batching points' BVH8s built on the GPU: other trees, the same scene -- same batching points, same visits at the top
    // level, the same hits but for rays that meet two triangles a last bit apart (which one a traversal keeps depends on the tree)
    AdapterOptions onDevice = one;
    onDevice.buildOnDevice = true;
    const Outcome dv = runScene(shapes, onDevice, n, 4);
    size_t differ = 0;
    for (int i = 0; i < n; ++i)
        differ += dv.shape[size_t(i)] != a.shape[size_t(i)] || dv.prim[size_t(i)] != a.prim[size_t(i)] || dv.tBits[size_t(i)] != a.tBits[size_t(i)]
            || dv.shadow[size_t(i)] != a.shadow[size_t(i)];
    const bool okDv = dv.traced == a.traced && differ <= size_t(n) / 2000 && dv.counters.svdag_tests == a.counters.svdag_tests;
    std::printf("one context, BVH8s built on the device: %s (%zu of %d rays differ, traced %zu)\n", okDv ? "same scene" : "DIFFERENT", differ, n, dv.traced);
    // the device-built trees are the same wherever they end up: built and kept by each of two contexts, or returned to the host
    // and streamed through a cache, or sharded over two contexts -- identical to the single context that built and kept them
    auto sameAsDv = [&](const Outcome& x, const char* name) {
        const bool ok = x.shape == dv.shape && x.prim == dv.prim && x.tBits == dv.tBits && x.visits == dv.visits && x.shadow == dv.shadow && x.nz == dv.nz
            && x.traced == dv.traced && x.counters.svdag_tests == dv.counters.svdag_tests && x.counters.leaf_visits == dv.counters.leaf_visits;
        std::printf("%s: %s\n", name, ok ? "identical" : "DIFFERENT");
        return ok;
    };
    AdapterOptions onDeviceTwo = two, onDeviceStreamed = streamed, onDeviceSharded = sharded;
    onDeviceTwo.buildOnDevice = onDeviceStreamed.buildOnDevice = onDeviceSharded.buildOnDevice = true;
    const bool okDv2 = sameAsDv(runScene(shapes, onDeviceTwo, n, 4), "two contexts, each building and keeping its copy of the trees");
    const bool okDv3 = sameAsDv(runScene(shapes, onDeviceStreamed, n, 4), "device-built trees returned to the host and streamed through a 1 MiB cache");
    const bool okDv4 = sameAsDv(runScene(shapes, onDeviceSharded, n, 4), "device-built trees returned to the host and sharded over two contexts");
    if (!okS || !okS2 || !refused || !okD || !okDS || !okDv || !okDv2 || !okDv3 || !okDv4)
        return 1;
    if (hits < size_t(n) / 2 || shadowed == 0 || shadowed == hits || !okB || !okC || !okR || c.counters.records_routed == 0)
        return 1;
    std::printf("adapter multi ok\n");
    return 0;
}

static int throughput(size_t n, unsigned threads)
{
    std::vector<TriangleShape> shapes;
    shapes.push_back(terrain(512)); // 524288 triangles
    AdapterOptions opt;
    GpuAccelerationStructureBuilder builder(&shapes, 32768, 32, opt);
    using Accel = GpuAccelerationStructure<BounceState, ShadowState>;
    struct alignas(64) Cell {
        std::atomic<uint64_t> v { 0 };
    };
    std::vector<Cell> hitCells(64), missCells(64);
    const auto t0 = std::chrono::steady_clock::now();
    Accel accel = builder.build<BounceState, ShadowState>(
        [&](const Ray&, const RayHit&, const BounceState& s) { hitCells[size_t(s.pixel) & 63].v.fetch_add(1, std::memory_order_relaxed); },
        [&](const Ray&, const BounceState& s) { missCells[size_t(s.pixel) & 63].v.fetch_add(1, std::memory_order_relaxed); },
        [&](const Ray&, const ShadowState&) {}, [&](const Ray&, const ShadowState&) {});
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<std::thread> pushers;
    for (unsigned t = 0; t < threads; ++t)
        pushers.emplace_back([&, t] {
            const size_t lo = n * t / threads, hi = n * (t + 1) / threads;
            for (size_t i = lo; i < hi; ++i)
                accel.intersect(cameraRay(int(i)), BounceState { int(i), 0 });
        });
    for (auto& th : pushers)
        th.join();
    const auto t2 = std::chrono::steady_clock::now();
    const size_t traced = accel.flush();
    const auto t3 = std::chrono::steady_clock::now();
    uint64_t hits = 0, misses = 0;
    for (size_t i = 0; i < 64; ++i) {
        hits += hitCells[i].v.load();
        misses += missCells[i].v.load();
    }
    // the same rays with results delivered in runs (one handler call per <= 8192 rays; nothing computed per ray unless asked for) ...
    std::atomic<uint64_t> bHits { 0 }, bMisses { 0 };
    accel.setBatchHandlers(
        [&](const RayBatch<BounceState>& b) {
            uint64_t h = 0;
            for (size_t i = 0; i < b.n; ++i)
                h += b.isHit(i) ? 1u : 0u;
            bHits.fetch_add(h, std::memory_order_relaxed);
            bMisses.fetch_add(b.n - h, std::memory_order_relaxed);
        },
        [&](const RayBatch<ShadowState>&) {});
    auto pushAll = [&] {
        std::vector<std::thread> ps;
        for (unsigned t = 0; t < threads; ++t)
            ps.emplace_back([&, t] {
                const size_t lo = n * t / threads, hi = n * (t + 1) / threads;
                for (size_t i = lo; i < hi; ++i)
                    accel.intersect(cameraRay(int(i)), BounceState { int(i), 0 });
            });
        for (auto& th : ps)
            th.join();
    };
    pushAll();
    const auto t4 = std::chrono::steady_clock::now();
    const size_t tracedBatch = accel.flush();
    const auto t5 = std::chrono::steady_clock::now();
    // ... and handed over as arrays too (intersectBatch: what a wavefront integrator holds anyway)
    std::vector<std::vector<float>> soaOf(threads);
    std::vector<std::vector<BounceState>> statesOf(threads);
    {
        std::vector<std::thread> ps;
        for (unsigned t = 0; t < threads; ++t)
            ps.emplace_back([&, t] {
                const size_t lo = n * t / threads, hi = n * (t + 1) / threads, m = hi - lo;
                soaOf[t].resize(8 * m);
                for (size_t i = lo; i < hi; ++i) {
                    const Ray r = cameraRay(int(i));
                    float* a = soaOf[t].data() + (i - lo);
                    a[0] = r.origin[0], a[m] = r.origin[1], a[2 * m] = r.origin[2], a[3 * m] = r.direction[0], a[4 * m] = r.direction[1], a[5 * m] = r.direction[2];
                    a[6 * m] = r.tnear, a[7 * m] = r.tfar;
                    statesOf[t].push_back(BounceState { int(i), 0 });
                }
            });
        for (auto& th : ps)
            th.join();
    }
    const auto t6 = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> ps;
        for (unsigned t = 0; t < threads; ++t)
            ps.emplace_back([&, t] {
                const size_t m = statesOf[t].size();
                const float* a = soaOf[t].data();
                accel.intersectBatch(pgpu_rays { a, a + m, a + 2 * m, a + 3 * m, a + 4 * m, a + 5 * m, a + 6 * m, a + 7 * m }, m, statesOf[t].data());
            });
        for (auto& th : ps)
            th.join();
    }
    const auto t7 = std::chrono::steady_clock::now();
    const size_t tracedSoa = accel.flush();
    const auto t8 = std::chrono::steady_clock::now();
    const double build = std::chrono::duration<double>(t1 - t0).count(), push = std::chrono::duration<double>(t2 - t1).count(),
                 flush = std::chrono::duration<double>(t3 - t2).count(), flushBatch = std::chrono::duration<double>(t5 - t4).count(),
                 pushSoa = std::chrono::duration<double>(t7 - t6).count(), flushSoa = std::chrono::duration<double>(t8 - t7).count();
    std::printf("{\"adapter_throughput\": {\"rays\": %zu, \"push_threads\": %u, \"host_threads\": %u, \"batching_points\": %zu, \"build_s\": %.3f, "
                "\"push_Mrays_s\": %.1f, \"flush_Mrays_s\": %.1f, \"end_to_end_Mrays_s\": %.1f, \"flush_batch_handlers_Mrays_s\": %.1f, "
                "\"push_arrays_Mrays_s\": %.1f, \"flush_arrays_batch_handlers_Mrays_s\": %.1f, \"end_to_end_arrays_Mrays_s\": %.1f, \"hits\": %llu, \"misses\": %llu}}\n",
        traced, threads, accel.hostThreads(), accel.numBatchingPoints(), build, double(n) / push * 1e-6, double(n) / flush * 1e-6,
        double(n) / (push + flush) * 1e-6, double(n) / flushBatch * 1e-6, double(n) / pushSoa * 1e-6, double(n) / flushSoa * 1e-6,
        double(n) / (pushSoa + flushSoa) * 1e-6, (unsigned long long)hits, (unsigned long long)misses);
    if (traced != n || hits + misses != n || hits == 0 || misses < n / 10)
        return 1;
    if (tracedBatch != n || tracedSoa != n || bHits.load() != 2 * hits || bMisses.load() != 2 * misses) {
        std::printf("batch handlers disagree with the per-ray ones: %llu / %llu hits\n", (unsigned long long)bHits.load(), (unsigned long long)(2 * hits));
        return 1;
    }
    std::printf("adapter throughput ok\n");
    return 0;
}

int main(int argc, char** argv)
{
    try {
        if (argc > 1 && std::strcmp(argv[1], "multi") == 0)
            return multi();
        if (argc > 1 && std::strcmp(argv[1], "throughput") == 0)
            return throughput(argc > 2 ? size_t(std::atof(argv[2])) : size_t(10000000), argc > 3 ? unsigned(std::atoi(argv[3])) : 16u);
        return basic();
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return std::string(e.what()).find("pgpu_init") != std::string::npos ? 2 : 1;
    }
}
