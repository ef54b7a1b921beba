// Header-only host adapter over the C ABI (include/pandora_hip.h). It keeps the shape of the reference's
// interface so that integrator code written against
//     pandora::AccelerationStructure<HitRayState, AnyHitRayState>          (traversal/acceleration_structure.h:6-11)
//     pandora::OfflineBatchingAccelerationStructureBuilder::build(...)     (offline_batching_acceleration_structure.h:106-126,307-386)
// can switch over: per-ray intersect()/intersectAny() from any thread, results delivered to the four
// hit / miss / anyHit / anyMiss handlers exactly once per ray. Where the reference enqueues into TaskGraph
// task handles and traces when TaskGraph::run() drains the queues (task_graph.cpp:19-56), this adapter
// buffers rays and traces when flush() is called.
//
// Threads. intersect()/intersectAny() append to a buffer that belongs to the calling thread: no lock, no shared cache line
// per ray (the reference's callers are TBB workers, stream/task_graph.h:130-156). flush() is called by one thread while no
// other thread is inside intersect()/intersectAny() -- the same quiescence TaskGraph::run() starts from -- and does its
// host work on a pool of worker threads: AoS -> SoA into pinned staging memory, and the handlers, which therefore run
// concurrently on several threads, as the reference's do, and may call intersect()/intersectAny() again (next bounce:
// traced by the next flush()). The rays go to the GPU in batches; while the GPU traces batch b the pool converts batch
// b + 1 and runs the handlers of batch b - 1.
//
// Several GPUs, one process: AdapterOptions::devices. Replicated (default): every GPU holds the whole scene and traces its
// share of the rays. LeafSharded: the batching points are spread over the GPUs' HBM (BASELINE configs[3]) and
// pgpu_trace_sharded routes parked rays to the owners over RCCL (or, with hostStagedExchange, through host memory --
// for several contexts on one GPU, which RCCL refuses).
//
// On a hit the handler receives the ray with tfar = hit distance, the hit record (shape id, primitive id,
// geometric uv and the geometric normal flipped towards the ray, as triangle_intersect.cpp:455-461) and
// the caller's state. Errors surface as std::runtime_error like the reference's ALWAYS_ASSERT/THROW_ERROR
// (utility/error_handling.h:7-37); the C ABI underneath never throws.
#pragma once
#include "pandora_hip.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace pandora_amd {

struct Ray { // graphics_core/ray.h:14-40
    float origin[3] { 0, 0, 0 };
    float direction[3] { 0, 0, 1 };
    float tnear = 0.0f;
    float tfar = std::numeric_limits<float>::max();
    uint64_t numTopLevelIntersections = 0;
};

struct RayHit { // graphics_core/ray.h:42-51
    float geometricNormal[3] { 0, 0, 0 };
    float geometricUV[2] { 0, 0 };
    uint32_t shapeID = 0; // index into the builder's shape list (stands in for pSceneObject)
    uint32_t primitiveID = 0;
};

struct TriangleShape { // shapes/triangle.h:78-87: indexed triangles
    std::vector<float> positions;  // 3 per vertex
    std::vector<uint32_t> indices; // 3 per triangle
};

enum class MultiGpu {
    Replicated, // every GPU holds the whole scene; the rays are split
    LeafSharded // batching points spread over the GPUs, parked rays routed to the owner (pgpu_trace_sharded)
};

struct AdapterOptions {
    std::vector<int> devices { 0 };
    MultiGpu multiGpu = MultiGpu::Replicated;
    bool hostStagedExchange = false; // LeafSharded: exchange through host memory inside the process instead of RCCL
    unsigned hostThreads = 0;        // worker threads of flush(); 0 = min(16, hardware threads)
    size_t batchRays = size_t(1) << 20; // rays per GPU call
    int triangleTest = PGPU_TRI_MT;
    // The reference builder's botLevelBVHCacheSize (offline_batching_acceleration_structure.h:108-109), in bytes: 0 = every
    // batching point's BVH is resident in HBM; otherwise the BVHs stay in pinned host memory and are streamed through an HBM
    // cache of this size (pgpu_stage_bvh8 / pgpu_set_leaf_cache; it must hold the largest one). Not with LeafSharded.
    size_t botLevelBVHCacheBytes = 0;
    // pgpu_set_svdag_policy(ctx, 1): the SVDAG early-out only for batching points whose geometry is not in the GPU's HBM (streamed
    // from host memory, or owned by another GPU); resident points are visited directly. Same hits; Ray::numTopLevelIntersections
    // and the SVDAG counters differ from the reference's.
    bool svdagOnlyForNonResident = false;
    // The batching points' BVH8s built by the first device's GPU, all in one pass (pgpu_build_bvh8_device), instead of one
    // pgpu_build_bvh8 per point on the host's threads: the scene is ready sooner, a ray pays a few per cent more.
    // With every batching point resident on every device (no cache budget, not LeafSharded) each device builds its own copy and
    // keeps it (pgpu_build_upload_bvh8_device: the trees never exist on the host; the build is deterministic, so the copies are
    // the same); otherwise the trees come back to the host, and staging and the owners' uploads work as with the host builder's.
    bool buildOnDevice = false;
};

inline void check(int rc, pgpu_ctx* ctx, const char* what)
{
    if (rc != PGPU_OK)
        throw std::runtime_error(std::string(what) + ": " + (ctx ? pgpu_last_error(ctx) : "error ") + " (code " + std::to_string(rc) + ")");
}

namespace detail {

    // A fixed set of worker threads. A job is fn(0) .. fn(pieces - 1); the workers draw piece numbers from the job's
    // cursor, jobs are served in the order they were submitted.
    class WorkerPool {
    public:
        struct Job {
            std::function<void(size_t)> fn;
            size_t pieces = 0;
            std::atomic<size_t> next { 0 }, done { 0 };
            std::mutex m;
            std::condition_variable cv;
            std::exception_ptr error;
        };
        using Ticket = std::shared_ptr<Job>;

        explicit WorkerPool(unsigned n)
        {
            for (unsigned i = 0; i < std::max(1u, n); ++i)
                m_threads.emplace_back([this] { run(); });
        }
        ~WorkerPool()
        {
            {
                std::lock_guard<std::mutex> l(m_mutex);
                m_stop = true;
            }
            m_cv.notify_all();
            for (auto& t : m_threads)
                t.join();
        }
        unsigned size() const { return unsigned(m_threads.size()); }

        Ticket submit(size_t pieces, std::function<void(size_t)> fn)
        {
            auto job = std::make_shared<Job>();
            job->fn = std::move(fn);
            job->pieces = pieces;
            if (pieces == 0)
                return job;
            {
                std::lock_guard<std::mutex> l(m_mutex);
                m_queue.push_back(job);
            }
            m_cv.notify_all();
            return job;
        }
        // blocks until every piece has run; rethrows the first exception a piece threw
        static void wait(const Ticket& job)
        {
            if (!job)
                return;
            std::unique_lock<std::mutex> l(job->m);
            job->cv.wait(l, [&] { return job->done.load() >= job->pieces; });
            if (job->error)
                std::rethrow_exception(job->error);
        }

    private:
        void run()
        {
            for (;;) {
                Ticket job;
                {
                    std::unique_lock<std::mutex> l(m_mutex);
                    m_cv.wait(l, [&] { return m_stop || !m_queue.empty(); });
                    if (m_queue.empty())
                        return; // (stopping, and nothing left to do)
                    job = m_queue.front();
                }
                const size_t i = job->next.fetch_add(1);
                if (i >= job->pieces) { // all pieces handed out: the next job's turn
                    std::lock_guard<std::mutex> l(m_mutex);
                    if (!m_queue.empty() && m_queue.front() == job)
                        m_queue.pop_front();
                    continue;
                }
                try {
                    job->fn(i);
                } catch (...) {
                    std::lock_guard<std::mutex> l(job->m);
                    if (!job->error)
                        job->error = std::current_exception();
                }
                if (job->done.fetch_add(1) + 1 == job->pieces) {
                    std::lock_guard<std::mutex> l(job->m);
                    job->cv.notify_all();
                }
            }
        }

        std::vector<std::thread> m_threads;
        std::deque<Ticket> m_queue;
        std::mutex m_mutex;
        std::condition_variable m_cv;
        bool m_stop = false;
    };

    class Barrier { // reusable, for the few device threads of one flush
    public:
        explicit Barrier(unsigned n)
            : m_n(n)
        {
        }
        void arrive()
        {
            std::unique_lock<std::mutex> l(m_m);
            const uint64_t phase = m_phase;
            if (++m_count == m_n) {
                m_count = 0;
                ++m_phase;
                m_cv.notify_all();
            } else {
                m_cv.wait(l, [&] { return m_phase != phase; });
            }
        }

    private:
        unsigned m_n, m_count = 0;
        uint64_t m_phase = 0;
        std::mutex m_m;
        std::condition_variable m_cv;
    };

    // pgpu_transport between the contexts of one process, through host memory: what sharded.HostStagedTransport does between
    // processes. Every rank's device thread enters each collective; buffers are staged in `slot[rank]`.
    struct ThreadExchange {
        explicit ThreadExchange(unsigned world)
            : barrier(world)
            , slot(world)
            , offsets(world)
        {
        }
        Barrier barrier;
        std::vector<std::vector<unsigned char>> slot;
        std::vector<std::vector<uint64_t>> offsets; // exchange: byte offset of the rows for rank d inside slot[rank]
        std::atomic<int> failed { 0 };
    };
    struct ExchangeEnd {
        ThreadExchange* x = nullptr;
        unsigned rank = 0, world = 0;
        pgpu_ctx* ctx = nullptr;

        bool pull(void* host, const void* dev, size_t bytes, void* stream) const
        {
            return pgpu_copy(ctx, host, dev, bytes, stream) == PGPU_OK && pgpu_stream_sync(ctx, stream) == PGPU_OK;
        }
        // every rank leaves through the same two barriers whatever happened, then all report the failure
        int finish(bool ok) const
        {
            if (!ok)
                x->failed.store(1);
            x->barrier.arrive();
            return x->failed.load();
        }
        static int allgather(void* user, const uint32_t* send, uint32_t* recv, size_t n, void* stream)
        {
            const ExchangeEnd& e = *static_cast<ExchangeEnd*>(user);
            e.x->slot[e.rank].resize(4 * n);
            bool ok = e.pull(e.x->slot[e.rank].data(), send, 4 * n, stream);
            e.x->barrier.arrive();
            for (unsigned r = 0; ok && r < e.world; ++r)
                ok = e.x->slot[r].size() == 4 * n && pgpu_copy(e.ctx, recv + size_t(r) * n, e.x->slot[r].data(), 4 * n, stream) == PGPU_OK;
            ok = ok && pgpu_stream_sync(e.ctx, stream) == PGPU_OK;
            return e.finish(ok);
        }
        static int exchange(void* user, const void* send, const uint32_t* sendCounts, void* recv, const uint32_t* recvCounts, size_t rowBytes, void* stream)
        {
            const ExchangeEnd& e = *static_cast<ExchangeEnd*>(user);
            std::vector<uint64_t>& off = e.x->offsets[e.rank];
            off.assign(e.world + 1, 0);
            for (unsigned d = 0; d < e.world; ++d)
                off[d + 1] = off[d] + uint64_t(sendCounts[d]) * rowBytes;
            e.x->slot[e.rank].resize(size_t(off[e.world]));
            bool ok = e.pull(e.x->slot[e.rank].data(), send, size_t(off[e.world]), stream);
            e.x->barrier.arrive();
            size_t at = 0;
            for (unsigned s = 0; ok && s < e.world; ++s) {
                const size_t bytes = size_t(recvCounts[s]) * rowBytes;
                const std::vector<uint64_t>& so = e.x->offsets[s];
                ok = so.size() == e.world + 1 && so[e.rank + 1] - so[e.rank] == bytes
                    && pgpu_copy(e.ctx, static_cast<unsigned char*>(recv) + at, e.x->slot[s].data() + so[e.rank], bytes, stream) == PGPU_OK;
                at += bytes;
            }
            ok = ok && pgpu_stream_sync(e.ctx, stream) == PGPU_OK;
            return e.finish(ok);
        }
        static int allreduce(void* user, unsigned long long* buf, size_t n, void* stream)
        {
            const ExchangeEnd& e = *static_cast<ExchangeEnd*>(user);
            e.x->slot[e.rank].resize(8 * n);
            bool ok = e.pull(e.x->slot[e.rank].data(), buf, 8 * n, stream);
            e.x->barrier.arrive();
            std::vector<unsigned long long> sum(n, 0);
            for (unsigned r = 0; ok && r < e.world; ++r) {
                ok = e.x->slot[r].size() == 8 * n;
                for (size_t i = 0; ok && i < n; ++i) {
                    unsigned long long v;
                    std::memcpy(&v, e.x->slot[r].data() + 8 * i, 8);
                    sum[i] += v;
                }
            }
            ok = ok && pgpu_copy(e.ctx, buf, sum.data(), 8 * n, stream) == PGPU_OK && pgpu_stream_sync(e.ctx, stream) == PGPU_OK;
            return e.finish(ok);
        }
    };

    inline uint64_t nextAdapterId()
    {
        static std::atomic<uint64_t> counter { 1 };
        return counter.fetch_add(1);
    }

} // namespace detail

// What a batch handler gets (GpuAccelerationStructure::setBatchHandlers): a run of rays that were buffered one behind the other --
// by one thread's intersect() calls or by one intersectBatch() -- with their results as arrays. One std::function call per run
// (up to 8192 rays) instead of one per ray, and nothing is computed for a ray the handler does not look at: the per-ray handlers
// are handed a RayHit with the geometric normal of every hit (what the reference's leaf computes for every TENTATIVE hit,
// offline_bvh_cache.cpp:48-49), which costs three random reads into the mesh per ray; here hit(i) does that on request.
template <typename State>
struct RayBatch {
    size_t n = 0;
    const Ray* rays = nullptr;      // the n rays as they were handed to intersect(); nullptr for rays that came through intersectBatch
    const float* soa[8] {};         // ... whose origin / direction / tnear / tfar arrays (n values each) are here instead
    const State* states = nullptr;  // n ray states
    // closest hit (null for an any-hit batch): hit distance (a miss keeps the ray's tfar), leaf-object index (PGPU_MISS = miss),
    // batching point, barycentrics
    const float* t = nullptr;
    const uint32_t* obj = nullptr;
    const uint32_t* leaf = nullptr;
    const float* u = nullptr;
    const float* v = nullptr;
    const uint32_t* occluded = nullptr; // any hit (null for a closest-hit batch): 1 = something lies inside [tnear, tfar]
    const uint32_t* visits = nullptr;   // Ray::numTopLevelIntersections gained by this trace (two-level scenes), else nullptr
    std::function<RayHit(const Ray&, uint32_t, uint32_t, float, float)> const* surfaceFn = nullptr;

    bool isHit(size_t i) const { return obj ? obj[i] != PGPU_MISS : occluded[i] != 0u; }
    Ray ray(size_t i) const // as the per-ray handlers get it: tfar = hit distance (-inf for an occluded shadow ray), visits added
    {
        Ray r;
        if (rays) {
            r = rays[i];
        } else {
            for (int a = 0; a < 3; ++a) {
                r.origin[a] = soa[a][i];
                r.direction[a] = soa[3 + a][i];
            }
            r.tnear = soa[6][i];
            r.tfar = soa[7][i];
        }
        if (visits)
            r.numTopLevelIntersections += visits[i];
        if (obj && obj[i] != PGPU_MISS)
            r.tfar = t[i];
        if (occluded && occluded[i])
            r.tfar = -std::numeric_limits<float>::infinity();
        return r;
    }
    RayHit hit(size_t i) const { return (*surfaceFn)(ray(i), leaf[i], obj[i], u[i], v[i]); } // (closest-hit batches, isHit(i) only)
};

template <typename HitRayState, typename AnyHitRayState>
class GpuAccelerationStructure {
public:
    using OnHit = std::function<void(const Ray&, const RayHit&, const HitRayState&)>;
    using OnMiss = std::function<void(const Ray&, const HitRayState&)>;
    using OnAnyHit = std::function<void(const Ray&, const AnyHitRayState&)>;
    using OnAnyMiss = std::function<void(const Ray&, const AnyHitRayState&)>;
    using OnClosestBatch = std::function<void(const RayBatch<HitRayState>&)>;
    using OnAnyBatch = std::function<void(const RayBatch<AnyHitRayState>&)>;

    GpuAccelerationStructure(GpuAccelerationStructure&&) noexcept = default;
    GpuAccelerationStructure(const GpuAccelerationStructure&) = delete;
    ~GpuAccelerationStructure()
    {
        if (!m)
            return;
        m->pool.reset(); // (no handler may outlive the contexts)
        for (Device& d : m->devices)
            d.release();
    }

    // AccelerationStructure::intersect (acceleration_structure.h:8): callable concurrently, lock-free per ray
    void intersect(const Ray& ray, const HitRayState& state) const
    {
        ThreadBuf& b = local();
        b.rays.push_back(ray);
        b.states.push_back(state);
    }
    // AccelerationStructure::intersectAny (acceleration_structure.h:9)
    void intersectAny(const Ray& ray, const AnyHitRayState& state) const
    {
        ThreadBuf& b = local();
        b.anyRays.push_back(ray);
        b.anyStates.push_back(state);
    }

    // The same for callers that hold their rays as arrays already (a wavefront integrator): n rays as eight arrays and n states, copied
    // into the calling thread's buffer as they are -- no AoS detour on the way to the GPU, which wants exactly these arrays.
    void intersectBatch(const pgpu_rays& rays, size_t n, const HitRayState* states) const { local().chunks.emplace_back(new SoaChunk<HitRayState>(rays, n, states)); }
    void intersectAnyBatch(const pgpu_rays& rays, size_t n, const AnyHitRayState* states) const { local().anyChunks.emplace_back(new SoaChunk<AnyHitRayState>(rays, n, states)); }

    // Results in runs instead of one call per ray (see RayBatch). A kind whose batch handler is set no longer calls its per-ray
    // handlers; pass an empty function to go back. Call between flushes.
    void setBatchHandlers(OnClosestBatch closest, OnAnyBatch any)
    {
        m->onClosestBatch = std::move(closest);
        m->onAnyBatch = std::move(any);
    }

    // rays waiting for the next flush() (call it while no thread is inside intersect()/intersectAny())
    size_t pending() const
    {
        std::lock_guard<std::mutex> l(m->registryMutex);
        size_t n = 0;
        for (auto& kv : m->byThread) {
            n += kv.second->rays.size() + kv.second->anyRays.size();
            for (auto& ch : kv.second->chunks)
                n += ch->n;
            for (auto& ch : kv.second->anyChunks)
                n += ch->n;
        }
        return n;
    }

    // Traces everything buffered so far and runs the handlers (on the worker pool, concurrently). Returns the number of
    // rays traced. Rays the handlers add are traced by the next flush().
    size_t flush()
    {
        // take what the threads have buffered; the vectors stay alive (and unmoved) until the handlers have run
        std::vector<std::unique_ptr<ThreadBuf>> taken;
        {
            std::lock_guard<std::mutex> l(m->registryMutex);
            for (auto& kv : m->byThread) {
                ThreadBuf& b = *kv.second;
                if (b.rays.empty() && b.anyRays.empty() && b.chunks.empty() && b.anyChunks.empty())
                    continue;
                taken.emplace_back(new ThreadBuf());
                taken.back()->rays.swap(b.rays);
                taken.back()->states.swap(b.states);
                taken.back()->anyRays.swap(b.anyRays);
                taken.back()->anyStates.swap(b.anyStates);
                taken.back()->chunks.swap(b.chunks);
                taken.back()->anyChunks.swap(b.anyChunks);
            }
        }
        Gathered closest, any;
        for (auto& t : taken) {
            closest.add(t->rays.data(), nullptr, t->rays.size(), t->states.data(), nullptr);
            any.add(t->anyRays.data(), nullptr, t->anyRays.size(), nullptr, t->anyStates.data());
            for (auto& ch : t->chunks)
                closest.add(nullptr, ch->data.data(), ch->n, ch->states.data(), nullptr);
            for (auto& ch : t->anyChunks)
                any.add(nullptr, ch->data.data(), ch->n, nullptr, ch->states.data());
        }
        const size_t total = closest.total() + any.total();
        if (total == 0)
            return 0;

        const size_t nDev = m->devices.size();
        std::vector<pgpu_counters> perDevice(nDev);
        std::vector<std::exception_ptr> errors(nDev);
        auto drive = [&](size_t d) {
            try {
                perDevice[d] = runDevice(d, closest, any);
            } catch (...) {
                errors[d] = std::current_exception();
            }
        };
        if (nDev == 1) {
            drive(0);
        } else {
            std::vector<std::thread> drivers;
            for (size_t d = 0; d < nDev; ++d)
                drivers.emplace_back(drive, d);
            for (auto& t : drivers)
                t.join();
        }
        for (auto& e : errors)
            if (e)
                std::rethrow_exception(e);
        m->counters = pgpu_counters {};
        for (const pgpu_counters& c : perDevice) {
            m->counters.svdag_tests += c.svdag_tests;
            m->counters.svdag_culled += c.svdag_culled;
            m->counters.leaf_visits += c.leaf_visits;
            m->counters.rounds = std::max(m->counters.rounds, c.rounds);
            m->counters.records_routed += c.records_routed;
            m->counters.exchange_bytes += c.exchange_bytes;
        }
        return total;
    }

    const pgpu_counters& counters() const { return m->counters; } // g_stats.svdag.* of the last flush, summed over batches and GPUs
    size_t numBatchingPoints() const { return m->objsOfLeaf.size(); }
    size_t numDevices() const { return m->devices.size(); }
    unsigned hostThreads() const { return m->pool->size(); }

private:
    friend class GpuAccelerationStructureBuilder;

    template <typename State>
    struct SoaChunk { // one intersectBatch call: the eight arrays one behind the other (n values each) and the states
        size_t n;
        std::vector<float> data;
        std::vector<State> states;
        SoaChunk(const pgpu_rays& r, size_t count, const State* st)
            : n(count)
            , data(8 * count)
            , states(st, st + count)
        {
            const float* const src[8] = { r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, r.tnear, r.tfar };
            for (int a = 0; a < 8; ++a)
                std::memcpy(data.data() + size_t(a) * n, src[a], n * sizeof(float));
        }
    };
    struct ThreadBuf {
        std::vector<Ray> rays, anyRays;
        std::vector<HitRayState> states;
        std::vector<AnyHitRayState> anyStates;
        std::vector<std::unique_ptr<SoaChunk<HitRayState>>> chunks;
        std::vector<std::unique_ptr<SoaChunk<AnyHitRayState>>> anyChunks;
    };

    // the buffered rays of one kind, as a list of per-thread segments addressed by one running index
    struct Gathered {
        struct Seg {
            const Ray* rays;  // AoS rays (intersect), or
            const float* soa; // eight arrays of n values one behind the other (intersectBatch)
            const HitRayState* states;
            const AnyHitRayState* anyStates;
            size_t first, n;
            Ray ray(size_t k) const
            {
                if (rays)
                    return rays[k];
                Ray r;
                for (int a = 0; a < 3; ++a) {
                    r.origin[a] = soa[size_t(a) * n + k];
                    r.direction[a] = soa[size_t(3 + a) * n + k];
                }
                r.tnear = soa[6 * n + k];
                r.tfar = soa[7 * n + k];
                return r;
            }
        };
        std::vector<Seg> segs;
        void add(const Ray* rays, const float* soa, size_t n, const HitRayState* s, const AnyHitRayState* as)
        {
            if (n)
                segs.push_back({ rays, soa, s, as, total(), n });
        }
        // calls f(segment, first index inside the segment, count, running index of the first) for the runs that make up [lo, hi)
        template <typename F>
        void eachRun(size_t lo, size_t hi, F&& f) const
        {
            if (lo >= hi)
                return;
            size_t s = size_t(std::upper_bound(segs.begin(), segs.end(), lo, [](size_t v, const Seg& g) { return v < g.first; }) - segs.begin()) - 1;
            for (size_t i = lo; i < hi; ++s) {
                const Seg& g = segs[s];
                const size_t k = i - g.first, count = std::min(g.n - k, hi - i);
                f(g, k, count, i);
                i += count;
            }
        }
        size_t total() const { return segs.empty() ? 0 : segs.back().first + segs.back().n; }
        // calls f(segment, index inside the segment, running index) for the running indices [lo, hi)
        template <typename F>
        void each(size_t lo, size_t hi, F&& f) const
        {
            if (lo >= hi)
                return;
            size_t s = size_t(std::upper_bound(segs.begin(), segs.end(), lo, [](size_t v, const Seg& g) { return v < g.first; }) - segs.begin()) - 1;
            for (size_t i = lo; i < hi; ++s) {
                const Seg& g = segs[s];
                for (size_t k = i - g.first; k < g.n && i < hi; ++k, ++i)
                    f(g, k, i);
            }
        }
    };

    // pinned host + device memory for one batch in flight
    struct Stage {
        float* hostRays = nullptr; // 8 x cap
        float* devRays = nullptr;
        uint32_t* hostOut = nullptr; // t, obj, leaf, u, v, visits, hit: 7 x cap
        uint32_t* devOut = nullptr;
        detail::WorkerPool::Ticket handlers; // of the batch that used this stage last
    };
    struct Device {
        pgpu_ctx* ctx = nullptr;
        void* copyStream = nullptr;
        size_t cap = 0;
        Stage stage[2];
        uint32_t singleBvh = 0; // single-level scenes
        std::unique_ptr<detail::ExchangeEnd> exchangeEnd;

        void reserve(size_t n)
        {
            if (n <= cap)
                return;
            for (Stage& s : stage) {
                free(s);
                check(pgpu_host_alloc(ctx, 8 * n * sizeof(float), reinterpret_cast<void**>(&s.hostRays)), ctx, "pgpu_host_alloc");
                check(pgpu_device_alloc(ctx, 8 * n * sizeof(float), reinterpret_cast<void**>(&s.devRays)), ctx, "pgpu_device_alloc");
                check(pgpu_host_alloc(ctx, 7 * n * sizeof(uint32_t), reinterpret_cast<void**>(&s.hostOut)), ctx, "pgpu_host_alloc");
                check(pgpu_device_alloc(ctx, 7 * n * sizeof(uint32_t), reinterpret_cast<void**>(&s.devOut)), ctx, "pgpu_device_alloc");
            }
            cap = n;
        }
        void free(Stage& s)
        {
            (void)pgpu_host_free(ctx, s.hostRays);
            (void)pgpu_device_free(ctx, s.devRays);
            (void)pgpu_host_free(ctx, s.hostOut);
            (void)pgpu_device_free(ctx, s.devOut);
            s.hostRays = s.devRays = nullptr;
            s.hostOut = s.devOut = nullptr;
        }
        void release()
        {
            if (!ctx)
                return;
            for (Stage& s : stage)
                free(s);
            (void)pgpu_stream_destroy(ctx, copyStream);
            pgpu_shutdown(ctx);
            ctx = nullptr;
        }
    };

    struct Impl {
        uint64_t id = detail::nextAdapterId();
        std::vector<Device> devices;
        bool twoLevel = false, sharded = false;
        int triMode = PGPU_TRI_MT;
        size_t batchRays = size_t(1) << 20;
        std::vector<std::vector<pgpu_leafobj>> objsOfLeaf; // final leaf slot -> leaf-object table
        const std::vector<TriangleShape>* shapes = nullptr;
        OnHit onHit;
        OnMiss onMiss;
        OnAnyHit onAnyHit;
        OnAnyMiss onAnyMiss;
        OnClosestBatch onClosestBatch;
        OnAnyBatch onAnyBatch;
        std::function<RayHit(const Ray&, uint32_t, uint32_t, float, float)> surfaceFn; // RayBatch::hit
        pgpu_counters counters {};
        std::unique_ptr<detail::WorkerPool> pool;
        std::unique_ptr<detail::ThreadExchange> exchange;
        std::mutex registryMutex;
        std::map<std::thread::id, std::unique_ptr<ThreadBuf>> byThread;
    };

    GpuAccelerationStructure()
        : m(new Impl())
    {
        Impl* const impl = m.get(); // (stays valid when the adapter object is moved)
        m->surfaceFn = [impl](const Ray& r, uint32_t leaf, uint32_t obj, float u, float v) { return surfaceOf(*impl, r, leaf, obj, u, v); };
    }

    // The calling thread's buffer. The last adapter a thread used is remembered per thread, so the common case is one
    // comparison; a thread that alternates between adapters pays a map lookup under the registry lock per switch.
    ThreadBuf& local() const
    {
        thread_local uint64_t cachedId = 0;
        thread_local ThreadBuf* cached = nullptr;
        if (cachedId != m->id) {
            std::lock_guard<std::mutex> l(m->registryMutex);
            std::unique_ptr<ThreadBuf>& slot = m->byThread[std::this_thread::get_id()];
            if (!slot)
                slot.reset(new ThreadBuf());
            cached = slot.get();
            cachedId = m->id;
        }
        return *cached;
    }

    static constexpr size_t kPiece = 8192; // rays per unit of host work

    // One GPU's share of a flush: its slice of the closest-hit rays, then of the any-hit rays, in batches through the
    // two stages. In a sharded scene every device makes the same sequence of (collective) calls, empty batches included.
    pgpu_counters runDevice(size_t d, const Gathered& closest, const Gathered& any)
    {
        Device& dev = m->devices[d];
        const size_t nDev = m->devices.size();
        struct Batch {
            int kind;
            size_t lo, hi;
        };
        std::vector<Batch> batches;
        auto cut = [&](int kind, size_t total) {
            const size_t per = (total + nDev - 1) / nDev; // the largest share decides how many calls everybody makes
            const size_t lo = std::min(total, d * per), hi = std::min(total, lo + per);
            const size_t calls = (per + m->batchRays - 1) / m->batchRays;
            for (size_t c = 0; c < calls; ++c) {
                const size_t a = std::min(hi, lo + c * m->batchRays), b = std::min(hi, a + m->batchRays);
                if (a < b || m->sharded)
                    batches.push_back({ kind, a, b });
            }
        };
        cut(PGPU_CLOSEST, closest.total());
        cut(PGPU_ANY, any.total());
        size_t largest = 0;
        for (const Batch& b : batches)
            largest = std::max(largest, b.hi - b.lo);
        dev.reserve(std::max<size_t>(largest, 1));

        pgpu_counters sum {};
        auto prepare = [&](size_t b) { // AoS -> SoA into the stage's pinned memory (pool), then to the device
            const Batch& bt = batches[b];
            Stage& st = dev.stage[b & 1];
            const size_t n = bt.hi - bt.lo;
            const Gathered& src = bt.kind == PGPU_CLOSEST ? closest : any;
            float* const soa = st.hostRays;
            detail::WorkerPool::wait(m->pool->submit((n + kPiece - 1) / kPiece, [&, n, soa](size_t piece) {
                const size_t lo = bt.lo + piece * kPiece, hi = std::min(bt.hi, lo + kPiece);
                src.eachRun(lo, hi, [&](const typename Gathered::Seg& g, size_t k0, size_t count, size_t i0) {
                    const size_t j0 = i0 - bt.lo;
                    if (!g.rays) { // handed over as arrays: eight copies
                        for (size_t a = 0; a < 8; ++a)
                            std::memcpy(soa + a * n + j0, g.soa + a * g.n + k0, count * sizeof(float));
                        return;
                    }
                    for (size_t c = 0; c < count; ++c) {
                        const Ray& r = g.rays[k0 + c];
                        const size_t j = j0 + c;
                        soa[j] = r.origin[0];
                        soa[n + j] = r.origin[1];
                        soa[2 * n + j] = r.origin[2];
                        soa[3 * n + j] = r.direction[0];
                        soa[4 * n + j] = r.direction[1];
                        soa[5 * n + j] = r.direction[2];
                        soa[6 * n + j] = r.tnear;
                        soa[7 * n + j] = r.tfar;
                    }
                });
            }));
            check(pgpu_copy(dev.ctx, st.devRays, st.hostRays, 8 * n * sizeof(float), dev.copyStream), dev.ctx, "pgpu_copy (rays)");
            check(pgpu_stream_sync(dev.ctx, dev.copyStream), dev.ctx, "pgpu_stream_sync");
        };
        auto trace = [&](size_t b) {
            const Batch& bt = batches[b];
            Stage& st = dev.stage[b & 1];
            const size_t n = bt.hi - bt.lo;
            float* const f = st.devRays;
            uint32_t* const o = st.devOut;
            const pgpu_rays pr { f, f + n, f + 2 * n, f + 3 * n, f + 4 * n, f + 5 * n, f + 6 * n, f + 7 * n };
            pgpu_hits ph { reinterpret_cast<float*>(o), o + n, o + 2 * n, reinterpret_cast<float*>(o + 3 * n), reinterpret_cast<float*>(o + 4 * n), o + 5 * n, nullptr };
            uint32_t* const anyOut = o + 6 * n;
            const bool closestKind = bt.kind == PGPU_CLOSEST;
            if (!closestKind)
                ph = pgpu_hits { nullptr, nullptr, nullptr, nullptr, nullptr, o + 5 * n, nullptr };
            pgpu_counters c {};
            if (m->sharded)
                check(pgpu_trace_sharded(dev.ctx, bt.kind, m->triMode, &pr, n, &ph, closestKind ? nullptr : anyOut, &c, nullptr), dev.ctx, "pgpu_trace_sharded");
            else if (n == 0)
                return;
            else if (m->twoLevel)
                check(pgpu_trace(dev.ctx, bt.kind, m->triMode, &pr, n, PGPU_MEM_DEVICE, &ph, closestKind ? nullptr : anyOut, &c, nullptr), dev.ctx, "pgpu_trace");
            else {
                check(pgpu_trace_bvh8(dev.ctx, dev.singleBvh, bt.kind, m->triMode, &pr, n, PGPU_MEM_DEVICE, closestKind ? &ph : nullptr,
                          closestKind ? nullptr : anyOut, nullptr),
                    dev.ctx, "pgpu_trace_bvh8");
                // (single level: every ray meets the one "batching point" once; the kernel keeps no visit counts)
            }
            check(pgpu_sync(dev.ctx, nullptr), dev.ctx, "pgpu_sync");
            sum.svdag_tests += c.svdag_tests;
            sum.svdag_culled += c.svdag_culled;
            sum.leaf_visits += c.leaf_visits;
            sum.rounds = std::max(sum.rounds, c.rounds);
            sum.records_routed += c.records_routed;
            sum.exchange_bytes += c.exchange_bytes;
            if (n) {
                check(pgpu_copy(dev.ctx, st.hostOut, st.devOut, 7 * n * sizeof(uint32_t), dev.copyStream), dev.ctx, "pgpu_copy (results)");
                check(pgpu_stream_sync(dev.ctx, dev.copyStream), dev.ctx, "pgpu_stream_sync");
            }
        };
        auto deliver = [&](size_t b) -> detail::WorkerPool::Ticket { // the handlers of batch b, queued on the pool
            const Batch bt = batches[b];
            const size_t n = bt.hi - bt.lo;
            const uint32_t* const o = dev.stage[b & 1].hostOut;
            const bool visitsValid = m->twoLevel;
            Impl* const impl = m.get();
            auto fillBatch = [impl, bt, n, o, visitsValid](auto& rb, const typename Gathered::Seg& g, size_t k0, size_t count, size_t i0) {
                const size_t j0 = i0 - bt.lo;
                rb.n = count;
                rb.rays = g.rays ? g.rays + k0 : nullptr;
                for (size_t a = 0; a < 8; ++a)
                    rb.soa[a] = g.rays ? nullptr : g.soa + a * g.n + k0;
                rb.visits = visitsValid ? o + 5 * n + j0 : nullptr;
                rb.surfaceFn = &impl->surfaceFn;
            };
            if (bt.kind == PGPU_CLOSEST && m->onClosestBatch)
                return m->pool->submit((n + kPiece - 1) / kPiece, [this, bt, n, o, fillBatch, &closest](size_t piece) {
                    const size_t lo = bt.lo + piece * kPiece, hi = std::min(bt.hi, lo + kPiece);
                    closest.eachRun(lo, hi, [&](const typename Gathered::Seg& g, size_t k0, size_t count, size_t i0) {
                        const size_t j0 = i0 - bt.lo;
                        RayBatch<HitRayState> rb;
                        fillBatch(rb, g, k0, count, i0);
                        rb.states = g.states + k0;
                        rb.t = reinterpret_cast<const float*>(o) + j0;
                        rb.obj = o + n + j0;
                        rb.leaf = o + 2 * n + j0;
                        rb.u = reinterpret_cast<const float*>(o + 3 * n) + j0;
                        rb.v = reinterpret_cast<const float*>(o + 4 * n) + j0;
                        m->onClosestBatch(rb);
                    });
                });
            if (bt.kind == PGPU_ANY && m->onAnyBatch)
                return m->pool->submit((n + kPiece - 1) / kPiece, [this, bt, n, o, fillBatch, &any](size_t piece) {
                    const size_t lo = bt.lo + piece * kPiece, hi = std::min(bt.hi, lo + kPiece);
                    any.eachRun(lo, hi, [&](const typename Gathered::Seg& g, size_t k0, size_t count, size_t i0) {
                        RayBatch<AnyHitRayState> rb;
                        fillBatch(rb, g, k0, count, i0);
                        rb.states = g.anyStates + k0;
                        rb.occluded = o + 6 * n + (i0 - bt.lo);
                        m->onAnyBatch(rb);
                    });
                });
            if (bt.kind == PGPU_CLOSEST)
                return m->pool->submit((n + kPiece - 1) / kPiece, [this, bt, n, o, visitsValid, &closest](size_t piece) {
                    const size_t lo = bt.lo + piece * kPiece, hi = std::min(bt.hi, lo + kPiece);
                    const float* const t = reinterpret_cast<const float*>(o);
                    const float* const u = reinterpret_cast<const float*>(o + 3 * n);
                    const float* const v = reinterpret_cast<const float*>(o + 4 * n);
                    closest.each(lo, hi, [&](const typename Gathered::Seg& g, size_t k, size_t i) {
                        const size_t j = i - bt.lo;
                        Ray r = g.ray(k);
                        if (visitsValid)
                            r.numTopLevelIntersections += o[5 * n + j];
                        if (o[n + j] == PGPU_MISS) {
                            m->onMiss(r, g.states[k]);
                            return;
                        }
                        r.tfar = t[j];
                        m->onHit(r, surface(r, o[2 * n + j], o[n + j], u[j], v[j]), g.states[k]);
                    });
                });
            return m->pool->submit((n + kPiece - 1) / kPiece, [this, bt, n, o, visitsValid, &any](size_t piece) {
                const size_t lo = bt.lo + piece * kPiece, hi = std::min(bt.hi, lo + kPiece);
                any.each(lo, hi, [&](const typename Gathered::Seg& g, size_t k, size_t i) {
                    const size_t j = i - bt.lo;
                    Ray r = g.ray(k);
                    if (visitsValid)
                        r.numTopLevelIntersections += o[5 * n + j];
                    if (o[6 * n + j]) {
                        r.tfar = -std::numeric_limits<float>::infinity(); // wive_bvh8_impl.h:194
                        m->onAnyHit(r, g.anyStates[k]);
                    } else {
                        m->onAnyMiss(r, g.anyStates[k]);
                    }
                });
            });
        };

        // b - 1: handlers (pool) | b: GPU (this thread) | b + 1: conversion + upload (helper thread + pool)
        const size_t nb = batches.size();
        std::exception_ptr failure;
        try {
            if (nb)
                prepare(0);
            for (size_t b = 0; b < nb; ++b) {
                std::future<void> next;
                if (b + 1 < nb)
                    next = std::async(std::launch::async, prepare, b + 1);
                try {
                    Stage& st = dev.stage[b & 1];
                    detail::WorkerPool::wait(st.handlers); // batch b - 2 read its results from this stage
                    st.handlers.reset();
                    trace(b);
                    st.handlers = deliver(b);
                } catch (...) {
                    if (next.valid())
                        next.wait();
                    throw;
                }
                if (next.valid())
                    next.get();
            }
        } catch (...) {
            failure = std::current_exception();
        }
        for (Stage& st : dev.stage) { // the handlers read `closest` / `any` and the stages: never leave before they are done
            try {
                detail::WorkerPool::wait(st.handlers);
            } catch (...) {
                if (!failure)
                    failure = std::current_exception();
            }
            st.handlers.reset();
        }
        if (failure)
            std::rethrow_exception(failure);
        return sum;
    }

    // what the reference's leaf does on every tentative hit (offline_bvh_cache.cpp:48-49), done once here
    RayHit surface(const Ray& r, uint32_t leaf, uint32_t obj, float u, float v) const { return surfaceOf(*m, r, leaf, obj, u, v); }
    static RayHit surfaceOf(const Impl& im, const Ray& r, uint32_t leaf, uint32_t obj, float u, float v)
    {
        const pgpu_leafobj lo = im.objsOfLeaf[leaf][obj];
        const TriangleShape& s = (*im.shapes)[lo.mesh];
        const float* p0 = &s.positions[3 * size_t(s.indices[3 * size_t(lo.prim)])];
        const float* p1 = &s.positions[3 * size_t(s.indices[3 * size_t(lo.prim) + 1])];
        const float* p2 = &s.positions[3 * size_t(s.indices[3 * size_t(lo.prim) + 2])];
        const float e1[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] }, e2[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
        float n[3] = { e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0], e1[0] * e2[1] - e2[0] * e1[1] };
        const float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        for (float& c : n)
            c /= len;
        if (-(n[0] * r.direction[0] + n[1] * r.direction[1] + n[2] * r.direction[2]) < 0)
            for (float& c : n)
                c = -c;
        RayHit h;
        for (int a = 0; a < 3; ++a)
            h.geometricNormal[a] = n[a];
        h.geometricUV[0] = u;
        h.geometricUV[1] = v;
        h.shapeID = lo.mesh;
        h.primitiveID = lo.prim;
        return h;
    }

    std::unique_ptr<Impl> m;
};

// Counterpart of OfflineBatchingAccelerationStructureBuilder (offline_batching...h:106-126): same knobs
// (primitivesPerBatchingPoint, svdagRes; 0 disables the SVDAG early-out as in :323; AdapterOptions::botLevelBVHCacheBytes = its
// botLevelBVHCacheSize). The geometry cache and the task graph of the reference have no role here.
class GpuAccelerationStructureBuilder {
public:
    GpuAccelerationStructureBuilder(const std::vector<TriangleShape>* shapes, unsigned primitivesPerBatchingPoint, unsigned svdagRes,
        int device = 0, int triangleTest = PGPU_TRI_MT)
        : m_shapes(shapes)
        , m_primsPerPoint(primitivesPerBatchingPoint)
        , m_svdagRes(svdagRes)
    {
        m_options.devices = { device };
        m_options.triangleTest = triangleTest;
        validate();
    }
    GpuAccelerationStructureBuilder(const std::vector<TriangleShape>* shapes, unsigned primitivesPerBatchingPoint, unsigned svdagRes, AdapterOptions options)
        : m_shapes(shapes)
        , m_primsPerPoint(primitivesPerBatchingPoint)
        , m_svdagRes(svdagRes)
        , m_options(std::move(options))
    {
        validate();
    }

    template <typename HitRayState, typename AnyHitRayState>
    GpuAccelerationStructure<HitRayState, AnyHitRayState> build(
        typename GpuAccelerationStructure<HitRayState, AnyHitRayState>::OnHit hit,
        typename GpuAccelerationStructure<HitRayState, AnyHitRayState>::OnMiss miss,
        typename GpuAccelerationStructure<HitRayState, AnyHitRayState>::OnAnyHit anyHit,
        typename GpuAccelerationStructure<HitRayState, AnyHitRayState>::OnAnyMiss anyMiss)
    {
        using Accel = GpuAccelerationStructure<HitRayState, AnyHitRayState>;
        Accel a;
        auto& m = *a.m;
        m.shapes = m_shapes;
        m.triMode = m_options.triangleTest;
        m.batchRays = std::max<size_t>(1024, m_options.batchRays);
        m.onHit = std::move(hit);
        m.onMiss = std::move(miss);
        m.onAnyHit = std::move(anyHit);
        m.onAnyMiss = std::move(anyMiss);
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        m.pool.reset(new detail::WorkerPool(m_options.hostThreads ? m_options.hostThreads : std::min(16u, hw)));
        const size_t nDev = m_options.devices.size();
        m.devices.resize(nDev);
        for (size_t d = 0; d < nDev; ++d) {
            check(pgpu_init(m_options.devices[d], &m.devices[d].ctx), nullptr, "pgpu_init");
            check(pgpu_stream_create(m.devices[d].ctx, &m.devices[d].copyStream), m.devices[d].ctx, "pgpu_stream_create");
        }
        pgpu_ctx* const ctx0 = m.devices[0].ctx;

        // one merged vertex/index view for the builders; remember (shape, prim) of every merged triangle
        std::vector<float> pos;
        std::vector<uint32_t> idx;
        std::vector<pgpu_leafobj> triSource;
        for (uint32_t s = 0; s < m_shapes->size(); ++s) {
            const TriangleShape& sh = (*m_shapes)[s];
            for (size_t d = 0; d < nDev; ++d) {
                uint32_t meshId = 0;
                check(pgpu_upload_mesh(m.devices[d].ctx, sh.positions.data(), sh.positions.size() / 3, sh.indices.data(), sh.indices.size() / 3, &meshId), m.devices[d].ctx, "pgpu_upload_mesh");
                if (meshId != s)
                    throw std::runtime_error("GpuAccelerationStructureBuilder: mesh ids are expected to follow the shape order");
            }
            const uint32_t base = uint32_t(pos.size() / 3);
            pos.insert(pos.end(), sh.positions.begin(), sh.positions.end());
            for (size_t i = 0; i < sh.indices.size(); ++i)
                idx.push_back(sh.indices[i] + base);
            for (uint32_t p = 0; p < sh.indices.size() / 3; ++p)
                triSource.push_back({ s, p });
        }
        const size_t nv = pos.size() / 3, nt = idx.size() / 3;
        std::vector<uint32_t> groupOf(nt);
        uint32_t nGroups = 0;
        check(pgpu_partition_prims(pos.data(), nv, idx.data(), nt, m_primsPerPoint, groupOf.data(), &nGroups), ctx0, "pgpu_partition_prims");
        std::vector<std::vector<uint32_t>> groups(nGroups);
        for (uint32_t p = 0; p < nt; ++p)
            groups[groupOf[p]].push_back(p);

        // host side of every batching point on the pool: its BVH8, leaf-object table and bounds
        std::vector<pgpu_host_bvh8*> built(nGroups, nullptr);
        struct FreeBuilt {
            std::vector<pgpu_host_bvh8*>& v;
            ~FreeBuilt()
            {
                for (pgpu_host_bvh8* hb : v)
                    if (hb)
                        pgpu_host_bvh8_free(hb);
            }
        } freeBuilt { built };
        std::vector<std::vector<pgpu_leafobj>> objs(nGroups);
        std::vector<float> bounds(6 * size_t(nGroups));
        // (LeafSharded needs two batching points at least, like everything two-level: decided below, the same way)
        const bool residentBuild = m_options.buildOnDevice && m_options.botLevelBVHCacheBytes == 0 && !(nGroups >= 2 && m_options.multiGpu == MultiGpu::LeafSharded);
        std::vector<std::vector<uint32_t>> residentIds;
        if (m_options.buildOnDevice) {
            std::vector<uint32_t> flat;
            std::vector<uint64_t> offs { 0 };
            flat.reserve(nt);
            for (auto& g : groups) {
                flat.insert(flat.end(), g.begin(), g.end());
                offs.push_back(flat.size());
            }
            if (residentBuild) {
                residentIds.assign(nDev, std::vector<uint32_t>(nGroups, PGPU_NO_BVH));
                for (size_t d = 0; d < nDev; ++d)
                    check(pgpu_build_upload_bvh8_device(m.devices[d].ctx, pos.data(), nv, idx.data(), nt, flat.data(), offs.data(), nGroups, 4, residentIds[d].data(), nullptr),
                        m.devices[d].ctx, "pgpu_build_upload_bvh8_device");
            } else {
                check(pgpu_build_bvh8_device(ctx0, pos.data(), nv, idx.data(), nt, flat.data(), offs.data(), nGroups, 4, built.data(), nullptr), ctx0, "pgpu_build_bvh8_device");
            }
        }
        detail::WorkerPool::wait(m.pool->submit(nGroups, [&](size_t g) {
            if (!m_options.buildOnDevice && pgpu_build_bvh8(pos.data(), nv, idx.data(), nt, groups[g].data(), groups[g].size(), 4, &built[g]) != PGPU_OK)
                throw std::runtime_error("pgpu_build_bvh8 failed for batching point " + std::to_string(g));
            for (uint32_t p : groups[g])
                objs[g].push_back(triSource[p]);
            float* b = &bounds[6 * g];
            for (int c = 0; c < 3; ++c) {
                b[c] = std::numeric_limits<float>::max();
                b[3 + c] = -std::numeric_limits<float>::max();
            }
            for (uint32_t p : groups[g])
                for (int k = 0; k < 3; ++k)
                    for (int c = 0; c < 3; ++c) {
                        const float x = pos[3 * size_t(idx[3 * size_t(p) + size_t(k)]) + size_t(c)];
                        b[c] = std::min(b[c], x);
                        b[3 + c] = std::max(b[3 + c], x);
                    }
        }));

        m.twoLevel = nGroups >= 2;
        // (LeafSharded with one device is allowed: one rank owns every batching point and the whole sharded loop, communicator
        // included, runs -- how the RCCL path is exercised on a one-GPU box)
        m.sharded = m.twoLevel && m_options.multiGpu == MultiGpu::LeafSharded;
        // owner of every batching point (sharded): heaviest first onto the least loaded GPU, as sharded.assign_leaves
        std::vector<uint32_t> ownerOfGroup(nGroups, 0);
        if (m.sharded) {
            std::vector<uint32_t> byWeight(nGroups);
            for (uint32_t g = 0; g < nGroups; ++g)
                byWeight[g] = g;
            std::stable_sort(byWeight.begin(), byWeight.end(), [&](uint32_t x, uint32_t y) { return groups[x].size() > groups[y].size(); });
            std::vector<uint64_t> load(nDev, 0);
            for (uint32_t g : byWeight) {
                const size_t r = size_t(std::min_element(load.begin(), load.end()) - load.begin());
                ownerOfGroup[g] = uint32_t(r);
                load[r] += groups[g].size();
            }
        }
        const bool streamed = m.twoLevel && m_options.botLevelBVHCacheBytes > 0; // (a single batching point is traced single-level: resident)
        std::vector<std::vector<uint32_t>> bvhIds(nDev, std::vector<uint32_t>(nGroups, PGPU_NO_BVH));
        for (size_t d = 0; d < nDev; ++d)
            for (uint32_t g = 0; g < nGroups; ++g) {
                if (residentBuild) {
                    bvhIds[d][g] = residentIds[d][g];
                    continue;
                }
                if (m.sharded && ownerOfGroup[g] != d)
                    continue;
                size_t nNodes = 0, nIdx = 0;
                const void* nodes = pgpu_host_bvh8_nodes(built[g], &nNodes);
                const uint32_t* li = pgpu_host_bvh8_leaf_indices(built[g], &nIdx);
                if (streamed)
                    check(pgpu_stage_bvh8(m.devices[d].ctx, nodes, nNodes, li, nIdx, pgpu_host_bvh8_root(built[g]), objs[g].data(), objs[g].size(), &bvhIds[d][g]),
                        m.devices[d].ctx, "pgpu_stage_bvh8");
                else
                    check(pgpu_upload_bvh8(m.devices[d].ctx, nodes, nNodes, li, nIdx, pgpu_host_bvh8_root(built[g]), objs[g].data(), objs[g].size(), &bvhIds[d][g]),
                        m.devices[d].ctx, "pgpu_upload_bvh8");
            }
        if (streamed)
            for (size_t d = 0; d < nDev; ++d)
                check(pgpu_set_leaf_cache(m.devices[d].ctx, m_options.botLevelBVHCacheBytes), m.devices[d].ctx, "pgpu_set_leaf_cache");
        if (!m.twoLevel) {
            // the reference refuses a single batching point (pauseable_bvh4.h:180); serve it single-level
            for (size_t d = 0; d < nDev; ++d)
                m.devices[d].singleBvh = bvhIds[d][0];
            m.objsOfLeaf = { objs[0] };
            return a;
        }
        pgpu_host_svdags* hs = nullptr;
        if (m_svdagRes > 0) {
            std::vector<uint32_t> flat;
            std::vector<uint64_t> offs { 0 };
            for (auto& g : groups) {
                flat.insert(flat.end(), g.begin(), g.end());
                offs.push_back(flat.size());
            }
            // voxeliser + octrees + DAG merge on the first GPU (resolutions above 256 only exist in the host builder)
            if (m_svdagRes <= 256 && nGroups <= 65535)
                check(pgpu_build_svdags_device(ctx0, pos.data(), nv, idx.data(), nt, flat.data(), offs.data(), nGroups, m_svdagRes, &hs), ctx0, "pgpu_build_svdags_device");
            else
                check(pgpu_build_svdags(pos.data(), nv, idx.data(), nt, flat.data(), offs.data(), nGroups, m_svdagRes, &hs), ctx0, "pgpu_build_svdags");
        }
        pgpu_host_bvh4* top = nullptr;
        int rc = pgpu_build_top_bvh4(bounds.data(), nGroups, &top);
        if (rc != PGPU_OK && hs)
            pgpu_host_svdags_free(hs);
        check(rc, ctx0, "pgpu_build_top_bvh4");
        size_t nNodes = 0, nLeaves = 0;
        const void* nodes = pgpu_host_bvh4_nodes(top, &nNodes);
        const uint32_t* order = pgpu_host_bvh4_leaf_order(top, &nLeaves);
        m.objsOfLeaf.resize(nLeaves);
        std::vector<uint32_t> ownerOfLeaf(nLeaves);
        for (size_t k = 0; k < nLeaves; ++k) {
            m.objsOfLeaf[k] = objs[order[k]];
            ownerOfLeaf[k] = ownerOfGroup[order[k]];
        }
        pgpu_ctx* failedCtx = nullptr;
        const char* failedWhat = "";
        for (size_t d = 0; d < nDev && rc == PGPU_OK; ++d) {
            pgpu_ctx* const c = m.devices[d].ctx;
            failedCtx = c;
            if (hs) {
                size_t nw = 0, nd = 0;
                const uint32_t* w = pgpu_host_svdags_words(hs, &nw);
                const pgpu_svdag_desc* dd = pgpu_host_svdags_descs(hs, &nd);
                rc = pgpu_upload_svdags(c, w, nw, dd, nd);
                failedWhat = "pgpu_upload_svdags";
            }
            std::vector<pgpu_leaf> leaves(nLeaves);
            for (size_t k = 0; k < nLeaves; ++k)
                leaves[k] = { bvhIds[d][order[k]], m_svdagRes > 0 ? order[k] : PGPU_NO_SVDAG };
            if (rc == PGPU_OK) {
                rc = pgpu_upload_top_bvh4(c, nodes, nNodes, pgpu_host_bvh4_root(top), leaves.data(), nLeaves);
                failedWhat = "pgpu_upload_top_bvh4";
            }
        }
        pgpu_host_bvh4_free(top);
        if (hs)
            pgpu_host_svdags_free(hs);
        check(rc, failedCtx, failedWhat);
        if (m_options.svdagOnlyForNonResident)
            for (size_t d = 0; d < nDev; ++d)
                check(pgpu_set_svdag_policy(m.devices[d].ctx, 1), m.devices[d].ctx, "pgpu_set_svdag_policy");
        if (m.sharded) {
            joinCommunicator(m.devices, m.exchange);
            for (size_t d = 0; d < nDev; ++d) // (the owner table is checked against the communicator's size)
                check(pgpu_set_leaf_owners(m.devices[d].ctx, ownerOfLeaf.data(), nLeaves), m.devices[d].ctx, "pgpu_set_leaf_owners");
        }
        return a;
    }

private:
    void validate() const
    {
        if (!m_shapes || m_shapes->empty())
            throw std::runtime_error("GpuAccelerationStructureBuilder: no shapes");
        if (m_options.devices.empty())
            throw std::runtime_error("GpuAccelerationStructureBuilder: no devices");
        if (m_options.botLevelBVHCacheBytes > 0 && m_options.multiGpu == MultiGpu::LeafSharded)
            throw std::runtime_error("GpuAccelerationStructureBuilder: botLevelBVHCacheBytes (streamed leaves) cannot be combined with MultiGpu::LeafSharded");
    }

    // pgpu_comm_init is collective: one thread per context joins
    template <typename DeviceT>
    void joinCommunicator(std::vector<DeviceT>& devices, std::unique_ptr<detail::ThreadExchange>& exchange) const
    {
        const unsigned world = unsigned(devices.size());
        unsigned char uid[128] = {};
        if (m_options.hostStagedExchange)
            exchange.reset(new detail::ThreadExchange(world));
        else
            check(pgpu_comm_unique_id(uid), nullptr, "pgpu_comm_unique_id");
        std::vector<int> rcs(world, PGPU_OK);
        std::vector<std::thread> joiners;
        for (unsigned r = 0; r < world; ++r)
            joiners.emplace_back([&, r] {
                DeviceT& d = devices[r];
                if (m_options.hostStagedExchange) {
                    d.exchangeEnd.reset(new detail::ExchangeEnd());
                    d.exchangeEnd->x = exchange.get();
                    d.exchangeEnd->rank = r;
                    d.exchangeEnd->world = world;
                    d.exchangeEnd->ctx = d.ctx;
                    const pgpu_transport t { d.exchangeEnd.get(), &detail::ExchangeEnd::allgather, &detail::ExchangeEnd::exchange, &detail::ExchangeEnd::allreduce };
                    rcs[r] = pgpu_comm_init_transport(d.ctx, int(r), int(world), &t);
                } else {
                    rcs[r] = pgpu_comm_init(d.ctx, int(r), int(world), uid);
                }
                if (rcs[r] == PGPU_OK)
                    rcs[r] = pgpu_comm_selftest(d.ctx);
            });
        for (auto& t : joiners)
            t.join();
        for (unsigned r = 0; r < world; ++r)
            check(rcs[r], devices[r].ctx, "pgpu_comm_init / pgpu_comm_selftest");
    }

    const std::vector<TriangleShape>* m_shapes;
    unsigned m_primsPerPoint, m_svdagRes;
    AdapterOptions m_options;
};

} // namespace pandora_amd
