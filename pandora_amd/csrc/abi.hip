// C-ABI implementation: device context, scene upload, batch tracing. See include/pandora_hip.h.
#include "context.h"
#include "trace_ranges.h"
#include <dlfcn.h>
#include <cstdlib>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>

using namespace pandora_hip;

namespace pandora_hip {

const RoctxApi& roctxApi()
{
    static const RoctxApi api = [] {
        RoctxApi a;
        const char* env = std::getenv("PANDORA_ROCTX");
        if (env && env[0] == '0')
            return a;
        // rocprofv3's roctx (ROCm 7), then the roctracer one; an already loaded copy (the profiler preloads its own) is found first
        for (const char* name : { "librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so" }) {
            void* lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (!lib)
                lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!lib)
                continue;
            a.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
            a.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
            if (a.push && a.pop)
                return a;
            a.push = nullptr;
            a.pop = nullptr;
        }
        return a;
    }();
    return api;
}

int fail(pgpu_ctx* c, int code, const char* what, hipError_t e)
{
    if (c) {
        // (the memory helpers of memory_abi.hip may fail on another thread while a trace call fails on this one)
        static std::mutex errMutex;
        std::lock_guard<std::mutex> lock(errMutex);
        c->err = what;
        if (e != hipSuccess) {
            c->err += ": ";
            c->err += hipGetErrorString(e);
        }
    }
    return code;
}

int ensure(pgpu_ctx* c, DeviceBuffer& b, size_t bytes)
{
    if (b.bytes >= bytes)
        return PGPU_OK;
    if (b.ptr)
        HIP_TRY(c, hipFree(b.ptr));
    b.ptr = nullptr;
    b.bytes = 0;
    HIP_TRY(c, hipMalloc(&b.ptr, bytes));
    b.bytes = bytes;
    return PGPU_OK;
}

int syncBvhTable(pgpu_ctx* c)
{
    if (!c->bvhTableDirty)
        return PGPU_OK;
    if (c->bvhTableCap < c->bvhs.size()) {
        if (c->bvhTable)
            HIP_TRY(c, hipFree(c->bvhTable));
        c->bvhTable = nullptr;
        c->bvhTableCap = std::max<size_t>(64, c->bvhs.size() * 2);
        HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->bvhTable), c->bvhTableCap * sizeof(Bvh8Dev)));
    }
    HIP_TRY(c, hipMemcpy(c->bvhTable, c->bvhs.data(), c->bvhs.size() * sizeof(Bvh8Dev), hipMemcpyHostToDevice));
    c->bvhTableDirty = false;
    return PGPU_OK;
}

// Staged BVHs lose their place in the cache and the arena is freed: the cache is rebuilt (empty) at the next pgpu_trace.
int dropLeafCache(pgpu_ctx* c)
{
    if (c->cacheArena || !c->ring.empty())
        HIP_TRY(c, hipDeviceSynchronize());
    for (size_t i = 0; i < c->bvhs.size(); ++i)
        if (i < c->staged.size() && c->staged[i].host) {
            c->staged[i].resident = false;
            c->staged[i].lastLaunch = -1;
            c->bvhs[i].nodes = nullptr;
            c->bvhs[i].tris = nullptr;
        }
    c->ring.clear();
    c->ringWrite = 0;
    if (c->cacheArena)
        HIP_TRY(c, hipFree(c->cacheArena));
    c->cacheArena = nullptr;
    c->cacheBytes = 0;
    c->bvhTableDirty = true;
    return PGPU_OK;
}

} // namespace pandora_hip

namespace {

template <typename T>
int uploadArray(pgpu_ctx* c, const T* host, size_t n, T** dev)
{
    *dev = nullptr;
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(n * sizeof(T), 16);
    HIP_TRY(c, hipMalloc(&p, bytes));
    c->owned.push_back(p);
    if (n)
        HIP_TRY(c, hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *dev = static_cast<T*>(p);
    return PGPU_OK;
}

// Longest chain of nested inner nodes; also validates every handle reachable from the root and marks the nodes and leaf
// slots the tree uses (the reference's allocators leave uninitialised gaps between their blocks,
// contiguous_allocator_ts.h:151-198, so whatever is not reachable is not looked at). Returns -1 on a malformed tree.
int bvh8Depth(const unsigned char* nodes, size_t nNodes, size_t nIdx, uint32_t root, std::vector<uint8_t>& nodeUsed, std::vector<uint8_t>& slotUsed)
{
    struct E {
        uint32_t h;
        int d;
    };
    nodeUsed.assign(nNodes, 0);
    slotUsed.assign(nIdx, 0);
    std::vector<E> todo { { root, 0 } };
    int deepest = 0;
    size_t visited = 0;
    while (!todo.empty()) {
        const E e = todo.back();
        todo.pop_back();
        const uint32_t flags = e.h >> 29, index = e.h & 0x1FFFFFFFu;
        if (flags == 2u) {
            if (index >= nNodes || ++visited > nNodes)
                return -1;
            nodeUsed[index] = 1;
            deepest = std::max(deepest, e.d + 1);
            const uint32_t* kids = reinterpret_cast<const uint32_t*>(nodes + size_t(index) * 256 + 192);
            for (int i = 0; i < 8; ++i)
                if (kids[i] != 0u)
                    todo.push_back({ kids[i], e.d + 1 });
        } else if (flags & 4u) {
            const uint32_t count = (flags & 3u) + 1u;
            if (size_t(index) + count > nIdx)
                return -1;
            for (uint32_t k = 0; k < count; ++k)
                slotUsed[index + k] = 1;
        } else {
            return -1; // reserved flag patterns 001 / 011 (000 = empty is skipped above)
        }
    }
    return deepest;
}

// The kernel's fast slab test assumes min <= max and finite bounds for every present child.
bool bvh8NeedsExact(const unsigned char* nodes, size_t nNodes, const std::vector<uint8_t>& nodeUsed)
{
    for (size_t i = 0; i < nNodes; ++i) {
        if (!nodeUsed[i])
            continue;
        const float* f = reinterpret_cast<const float*>(nodes + i * 256);
        const uint32_t* kids = reinterpret_cast<const uint32_t*>(f) + 48;
        for (int c = 0; c < 8; ++c) {
            if (kids[c] == 0u)
                continue;
            for (int a = 0; a < 3; ++a) {
                const float lo = f[16 * a + c], hi = f[16 * a + 8 + c];
                if (!(lo <= hi) || !(std::fabs(lo) <= 3.4028234663852886e38f) || !(std::fabs(hi) <= 3.4028234663852886e38f))
                    return true;
            }
        }
    }
    return false;
}

} // namespace

namespace pandora_hip {
int readKernelStats(pgpu_ctx* c)
{
    unsigned long long s[6];
    HIP_TRY(c, hipMemcpy(s, c->statsDev, sizeof(s), hipMemcpyDeviceToHost));
    c->counters.inner_visits = s[0];
    c->counters.leaf_pops = s[1];
    c->counters.tri_tests = s[2];
    c->counters.wave_steps = s[3];
    c->counters.leaf_passes = s[4];
    c->counters.refill_passes = s[5];
    return PGPU_OK;
}

void planBvh8LaunchShape(const pgpu_ctx* c, uint32_t depth, size_t n_work, Bvh8TraceArgs& a, uint32_t* blocks)
{
    const uint32_t need = stackEntriesForDepth(depth);
    const uint32_t window = stackWindow(need, c->stackWindow);
    // Layout. Instance leaves: quad only. Automatic: quad (16 rays per wave) for a whole batch through one BVH, where the kernel is
    // bound by instruction issue; pair (32 rays per wave) for the work lists of the two-level loops, whose short traversals are
    // bound by the latency of fetching rays and roots -- twice the rays in flight per wave measured 6 % faster there.
    const uint32_t lanes = c->hasInstances ? 4u : (c->lanesPerRay ? c->lanesPerRay : (a.work_ray ? 2u : 4u));
    a.lanes_per_ray = lanes;
    a.instanced = c->hasInstances ? 1u : 0u;
    a.stack_entries = window;
    a.refill_min = c->refillMin ? c->refillMin : (lanes == 2u ? 6u : 3u);
    a.n_bvhs = uint32_t(c->bvhs.size());
    if (a.chunk == 0)
        a.chunk = n_work >= (size_t(1) << 18) ? 128u : 64u; // automatic: large batches amortise the cursor atomics over more rays (measured)
    a.range_size = uint32_t((((n_work + 7) / 8 + a.chunk - 1) / a.chunk) * a.chunk);
    a.leaf_min = c->leafMin ? c->leafMin : (lanes == 2u ? 8u : 4u);
    a.spill_entries = need > window ? need - window : 0u;
    a.spill = nullptr;
    const uint32_t raysPerBlock = 256u / lanes; // 4 waves x 16 (or 32) rays
    const uint32_t maxBlocks = uint32_t(c->numCUs) * std::min(blocksPerCU(c->blocksPerCU, window, lanes), c->hasInstances ? 6u : 8u);
    const uint32_t wantBlocks = uint32_t((n_work + 2 * raysPerBlock - 1) / (2 * raysPerBlock));
    *blocks = std::max(1u, std::min(maxBlocks, wantBlocks));
}

int planBvh8Launch(pgpu_ctx* c, uint32_t depth, size_t n_work, Bvh8TraceArgs& a, uint32_t* blocks, DeviceBuffer* spill)
{
    planBvh8LaunchShape(c, depth, n_work, a, blocks);
    if (a.spill_entries) {
        DeviceBuffer& sb = spill ? *spill : c->spill;
        const uint32_t raysPerBlock = 256u / a.lanes_per_ray;
        const size_t bytes = size_t(*blocks) * raysPerBlock * a.spill_entries * 8u;
        const int rc = ensure(c, sb, bytes);
        if (rc != PGPU_OK)
            return rc;
        a.spill = static_cast<unsigned long long*>(sb.ptr);
    }
    return PGPU_OK;
}
}

extern "C" {

const char* pgpu_version(void) { return "pandora_hip 0.1 (gfx950)"; }

int pgpu_init(int device_id, pgpu_ctx** out)
{
    if (!out)
        return PGPU_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return PGPU_ERR_NO_DEVICE;
    if (device_id < 0 || device_id >= count)
        return PGPU_ERR_INVALID;
    pgpu_ctx* c = new (std::nothrow) pgpu_ctx();
    if (!c)
        return PGPU_ERR_INVALID;
    c->device = device_id;
    hipError_t e = hipSetDevice(device_id);
    if (e == hipSuccess)
        e = hipStreamCreateWithFlags(&c->stream, hipStreamDefault); // blocking: ordered against the legacy null stream (header: "streams")
    if (e == hipSuccess)
        e = hipEventCreate(&c->ev0);
    if (e == hipSuccess)
        e = hipEventCreate(&c->ev1);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&c->cursor), 512); // (two sets of 8 cursors: c->cursor, c->cursor + 64)
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&c->statsDev), 64);
    if (e == hipSuccess)
        e = hipMemset(c->statsDev, 0, 64);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&c->topCounters), 64);
    if (e == hipSuccess)
        e = hipMemset(c->topCounters, 0, 64);
    hipDeviceProp_t prop;
    if (e == hipSuccess)
        e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) {
        pgpu_shutdown(c); // (releases whatever the steps above did create)
        return PGPU_ERR_HIP;
    }
    c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = std::getenv("PANDORA_LEAF_DRAIN")) // (tuning knob of the streamed loop's queue policy, top_abi.hip; results do not depend on it)
        c->leafDrainRatio = float(std::atof(e));
    if (const char* e = std::getenv("PANDORA_LEAF_INPLACE"))
        c->leafInPlaceBytesPerRay = uint32_t(std::atol(e));
    if (const char* e = std::getenv("PANDORA_LEAF_PREFETCH"))
        c->leafPrefetchBytesPerRay = float(std::atof(e));
    if (initStaticScratch(c) != PGPU_OK) {
        pgpu_shutdown(c);
        return PGPU_ERR_HIP;
    }
    *out = c;
    return PGPU_OK;
}

void pgpu_shutdown(pgpu_ctx* c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    destroyComm(c);
    destroyLanes(c);
    if (c->lanesFork)
        (void)hipEventDestroy(c->lanesFork);
    for (void* p : c->owned)
        (void)hipFree(p);
    if (c->bvhTable)
        (void)hipFree(c->bvhTable);
    if (c->cursor)
        (void)hipFree(c->cursor);
    if (c->statsDev)
        (void)hipFree(c->statsDev);
    if (c->topCounters)
        (void)hipFree(c->topCounters);
    for (StagedBvh& s : c->staged)
        if (s.host)
            (void)hipHostFree(s.host);
    for (hipEvent_t e : c->launchEvents)
        (void)hipEventDestroy(e);
    if (c->cacheArena)
        (void)hipFree(c->cacheArena);
    if (c->copyDone)
        (void)hipEventDestroy(c->copyDone);
    if (c->leafHostBuf)
        (void)hipHostFree(c->leafHostBuf);
    if (c->copyTableHost)
        (void)hipHostFree(c->copyTableHost);
    if (c->copyTableDev)
        (void)hipFree(c->copyTableDev);
    for (hipEvent_t e : c->roundEv)
        if (e)
            (void)hipEventDestroy(e);
    if (c->roundCounts)
        (void)hipHostFree(c->roundCounts);
    if (c->copyStream)
        (void)hipStreamDestroy(c->copyStream);
    if (c->auxStream)
        (void)hipStreamDestroy(c->auxStream);
    for (hipEvent_t e : { c->auxFork, c->auxJoin })
        if (e)
            (void)hipEventDestroy(e);
    for (DeviceBuffer* b : { &c->stageRays, &c->stageHits, &c->work, &c->spill, &c->blockWork, &c->routeWork, &c->compactWork, &c->shardWork, &c->shardSend, &c->shardRecv, &c->spillB, &c->tableView })
        if (b->ptr)
            (void)hipFree(b->ptr);
    if (c->ev0)
        (void)hipEventDestroy(c->ev0);
    if (c->ev1)
        (void)hipEventDestroy(c->ev1);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* pgpu_last_error(const pgpu_ctx* c) { return c ? c->err.c_str() : "null context"; }
float pgpu_last_kernel_ms(const pgpu_ctx* c) { return c ? c->lastKernelMs : 0.0f; }

int pgpu_set_stats(pgpu_ctx* c, int enabled)
{
    if (!c)
        return PGPU_ERR_INVALID;
    c->stats = enabled != 0;
    return PGPU_OK;
}

int pgpu_set_keep_shading(pgpu_ctx* c, int on)
{
    if (!c)
        return PGPU_ERR_INVALID;
    c->keepShading = on != 0;
    return PGPU_OK;
}

int pgpu_set_tuning(pgpu_ctx* c, int blocks_per_cu, int chunk)
{
    if (!c || blocks_per_cu < 0 || blocks_per_cu > 8 || (chunk != 0 && chunk < 8) || chunk > 65536)
        return PGPU_ERR_INVALID;
    c->blocksPerCU = blocks_per_cu;
    c->chunk = uint32_t(chunk);
    return PGPU_OK;
}

int pgpu_set_kernel_variant(pgpu_ctx* c, int refill_min, int stack_window, int leaf_min)
{
    if (!c || refill_min < 0 || refill_min > 32 || stack_window < 0 || stack_window > 96
        || leaf_min < 0 || leaf_min > 32)
        return PGPU_ERR_INVALID;
    c->leafMin = uint32_t(leaf_min);
    c->refillMin = uint32_t(refill_min);
    c->stackWindow = stack_window;
    return PGPU_OK;
}

int pgpu_set_round_lookahead(pgpu_ctx* c, int on)
{
    if (!c)
        return PGPU_ERR_INVALID;
    c->roundLookahead = on != 0;
    return PGPU_OK;
}

int pgpu_set_ray_layout(pgpu_ctx* c, int lanes_per_ray)
{
    if (!c || (lanes_per_ray != 0 && lanes_per_ray != 2 && lanes_per_ray != 4))
        return PGPU_ERR_INVALID;
    c->lanesPerRay = uint32_t(lanes_per_ray);
    return PGPU_OK;
}

int pgpu_get_counters(pgpu_ctx* c, pgpu_counters* out)
{
    if (!c || !out)
        return PGPU_ERR_INVALID;
    *out = c->counters;
    return PGPU_OK;
}

int pgpu_upload_mesh(pgpu_ctx* c, const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles,
    uint32_t* mesh_id)
{
    const TraceRange range_("pgpu_upload_mesh");
    if (!c || !positions || !indices || !mesh_id || n_vertices == 0 || n_triangles == 0)
        return fail(c, PGPU_ERR_INVALID, "pgpu_upload_mesh: null or empty argument");
    for (size_t i = 0; i < 3 * n_triangles; ++i)
        if (indices[i] >= n_vertices)
            return fail(c, PGPU_ERR_INVALID, "pgpu_upload_mesh: vertex index out of range");
    HostMesh m;
    m.pos.assign(positions, positions + 3 * n_vertices);
    m.idx.assign(indices, indices + 3 * n_triangles);
    c->meshes.push_back(std::move(m));
    *mesh_id = uint32_t(c->meshes.size() - 1);
    return PGPU_OK;
}

// Validation shared by pgpu_upload_bvh8 and pgpu_stage_bvh8 + the gather of every leaf slot's triangle (see Bvh8Dev in kernels.h).
static int prepareBvh8(pgpu_ctx* c, const char* who, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, const pgpu_instance* instances, size_t n_instances, std::vector<float>& tris,
    std::vector<InstDev>& inst, Bvh8Dev& d, std::vector<float>* shade = nullptr)
{
    auto bad = [&](int code, const char* what) {
        c->err = std::string(who) + ": " + what;
        return code;
    };
    if (!leaf_indices || !objs || n_indices == 0 || n_objs == 0 || (n_nodes && !nodes256))
        return bad(PGPU_ERR_INVALID, "null or empty argument");
    if (n_nodes >= (1u << 29) || n_indices >= (1u << 29))
        return bad(PGPU_ERR_INVALID, "more than 2^29 nodes or leaf indices");
    std::vector<uint8_t> nodeUsed, slotUsed;
    const int depth = bvh8Depth(static_cast<const unsigned char*>(nodes256), n_nodes, n_indices, root_handle, nodeUsed, slotUsed);
    if (depth < 0)
        return bad(PGPU_ERR_INVALID, "malformed tree (handle out of range or cycle)");
    if (7 * depth + 8 > 96) // the 96-entry traversal stack of wive_bvh8_impl.h:74-75
        return bad(PGPU_ERR_DEPTH, "more than 12 nested inner nodes");
    tris.assign((n_indices + 1) * 12, 0.0f); // (one zero record of padding: the pair kernel reads records two at a time)
    if (shade)
        shade->assign(n_objs * 9, 0.0f);
    std::vector<uint8_t> instSlot(n_indices, 0);
    uint32_t childDepth = 0;
    inst.clear();
    for (size_t i = 0; i < n_instances; ++i) {
        const pgpu_instance& in = instances[i];
        if (in.bvh_id >= c->bvhs.size())
            return bad(PGPU_ERR_INVALID, "instance of an unknown bvh id");
        if (c->bvhs[in.bvh_id].n_inst != 0)
            return bad(PGPU_ERR_DEPTH, "instance of a BVH that holds instances itself (one level of instancing is supported)");
        if (in.bvh_id < c->staged.size() && c->staged[in.bvh_id].host)
            return bad(PGPU_ERR_INVALID, "instance of a BVH staged in host memory (instanced prototypes must be resident: pgpu_upload_bvh8)");
        InstDev id {};
        std::memcpy(id.m, in.world_to_local, sizeof(id.m));
        id.child = in.bvh_id;
        inst.push_back(id);
        childDepth = std::max(childDepth, c->bvhs[in.bvh_id].depth);
    }
    for (size_t s = 0; s < n_indices; ++s) {
        if (!slotUsed[s])
            continue; // a gap between allocator blocks: no leaf owns it, its record stays a degenerate zero triangle
        const uint32_t obj = leaf_indices[s];
        if (obj >= n_objs)
            return bad(PGPU_ERR_INVALID, "leaf index beyond the leaf-object table");
        const pgpu_leafobj lo = objs[obj];
        if (lo.mesh == PGPU_INSTANCE) {
            // OfflineBVHLeaf::Instance: a degenerate (all-zero) triangle that never hits, tagged with its table entry
            if (!instances || lo.prim >= n_instances)
                return bad(PGPU_ERR_INVALID, "instance index beyond the instance table");
            const uint32_t tag = 0x80000000u | lo.prim;
            std::memcpy(&tris[s * 12 + 3], &obj, 4);
            std::memcpy(&tris[s * 12 + 7], &tag, 4);
            instSlot[s] = 1;
            continue;
        }
        if (lo.mesh >= c->meshes.size())
            return bad(PGPU_ERR_INVALID, "unknown mesh id");
        const HostMesh& m = c->meshes[lo.mesh];
        if (size_t(lo.prim) * 3 + 2 >= m.idx.size())
            return bad(PGPU_ERR_INVALID, "primitive id out of range");
        float* t = &tris[s * 12];
        for (int k = 0; k < 3; ++k) {
            const float* p = &m.pos[size_t(m.idx[size_t(lo.prim) * 3 + size_t(k)]) * 3];
            t[4 * k + 0] = p[0];
            t[4 * k + 1] = p[1];
            t[4 * k + 2] = p[2];
            if (shade)
                std::memcpy(&(*shade)[size_t(obj) * 9 + size_t(k) * 3], p, 12);
        }
        std::memcpy(&t[3], &obj, 4);
    }
    if (n_instances) {
        // the kernel enters an instance from a leaf that holds nothing else (the reference tests a leaf's objects one after
        // the other, wive_bvh8_impl.h:272-280; a nested traversal in the middle of a 4-wide triangle pass has no place here)
        auto leafOk = [&](uint32_t h) {
            if ((h >> 29) < 4u)
                return true;
            const uint32_t index = h & 0x1FFFFFFFu, count = ((h >> 29) & 3u) + 1u;
            if (count == 1u)
                return true;
            for (uint32_t k = 0; k < count; ++k)
                if (instSlot[index + k])
                    return false;
            return true;
        };
        bool ok = leafOk(root_handle);
        for (size_t i = 0; i < n_nodes && ok; ++i) {
            if (!nodeUsed[i])
                continue;
            const uint32_t* kids = reinterpret_cast<const uint32_t*>(static_cast<const unsigned char*>(nodes256) + i * 256 + 192);
            for (int k = 0; k < 8 && ok; ++k)
                ok = leafOk(kids[k]);
        }
        if (!ok)
            return bad(PGPU_ERR_INVALID, "an instance shares its leaf with other objects (build with pgpu_build_bvh8_boxes and the `alone` flag)");
    }
    d = Bvh8Dev {};
    d.root = root_handle;
    d.n_nodes = uint32_t(n_nodes);
    d.n_slots = uint32_t(n_indices);
    // with instances the per-ray stack holds the outer traversal's entries below the nested one's: (7 d1 + 8) + (7 d2 + 8)
    d.depth = n_instances ? uint32_t(depth) + childDepth + 2u : uint32_t(depth);
    d.n_inst = uint32_t(n_instances);
    d.needs_exact = bvh8NeedsExact(static_cast<const unsigned char*>(nodes256), n_nodes, nodeUsed) ? 1u : 0u;
    return PGPU_OK;
}

int pgpu_upload_bvh8(pgpu_ctx* c, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, uint32_t* bvh_id)
{
    return pgpu_upload_bvh8_instanced(c, nodes256, n_nodes, leaf_indices, n_indices, root_handle, objs, n_objs, nullptr, 0, bvh_id);
}

int pgpu_upload_bvh8_instanced(pgpu_ctx* c, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, const pgpu_instance* instances, size_t n_instances, uint32_t* bvh_id)
{
    const TraceRange range_("pgpu_upload_bvh8");
    if (!c || !bvh_id || (n_instances && !instances))
        return fail(c, PGPU_ERR_INVALID, "pgpu_upload_bvh8: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<float> tris, shade;
    std::vector<InstDev> inst;
    Bvh8Dev d {};
    int rc = prepareBvh8(c, "pgpu_upload_bvh8", nodes256, n_nodes, leaf_indices, n_indices, root_handle, objs, n_objs, instances, n_instances, tris, inst, d,
        c->keepShading ? &shade : nullptr);
    if (rc != PGPU_OK)
        return rc;
    unsigned char* devNodes = nullptr;
    float* devTris = nullptr;
    if (c->keepShading) {
        float* devShade = nullptr;
        rc = uploadArray(c, shade.data(), shade.size(), &devShade);
        if (rc != PGPU_OK)
            return rc;
        d.shade = devShade;
    }
    if (!inst.empty()) {
        InstDev* devInst = nullptr;
        rc = uploadArray(c, inst.data(), inst.size(), &devInst);
        if (rc != PGPU_OK)
            return rc;
        d.inst = reinterpret_cast<const unsigned char*>(devInst);
        c->hasInstances = true;
    }
    rc = uploadArray(c, static_cast<const unsigned char*>(nodes256), n_nodes * 256, &devNodes);
    if (rc != PGPU_OK)
        return rc;
    rc = uploadArray(c, tris.data(), tris.size(), &devTris);
    if (rc != PGPU_OK)
        return rc;
    d.nodes = devNodes;
    d.tris = reinterpret_cast<const unsigned char*>(devTris);
    c->bvhs.push_back(d);
    c->staged.resize(c->bvhs.size());
    c->bvhTableDirty = true;
    *bvh_id = uint32_t(c->bvhs.size() - 1);
    return syncBvhTable(c); // the device table is complete after every upload: trace calls need not allocate for it
}

int pgpu_stage_bvh8(pgpu_ctx* c, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, uint32_t* bvh_id)
{
    return pgpu_stage_bvh8_instanced(c, nodes256, n_nodes, leaf_indices, n_indices, root_handle, objs, n_objs, nullptr, 0, bvh_id);
}

int pgpu_stage_bvh8_instanced(pgpu_ctx* c, const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_indices,
    uint32_t root_handle, const pgpu_leafobj* objs, size_t n_objs, const pgpu_instance* instances, size_t n_instances, uint32_t* bvh_id)
{
    const TraceRange range_("pgpu_stage_bvh8");
    if (!c || !bvh_id || (n_instances && !instances))
        return fail(c, PGPU_ERR_INVALID, "pgpu_stage_bvh8: null argument");
    HIP_TRY(c, hipSetDevice(c->device));
    std::vector<float> tris;
    std::vector<InstDev> inst;
    Bvh8Dev d {};
    const int rc = prepareBvh8(c, "pgpu_stage_bvh8", nodes256, n_nodes, leaf_indices, n_indices, root_handle, objs, n_objs, instances, n_instances, tris, inst, d);
    if (rc != PGPU_OK)
        return rc;
    // The cache's slots were sized for the images staged so far: start over (it is rebuilt at the next pgpu_trace), or
    // an image larger than a slot would overrun it.
    const int rd = dropLeafCache(c);
    if (rd != PGPU_OK)
        return rd;
    StagedBvh s;
    s.nodeBytes = (n_nodes * 256 + 255) & ~size_t(255);
    s.triBytes = (tris.size() * 4 + 255) & ~size_t(255);
    s.instBytes = (inst.size() * sizeof(InstDev) + 255) & ~size_t(255);
    HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&s.host), std::max<size_t>(s.nodeBytes + s.triBytes + s.instBytes, 256), hipHostMallocDefault));
    std::memset(s.host, 0, s.nodeBytes + s.triBytes + s.instBytes);
    if (n_nodes)
        std::memcpy(s.host, nodes256, n_nodes * 256);
    std::memcpy(s.host + s.nodeBytes, tris.data(), tris.size() * 4);
    if (!inst.empty()) {
        std::memcpy(s.host + s.nodeBytes + s.triBytes, inst.data(), inst.size() * sizeof(InstDev));
        c->hasInstances = true;
    }
    c->bvhs.push_back(d); // nodes / tris stay null until the BVH sits in a cache slot
    c->staged.resize(c->bvhs.size());
    c->staged.back() = s;
    c->nStaged++;
    c->bvhTableDirty = true;
    *bvh_id = uint32_t(c->bvhs.size() - 1);
    return PGPU_OK;
}

int pgpu_set_leaf_cache(pgpu_ctx* c, size_t bytes)
{
    if (!c)
        return PGPU_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    const int rc = dropLeafCache(c);
    if (rc != PGPU_OK)
        return rc;
    c->cacheBudget = bytes;
    return PGPU_OK;
}

int pgpu_set_leaf_hold(pgpu_ctx* c, uint32_t image_bytes_per_ray)
{
    if (!c)
        return PGPU_ERR_INVALID;
    c->leafHoldBytesPerRay = image_bytes_per_ray;
    return PGPU_OK;
}

int pgpu_set_leaf_in_place(pgpu_ctx* c, uint32_t image_bytes_per_ray)
{
    if (!c)
        return PGPU_ERR_INVALID;
    c->leafInPlaceBytesPerRay = image_bytes_per_ray;
    return PGPU_OK;
}

int pgpu_set_leaf_copy(pgpu_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 1)
        return fail(c, PGPU_ERR_INVALID, "pgpu_set_leaf_copy: mode is 0 (gather kernel) or 1 (one hipMemcpyAsync per image)");
    c->leafCopyMode = mode;
    return PGPU_OK;
}

int pgpu_trace_bvh8(pgpu_ctx* c, uint32_t bvh_id, int kind, int tri_mode, const pgpu_rays* rays, size_t n, int mem,
    const pgpu_hits* hits, uint32_t* any_hit, void* stream_)
{
    const TraceRange range_("pgpu_trace_bvh8");
    if (!c || !rays)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8: null argument");
    if (bvh_id >= c->bvhs.size())
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_bvh8: unknown bvh id");
    if (bvh_id < c->staged.size() && c->staged[bvh_id].host)
        return fail(c, PGPU_ERR_NO_SCENE, "pgpu_trace_bvh8: this BVH is staged in host memory (pgpu_stage_bvh8); only pgpu_trace streams it");
    if (kind != PGPU_CLOSEST && kind != PGPU_ANY)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8: kind must be PGPU_CLOSEST or PGPU_ANY");
    if (tri_mode != PGPU_TRI_MT && tri_mode != PGPU_TRI_WATERTIGHT)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8: unknown triangle test");
    if (kind == PGPU_CLOSEST && (!hits || !hits->t || !hits->obj || !hits->leaf || !hits->u || !hits->v))
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8: closest-hit needs t, obj, leaf, u, v arrays");
    if (kind == PGPU_ANY && !any_hit)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8: any-hit needs the any_hit array");
    if (n > 0x3FFFFFFFull)
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8: batch too large");
    const float* rp[8] = { rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->tnear, rays->tfar };
    for (const float* p : rp)
        if (!p && n)
            return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8: null ray array");
    if (n == 0)
        return PGPU_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = syncBvhTable(c);
    if (rc != PGPU_OK)
        return rc;
    hipStream_t stream = (mem == PGPU_MEM_DEVICE && stream_) ? static_cast<hipStream_t>(stream_) : c->stream;

    Bvh8TraceArgs a {};
    float* hostOut[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
    if (mem == PGPU_MEM_HOST) {
        rc = ensure(c, c->stageRays, n * 8 * sizeof(float));
        if (rc != PGPU_OK)
            return rc;
        rc = ensure(c, c->stageHits, n * 7 * sizeof(float));
        if (rc != PGPU_OK)
            return rc;
        float* dr = static_cast<float*>(c->stageRays.ptr);
        for (int i = 0; i < 8; ++i)
            HIP_TRY(c, hipMemcpyAsync(dr + size_t(i) * n, rp[i], n * sizeof(float), hipMemcpyHostToDevice, stream));
        for (int i = 0; i < 8; ++i)
            rp[i] = dr + size_t(i) * n;
        float* dh = static_cast<float*>(c->stageHits.ptr);
        for (int i = 0; i < 5; ++i)
            hostOut[i] = dh + size_t(i) * n;
        a.t_out = hostOut[0];
        a.obj_out = reinterpret_cast<uint32_t*>(hostOut[1]);
        a.leaf_out = reinterpret_cast<uint32_t*>(hostOut[2]);
        a.u_out = hostOut[3];
        a.v_out = hostOut[4];
        a.any_out = reinterpret_cast<uint32_t*>(dh + size_t(5) * n);
        if (kind == PGPU_CLOSEST && hits->inst)
            a.inst_out = reinterpret_cast<uint32_t*>(dh + size_t(6) * n);
    } else if (mem == PGPU_MEM_DEVICE) {
        if (kind == PGPU_CLOSEST) {
            a.t_out = hits->t;
            a.obj_out = hits->obj;
            a.leaf_out = hits->leaf;
            a.u_out = hits->u;
            a.v_out = hits->v;
            a.inst_out = hits->inst;
        }
        a.any_out = any_hit;
    } else {
        return fail(c, PGPU_ERR_INVALID, "pgpu_trace_bvh8: mem must be PGPU_MEM_HOST or PGPU_MEM_DEVICE");
    }
    a.bvhs = c->bvhTable;
    a.work_ray = nullptr;
    a.work_bvh = nullptr;
    a.fixed_bvh = bvh_id;
    a.n_work = uint32_t(n);
    a.ox = rp[0];
    a.oy = rp[1];
    a.oz = rp[2];
    a.dx = rp[3];
    a.dy = rp[4];
    a.dz = rp[5];
    a.tnear = rp[6];
    a.tfar_in = rp[7];
    a.cursor = c->cursor;
    a.chunk = c->chunk;
    a.init_miss = 1;
    a.stats = c->stats ? c->statsDev : nullptr;

    HIP_TRY(c, hipMemsetAsync(c->cursor, 0, 256, stream));
    if (c->stats)
        HIP_TRY(c, hipMemsetAsync(c->statsDev, 0, 64, stream));
    if (a.inst_out && !c->hasInstances)
        HIP_TRY(c, hipMemsetAsync(a.inst_out, 0xFF, n * 4, stream)); // (no instance leaves anywhere: the plain kernel does not write it)
    uint32_t blocks = 1;
    rc = pandora_hip::planBvh8Launch(c, c->bvhs[bvh_id].depth, n, a, &blocks);
    if (rc != PGPU_OK)
        return rc;
    HIP_TRY(c, hipEventRecord(c->ev0, stream));
    HIP_TRY(c, launchBvh8Trace(a, kind, tri_mode, c->stats, blocks, stream));
    HIP_TRY(c, hipEventRecord(c->ev1, stream));
    c->timingStale = false;

    if (mem == PGPU_MEM_HOST) {
        if (kind == PGPU_CLOSEST) {
            void* dst[5] = { hits->t, hits->obj, hits->leaf, hits->u, hits->v };
            for (int i = 0; i < 5; ++i)
                HIP_TRY(c, hipMemcpyAsync(dst[i], hostOut[i], n * 4, hipMemcpyDeviceToHost, stream));
            if (hits->inst)
                HIP_TRY(c, hipMemcpyAsync(hits->inst, a.inst_out, n * 4, hipMemcpyDeviceToHost, stream));
            if (hits->visits)
                std::memset(hits->visits, 0, n * 4);
        } else {
            HIP_TRY(c, hipMemcpyAsync(any_hit, a.any_out, n * 4, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(c, hipStreamSynchronize(stream));
        HIP_TRY(c, hipEventElapsedTime(&c->lastKernelMs, c->ev0, c->ev1));
        if (c->stats) {
            c->counters = pgpu_counters {};
            const int rs = pandora_hip::readKernelStats(c);
            if (rs != PGPU_OK)
                return rs;
        }
    }
    return PGPU_OK;
}

// Blocks until the context's last device-memory trace has finished and refreshes the timing / counters.
int pgpu_sync(pgpu_ctx* c, void* stream_)
{
    if (!c)
        return PGPU_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t stream = stream_ ? static_cast<hipStream_t>(stream_) : c->stream;
    HIP_TRY(c, hipStreamSynchronize(stream));
    if (!c->timingStale && hipEventQuery(c->ev1) == hipSuccess)
        (void)hipEventElapsedTime(&c->lastKernelMs, c->ev0, c->ev1);
    if (c->topCountersDirty) {
        // SVDAG counters of the pgpu_top_episode calls since the last sync (g_stats.svdag.*): they accumulate
        unsigned long long t[2] = { 0, 0 };
        HIP_TRY(c, hipMemcpy(t, c->topCounters, sizeof(t), hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemset(c->topCounters, 0, 64));
        c->counters.svdag_tests += t[0];
        c->counters.svdag_culled += t[1];
        c->topCountersDirty = false;
    }
    if (c->stats) {
        const int rs = pandora_hip::readKernelStats(c);
        if (rs != PGPU_OK)
            return rs;
    }
    return PGPU_OK;
}
}
