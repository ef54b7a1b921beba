// Header-only host adapter over the C ABI (include/pandora_hip.h). It keeps the shape of the reference's
// interface so that integrator code written against
//     pandora::AccelerationStructure<HitRayState, AnyHitRayState>          (traversal/acceleration_structure.h:6-11)
//     pandora::OfflineBatchingAccelerationStructureBuilder::build(...)     (offline_batching_acceleration_structure.h:106-126,307-386)
// can switch over: per-ray intersect()/intersectAny() from any thread, results delivered to the four
// hit / miss / anyHit / anyMiss handlers exactly once per ray. Where the reference enqueues into TaskGraph
// task handles and traces when TaskGraph::run() drains the queues (task_graph.cpp:19-56), this adapter
// buffers rays and traces when flush() is called; handlers run on the flushing thread.
//
// On a hit the handler receives the ray with tfar = hit distance, the hit record (shape id, primitive id,
// geometric uv and the geometric normal flipped towards the ray, as triangle_intersect.cpp:455-461) and
// the caller's state. Errors surface as std::runtime_error like the reference's ALWAYS_ASSERT/THROW_ERROR
// (utility/error_handling.h:7-37); the C ABI underneath never throws.
#pragma once
#include "pandora_hip.h"
#include <cmath>
#include <cstdint>
#include <functional>
#include <limits>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
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

inline void check(int rc, pgpu_ctx* ctx, const char* what)
{
    if (rc != PGPU_OK)
        throw std::runtime_error(std::string(what) + ": " + (ctx ? pgpu_last_error(ctx) : "error ") + " (code " + std::to_string(rc) + ")");
}

template <typename HitRayState, typename AnyHitRayState>
class GpuAccelerationStructure {
public:
    using OnHit = std::function<void(const Ray&, const RayHit&, const HitRayState&)>;
    using OnMiss = std::function<void(const Ray&, const HitRayState&)>;
    using OnAnyHit = std::function<void(const Ray&, const AnyHitRayState&)>;
    using OnAnyMiss = std::function<void(const Ray&, const AnyHitRayState&)>;

    GpuAccelerationStructure(GpuAccelerationStructure&& o) noexcept
        : m_ctx(o.m_ctx)
        , m_twoLevel(o.m_twoLevel)
        , m_triMode(o.m_triMode)
        , m_bvhOfLeaf(std::move(o.m_bvhOfLeaf))
        , m_objsOfLeaf(std::move(o.m_objsOfLeaf))
        , m_shapes(o.m_shapes)
        , m_onHit(std::move(o.m_onHit))
        , m_onMiss(std::move(o.m_onMiss))
        , m_onAnyHit(std::move(o.m_onAnyHit))
        , m_onAnyMiss(std::move(o.m_onAnyMiss))
        , m_counters(o.m_counters)
        , m_mutex(std::move(o.m_mutex))
        , m_rays(std::move(o.m_rays))
        , m_anyRays(std::move(o.m_anyRays))
        , m_states(std::move(o.m_states))
        , m_anyStates(std::move(o.m_anyStates))
        , m_soa(std::move(o.m_soa))
    {
        o.m_ctx = nullptr; // the context has exactly one owner
    }
    GpuAccelerationStructure(const GpuAccelerationStructure&) = delete;
    ~GpuAccelerationStructure()
    {
        if (m_ctx)
            pgpu_shutdown(m_ctx);
    }

    // AccelerationStructure::intersect (acceleration_structure.h:8): callable concurrently
    void intersect(const Ray& ray, const HitRayState& state) const
    {
        std::lock_guard<std::mutex> l(*m_mutex);
        m_rays.push_back(ray);
        m_states.push_back(state);
    }
    // AccelerationStructure::intersectAny (acceleration_structure.h:9)
    void intersectAny(const Ray& ray, const AnyHitRayState& state) const
    {
        std::lock_guard<std::mutex> l(*m_mutex);
        m_anyRays.push_back(ray);
        m_anyStates.push_back(state);
    }

    size_t pending() const
    {
        std::lock_guard<std::mutex> l(*m_mutex);
        return m_rays.size() + m_anyRays.size();
    }

    // Traces everything buffered so far and runs the handlers; handlers may call intersect()/intersectAny()
    // again (next bounce) -- those rays are traced by the next flush(). Returns the number of rays traced.
    size_t flush()
    {
        std::vector<Ray> rays, anyRays;
        std::vector<HitRayState> states;
        std::vector<AnyHitRayState> anyStates;
        {
            std::lock_guard<std::mutex> l(*m_mutex);
            rays.swap(m_rays);
            states.swap(m_states);
            anyRays.swap(m_anyRays);
            anyStates.swap(m_anyStates);
        }
        if (!rays.empty()) {
            const size_t n = rays.size();
            soa(rays);
            std::vector<float> t(n), u(n), v(n);
            std::vector<uint32_t> obj(n), leaf(n), visits(n);
            pgpu_rays pr = raysStruct(n);
            pgpu_hits ph { t.data(), obj.data(), leaf.data(), u.data(), v.data(), visits.data() };
            if (m_twoLevel)
                check(pgpu_trace(m_ctx, PGPU_CLOSEST, m_triMode, &pr, n, PGPU_MEM_HOST, &ph, nullptr, &m_counters, nullptr), m_ctx, "pgpu_trace");
            else
                check(pgpu_trace_bvh8(m_ctx, m_bvhOfLeaf[0], PGPU_CLOSEST, m_triMode, &pr, n, PGPU_MEM_HOST, &ph, nullptr, nullptr), m_ctx, "pgpu_trace_bvh8");
            for (size_t i = 0; i < n; ++i) {
                Ray r = rays[i];
                r.numTopLevelIntersections += visits[i];
                if (obj[i] == PGPU_MISS) {
                    m_onMiss(r, states[i]);
                    continue;
                }
                r.tfar = t[i];
                m_onHit(r, surface(r, leaf[i], obj[i], u[i], v[i]), states[i]);
            }
        }
        if (!anyRays.empty()) {
            const size_t n = anyRays.size();
            soa(anyRays);
            std::vector<uint32_t> hit(n), visits(n);
            pgpu_rays pr = raysStruct(n);
            pgpu_hits ph { nullptr, nullptr, nullptr, nullptr, nullptr, visits.data() };
            if (m_twoLevel)
                check(pgpu_trace(m_ctx, PGPU_ANY, m_triMode, &pr, n, PGPU_MEM_HOST, &ph, hit.data(), &m_counters, nullptr), m_ctx, "pgpu_trace");
            else
                check(pgpu_trace_bvh8(m_ctx, m_bvhOfLeaf[0], PGPU_ANY, m_triMode, &pr, n, PGPU_MEM_HOST, nullptr, hit.data(), nullptr), m_ctx, "pgpu_trace_bvh8");
            for (size_t i = 0; i < n; ++i) {
                Ray r = anyRays[i];
                r.numTopLevelIntersections += visits[i];
                if (hit[i]) {
                    r.tfar = -std::numeric_limits<float>::infinity(); // wive_bvh8_impl.h:194
                    m_onAnyHit(r, anyStates[i]);
                } else {
                    m_onAnyMiss(r, anyStates[i]);
                }
            }
        }
        return rays.size() + anyRays.size();
    }

    const pgpu_counters& counters() const { return m_counters; } // g_stats.svdag.* of the last flush
    size_t numBatchingPoints() const { return m_bvhOfLeaf.size(); }

private:
    friend class GpuAccelerationStructureBuilder;
    GpuAccelerationStructure() = default;

    void soa(const std::vector<Ray>& rays)
    {
        const size_t n = rays.size();
        m_soa.resize(8 * n);
        for (size_t i = 0; i < n; ++i) {
            for (int a = 0; a < 3; ++a) {
                m_soa[size_t(a) * n + i] = rays[i].origin[a];
                m_soa[size_t(3 + a) * n + i] = rays[i].direction[a];
            }
            m_soa[6 * n + i] = rays[i].tnear;
            m_soa[7 * n + i] = rays[i].tfar;
        }
    }
    pgpu_rays raysStruct(size_t n) const
    {
        const float* b = m_soa.data();
        return pgpu_rays { b, b + n, b + 2 * n, b + 3 * n, b + 4 * n, b + 5 * n, b + 6 * n, b + 7 * n };
    }
    // what the reference's leaf does on every tentative hit (offline_bvh_cache.cpp:48-49), done once here
    RayHit surface(const Ray& r, uint32_t leaf, uint32_t obj, float u, float v) const
    {
        const pgpu_leafobj lo = m_objsOfLeaf[leaf][obj];
        const TriangleShape& s = (*m_shapes)[lo.mesh];
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

    pgpu_ctx* m_ctx = nullptr;
    bool m_twoLevel = false;
    int m_triMode = PGPU_TRI_MT;
    std::vector<uint32_t> m_bvhOfLeaf;                 // final leaf slot -> bvh id
    std::vector<std::vector<pgpu_leafobj>> m_objsOfLeaf; // final leaf slot -> leaf-object table
    const std::vector<TriangleShape>* m_shapes = nullptr;
    OnHit m_onHit;
    OnMiss m_onMiss;
    OnAnyHit m_onAnyHit;
    OnAnyMiss m_onAnyMiss;
    pgpu_counters m_counters {};
    std::unique_ptr<std::mutex> m_mutex { new std::mutex() };
    mutable std::vector<Ray> m_rays, m_anyRays;
    mutable std::vector<HitRayState> m_states;
    mutable std::vector<AnyHitRayState> m_anyStates;
    std::vector<float> m_soa;
};

// Counterpart of OfflineBatchingAccelerationStructureBuilder (offline_batching...h:106-126): same knobs
// (primitivesPerBatchingPoint, svdagRes; 0 disables the SVDAG early-out as in :323). The geometry cache,
// task graph and BVH cache size of the reference have no role here: everything is resident in HBM.
class GpuAccelerationStructureBuilder {
public:
    GpuAccelerationStructureBuilder(const std::vector<TriangleShape>* shapes, unsigned primitivesPerBatchingPoint, unsigned svdagRes,
        int device = 0, int triangleTest = PGPU_TRI_MT)
        : m_shapes(shapes)
        , m_primsPerPoint(primitivesPerBatchingPoint)
        , m_svdagRes(svdagRes)
        , m_device(device)
        , m_triMode(triangleTest)
    {
        if (!shapes || shapes->empty())
            throw std::runtime_error("GpuAccelerationStructureBuilder: no shapes");
    }

    template <typename HitRayState, typename AnyHitRayState>
    GpuAccelerationStructure<HitRayState, AnyHitRayState> build(
        typename GpuAccelerationStructure<HitRayState, AnyHitRayState>::OnHit hit,
        typename GpuAccelerationStructure<HitRayState, AnyHitRayState>::OnMiss miss,
        typename GpuAccelerationStructure<HitRayState, AnyHitRayState>::OnAnyHit anyHit,
        typename GpuAccelerationStructure<HitRayState, AnyHitRayState>::OnAnyMiss anyMiss)
    {
        GpuAccelerationStructure<HitRayState, AnyHitRayState> a;
        a.m_shapes = m_shapes;
        a.m_triMode = m_triMode;
        a.m_onHit = std::move(hit);
        a.m_onMiss = std::move(miss);
        a.m_onAnyHit = std::move(anyHit);
        a.m_onAnyMiss = std::move(anyMiss);
        check(pgpu_init(m_device, &a.m_ctx), nullptr, "pgpu_init");

        // one merged vertex/index view for the builders; remember (shape, prim) of every merged triangle
        std::vector<float> pos;
        std::vector<uint32_t> idx;
        std::vector<pgpu_leafobj> triSource;
        for (uint32_t s = 0; s < m_shapes->size(); ++s) {
            const TriangleShape& sh = (*m_shapes)[s];
            uint32_t meshId = 0;
            check(pgpu_upload_mesh(a.m_ctx, sh.positions.data(), sh.positions.size() / 3, sh.indices.data(), sh.indices.size() / 3, &meshId), a.m_ctx, "pgpu_upload_mesh");
            const uint32_t base = uint32_t(pos.size() / 3);
            pos.insert(pos.end(), sh.positions.begin(), sh.positions.end());
            for (size_t i = 0; i < sh.indices.size(); ++i)
                idx.push_back(sh.indices[i] + base);
            for (uint32_t p = 0; p < sh.indices.size() / 3; ++p)
                triSource.push_back({ meshId, p });
        }
        const size_t nv = pos.size() / 3, nt = idx.size() / 3;
        std::vector<uint32_t> groupOf(nt);
        uint32_t nGroups = 0;
        check(pgpu_partition_prims(pos.data(), nv, idx.data(), nt, m_primsPerPoint, groupOf.data(), &nGroups), a.m_ctx, "pgpu_partition_prims");
        std::vector<std::vector<uint32_t>> groups(nGroups);
        for (uint32_t p = 0; p < nt; ++p)
            groups[groupOf[p]].push_back(p);

        std::vector<uint32_t> bvhIds(nGroups);
        std::vector<std::vector<pgpu_leafobj>> objs(nGroups);
        std::vector<float> bounds(6 * size_t(nGroups));
        for (uint32_t g = 0; g < nGroups; ++g) {
            pgpu_host_bvh8* hb = nullptr;
            check(pgpu_build_bvh8(pos.data(), nv, idx.data(), nt, groups[g].data(), groups[g].size(), 4, &hb), a.m_ctx, "pgpu_build_bvh8");
            for (uint32_t p : groups[g])
                objs[g].push_back(triSource[p]);
            size_t nNodes = 0, nIdx = 0;
            const void* nodes = pgpu_host_bvh8_nodes(hb, &nNodes);
            const uint32_t* li = pgpu_host_bvh8_leaf_indices(hb, &nIdx);
            const int rc = pgpu_upload_bvh8(a.m_ctx, nodes, nNodes, li, nIdx, pgpu_host_bvh8_root(hb), objs[g].data(), objs[g].size(), &bvhIds[g]);
            pgpu_host_bvh8_free(hb);
            check(rc, a.m_ctx, "pgpu_upload_bvh8");
            float* b = &bounds[6 * size_t(g)];
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
        }
        if (nGroups < 2) {
            // the reference refuses a single batching point (pauseable_bvh4.h:180); serve it single-level
            a.m_twoLevel = false;
            a.m_bvhOfLeaf = { bvhIds[0] };
            a.m_objsOfLeaf = { objs[0] };
            return a;
        }
        if (m_svdagRes > 0) {
            std::vector<uint32_t> flat;
            std::vector<uint64_t> offs { 0 };
            for (auto& g : groups) {
                flat.insert(flat.end(), g.begin(), g.end());
                offs.push_back(flat.size());
            }
            pgpu_host_svdags* hs = nullptr;
            check(pgpu_build_svdags(pos.data(), nv, idx.data(), nt, flat.data(), offs.data(), nGroups, m_svdagRes, &hs), a.m_ctx, "pgpu_build_svdags");
            size_t nw = 0, nd = 0;
            const uint32_t* w = pgpu_host_svdags_words(hs, &nw);
            const pgpu_svdag_desc* d = pgpu_host_svdags_descs(hs, &nd);
            const int rc = pgpu_upload_svdags(a.m_ctx, w, nw, d, nd);
            pgpu_host_svdags_free(hs);
            check(rc, a.m_ctx, "pgpu_upload_svdags");
        }
        pgpu_host_bvh4* top = nullptr;
        check(pgpu_build_top_bvh4(bounds.data(), nGroups, &top), a.m_ctx, "pgpu_build_top_bvh4");
        size_t nNodes = 0, nLeaves = 0;
        const void* nodes = pgpu_host_bvh4_nodes(top, &nNodes);
        const uint32_t* order = pgpu_host_bvh4_leaf_order(top, &nLeaves);
        std::vector<pgpu_leaf> leaves(nLeaves);
        a.m_bvhOfLeaf.resize(nLeaves);
        a.m_objsOfLeaf.resize(nLeaves);
        for (size_t k = 0; k < nLeaves; ++k) {
            leaves[k] = { bvhIds[order[k]], m_svdagRes > 0 ? order[k] : PGPU_NO_SVDAG };
            a.m_bvhOfLeaf[k] = bvhIds[order[k]];
            a.m_objsOfLeaf[k] = objs[order[k]];
        }
        const int rc = pgpu_upload_top_bvh4(a.m_ctx, nodes, nNodes, pgpu_host_bvh4_root(top), leaves.data(), nLeaves);
        pgpu_host_bvh4_free(top);
        check(rc, a.m_ctx, "pgpu_upload_top_bvh4");
        a.m_twoLevel = true;
        return a;
    }

private:
    const std::vector<TriangleShape>* m_shapes;
    unsigned m_primsPerPoint, m_svdagRes;
    int m_device, m_triMode;
};

} // namespace pandora_amd
