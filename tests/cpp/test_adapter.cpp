// Exercises the C++ host adapter the way an integrator uses the reference's interface:
// per-ray intersect()/intersectAny(), results through the four handlers, two bounces through flush().
// Exit code 0 = pass, 2 = no GPU (reported cleanly), 1 = failure.
#include "pandora_amd/gpu_acceleration_structure.hpp"
#include <atomic>
#include <cstdio>
#include <thread>

using namespace pandora_amd;

struct BounceState {
    int pixel;
    int depth;
};
struct ShadowState {
    int pixel;
};

static TriangleShape quad(float z, float half)
{
    TriangleShape s;
    s.positions = { -half, -half, z, half, -half, z, half, half, z, -half, half, z };
    s.indices = { 0, 1, 2, 0, 2, 3 };
    return s;
}

int main()
{
    std::vector<TriangleShape> shapes;
    for (int i = 0; i < 8; ++i)
        shapes.push_back(quad(-1.0f - float(i), 1.0f + 0.25f * float(i)));
    std::atomic<int> hits { 0 }, misses { 0 }, anyHits { 0 }, anyMisses { 0 }, bad { 0 };
    try {
        GpuAccelerationStructureBuilder builder(&shapes, 4, 16);
        using Accel = GpuAccelerationStructure<BounceState, ShadowState>;
        Accel* self = nullptr;
        Accel accel = builder.build<BounceState, ShadowState>(
            [&](const Ray& r, const RayHit& h, const BounceState& s) {
                ++hits;
                // rays start at z = +1 looking down -z: nearest quad is shape 0 at z = -1, t = 2
                if (s.depth == 0 && (h.shapeID != 0 || std::fabs(r.tfar - 2.0f) > 1e-6f || h.geometricNormal[2] <= 0))
                    ++bad;
                if (s.depth == 0) { // spawn a shadow ray back towards the origin side and one bounce
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
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return std::string(e.what()).find("pgpu_init") != std::string::npos ? 2 : 1;
    }
    std::printf("adapter ok\n");
    return 0;
}
