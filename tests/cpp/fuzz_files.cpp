// Host-side sanitizer fuzz of the flatbuffer readers (built with -fsanitize=address,undefined by tests/test_file_fuzz.py):
// mutated and truncated BVH8 / triangle-mesh files must be rejected or read without touching memory outside the buffer.
#include "../../pandora_amd/host/bvh8_file.cpp"
#include <cstdio>
#include <random>

int main()
{
    std::mt19937 rng(12345);
    std::vector<float> pos(300), nrm(300), tex(200);
    std::vector<uint32_t> idx(150);
    for (auto& x : pos) x = float(rng() % 1000) * 0.01f;
    for (auto& x : idx) x = rng() % 100;
    void* mesh = nullptr;
    size_t meshSize = 0;
    if (pgpu_mesh_serialize(pos.data(), 100, idx.data(), 50, nrm.data(), tex.data(), &mesh, &meshSize) != PGPU_OK)
        return 2;
    std::vector<unsigned char> nodes(256 * 5, 0);
    std::vector<uint32_t> leafIdx(40, 1);
    void* bvh = nullptr;
    size_t bvhSize = 0;
    if (pgpu_bvh8_serialize(nodes.data(), 5, leafIdx.data(), 40, 0x40000000u, 7, &bvh, &bvhSize) != PGPU_OK)
        return 3;
    unsigned long long accepted = 0, sum = 0;
    for (int it = 0; it < 200000; ++it) {
        const bool isMesh = it & 1;
        const unsigned char* src = static_cast<const unsigned char*>(isMesh ? mesh : bvh);
        const size_t full = isMesh ? meshSize : bvhSize;
        size_t n = (rng() % 8 == 0) ? rng() % (full + 1) : full;
        // exactly-sized heap copy so that any out-of-bounds read trips the sanitizer
        std::vector<unsigned char> buf(src, src + n);
        const int flips = int(rng() % 4);
        for (int f = 0; f < flips && n; ++f)
            buf[rng() % std::min<size_t>(n, 128)] = (unsigned char)rng();
        if (isMesh) {
            const float *p, *nr, *tx;
            const uint32_t* ix;
            size_t np, ni, nn, nt;
            float b[6];
            if (pgpu_mesh_deserialize(buf.data(), n, &p, &np, &ix, &ni, &nr, &nn, &tx, &nt, b) == PGPU_OK) {
                ++accepted;
                for (size_t i = 0; i < np * 3; ++i) sum += (unsigned long long)(p[i] != 0.0f);
                for (size_t i = 0; i < ni * 3; ++i) sum += ix[i];
                for (size_t i = 0; i < nn * 3; ++i) sum += (unsigned long long)(nr[i] != 0.0f);
                for (size_t i = 0; i < nt * 2; ++i) sum += (unsigned long long)(tx[i] != 0.0f);
            }
        } else {
            const void* nd;
            const uint32_t* li;
            size_t nn, nl;
            uint32_t root;
            uint64_t nobj;
            if (pgpu_bvh8_deserialize(buf.data(), n, &nd, &nn, &li, &nl, &root, &nobj) == PGPU_OK) {
                ++accepted;
                for (size_t i = 0; i < nn * 256; ++i) sum += static_cast<const unsigned char*>(nd)[i];
                for (size_t i = 0; i < nl; ++i) { uint32_t w; std::memcpy(&w, reinterpret_cast<const unsigned char*>(li) + 4 * i, 4); sum += w; }
            }
        }
    }
    pgpu_bytes_free(mesh);
    pgpu_bytes_free(bvh);
    std::printf("ok accepted=%llu checksum=%llu\n", accepted, sum);
    return 0;
}
