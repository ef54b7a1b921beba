// On-disk forms on either side of the traversal path: the cached bottom-level BVH8 and the triangle mesh chunk (at the end of this file).
// BVH8: the flatbuffer the reference writes for a cached BVH
//   table WiVeBVH8 { innerNodeAllocator, leafIndexAllocator : ContiguousAllocator; compressedRootHandle : uint32; numLeafObjects : ulong }
//   table ContiguousAllocator { maxSize, blockSize, currentSize : uint32; data : [byte] }
// (pandora/flatbuffers/wive_bvh8.fbs, contiguous_allocator.fbs; written by WiVeBVH8::serialize, wive_bvh8_impl.h:37-49,
// ContiguousAllocatorTS::serialize, contiguous_allocator_ts.h:200-205 and CachedBVH::serialize, offline_bvh_cache.cpp:114-122; read back with
// GetWiVeBVH8, offline_bvh_cache.cpp:138-143). `data` is the raw array of 256-byte nodes / of u32 leaf indices and `currentSize` counts
// elements, so the node bytes go to pgpu_upload_bvh8 unchanged.
//
// flatc and the flatbuffers library are not in this image. The writer follows the FlatBuffers binary format and the construction order of
// FlatBufferBuilder + the flatc-generated Create functions (buffer built back to front; fields pushed largest first, equal sizes in reverse
// declaration order; scalars equal to their default left out; identical vtables shared). The reader accepts any conforming buffer (absent
// fields read as 0, any field order) and bounds-checks every offset. Byte-identity with a flatc-built file: parity unpinned (nothing here
// can produce one).
#include "pandora_hip.h"
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

// Built back to front like FlatBufferBuilder: positions are distances from the END of the buffer.
class BackwardBuilder {
public:
    explicit BackwardBuilder(size_t capacity)
        : m_buf(capacity, 0)
        , m_head(capacity)
    {
    }
    size_t size() const { return m_buf.size() - m_head; }
    unsigned char* at(size_t pos) { return m_buf.data() + (m_buf.size() - pos); }
    static size_t padding(size_t size, size_t align) { return (~size + 1) & (align - 1); }
    void pad(size_t n) { m_head -= n; } // (zero-filled already)
    void align(size_t a)
    {
        m_minAlign = std::max(m_minAlign, a);
        pad(padding(size(), a));
    }
    void preAlign(size_t len, size_t a)
    {
        m_minAlign = std::max(m_minAlign, a);
        pad(padding(size() + len, a));
    }
    void pushBytes(const void* p, size_t n)
    {
        m_head -= n;
        if (n)
            std::memcpy(m_buf.data() + m_head, p, n);
    }
    template <typename T>
    size_t push(T v)
    {
        align(sizeof(T));
        pushBytes(&v, sizeof(T)); // little-endian host, as the format requires
        return size();
    }
    size_t createByteVector(const void* p, size_t n)
    {
        preAlign(n, 4); // so that the length prefix ends up 4-aligned
        pushBytes(p, n);
        return push<uint32_t>(uint32_t(n));
    }
    // CreateVectorOfStructs for structs of 4-byte scalars: `count` elements of `elemBytes` each, length prefix = count
    size_t createStructVector(const void* p, size_t count, size_t elemBytes)
    {
        preAlign(count * elemBytes, 4);
        pushBytes(p, count * elemBytes);
        return push<uint32_t>(uint32_t(count));
    }
    // AddStruct: the struct is stored inline in the table
    void addStruct(uint16_t voffset, const void* p, size_t bytes, size_t alignment)
    {
        align(alignment);
        pushBytes(p, bytes);
        m_fields.push_back({ voffset, size() });
    }
    void startTable()
    {
        m_tableStart = size();
        m_fields.clear();
    }
    template <typename T>
    void addScalar(uint16_t voffset, T v)
    {
        if (v == T(0))
            return; // defaults are not stored (force_defaults is off in the reference's builder)
        m_fields.push_back({ voffset, push<T>(v) });
    }
    void addOffset(uint16_t voffset, size_t target)
    {
        align(4);
        m_fields.push_back({ voffset, push<uint32_t>(uint32_t(size() - target + 4)) });
    }
    size_t endTable()
    {
        const size_t tablePos = push<int32_t>(0); // vtable offset, patched below
        uint16_t vtSize = 4;
        for (const Field& f : m_fields)
            vtSize = std::max<uint16_t>(vtSize, uint16_t(f.voffset + 2));
        std::vector<unsigned char> vt(vtSize, 0);
        const uint16_t objSize = uint16_t(tablePos - m_tableStart);
        std::memcpy(&vt[0], &vtSize, 2);
        std::memcpy(&vt[2], &objSize, 2);
        for (const Field& f : m_fields) {
            const uint16_t pos = uint16_t(tablePos - f.at);
            std::memcpy(&vt[f.voffset], &pos, 2);
        }
        size_t vtPos = 0;
        for (size_t known : m_vtables) { // an identical vtable written earlier is shared
            uint16_t knownSize;
            std::memcpy(&knownSize, at(known), 2);
            if (knownSize == vtSize && std::memcmp(at(known), vt.data(), vtSize) == 0) {
                vtPos = known;
                break;
            }
        }
        if (vtPos == 0) {
            pushBytes(vt.data(), vt.size());
            vtPos = size();
            m_vtables.push_back(vtPos);
        }
        const int32_t soffset = int32_t(int64_t(vtPos) - int64_t(tablePos));
        std::memcpy(at(tablePos), &soffset, 4);
        return tablePos;
    }
    void finish(size_t root)
    {
        preAlign(4, m_minAlign);
        align(4);
        push<uint32_t>(uint32_t(size() - root + 4));
    }
    const unsigned char* data() const { return m_buf.data() + m_head; }

private:
    struct Field {
        uint16_t voffset;
        size_t at;
    };
    std::vector<unsigned char> m_buf;
    size_t m_head, m_tableStart = 0, m_minAlign = 1;
    std::vector<Field> m_fields;
    std::vector<size_t> m_vtables;
};

constexpr uint16_t fieldVoffset(unsigned index) { return uint16_t(4 + 2 * index); }

size_t writeAllocator(BackwardBuilder& b, const void* data, size_t bytes, uint32_t count, uint32_t blockSize)
{
    const size_t vec = b.createByteVector(data, bytes);
    b.startTable(); // CreateContiguousAllocator: add_data, add_currentSize, add_blockSize, add_maxSize
    b.addOffset(fieldVoffset(3), vec);
    b.addScalar<uint32_t>(fieldVoffset(2), count);
    b.addScalar<uint32_t>(fieldVoffset(1), blockSize);
    b.addScalar<uint32_t>(fieldVoffset(0), count); // maxSize == currentSize, as after compact() (contiguous_allocator_ts.h:160-198)
    return b.endTable();
}

// ---- reader
struct View {
    const unsigned char* p;
    size_t n;
    bool ok(size_t off, size_t len) const { return off <= n && len <= n - off; }
    template <typename T>
    bool read(size_t off, T* out) const
    {
        if (!ok(off, sizeof(T)))
            return false;
        std::memcpy(out, p + off, sizeof(T));
        return true;
    }
};

struct Table {
    size_t pos = 0, vt = 0;
    uint16_t vtSize = 0, objSize = 0;
};

bool openTable(const View& v, size_t pos, Table* t)
{
    int32_t so;
    if (!v.read(pos, &so))
        return false;
    const int64_t vt = int64_t(pos) - so;
    if (vt < 0 || !v.ok(size_t(vt), 4))
        return false;
    t->pos = pos;
    t->vt = size_t(vt);
    if (!v.read(t->vt, &t->vtSize) || !v.read(t->vt + 2, &t->objSize) || t->vtSize < 4 || (t->vtSize & 1) || !v.ok(t->vt, t->vtSize)
        || t->objSize < 4 || !v.ok(pos, t->objSize))
        return false;
    return true;
}

// offset of field `index` inside the buffer, 0 when the field is absent (default value)
bool fieldAt(const View& v, const Table& t, unsigned index, size_t width, size_t* at)
{
    *at = 0;
    const size_t slot = 4 + 2 * size_t(index);
    if (slot + 2 > t.vtSize)
        return true;
    uint16_t off;
    if (!v.read(t.vt + slot, &off))
        return false;
    if (off == 0)
        return true;
    if (size_t(off) + width > t.objSize)
        return false;
    *at = t.pos + off;
    return true;
}

template <typename T>
bool scalarField(const View& v, const Table& t, unsigned index, T* out)
{
    size_t at;
    *out = T(0);
    if (!fieldAt(v, t, index, sizeof(T), &at))
        return false;
    return at == 0 || v.read(at, out);
}

bool offsetField(const View& v, const Table& t, unsigned index, size_t* target)
{
    size_t at;
    *target = 0;
    if (!fieldAt(v, t, index, 4, &at))
        return false;
    if (at == 0)
        return true;
    uint32_t rel;
    if (!v.read(at, &rel) || rel == 0 || !v.ok(at, rel))
        return false;
    *target = at + rel;
    return true;
}

struct AllocatorView {
    uint32_t maxSize = 0, blockSize = 0, currentSize = 0;
    const unsigned char* data = nullptr;
    uint32_t bytes = 0;
};

bool readAllocator(const View& v, size_t pos, AllocatorView* a)
{
    Table t;
    size_t vec;
    if (pos == 0 || !openTable(v, pos, &t) || !scalarField(v, t, 0, &a->maxSize) || !scalarField(v, t, 1, &a->blockSize)
        || !scalarField(v, t, 2, &a->currentSize) || !offsetField(v, t, 3, &vec))
        return false;
    if (vec == 0)
        return a->currentSize == 0;
    if ((vec & 3u) != 0 || !v.read(vec, &a->bytes) || !v.ok(vec + 4, a->bytes)) // (the format aligns every vector to 4 bytes)
        return false;
    a->data = v.p + vec + 4;
    return true;
}

} // namespace

extern "C" {

int pgpu_bvh8_serialize(const void* nodes256, size_t n_nodes, const uint32_t* leaf_indices, size_t n_leaf_indices, uint32_t root_handle,
    uint64_t num_leaf_objects, void** out_bytes, size_t* out_size)
{
    if (!out_bytes || !out_size || (n_nodes && !nodes256) || (n_leaf_indices && !leaf_indices) || n_nodes > 0x00FFFFFFu
        || n_leaf_indices > 0x3FFFFFFFu)
        return PGPU_ERR_INVALID;
    *out_bytes = nullptr;
    *out_size = 0;
    const size_t nodeBytes = n_nodes * 256, idxBytes = n_leaf_indices * 4;
    BackwardBuilder b(nodeBytes + idxBytes + 256);
    // WiVeBVH8::serialize: inner-node allocator, leaf-index allocator, then the root table
    const size_t inner = writeAllocator(b, nodes256, nodeBytes, uint32_t(n_nodes), 16);     // blockSize 16, wive_bvh8_impl.h:14
    const size_t leaf = writeAllocator(b, leaf_indices, idxBytes, uint32_t(n_leaf_indices), 4096); // default blockSize, contiguous_allocator_ts.h:21
    b.startTable(); // CreateWiVeBVH8: add_numLeafObjects (8 bytes first), add_compressedRootHandle, add_leafIndexAllocator, add_innerNodeAllocator
    b.addScalar<uint64_t>(fieldVoffset(3), num_leaf_objects);
    b.addScalar<uint32_t>(fieldVoffset(2), root_handle);
    b.addOffset(fieldVoffset(1), leaf);
    b.addOffset(fieldVoffset(0), inner);
    b.finish(b.endTable());
    void* out = std::malloc(std::max<size_t>(b.size(), 1));
    if (!out)
        return PGPU_ERR_INVALID;
    std::memcpy(out, b.data(), b.size());
    *out_bytes = out;
    *out_size = b.size();
    return PGPU_OK;
}

void pgpu_bytes_free(void* bytes) { std::free(bytes); }

int pgpu_bvh8_deserialize(const void* bytes, size_t size, const void** nodes256, size_t* n_nodes, const uint32_t** leaf_indices,
    size_t* n_leaf_indices, uint32_t* root_handle, uint64_t* num_leaf_objects)
{
    if (!bytes || !nodes256 || !n_nodes || !leaf_indices || !n_leaf_indices || !root_handle || !num_leaf_objects
        || (reinterpret_cast<uintptr_t>(bytes) & 3u) != 0)
        return PGPU_ERR_INVALID; // (the buffer itself must be 4-byte aligned, as any allocation or file mapping is)
    const View v { static_cast<const unsigned char*>(bytes), size };
    uint32_t rootRel;
    Table root;
    size_t innerPos, leafPos;
    AllocatorView inner, leaf;
    if (!v.read(0, &rootRel) || rootRel < 4 || !openTable(v, rootRel, &root) || !offsetField(v, root, 0, &innerPos) || !offsetField(v, root, 1, &leafPos)
        || !scalarField(v, root, 2, root_handle) || !scalarField(v, root, 3, num_leaf_objects) || !readAllocator(v, innerPos, &inner)
        || !readAllocator(v, leafPos, &leaf))
        return PGPU_ERR_INVALID;
    // currentSize counts elements (contiguous_allocator_ts.h:86-94): it must fit the stored bytes
    if (uint64_t(inner.currentSize) * 256 > inner.bytes || uint64_t(leaf.currentSize) * 4 > leaf.bytes)
        return PGPU_ERR_INVALID;
    // the node array is read in place: 4-byte alignment is all the format gives, the consumers copy (pgpu_upload_bvh8 does)
    *nodes256 = inner.data;
    *n_nodes = inner.currentSize;
    *leaf_indices = reinterpret_cast<const uint32_t*>(leaf.data);
    *n_leaf_indices = leaf.currentSize;
    return PGPU_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Triangle mesh chunk: table TriangleMesh { indices : [Vec3u]; positions : [Vec3]; normals : [Vec3]; texCoords : [Vec2]; bounds : Bounds }
// (pandora/flatbuffers/triangle_mesh.fbs, data_types.fbs), written by TriangleShape::serialize (shapes/triangle_shape.cpp:520-553: vectors
// created in the order indices, positions, normals, texCoords; absent normals / texCoords are left out; bounds inline) and read by
// TriangleShape::loadSerialized (:557-611) -- the geometry the traversal path's leaves point at.
int pgpu_mesh_serialize(const float* positions, size_t n_vertices, const uint32_t* indices, size_t n_triangles, const float* normals,
    const float* texcoords, void** out_bytes, size_t* out_size)
{
    if (!out_bytes || !out_size || (n_vertices && !positions) || (n_triangles && !indices) || n_vertices > 0x0FFFFFFFu || n_triangles > 0x0FFFFFFFu)
        return PGPU_ERR_INVALID;
    *out_bytes = nullptr;
    *out_size = 0;
    float bounds[6] = { 3.402823466e38f, 3.402823466e38f, 3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f }; // Bounds::reset
    for (size_t i = 0; i < n_vertices; ++i)
        for (int a = 0; a < 3; ++a) {
            bounds[a] = std::min(bounds[a], positions[3 * i + size_t(a)]);
            bounds[3 + a] = std::max(bounds[3 + a], positions[3 * i + size_t(a)]);
        }
    BackwardBuilder b(n_triangles * 12 + n_vertices * (12 + (normals ? 12 : 0) + (texcoords ? 8 : 0)) + 256);
    const size_t vIdx = b.createStructVector(indices, n_triangles, 12);
    const size_t vPos = b.createStructVector(positions, n_vertices, 12);
    const size_t vNrm = normals ? b.createStructVector(normals, n_vertices, 12) : 0;
    const size_t vTex = texcoords ? b.createStructVector(texcoords, n_vertices, 8) : 0;
    b.startTable(); // CreateTriangleMesh: every field is 4 bytes wide to the sorter, so reverse declaration order
    b.addStruct(fieldVoffset(4), bounds, sizeof(bounds), 4);
    if (vTex)
        b.addOffset(fieldVoffset(3), vTex);
    if (vNrm)
        b.addOffset(fieldVoffset(2), vNrm);
    b.addOffset(fieldVoffset(1), vPos);
    b.addOffset(fieldVoffset(0), vIdx);
    b.finish(b.endTable());
    void* out = std::malloc(std::max<size_t>(b.size(), 1));
    if (!out)
        return PGPU_ERR_INVALID;
    std::memcpy(out, b.data(), b.size());
    *out_bytes = out;
    *out_size = b.size();
    return PGPU_OK;
}

int pgpu_mesh_deserialize(const void* bytes, size_t size, const float** positions, size_t* n_vertices, const uint32_t** indices,
    size_t* n_triangles, const float** normals, size_t* n_normals, const float** texcoords, size_t* n_texcoords, float* bounds6)
{
    if (!bytes || !positions || !n_vertices || !indices || !n_triangles || (reinterpret_cast<uintptr_t>(bytes) & 3u) != 0)
        return PGPU_ERR_INVALID; // (the buffer itself must be 4-byte aligned, as any allocation or file mapping is)
    const View v { static_cast<const unsigned char*>(bytes), size };
    uint32_t rootRel;
    Table root;
    if (!v.read(0, &rootRel) || rootRel < 4 || !openTable(v, rootRel, &root))
        return PGPU_ERR_INVALID;
    auto structVector = [&](unsigned field, size_t elemBytes, const unsigned char** data, size_t* count) {
        size_t vec;
        uint32_t n = 0;
        *data = nullptr;
        *count = 0;
        if (!offsetField(v, root, field, &vec))
            return false;
        if (vec == 0)
            return true; // absent: an empty vector
        if ((vec & 3u) != 0 || !v.read(vec, &n) || uint64_t(n) * elemBytes > size || !v.ok(vec + 4, size_t(n) * elemBytes))
            return false; // (the format aligns every vector to 4 bytes: the typed pointers handed back rely on it)
        *data = v.p + vec + 4;
        *count = n;
        return true;
    };
    const unsigned char *pIdx, *pPos, *pNrm, *pTex;
    size_t nIdx, nPos, nNrm, nTex;
    if (!structVector(0, 12, &pIdx, &nIdx) || !structVector(1, 12, &pPos, &nPos) || !structVector(2, 12, &pNrm, &nNrm) || !structVector(3, 8, &pTex, &nTex))
        return PGPU_ERR_INVALID;
    size_t at;
    if (!fieldAt(v, root, 4, 24, &at) || !v.ok(at, at ? 24 : 0))
        return PGPU_ERR_INVALID;
    if (bounds6) {
        if (at)
            std::memcpy(bounds6, v.p + at, 24);
        else
            std::memset(bounds6, 0, 24);
    }
    // (values are read in place: 4-byte alignment relative to the start of the buffer is what the format gives)
    *indices = reinterpret_cast<const uint32_t*>(pIdx);
    *n_triangles = nIdx;
    *positions = reinterpret_cast<const float*>(pPos);
    *n_vertices = nPos;
    if (normals)
        *normals = reinterpret_cast<const float*>(pNrm);
    if (n_normals)
        *n_normals = nNrm;
    if (texcoords)
        *texcoords = reinterpret_cast<const float*>(pTex);
    if (n_texcoords)
        *n_texcoords = nTex;
    return PGPU_OK;
}
}
