"""On-disk BVH8 (the reference's wive_bvh8.fbs flatbuffer): writer/reader round trip, the structure of what the writer
emits checked with an independent parser written from the FlatBuffers format description, a foreign (differently
laid out) buffer, and robustness against truncated / corrupted input. Byte-identity with flatc output: parity
unpinned (no flatc / flatbuffers library in the image)."""
import struct

import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa


@pytest.fixture(scope="module")
def small():
    mesh = pa.scenes.g_small()
    return mesh, pa.accel.build_bvh8(mesh.pos, mesh.idx)


def _table(buf, pos):
    (so,) = struct.unpack_from("<i", buf, pos)
    vt = pos - so
    vt_size, obj_size = struct.unpack_from("<HH", buf, vt)
    fields = [struct.unpack_from("<H", buf, vt + 4 + 2 * i)[0] for i in range((vt_size - 4) // 2)]
    return vt, obj_size, fields


def _parse(buf):
    (root,) = struct.unpack_from("<I", buf, 0)
    vt, _, f = _table(buf, root)
    out = {"root_pos": root, "root_vt": vt, "root_fields": f}

    def alloc(field):
        at = root + f[field]
        pos = at + struct.unpack_from("<I", buf, at)[0]
        avt, _, af = _table(buf, pos)
        vals = [struct.unpack_from("<I", buf, pos + af[i])[0] if i < len(af) and af[i] else 0 for i in range(3)]
        vat = pos + af[3]
        vec = vat + struct.unpack_from("<I", buf, vat)[0]
        (n,) = struct.unpack_from("<I", buf, vec)
        return {"pos": pos, "vt": avt, "max": vals[0], "block": vals[1], "cur": vals[2], "vec": vec, "bytes": bytes(buf[vec + 4:vec + 4 + n])}

    out["inner"], out["leaf"] = alloc(0), alloc(1)
    out["root_handle"] = struct.unpack_from("<I", buf, root + f[2])[0] if len(f) > 2 and f[2] else 0
    out["n_obj"] = struct.unpack_from("<Q", buf, root + f[3])[0] if len(f) > 3 and f[3] else 0
    return out


def test_round_trip_and_structure(small):
    mesh, hb = small
    data = pa.accel.bvh8_to_bytes(hb)
    p = _parse(data)
    assert p["inner"]["bytes"] == hb.nodes.tobytes() and p["leaf"]["bytes"] == hb.leaf_idx.tobytes()
    assert p["inner"]["cur"] == hb.n_nodes == p["inner"]["max"] and p["inner"]["block"] == 16
    assert p["leaf"]["cur"] == hb.leaf_idx.size and p["leaf"]["block"] == 4096
    assert p["root_handle"] == hb.root and p["n_obj"] == hb.prim_ids.size
    # alignment rules of the format: tables and vector length prefixes 4-aligned, the u64 8-aligned, the file a multiple of 8
    assert p["root_pos"] % 4 == 0 and p["inner"]["pos"] % 4 == 0 and p["leaf"]["pos"] % 4 == 0
    assert p["inner"]["vec"] % 4 == 0 and p["leaf"]["vec"] % 4 == 0
    assert (p["root_pos"] + p["root_fields"][3]) % 8 == 0 and len(data) % 8 == 0
    # FlatBufferBuilder shares identical vtables: both allocator tables use one
    assert p["inner"]["vt"] == p["leaf"]["vt"] != p["root_vt"]
    # build order (back to front): inner vector is the last thing in the file, the root offset the first
    assert p["inner"]["vec"] > p["inner"]["pos"] > p["leaf"]["vec"] > p["leaf"]["pos"] > p["root_pos"]
    back = pa.accel.bvh8_from_bytes(data, hb.prim_ids)
    assert np.array_equal(back.nodes.view(np.uint32), hb.nodes.view(np.uint32)) and np.array_equal(back.leaf_idx, hb.leaf_idx)
    assert (back.root, back.depth) == (hb.root, hb.depth)
    # and it still traces the same
    rays = pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), 3000, seed=9)
    a = ob.OracleBvh8.from_host(hb, mesh).trace(rays)
    b = ob.OracleBvh8.from_host(back, mesh).trace(rays)
    for k in ("t", "obj", "u", "v"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32))
    with pytest.raises(pa.capi.PandoraError):
        pa.accel.bvh8_from_bytes(data, hb.prim_ids[:-1])  # "Number of leaf objects does not match" (wive_bvh8_impl.h:31)


def test_reads_a_foreign_layout(small):
    """A conforming buffer laid out differently from ours: built front to back, private vtables, root handle 0 left out."""
    mesh, hb = small
    nodes, idx = hb.nodes.tobytes()[:512], hb.leaf_idx.tobytes()[:16]
    buf = bytearray(8)  # root offset + padding

    def vector(b):
        while len(buf) % 4:
            buf.append(0)
        at = len(buf)
        buf.extend(struct.pack("<I", len(b)) + b)
        return at

    def table(vt_fields, payload):
        # vtable first, then the table: soffset = table - vtable
        while len(buf) % 4:
            buf.append(0)
        vt = len(buf)
        buf.extend(struct.pack("<HH", 4 + 2 * len(vt_fields), 4 + len(payload)) + b"".join(struct.pack("<H", f) for f in vt_fields))
        while len(buf) % 8 != 4:
            buf.append(0)
        pos = len(buf)
        buf.extend(struct.pack("<i", pos - vt) + payload)
        return pos

    def allocator(data, count, block):
        # payload: [data offset @4][currentSize @8][maxSize @12][blockSize @16]; data offset patched afterwards
        pos = table([12, 16, 8, 4], struct.pack("<IIII", 0, count, count + 5, block))
        vec = vector(data)
        struct.pack_into("<I", buf, pos + 4, vec - (pos + 4))
        return pos

    # root: [numLeafObjects u64 @4][inner @12][leaf @16], compressedRootHandle absent (0)
    root = table([12, 16, 0, 4], struct.pack("<QII", 7, 0, 0))
    inner = allocator(nodes, 2, 16)
    leaf = allocator(idx, 4, 64)
    struct.pack_into("<I", buf, root + 12, inner - (root + 12))
    struct.pack_into("<I", buf, root + 16, leaf - (root + 16))
    struct.pack_into("<I", buf, 0, root)
    back = pa.accel.bvh8_from_bytes(bytes(buf), np.arange(7, dtype=np.uint32))
    assert back.n_nodes == 2 and back.nodes.tobytes() == nodes and back.leaf_idx.tobytes() == idx and back.root == 0


def test_rejects_damaged_buffers_without_crashing(small):
    mesh, hb = small
    data = pa.accel.bvh8_to_bytes(hb)
    head = len(data) - hb.nodes.nbytes - hb.leaf_idx.nbytes  # tables, vtables, offsets: everything but the payload tails
    for cut in list(range(0, 96)) + list(range(len(data) - 64, len(data))):
        try:
            pa.accel.bvh8_from_bytes(data[:cut], hb.prim_ids)
        except pa.capi.PandoraError:
            pass
    rng = np.random.default_rng(5)
    ok = 0
    for _ in range(400):
        b = bytearray(data)
        for _ in range(3):
            at = int(rng.integers(0, max(head, 64)))
            b[at] = int(rng.integers(0, 256))
        try:
            r = pa.accel.bvh8_from_bytes(bytes(b), hb.prim_ids)
            ok += 1
            assert r.n_nodes * 256 <= len(data) and r.leaf_idx.size * 4 <= len(data)
        except pa.capi.PandoraError:
            pass
    assert ok < 400  # most corruptions of the header region must be noticed


def test_reader_accepts_allocator_block_gaps():
    """The reference's ContiguousAllocatorTS hands out blocks per thread, so the arrays it serialises contain uninitialised
    gaps (contiguous_allocator_ts.h:151-198) and currentSize counts them. A file laid out that way must read back
    verbatim, and the tree in it must trace like the dense one (the gaps are never looked at)."""
    from bvh_helpers import with_gaps
    z, mesh, hb, orc = ob.load_golden_bvh8()
    dense = pa.accel.HostBvh8(np.ascontiguousarray(z["nodes"]), np.ascontiguousarray(z["leaf_idx"]), int(z["root"]), 0, np.ascontiguousarray(z["prim_ids"]))
    gapped = with_gaps(dense)
    assert gapped.n_nodes > dense.n_nodes and gapped.leaf_idx.size > dense.leaf_idx.size
    back = pa.accel.bvh8_from_bytes(pa.accel.bvh8_to_bytes(gapped), gapped.prim_ids)
    assert back.nodes.tobytes() == np.ascontiguousarray(gapped.nodes).tobytes() and np.array_equal(back.leaf_idx, gapped.leaf_idx)
    assert back.root == gapped.root and back.depth == pa.accel.host_bvh8_depth(dense.nodes, dense.root)
    got = ob.OracleBvh8.from_host(back, mesh).trace(z["rays"], threads=4)
    for k in ("t", "obj", "u", "v"):
        assert np.array_equal(got[k].view(np.uint32), z[f"tri0_{k}"].view(np.uint32)), k
