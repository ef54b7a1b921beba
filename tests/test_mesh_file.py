"""The reference's triangle mesh chunk (triangle_mesh.fbs): writer/reader round trip, the structure of what the writer emits
checked with an independent parser written from the FlatBuffers format description, a foreign (differently laid out)
buffer, and robustness against truncated / corrupted input. Byte-identity with flatc output: parity unpinned (no flatc /
flatbuffers library in the image)."""
import struct

import numpy as np
import pytest

import pandora_amd as pa


def _table(buf, pos):
    (so,) = struct.unpack_from("<i", buf, pos)
    vt = pos - so
    vt_size, obj_size = struct.unpack_from("<HH", buf, vt)
    return vt, obj_size, [struct.unpack_from("<H", buf, vt + 4 + 2 * i)[0] for i in range((vt_size - 4) // 2)]


def _parse(buf):
    (root,) = struct.unpack_from("<I", buf, 0)
    _, _, f = _table(buf, root)
    out = {"root": root, "fields": f}
    for name, field, fmt, width in (("idx", 0, "<u4", 3), ("pos", 1, "<f4", 3), ("normals", 2, "<f4", 3), ("texcoords", 3, "<f4", 2)):
        if field < len(f) and f[field]:
            at = root + f[field]
            vec = at + struct.unpack_from("<I", buf, at)[0]
            (n,) = struct.unpack_from("<I", buf, vec)
            out[name] = np.frombuffer(buf, fmt, n * width, vec + 4).reshape(n, width)
            out[name + "_at"] = vec
        else:
            out[name] = None
    out["bounds"] = np.frombuffer(buf, "<f4", 6, root + f[4]) if len(f) > 4 and f[4] else None
    return out


def _mesh(with_attrs):
    m = pa.scenes.g_small()
    rng = np.random.default_rng(4)
    nrm = rng.standard_normal((m.pos.shape[0], 3)).astype(np.float32) if with_attrs else None
    tex = rng.random((m.pos.shape[0], 2), np.float32) if with_attrs else None
    return m, nrm, tex


@pytest.mark.parametrize("with_attrs", [False, True])
def test_round_trip_and_structure(with_attrs):
    m, nrm, tex = _mesh(with_attrs)
    data = pa.accel.mesh_to_bytes(m.pos, m.idx, nrm, tex)
    p = _parse(data)
    assert np.array_equal(p["idx"], m.idx) and np.array_equal(p["pos"].view(np.uint32), m.pos.view(np.uint32))
    if with_attrs:
        assert np.array_equal(p["normals"], nrm) and np.array_equal(p["texcoords"], tex)
        # vectors are created in the order indices, positions, normals, texCoords and the buffer grows backwards
        assert p["idx_at"] > p["pos_at"] > p["normals_at"] > p["texcoords_at"] > p["root"]
    else:
        assert p["normals"] is None and p["texcoords"] is None  # left out, as the reference does for empty ones
    assert np.array_equal(p["bounds"], np.concatenate([m.pos.min(0), m.pos.max(0)]))
    assert p["root"] % 4 == 0 and all(p[k + "_at"] % 4 == 0 for k in ("idx", "pos")) and len(data) % 4 == 0
    back = pa.accel.mesh_from_bytes(data)
    assert np.array_equal(back["idx"], m.idx) and np.array_equal(back["pos"].view(np.uint32), m.pos.view(np.uint32))
    assert np.array_equal(back["bounds"].ravel(), p["bounds"])
    if with_attrs:
        assert np.array_equal(back["normals"], nrm) and np.array_equal(back["texcoords"], tex)
    else:
        assert back["normals"].shape == (0, 3) and back["texcoords"].shape == (0, 2)
    # the mesh that comes back builds the same tree
    a, b = pa.accel.build_bvh8(m.pos, m.idx), pa.accel.build_bvh8(back["pos"], back["idx"])
    assert np.array_equal(a.nodes.view(np.uint32), b.nodes.view(np.uint32)) and np.array_equal(a.leaf_idx, b.leaf_idx)


def test_empty_mesh():
    data = pa.accel.mesh_to_bytes(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))
    back = pa.accel.mesh_from_bytes(data)
    assert back["pos"].shape == (0, 3) and back["idx"].shape == (0, 3)


def test_reads_a_foreign_layout():
    """A conforming buffer laid out differently from ours: built front to back, fields in declaration order, no bounds, no normals."""
    pos = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 2, 2]])
    idx = np.uint32([[0, 1, 2], [1, 2, 3]])
    tex = np.float32([[0, 0], [1, 0], [0, 1], [1, 1]])
    buf = bytearray(4)                                   # root offset, patched below
    vt = len(buf)
    buf += struct.pack("<HHHHHH", 12, 16, 4, 8, 0, 12)   # vtable: size 12, object 16; indices @4, positions @8, normals absent, texCoords @12
    table = len(buf)
    buf += struct.pack("<iIII", table - vt, 0, 0, 0)
    for field_at, arr in ((table + 4, idx), (table + 8, pos), (table + 12, tex)):
        vec = len(buf)
        buf += struct.pack("<I", arr.shape[0]) + arr.tobytes()
        struct.pack_into("<I", buf, field_at, vec - field_at)
    struct.pack_into("<I", buf, 0, table)
    back = pa.accel.mesh_from_bytes(bytes(buf))
    assert np.array_equal(back["pos"], pos) and np.array_equal(back["idx"], idx) and np.array_equal(back["texcoords"], tex)
    assert back["normals"].shape == (0, 3) and not back["bounds"].any()


def test_truncated_and_corrupted_buffers_are_rejected_or_read_safely():
    m, nrm, tex = _mesh(True)
    data = pa.accel.mesh_to_bytes(m.pos, m.idx, nrm, tex)
    for cut in (0, 3, 4, 11, 40, len(data) // 2, len(data) - 1):
        with pytest.raises(pa.capi.PandoraError):
            pa.accel.mesh_from_bytes(data[:cut])
    rng = np.random.default_rng(0)
    header = 96  # offsets, vtable and table live at the front of the file
    for _ in range(300):
        bad = bytearray(data)
        for _ in range(rng.integers(1, 4)):
            bad[rng.integers(0, header)] = rng.integers(0, 256)
        try:
            back = pa.accel.mesh_from_bytes(bytes(bad))  # must not read outside the buffer (it copies what it returns)
            assert back["pos"].size <= len(data) and back["idx"].size <= len(data)
        except pa.capi.PandoraError:
            pass


def test_bad_arguments_are_rejected():
    import ctypes as C
    L = pa.capi.lib()
    out, n = C.c_void_p(), C.c_size_t()
    pos = np.zeros((3, 3), np.float32)
    idx = np.zeros((1, 3), np.uint32)
    assert L.pgpu_mesh_serialize(None, 3, idx.ctypes.data, 1, None, None, C.byref(out), C.byref(n)) == pa.capi.ERR_INVALID
    assert L.pgpu_mesh_serialize(pos.ctypes.data, 3, None, 1, None, None, C.byref(out), C.byref(n)) == pa.capi.ERR_INVALID
    assert L.pgpu_mesh_serialize(pos.ctypes.data, 3, idx.ctypes.data, 1, None, None, None, C.byref(n)) == pa.capi.ERR_INVALID
    assert L.pgpu_mesh_deserialize(None, 0, *([None] * 9)) == pa.capi.ERR_INVALID
    data = pa.accel.mesh_to_bytes(pos, idx)
    buf = np.frombuffer(data + b"\0" * 8, np.uint8)
    ptrs = [C.c_void_p() for _ in range(2)]
    cnts = [C.c_size_t() for _ in range(2)]
    # a buffer that does not start on a 4-byte boundary cannot hand out float / u32 pointers
    assert L.pgpu_mesh_deserialize(buf.ctypes.data + 1, len(data), C.byref(ptrs[0]), C.byref(cnts[0]), C.byref(ptrs[1]), C.byref(cnts[1]), None, None, None, None, None) == pa.capi.ERR_INVALID
    # the optional outputs may be left out
    assert L.pgpu_mesh_deserialize(buf.ctypes.data, len(data), C.byref(ptrs[0]), C.byref(cnts[0]), C.byref(ptrs[1]), C.byref(cnts[1]), None, None, None, None, None) == pa.capi.OK
    assert cnts[0].value == 3 and cnts[1].value == 1
