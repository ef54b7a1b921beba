"""GPU parity: the SVDAGs built on the device (pgpu_build_svdags_device: voxeliser, occupancy pyramid, hash-consing in a device
table) against the host builder (pgpu_build_svdags, the restatement of TriangleShape::voxelize + constructSVOBreadthFirst +
compressDAGs). The two lay their nodes out in different orders; equal must be: every DAG's voxels, bounds, tree shape (the leaf
promotions decide how many trips a cast takes), the merged array's word count, and the traces through them."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    c = pa.accel.Context(0)
    yield c
    c.close()


def _groups(mesh, per_point):
    group_of, ng = pa.accel.partition_prims(mesh.pos, mesh.idx, per_point)
    order = np.argsort(group_of, kind="stable").astype(np.uint32)
    counts = np.bincount(group_of, minlength=ng)
    starts = np.concatenate([[0], np.cumsum(counts)])
    return [np.ascontiguousarray(order[starts[g]:starts[g + 1]]) for g in range(ng)]


def _shape(words, root):
    """The tree under `root` as the sequence of descriptor words in depth-first child order (layout-independent)."""
    out, todo = [], [int(root)]
    while todo:
        n = todo.pop()
        d = int(words[n])
        out.append(d)
        inner = ((d >> 8) & 0xFF) & ~(d & 0xFF)
        k = bin(inner).count("1")
        todo.extend(int(x) for x in words[n + 1:n + 1 + k][::-1])
    return out


def _same_dags(host, dev, check_shape_of=()):
    assert dev.words.size == host.words.size, f"merged array: {dev.words.size} words on the device, {host.words.size} on the host"
    assert dev.words_before == host.words_before
    assert len(dev.descs) == len(host.descs)
    for g, (dh, dd) in enumerate(zip(host.descs, dev.descs)):
        assert np.array_equal(np.asarray(dh["bounds_min"]), np.asarray(dd["bounds_min"])), f"dag {g}: bounds differ"
        assert np.float32(dh["inv_extent"]).view(np.uint32) == np.float32(dd["inv_extent"]).view(np.uint32), f"dag {g}: extent differs"
        assert int(dh["resolution"]) == int(dd["resolution"])
        gh, gd = pa.accel.svdag_expand(host.words, dh), pa.accel.svdag_expand(dev.words, dd)
        assert np.array_equal(gh, gd), f"dag {g}: {np.count_nonzero(gh != gd)} grid words differ"
    for g in check_shape_of:
        assert _shape(host.words, host.descs[g]["root"]) == _shape(dev.words, dev.descs[g]["root"]), f"dag {g}: tree shape differs"


@pytest.mark.parametrize("res", [4, 16, 64])
def test_device_builder_matches_host_builder_small_room(ctx, res):
    mesh = pa.scenes.g_small()
    groups = _groups(mesh, 64)
    host = pa.accel.build_svdags(mesh.pos, mesh.idx, groups, res)
    dev = pa.accel.build_svdags(mesh.pos, mesh.idx, groups, res, device_ctx=ctx)
    _same_dags(host, dev, check_shape_of=range(len(groups)))


def test_device_builder_matches_host_builder_city_and_atrium(ctx):
    city = pa.scenes.g_city(6, 4000, seed=5)
    groups = _groups(city, 16384)
    host = pa.accel.build_svdags(city.pos, city.idx, groups, 64)
    dev = pa.accel.build_svdags(city.pos, city.idx, groups, 64, device_ctx=ctx)
    _same_dags(host, dev, check_shape_of=(0, len(groups) // 2, len(groups) - 1))
    atrium = pa.scenes.g_atrium(65536)
    groups = _groups(atrium, 8192)
    host = pa.accel.build_svdags(atrium.pos, atrium.idx, groups, 128)
    dev = pa.accel.build_svdags(atrium.pos, atrium.idx, groups, 128, device_ctx=ctx)
    _same_dags(host, dev, check_shape_of=(0,))


def test_device_builder_edge_cases(ctx):
    # one axis-aligned triangle alone (flat bounds), a sliver, a group that fills its whole cube, a point-sized triangle
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0],
                    [5, 5, 5], [5.0000005, 5, 5], [5, 9, 5.0000005],
                    [2, 2, 2], [2, 2, 2], [2, 2, 2]], np.float32)
    idx = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.uint32)
    cube = pa.scenes.g_small()
    pos = np.concatenate([pos, cube.pos + 20.0]).astype(np.float32)
    idx = np.concatenate([idx, cube.idx + 9]).astype(np.uint32)
    groups = [np.array([0], np.uint32), np.array([1], np.uint32), np.array([2], np.uint32), np.arange(3, idx.shape[0], dtype=np.uint32),
              np.array([0, 1, 2], np.uint32)]
    for res in (8, 32):
        host = pa.accel.build_svdags(pos, idx, groups, res)
        dev = pa.accel.build_svdags(pos, idx, groups, res, device_ctx=ctx)
        _same_dags(host, dev, check_shape_of=range(len(groups)))
    with pytest.raises(pa.accel.PandoraError):
        pa.accel.build_svdags(pos, idx, groups, 48, device_ctx=ctx)  # not a power of two
    with pytest.raises(pa.accel.PandoraError):
        pa.accel.build_svdags(pos, idx, [np.array([10**7], np.uint32)], 8, device_ctx=ctx)  # primitive out of range


def test_traces_through_device_built_dags_match_oracle(ctx):
    mesh = pa.scenes.g_small()
    tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=32, svdag_device_ctx=ctx)
    c2 = pa.accel.Context(0)
    try:
        tl.upload(c2, c2.upload_mesh(mesh.pos, mesh.idx))
        orc = ob.scene_from_two_level(tl, mesh)
        rays = pa.scenes.random_rays((-3, -1, -3), (3, 3, 3), 20000, seed=11)
        got, cnt = c2.trace(rays)
        want = orc.trace(rays, lanes=3)
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            assert np.array_equal(np.asarray(got[k]).view(np.uint32), np.asarray(want[k]).view(np.uint32)), k
        for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
            assert cnt[k] == want["counters"][k], k
        # and the same counters as with host-built DAGs: the casts take the same decisions
        tl_host = pa.accel.build_two_level(mesh.pos, mesh.idx, 64, svdag_res=32)
        c3 = pa.accel.Context(0)
        try:
            tl_host.upload(c3, c3.upload_mesh(mesh.pos, mesh.idx))
            _, cnt_host = c3.trace(rays)
            for k in ("svdag_tests", "svdag_culled", "leaf_visits"):
                assert cnt[k] == cnt_host[k], k
        finally:
            c3.close()
    finally:
        c2.close()
