"""pgpu_build_bvh8_device (pandora_amd/csrc/bvh_build.hip): the BVH8s of many batching points built on the GPU. Topology is the
builder's own (as the host builder's: the reference's comes from Embree), so the checks are the structural invariants of the
reference's layout, the octant words against the oracle's restatement of wive_bvh8_build8_impl.h:172-203, and traces through the
device-built trees: bit-exact against the oracle on the same tree, same nearest distance as through the host builder's tree."""
import numpy as np
import pytest

import oracle_binding as ob
import pandora_amd as pa

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


@pytest.fixture(scope="module")
def ctx():
    import torch  # (first: torch brings its own HIP runtime, and whichever library loads one first decides for the process)
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    c = pa.accel.Context(0)
    yield c
    c.close()


@pytest.fixture(params=["ploc", "lbvh"], autouse=True)
def binary_tree(request, monkeypatch):
    """Every test with both binary trees under the collapse: locally-ordered clustering (the default) and the radix tree."""
    monkeypatch.setenv("PANDORA_BVH8_DEVICE_TREE", request.param)
    return request.param


def check_tree(hb, mesh, max_leaf):
    """Every triangle of the group exactly once, handles in range, children packed at the front, stored child boxes equal to the
    exact bounds below them, absent children zero, octant words by the reference's rule. -> depth"""
    nu = hb.nodes.view(np.uint32)
    tri_lo = mesh.pos[mesh.idx].min(axis=1)
    tri_hi = mesh.pos[mesh.idx].max(axis=1)
    seen_nodes, slots = set(), []
    lo_of, hi_of = {}, {}
    order = []
    stack = [(hb.root, 0)]
    depth = 0
    while stack:
        h, d = stack.pop()
        flags, i = h >> 29, h & 0x1FFFFFFF
        if flags == 2:
            assert i < hb.n_nodes and i not in seen_nodes
            seen_nodes.add(i)
            order.append(i)
            depth = max(depth, d + 1)
            k = int((nu[i, 48:56] != 0).sum())
            assert k >= 2 and np.all(nu[i, 48:48 + k] != 0), "children are packed at the front"
            assert np.all(hb.nodes[i, 0:48].reshape(6, 8)[:, k:] == 0), "absent children are zero boxes"
            for c in range(k):
                stack.append((int(nu[i, 48 + c]), d + 1))
        else:
            assert flags & 4, "reserved handle flags"
            cnt = (flags & 3) + 1
            assert cnt <= max_leaf and i + cnt <= hb.leaf_idx.size
            slots.extend(range(i, i + cnt))
    assert len(seen_nodes) == hb.n_nodes
    assert sorted(slots) == list(range(hb.leaf_idx.size))
    assert sorted(hb.leaf_idx.tolist()) == list(range(hb.prim_ids.size))
    assert depth == hb.depth and 7 * depth + 8 <= 96

    def leaf_bounds(h):
        i, cnt = h & 0x1FFFFFFF, ((h >> 29) & 3) + 1
        prims = hb.prim_ids[hb.leaf_idx[i:i + cnt]]
        return tri_lo[prims].min(axis=0), tri_hi[prims].max(axis=0)

    for i in reversed(order):  # children before parents (a parent is pushed before its children)
        n = hb.nodes[i]
        k = int((nu[i, 48:56] != 0).sum())
        los, his = [], []
        for c in range(k):
            h = int(nu[i, 48 + c])
            lo, hi = (lo_of[h & 0x1FFFFFFF], hi_of[h & 0x1FFFFFFF]) if (h >> 29) == 2 else leaf_bounds(h)
            assert np.array_equal(np.array([n[c], n[16 + c], n[32 + c]], np.float32), lo.astype(np.float32)), (i, c)
            assert np.array_equal(np.array([n[8 + c], n[24 + c], n[40 + c]], np.float32), hi.astype(np.float32)), (i, c)
            los.append(lo)
            his.append(hi)
        lo_of[i], hi_of[i] = np.min(los, axis=0), np.max(his, axis=0)
        blo = np.ascontiguousarray(np.stack([n[0:8], n[16:24], n[32:40]], 1), np.float32)
        bhi = np.ascontiguousarray(np.stack([n[8:16], n[24:32], n[40:48]], 1), np.float32)
        want = np.zeros(8, np.uint32)
        ob.oracle().orc_bvh8_perm(blo.ctypes.data, bhi.ctypes.data, k, want.ctypes.data)
        assert np.array_equal(nu[i, 56:64], want), i
    return depth


@pytest.mark.parametrize("max_leaf", [1, 4])
def test_device_built_tree_has_the_reference_layout(ctx, max_leaf):
    mesh = pa.scenes.g_small()
    (hb,), host_built = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, None, max_leaf=max_leaf)
    assert host_built == 0
    check_tree(hb, mesh, max_leaf)


def test_traces_through_a_device_built_tree_match_the_oracle_and_the_host_tree(ctx):
    mesh = pa.scenes.g_atrium(40_000)
    (hb,), host_built = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)
    assert host_built == 0
    check_tree(hb, mesh, 4)
    host = pa.accel.build_bvh8(mesh.pos, mesh.idx)
    rays = pa.scenes.primary_rays((3, 3, 9), (28, 5, 9), (0, 1, 0), 70, 192, 192)
    rays = np.concatenate([rays, pa.scenes.random_rays(mesh.pos.min(0), mesh.pos.max(0), 30_000, seed=5)], axis=1)
    mid = ctx.upload_mesh(mesh.pos, mesh.idx)
    got = ctx.trace_bvh8(ctx.upload_bvh8(hb, mid), rays)
    want = ob.OracleBvh8.from_host(hb, mesh).trace(rays)
    for k in ("t", "obj", "u", "v"):
        assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
    ref = ob.OracleBvh8.from_host(host, mesh).trace(rays)
    # The nearest distance does not depend on the tree but for a handful of rays that differ in the last bits: two triangles an
    # ulp apart along the ray, and a node whose entry distance equals the current hit distance is skipped -- which of the two is
    # seen first depends on the tree (tests/test_builder.py compares the host builder's two constructions the same way).
    differ = np.nonzero(got["t"].view(np.uint32) != ref["t"].view(np.uint32))[0]
    assert differ.size <= 2e-4 * rays.shape[1], differ.size
    assert np.allclose(got["t"][differ], ref["t"][differ], rtol=1e-5, atol=0)
    any_got = ctx.trace_bvh8(ctx.upload_bvh8(hb, mid), rays, kind=pa.capi.ANY)["hit"]
    any_ref = ob.OracleBvh8.from_host(host, mesh).trace(rays, any_hit=True)["hit"]
    assert np.array_equal(any_got, ob.OracleBvh8.from_host(hb, mesh).trace(rays, any_hit=True)["hit"])
    assert np.count_nonzero(any_got != any_ref) <= 2e-4 * rays.shape[1]
    # quality: what a ray pays in this tree against the host builder's (binary SAH tree collapsed by dynamic programming)
    cd, ch = want["counters"], ref["counters"]
    assert cd["inner_visits"] < 1.6 * ch["inner_visits"], (cd, ch)


def test_many_groups_in_one_build_and_a_two_level_scene_on_them(ctx):
    mesh = pa.scenes.g_city(4, 6000)
    group_of, ng = pa.accel.partition_prims(mesh.pos, mesh.idx, 4000)
    groups = [np.nonzero(group_of == g)[0].astype(np.uint32) for g in range(ng)]
    assert ng > 8
    bvhs, host_built = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, groups)
    assert host_built == 0 and len(bvhs) == ng
    for hb, g in zip(bvhs, groups):
        assert np.array_equal(hb.prim_ids, g)
        check_tree(hb, mesh, 4)
    # the same groups one at a time give the same trees (a group's tree does not depend on its neighbours in the build)
    for gi in (0, ng // 2, ng - 1):
        (alone,), _ = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, [groups[gi]])
        assert np.array_equal(alone.nodes.view(np.uint32), bvhs[gi].nodes.view(np.uint32)) and np.array_equal(alone.leaf_idx, bvhs[gi].leaf_idx)
    c2 = pa.accel.Context(0)
    try:
        tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 4000, svdag_res=16, bvh_device_ctx=c2, svdag_device_ctx=c2)
        tl.upload(c2, c2.upload_mesh(mesh.pos, mesh.idx))
        lo, hi = mesh.pos.min(0), mesh.pos.max(0)
        rays = pa.scenes.random_rays(lo, hi, 60_000, seed=9)
        got, cnt = c2.trace(rays)
        want = ob.scene_from_two_level(tl, mesh).trace(rays)
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            assert np.array_equal(np.asarray(got[k]).view(np.uint32), want[k].view(np.uint32)), k
        assert (want["obj"].view(np.uint32) != MISS).mean() > 0.3
    finally:
        c2.close()


def test_degenerate_groups(ctx):
    rng = np.random.default_rng(3)
    # 5000 coincident triangles (every key equal: the positions break the ties, the tree is balanced), a single triangle,
    # three triangles (the root is a leaf handle), and a sliver of triangles along one axis
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    pos = [np.tile(tri, (5000, 1)), tri + 5, np.tile(tri, (3, 1)) + np.repeat(np.arange(3, dtype=np.float32), 3)[:, None] * 2 + 10]
    line = np.tile(tri * 1e-3, (700, 1))
    line[:, 0] += np.repeat(rng.uniform(0, 100, 700).astype(np.float32), 3)
    pos.append(line)
    pos = np.concatenate(pos).astype(np.float32)
    idx = np.arange(pos.shape[0], dtype=np.uint32).reshape(-1, 3)
    counts = [5000, 1, 3, 700]
    starts = np.concatenate([[0], np.cumsum(counts)])
    groups = [np.arange(starts[i], starts[i + 1], dtype=np.uint32) for i in range(4)]
    mesh = pa.scenes.Mesh(pos, idx)
    bvhs, host_built = pa.accel.build_bvh8_device(ctx, pos, idx, groups)
    assert host_built == 0
    for hb in bvhs:
        check_tree(hb, mesh, 4)
    assert bvhs[1].n_nodes == 0 and bvhs[1].root >> 29 == 4 and bvhs[2].n_nodes == 0 and bvhs[2].root >> 29 == 6
    assert bvhs[0].depth <= 6  # 5000 equal keys: 1250 leaves of four under a balanced tree
    with pytest.raises(pa.capi.PandoraError):
        pa.accel.build_bvh8_device(ctx, pos, idx, [np.array([idx.shape[0]], np.uint32)])  # triangle out of range
    with pytest.raises(pa.capi.PandoraError):
        pa.accel.build_bvh8_device(ctx, pos, idx, [groups[0], np.zeros(0, np.uint32)])  # empty group


def test_groups_too_deep_for_the_stack_go_to_the_host_builder(ctx, monkeypatch):
    mesh = pa.scenes.g_small()
    half = mesh.n_tris // 2
    groups = [np.arange(0, half, dtype=np.uint32), np.arange(half, mesh.n_tris, dtype=np.uint32), np.arange(0, 30, dtype=np.uint32)]
    monkeypatch.setenv("PANDORA_BVH8_DEVICE_LEVELS", "2")  # 2 levels hold 4 * 8 * 8 = 256 triangles at most
    bvhs, host_built = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, groups)
    assert host_built == 2 and bvhs[2].depth <= 2
    for hb, g in zip(bvhs, groups):
        assert np.array_equal(hb.prim_ids, g)
        check_tree(hb, mesh, 4)
    host = pa.accel.build_bvh8(mesh.pos, mesh.idx, groups[0])
    assert np.array_equal(host.nodes.view(np.uint32), bvhs[0].nodes.view(np.uint32)) and np.array_equal(host.leaf_idx, bvhs[0].leaf_idx)


def test_trees_built_and_left_on_the_device_trace_like_the_same_trees_uploaded(ctx):
    """pgpu_build_upload_bvh8_device: no host copy of the trees exists until pgpu_download_bvh8 fetches one; what comes back is
    what pgpu_build_bvh8_device returns for the same groups, and traces through the resident trees equal the oracle's."""
    mesh = pa.scenes.g_city(4, 6000)
    c2 = pa.accel.Context(0)
    try:
        c2.set_keep_shading(True)
        tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 4000, svdag_res=16, resident_ctx=c2, svdag_device_ctx=c2)
        assert tl.bvh_ids is not None and all(b is None for b in tl.bvhs)
        tl.upload(c2, 0)
        rays = pa.scenes.random_rays(mesh.pos.min(0), mesh.pos.max(0), 60_000, seed=11)
        got, cnt = c2.trace(rays)
        tl.fetch_bvhs(c2)
        ref, _ = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, tl.groups)
        for a, b in zip(tl.bvhs, ref):
            assert np.array_equal(a.nodes.view(np.uint32), b.nodes.view(np.uint32)) and np.array_equal(a.leaf_idx, b.leaf_idx)
            assert (a.root, a.depth) == (b.root, b.depth)
        want = ob.scene_from_two_level(tl, mesh).trace(rays)
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            assert np.array_equal(np.asarray(got[k]).view(np.uint32), want[k].view(np.uint32)), k
        any_got, _ = c2.trace(rays, kind=pa.capi.ANY)
        assert np.array_equal(np.asarray(any_got["hit"]), ob.scene_from_two_level(tl, mesh).trace(rays, any_hit=True)["hit"])
    finally:
        c2.close()


def test_single_resident_tree_and_the_host_fallback(ctx, monkeypatch):
    mesh = pa.scenes.g_small()
    rays = pa.scenes.primary_rays((0, 1, 3.2), (0, 1, 0), (0, 1, 0), 50, 96, 96)
    for levels, fell_back in ((None, 0), ("2", 1)):
        if levels:
            monkeypatch.setenv("PANDORA_BVH8_DEVICE_LEVELS", levels)
        c2 = pa.accel.Context(0)
        try:
            (bid,), host_built = c2.build_upload_bvh8_device(mesh.pos, mesh.idx)
            assert host_built == fell_back
            hb = c2.download_bvh8(bid, np.arange(mesh.n_tris, dtype=np.uint32))
            check_tree(hb, mesh, 4)
            got = c2.trace_bvh8(bid, rays)
            want = ob.OracleBvh8.from_host(hb, mesh).trace(rays)
            for k in ("t", "obj", "u", "v"):
                assert np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)), k
            assert (want["obj"] != ob.MISS).mean() > 0.5
        finally:
            c2.close()


def test_the_cut_on_the_device_is_a_partition_within_the_budget_and_traces_like_any_other(ctx):
    mesh = pa.scenes.g_city(4, 6000)
    for budget in (4000, 300, 1, 10**9):
        group_of, ng = pa.accel.partition_prims_device(ctx, mesh.pos, mesh.idx, budget)
        counts = np.bincount(group_of, minlength=ng)
        assert group_of.size == mesh.n_tris and counts.min() >= 1 and counts.max() <= budget and counts.sum() == mesh.n_tris
        assert ng == (1 if budget >= mesh.n_tris else ng) and (budget != 1 or ng == mesh.n_tris)
        again, ng2 = pa.accel.partition_prims_device(ctx, mesh.pos, mesh.idx, budget)
        assert ng2 == ng and np.array_equal(again, group_of)
    group_of, ng = pa.accel.partition_prims_device(ctx, mesh.pos, mesh.idx, 4000)
    host_ng = pa.accel.partition_prims(mesh.pos, mesh.idx, 4000)[1]
    assert ng < 3 * host_ng  # (an LBVH cut leaves small groups next to large siblings; it must not shatter the scene)
    # groups are spatially compact: the sum of the groups' box surfaces stays within a small factor of the host cut's
    def surface(gof, n):
        tri = mesh.pos[mesh.idx]
        order = np.argsort(gof, kind="stable")
        at = np.concatenate([[0], np.cumsum(np.bincount(gof, minlength=n))])[:-1]
        lo = np.minimum.reduceat(tri.min(axis=1)[order], at, axis=0)
        hi = np.maximum.reduceat(tri.max(axis=1)[order], at, axis=0)
        d = hi - lo
        return float((d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]).sum())
    assert surface(group_of, ng) < 3.0 * surface(*pa.accel.partition_prims(mesh.pos, mesh.idx, 4000))
    c2 = pa.accel.Context(0)
    try:
        tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 4000, svdag_res=16, cut_ctx=c2, resident_ctx=c2, svdag_device_ctx=c2)
        tl.upload(c2, 0)
        rays = pa.scenes.random_rays(mesh.pos.min(0), mesh.pos.max(0), 40_000, seed=13)
        got, _ = c2.trace(rays)
        want = ob.scene_from_two_level(tl.fetch_bvhs(c2), mesh).trace(rays)
        for k in ("t", "obj", "leaf", "u", "v", "visits"):
            assert np.array_equal(np.asarray(got[k]).view(np.uint32), want[k].view(np.uint32)), k
    finally:
        c2.close()


def _walk_only(hb):
    """Handles in range, every slot once (no geometry involved: for meshes whose bounds are not numbers)."""
    nu = hb.nodes.view(np.uint32)
    seen, slots, depth = set(), [], 0
    stack = [(hb.root, 0)]
    while stack:
        h, d = stack.pop()
        flags, i = h >> 29, h & 0x1FFFFFFF
        if flags == 2:
            assert i < hb.n_nodes and i not in seen
            seen.add(i)
            depth = max(depth, d + 1)
            stack.extend((int(c), d + 1) for c in nu[i, 48:56] if c != 0)
        else:
            assert flags & 4
            slots.extend(range(i, i + (flags & 3) + 1))
    assert len(seen) == hb.n_nodes and sorted(slots) == list(range(hb.leaf_idx.size)) and depth == hb.depth <= 12
    assert sorted(hb.leaf_idx.tolist()) == list(range(hb.prim_ids.size))


@pytest.mark.parametrize("max_leaf", [1, 4])
def test_hostile_meshes_through_the_device_builder(ctx, max_leaf):
    """The meshes and rays of tests/test_gpu_fuzz.py (zero-area, collinear, flat, huge, duplicated, tiny triangles; rays with zero /
    infinite / NaN components and inverted intervals) through device-built trees, returned and resident: valid trees, traces equal
    to the oracle's on the same tree bit for bit. Then vertices that are not numbers: the build must stay a tree and the trace must
    come back (what such triangles hit is nobody's contract)."""
    from test_gpu_fuzz import _bits, _nasty_mesh, _nasty_rays
    rng = np.random.default_rng(300 + max_leaf)
    pos, idx = _nasty_mesh(rng, 1600)
    mesh = pa.scenes.Mesh(pos, idx)
    third = idx.shape[0] // 3
    groups = [np.arange(0, third, dtype=np.uint32), np.arange(third, idx.shape[0], dtype=np.uint32)]
    bvhs, host_built = pa.accel.build_bvh8_device(ctx, pos, idx, groups, max_leaf=max_leaf)
    assert host_built == 0
    rays = _nasty_rays(rng, 30000)
    c2 = pa.accel.Context(0)
    try:
        ids, _ = c2.build_upload_bvh8_device(pos, idx, groups, max_leaf=max_leaf)
        with np.errstate(all="ignore"):
            for hb, bid, g in zip(bvhs, ids, groups):
                check_tree(hb, mesh, max_leaf)
                back = c2.download_bvh8(int(bid), g)
                assert np.array_equal(back.nodes.view(np.uint32), hb.nodes.view(np.uint32)) and np.array_equal(back.leaf_idx, hb.leaf_idx)
                orc = ob.OracleBvh8.from_host(hb, mesh)
                for tri in (pa.capi.TRI_MT, pa.capi.TRI_WATERTIGHT):
                    want = orc.trace(rays, tri_mode=tri, threads=8)
                    got = c2.trace_bvh8(int(bid), rays, tri_mode=tri)
                    for k in ("t", "obj", "u", "v"):
                        assert np.array_equal(_bits(got[k]), _bits(want[k])), (tri, k)
                    assert np.array_equal(c2.trace_bvh8(int(bid), rays, kind=pa.capi.ANY, tri_mode=tri)["hit"], orc.trace(rays, any_hit=True, tri_mode=tri, threads=8)["hit"])
        bad = pos.copy()
        sel = rng.random(bad.shape[0])
        bad[sel < 0.02] = np.nan
        bad[(sel >= 0.02) & (sel < 0.04), 1] = np.inf
        bad[(sel >= 0.04) & (sel < 0.05)] = -np.inf
        bad[(sel >= 0.05) & (sel < 0.06)] = np.float32(3e38)
        worse, _ = pa.accel.build_bvh8_device(ctx, bad, idx, groups, max_leaf=max_leaf)
        for hb in worse:
            _walk_only(hb)
        ids2, _ = c2.build_upload_bvh8_device(bad, idx, groups, max_leaf=max_leaf)
        for bid in ids2:
            out = c2.trace_bvh8(int(bid), rays)
            assert out["t"].shape[0] == rays.shape[1]
        all_nan = np.full_like(pos, np.nan)
        for hb in pa.accel.build_bvh8_device(ctx, all_nan, idx, groups, max_leaf=max_leaf)[0]:
            _walk_only(hb)
    finally:
        c2.close()


def test_both_collapses_give_valid_trees_and_the_tables_give_the_smaller_one(ctx, monkeypatch):
    """The collapse by cost tables (default) against the one that opens children by surface (PANDORA_BVH8_DEVICE_COLLAPSE=greedy; also
    the second pass for groups the tables make too deep -- forced here with a budget of 5 levels)."""
    mesh = pa.scenes.g_atrium(30_000)
    (by_tables,), _ = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)
    monkeypatch.setenv("PANDORA_BVH8_DEVICE_COLLAPSE", "greedy")
    (greedy,), _ = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)
    monkeypatch.delenv("PANDORA_BVH8_DEVICE_COLLAPSE")
    check_tree(by_tables, mesh, 4)
    check_tree(greedy, mesh, 4)
    assert by_tables.n_nodes < 0.9 * greedy.n_nodes
    rays = pa.scenes.random_rays(mesh.pos.min(0), mesh.pos.max(0), 20_000, seed=3)
    a = ob.OracleBvh8.from_host(by_tables, mesh).trace(rays)
    b = ob.OracleBvh8.from_host(greedy, mesh).trace(rays)
    assert np.count_nonzero(a["t"].view(np.uint32) != b["t"].view(np.uint32)) <= 2e-4 * rays.shape[1]
    assert a["counters"]["inner_visits"] < b["counters"]["inner_visits"] and a["counters"]["tri_tests"] < b["counters"]["tri_tests"]
    monkeypatch.setenv("PANDORA_BVH8_DEVICE_LEVELS", str(max(1, by_tables.depth - 1)))  # the tables' tree no longer fits: second pass (or the host)
    (second,), host_built = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx)
    check_tree(second, mesh, 4)
    assert second.depth <= by_tables.depth - 1 or host_built == 1


def test_a_clustering_that_does_not_finish_leaves_the_build_to_the_radix_tree(ctx, monkeypatch, binary_tree):
    """More rounds than the budget (1024; here 5, by PANDORA_BVH8_DEVICE_PLOC_ROUNDS): the radix tree is built instead, and the result
    is byte for byte what PANDORA_BVH8_DEVICE_TREE=lbvh gives."""
    if binary_tree != "ploc":
        pytest.skip("the radix tree has no rounds")
    mesh = pa.scenes.g_city(2, 3000)
    group_of, ng = pa.accel.partition_prims(mesh.pos, mesh.idx, 2000)
    groups = [np.nonzero(group_of == g)[0].astype(np.uint32) for g in range(ng)]
    monkeypatch.setenv("PANDORA_BVH8_DEVICE_PLOC_ROUNDS", "5")
    cut_short, host_built = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, groups)
    assert host_built == 0
    monkeypatch.delenv("PANDORA_BVH8_DEVICE_PLOC_ROUNDS")
    clustered, _ = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, groups)
    monkeypatch.setenv("PANDORA_BVH8_DEVICE_TREE", "lbvh")
    radix, _ = pa.accel.build_bvh8_device(ctx, mesh.pos, mesh.idx, groups)
    differs = 0
    for a, b, c, g in zip(cut_short, radix, clustered, groups):
        check_tree(a, mesh, 4)
        assert np.array_equal(a.nodes.view(np.uint32), b.nodes.view(np.uint32)) and np.array_equal(a.leaf_idx, b.leaf_idx)
        differs += a.n_nodes != c.n_nodes or not np.array_equal(a.leaf_idx, c.leaf_idx)
    assert differs > 0  # (the clustered trees are other trees)


@pytest.mark.parametrize("max_leaf", [1, 4])
def test_random_groups_of_every_small_size(ctx, max_leaf):
    """Groups of 1 ... 1000 triangles in random order, some made of copies of one triangle, some on a line: every group's tree has the
    reference's layout; a group that fits a leaf is a root handle, any other has fewer nodes than triangles."""
    rng = np.random.default_rng(17 + max_leaf)
    sizes = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 33, 64, 100, 257, 1000] * 3
    rng.shuffle(sizes)
    n = int(sum(sizes))
    pos = rng.uniform(-5, 5, (3 * n, 3)).astype(np.float32)
    idx = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    for g, s in enumerate(sizes):
        tri = slice(3 * starts[g], 3 * starts[g + 1])
        pos[tri] = pos[tri] * 0.2 + rng.uniform(-20, 20, 3).astype(np.float32)  # (groups apart from each other)
        if g % 5 == 1:
            pos[tri] = np.tile(pos[3 * starts[g]:3 * starts[g] + 3], (s, 1))  # copies of one triangle
        elif g % 5 == 2:
            pos[tri, 1:] = pos[3 * starts[g], 1:]  # on a line
    order = rng.permutation(n).astype(np.uint32)  # the groups name their triangles in no particular order
    rank = np.empty(n, np.int64)
    rank[order] = np.arange(n)
    groups = [np.ascontiguousarray(order[np.sort(rank[starts[g]:starts[g + 1]])]) for g in range(len(sizes))]
    for g in range(len(sizes)):
        assert np.array_equal(np.sort(groups[g]), np.arange(starts[g], starts[g + 1]))
    mesh = pa.scenes.Mesh(pos, idx)
    bvhs, host_built = pa.accel.build_bvh8_device(ctx, pos, idx, groups, max_leaf=max_leaf)
    assert host_built == 0
    for hb, g, s in zip(bvhs, groups, sizes):
        assert np.array_equal(hb.prim_ids, g)
        depth = check_tree(hb, mesh, max_leaf)
        if s <= max_leaf:
            assert hb.n_nodes == 0 and depth == 0
        elif s <= 8:
            assert 1 <= hb.n_nodes < s and depth < s  # (boxes without surface -- copies, a line -- cost nothing either way: any valid shape)
