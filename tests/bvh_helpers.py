"""Tree surgery shared by the BVH8 file tests (CPU) and the GPU config tests."""
import numpy as np

import pandora_amd as pa


def with_gaps(hb):
    """The same tree as the reference's block allocators would leave it (contiguous_allocator_ts.h:151-198): nodes and leaf
    indices in blocks with uninitialised gaps between them -- here filled with garbage that must never be looked at."""
    rng = np.random.default_rng(3)
    nodes_u = hb.nodes.view(np.uint32)
    n = hb.n_nodes
    new_of = np.arange(n) + (np.arange(n) // 5) * 3  # 3 junk nodes after every 5 real ones
    out_nodes = rng.integers(0, 2**32, size=(int(new_of[-1]) + 4, 64), dtype=np.uint64).astype(np.uint32)
    handles = nodes_u[:, 48:56]
    leaf_handles = np.unique(handles[(handles >> 29) >= 4])
    if (hb.root >> 29) >= 4:
        leaf_handles = np.union1d(leaf_handles, [hb.root])
    slot_of, cursor = {}, 7
    for h in leaf_handles:
        slot_of[int(h)] = cursor
        cursor += int((h >> 29) & 3) + 1 + int(rng.integers(0, 6))
    out_idx = rng.integers(0, 2**32, size=cursor + 9, dtype=np.uint64).astype(np.uint32)
    for h, s in slot_of.items():
        c = ((h >> 29) & 3) + 1
        out_idx[s:s + c] = hb.leaf_idx[(h & 0x1FFFFFFF):(h & 0x1FFFFFFF) + c]

    def remap(h):
        h = int(h)
        if h == 0:
            return 0
        if (h >> 29) == 2:
            return (2 << 29) | int(new_of[h & 0x1FFFFFFF])
        return (h & 0xE0000000) | slot_of[h]

    for i in range(n):
        row = nodes_u[i].copy()
        row[48:56] = [remap(h) for h in row[48:56]]
        out_nodes[new_of[i]] = row
    return pa.accel.HostBvh8(out_nodes.view(np.float32), out_idx, remap(hb.root), hb.depth, hb.prim_ids)
