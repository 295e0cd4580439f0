"""Structural invariants of the BVH buffers the host side of libmi355pt produces (no GPU, no oracle in the assertions): what
renderer.wgsl relies on when it walks them.  The BVH2 input comes from the oracle's LBVH restatement; everything asserted here
is a property of the buffers, not a comparison with another implementation."""
import numpy as np
import pytest

LEAF = 0x80000000
INVALID = 0xFFFFFFFF


def half(bits):
    return np.asarray(bits, np.uint16).view(np.float16).astype(np.float32)


def unpack_bounds(w):
    w = np.asarray(w, np.uint32)
    return (np.stack([half(w[..., 0] & 0xFFFF), half(w[..., 0] >> 16), half(w[..., 1] & 0xFFFF)], -1),
            np.stack([half(w[..., 1] >> 16), half(w[..., 2] & 0xFFFF), half(w[..., 2] >> 16)], -1))


def soup(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 1, 3), dtype=np.float32) * 2 - 1
    return (c + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * 0.2).astype(np.float32).reshape(-1)


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (7, 2), (300, 3), (20000, 4)])
def test_morton_sort_is_a_stable_sort_of_a_permutation(rt, n, seed):
    tris = soup(n, seed)
    if n >= 300:
        tris.reshape(n, 9)[5:25] = tris.reshape(n, 9)[100]        # equal codes: ties broken by triangle index
    m, t = rt.morton_sort(tris)
    assert m.dtype == np.uint32 and len(m) == len(t) == n
    assert np.array_equal(np.sort(t), np.arange(n, dtype=np.uint32))
    assert np.all(m[:-1] <= m[1:]) and np.all(m < (1 << 30))
    ties = m[:-1] == m[1:]
    assert np.all(t[:-1][ties] < t[1:][ties])


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (3, 2), (50, 3), (4097, 4), (30000, 5)])
def test_collapsed_bvh4_is_a_preorder_tree_that_contains_its_triangles(rt, orc, n, seed):
    tris = soup(n, seed)
    m, t = rt.morton_sort(tris)
    bvh2 = orc.build_lbvh2(tris, m, t)
    b4, n4 = rt.collapse_lbvh2_to_bvh4(bvh2, n)
    assert b4[0] == n4 and len(b4) == 1 + 8 * n4 and n <= n4 <= 2 * n - 1
    rec = b4[1:].reshape(n4, 8)
    leaf = (rec[:, 7] & LEAF) != 0
    # one leaf per triangle, leaves have no children, internal nodes have 2..4 of them packed to the front
    assert leaf.sum() == n and np.array_equal(np.sort(rec[leaf, 7] & 0x7FFFFFFF), np.arange(n, dtype=np.uint32))
    assert np.all(rec[leaf, 3:7] == INVALID) and np.all(rec[~leaf, 7] == 0)
    kids = rec[:, 3:7]
    valid = kids != INVALID
    cnt = valid.sum(1)
    assert np.all(cnt[~leaf] >= 2) and np.all(cnt[~leaf] <= 4)
    assert np.all(valid[:, :-1] >= valid[:, 1:])                      # no hole before a child
    # DFS pre-order: a node's first child is the next node, every node except the root has exactly one parent, ids grow downwards
    internal = np.nonzero(~leaf)[0]
    assert np.array_equal(kids[internal, 0], internal.astype(np.uint32) + 1)
    all_kids = kids[valid]
    assert np.array_equal(np.sort(all_kids), np.arange(1, n4, dtype=np.uint32))
    parent_of = np.repeat(np.arange(n4), 4).reshape(n4, 4)[valid]
    assert np.all(all_kids > parent_of)
    # bounds: a leaf's box contains its triangle; a parent's box contains its children's boxes
    mn, mx = unpack_bounds(rec[:, 0:3])
    tri = tris.reshape(n, 3, 3)[rec[leaf, 7] & 0x7FFFFFFF]
    assert np.all(mn[leaf] <= tri.min(1)) and np.all(mx[leaf] >= tri.max(1))
    # ... except where the reference's f16 re-encode (PathTracer.js:42-51) flushes a child bound below the f16 normal range
    # (|x| < 2^-14) to zero: a quirk of the reference that the collapse reproduces bit for bit
    tiny = np.float32(2.0 ** -14)
    for s in range(4):
        has = valid[:, s]
        c = kids[has, s]
        ok_mn = (mn[has] <= mn[c]) | ((np.abs(mn[c]) < tiny) & (mn[has] == 0))
        ok_mx = (mx[has] >= mx[c]) | ((np.abs(mx[c]) < tiny) & (mx[has] == 0))
        assert np.all(ok_mn) and np.all(ok_mx)


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (5, 2), (1000, 3)])
def test_bvh4_wide_keeps_bvh2_nodes_and_adopts_grandchildren(rt, orc, n, seed):
    tris = soup(n, seed)
    m, t = rt.morton_sort(tris)
    bvh2 = orc.build_lbvh2(tris, m, t)
    wide = rt.bvh2_to_bvh4_wide(bvh2)
    nn2 = 2 * n - 1
    assert wide[0] == nn2 and len(wide) == 1 + 8 * nn2
    r2, r4 = bvh2[1:].reshape(nn2, 6), wide[1:].reshape(nn2, 8)
    assert np.array_equal(r4[:, 0:3], r2[:, 0:3])                     # bounds copied verbatim (tests/test.cpp:106-196)
    leaf2 = (r2[:, 5] & LEAF) != 0
    assert np.array_equal(r4[leaf2, 7], r2[leaf2, 5]) and np.all(r4[leaf2, 3:7] == INVALID)
    for i in np.nonzero(~leaf2)[0][:200]:
        want = []
        for c in (r2[i, 3], r2[i, 4]):
            want += [c] if leaf2[c] else [r2[c, 3], r2[c, 4]]
        got = [k for k in r4[i, 3:7] if k != INVALID]
        assert got == [int(w) for w in want], i
