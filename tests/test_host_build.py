"""Host-side scene build of the product (libmi355pt, no GPU touched): checked against the
reference-generated golden vectors and, on larger seeded inputs, against the CPU oracle."""
import os
import re

import numpy as np
import pytest


def _u32(a):
    return np.asarray(a, dtype=np.uint64).astype(np.uint32)


def test_library_exports_every_declared_symbol(rt):
    """Both directions: every entry point include/mi355pt.h declares is exported, and the library's dynamic symbol table (`nm -D`)
    holds nothing that the header does not declare (the library is linked with -fvisibility=hidden and a version script)."""
    hdr = open(os.path.join(os.path.dirname(rt.__file__), "..", "include", "mi355pt.h")).read()
    declared = set(re.findall(r"^PT_API [^;(]*?\b(pt_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(rt.EXPORTS), declared ^ set(rt.EXPORTS)
    for name in declared:
        assert hasattr(rt.lib, name), name
    assert b"gfx950" in rt.lib.pt_version()
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", rt.LIB_PATH], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared, sorted(exported ^ declared)


def test_sizing(rt, golden_js):
    for s in golden_js["sizing"]:
        assert rt.compute_bvh2_sizing(s["numTris"]) == s["bvh2"]
        nn2 = s["bvh2"]["numNodes2"]
        assert rt.compute_bvh4_sizing(nn2) == s["bvh4_of_numNodes2"]


def test_morton_sort_golden(rt, golden_js):
    for case in golden_js["morton"]:
        tris = _u32(case["tris_f32_bits"]).view(np.float32)
        m, t = rt.morton_sort(tris)
        assert np.array_equal(m, _u32(case["mortonSorted"])), case["name"]
        assert np.array_equal(t, _u32(case["triIndexSorted"])), case["name"]


def test_collapse_golden(rt, golden_js):
    for case in golden_js["collapse"]:
        out, n4 = rt.collapse_lbvh2_to_bvh4(_u32(case["bvh2"]), case["numTris"])
        assert n4 == case["numNodes4"], case["name"]
        assert np.array_equal(out, _u32(case["bvh4"])), case["name"]


@pytest.mark.parametrize("n,seed", [(2, 1), (3, 2), (257, 3), (5000, 4), (40000, 5)])
def test_morton_collapse_wide_vs_oracle(rt, orc, n, seed):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 1, 3), dtype=np.float32) * 2 - 1
    tris = (c + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * 0.1).astype(np.float32).reshape(-1)
    if n > 100:  # duplicates exercise the index tie-break
        tris.reshape(n, 9)[10:20] = tris.reshape(n, 9)[30:40]
    m, t = rt.morton_sort(tris)
    om, ot = orc.morton_sort(tris)
    assert np.array_equal(m, om) and np.array_equal(t, ot)
    bvh2 = orc.build_lbvh2(tris, om, ot)
    b4, n4 = rt.collapse_lbvh2_to_bvh4(bvh2, n)
    ob4, on4 = orc.collapse_bvh4(bvh2, n)
    assert n4 == on4 and np.array_equal(b4, ob4)
    assert np.array_equal(rt.bvh2_to_bvh4_wide(bvh2), orc.bvh4_wide(bvh2))


def test_collapse_rejects_garbage(rt):
    bad = np.array([3, 0, 0, 0, 7, 9, 0] + [0] * 12, np.uint32)  # children out of range
    with pytest.raises(rt.PtError):
        rt.collapse_lbvh2_to_bvh4(bad, 2)


def test_bvh_file_roundtrip(rt, tmp_path):
    w = np.arange(1, 1000, dtype=np.uint32)
    p = str(tmp_path / "BVH2.bin")
    rt.write_u32_file(p, w)
    assert os.path.getsize(p) == w.size * 4
    assert np.array_equal(rt.read_u32_file(p), w)
    assert np.array_equal(np.fromfile(p, dtype="<u4"), w)  # raw LE u32 dump (src/server/api.js:27-31)
    with open(p, "ab") as f:
        f.write(b"x")
    with pytest.raises(rt.PtError):
        rt.read_u32_file(p)  # size not a multiple of 4 (tests/test.cpp:20)


@pytest.mark.parametrize("kind,n", [(0, 24), (0, 2001), (0, 50000), (1, 12000), (1, 30001)])
def test_procedural_scenes(rt, kind, n):
    t = rt.procedural_scene(kind, n).reshape(n, 3, 3)
    assert np.isfinite(t).all()
    assert t.min() >= -1.0 - 1e-6 and t.max() <= 1.0 + 1e-6
    ext = t.reshape(-1, 3).max(0) - t.reshape(-1, 3).min(0)
    assert abs(ext.max() - 2.0) < 1e-5          # normalize:"cube" (Scene.js:136-139)
    again = rt.procedural_scene(kind, n).reshape(n, 3, 3)
    assert np.array_equal(t, again)              # deterministic
    area = np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)
    assert (area > 0).mean() > 0.999


def test_dragon_class_is_closed(rt):
    # every edge of the base grid is shared by exactly two triangles (the T-junction splits aside)
    n = 2 * 40 * 6  # a*b grid with no remainder
    t = rt.procedural_scene(0, n).reshape(n, 3, 3)
    edges = {}
    for tri in t:
        k = [tuple(np.round(v, 6)) for v in tri]
        for a, b in ((0, 1), (1, 2), (2, 0)):
            e = tuple(sorted((k[a], k[b])))
            edges[e] = edges.get(e, 0) + 1
    assert set(edges.values()) == {2}


def test_tile_layout_partitions_frame(rt):
    for (w, h, count) in [(1920, 1080, 8), (256, 256, 2), (100, 60, 3)]:
        total = 0
        for r in range(count):
            nt, fl = rt.tile_layout(w, h, r, count)
            assert fl == nt * 256
            total += nt
        assert total == ((w + 7) // 8) * ((h + 7) // 8)


def test_no_device_fails_loudly(rt):
    import torch  # noqa: F401  (only to learn whether a GPU is visible)
    import torch.cuda
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rt.PtError) as e:
        rt.Context(0)
    assert e.value.code == 2 and "no CPU path" in str(e.value)
