"""Pins the CPU oracle against vectors produced by the reference's own code.

* tests/golden/pathtracer_js_golden.json -- made by importing the reference's
  src/libs/PathTracer.js under Node (tests/golden/gen_golden_js.js).
* oracle/_ref/bvh4_wide_ref -- the reference's tests/test.cpp compiled in place.
"""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import orc as orc_mod


def _u32(a):
    return np.asarray(a, dtype=np.uint64).astype(np.uint32)


def test_sizing_matches_reference(golden_js):
    # PathTracer.js:227-238
    for s in golden_js["sizing"]:
        n = s["numTris"]
        nn2 = 2 * n - 1 if n > 0 else 0
        assert s["bvh2"]["numNodes2"] == nn2
        assert s["bvh2"]["bytes"] == 4 * (1 + 6 * nn2)
        assert s["bvh4_of_numNodes2"]["bytes"] == 4 * (1 + 8 * nn2)


def test_morton_sort_matches_reference(orc, golden_js):
    # PathTracer.js:427-481
    assert len(golden_js["morton"]) >= 5
    for case in golden_js["morton"]:
        tris = _u32(case["tris_f32_bits"]).view(np.float32)
        m, t = orc.morton_sort(tris)
        assert np.array_equal(m, _u32(case["mortonSorted"])), case["name"]
        assert np.array_equal(t, _u32(case["triIndexSorted"])), case["name"]


def test_collapse_matches_reference(orc, golden_js):
    # PathTracer.js:506-667 incl. truncating f16 re-encode with subnormal flush
    assert len(golden_js["collapse"]) >= 8
    for case in golden_js["collapse"]:
        bvh2 = _u32(case["bvh2"])
        out, n4 = orc.collapse_bvh4(bvh2, case["numTris"])
        assert n4 == case["numNodes4"], case["name"]
        assert np.array_equal(out, _u32(case["bvh4"])), case["name"]


def test_f16_decode_all_patterns(orc):
    # PathTracer.js:16-40 == IEEE widening; numpy float16 is the independent check
    bits = np.arange(65536, dtype=np.uint16)
    ref = bits.view(np.float16).astype(np.float32)
    got = np.array([orc.lib.orc_f16_to_f32(int(b)) for b in bits], np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got[~nan].view(np.uint32), ref[~nan].view(np.uint32))


def test_f16_rtne_matches_numpy(orc):
    rng = np.random.default_rng(7)
    vals = np.concatenate([
        rng.standard_normal(20000).astype(np.float32) * np.float32(2.0),
        (rng.standard_normal(5000) * 1e-5).astype(np.float32),
        (rng.standard_normal(2000) * 3e4).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.99, 65520.0, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 2.99e-8, 6.1e-5, 6.097e-5], np.float32),
        np.arange(0, 65536, dtype=np.uint16).view(np.float16).astype(np.float32)[~np.isnan(np.arange(0, 65536, dtype=np.uint16).view(np.float16))],
    ])
    with np.errstate(over="ignore"):
        ref = vals.astype(np.float16).view(np.uint16)
    got = np.array([orc.lib.orc_f32_to_f16_rtne(float(v)) for v in vals], np.uint32).astype(np.uint16)
    assert np.array_equal(got, ref)


def test_f16_trunc_semantics(orc):
    # PathTracer.js:42-51: truncation, flush to signed zero below the normal range, saturate
    f = orc.lib.orc_f32_to_f16_trunc
    assert f(1.0) == 0x3C00
    assert f(1.0009765625) == 0x3C01          # exactly representable
    assert f(1.0019) == 0x3C01                # truncates (RTNE would give 0x3C02)
    assert f(-1.0019) == 0xBC01
    assert f(6.0e-5) == 0x0000                # f16-subnormal magnitude flushes to +0
    assert f(-6.0e-5) == 0x8000
    assert f(6.103515625e-5) == 0x0400        # smallest normal survives
    assert f(70000.0) == 0x7C00
    assert f(float("inf")) == 0x7C00
    assert f(float("-inf")) == 0xFC00


def test_increment_f16(orc):
    # BVHBuilder.wgsl:63-81
    inc = orc.lib.orc_increment_f16
    assert inc(1.0, 1) == np.float32(1.0009765625)
    assert inc(1.0, 0) == np.float32(0.99951171875)
    assert inc(-1.0, 0) == np.float32(-1.0009765625)
    assert inc(-1.0, 1) == np.float32(-0.99951171875)
    assert inc(0.0, 1) == np.float32(2.0 ** -24)
    # stepping down from +0 lands on -0 (ordered-u16 mapping), not on the negative subnormal
    z = np.float32(inc(0.0, 0))
    assert z == 0.0 and np.signbit(z)


REF_BIN = os.path.join(orc_mod.ORACLE_DIR, "_ref", "bvh4_wide_ref")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="oracle/_ref/bvh4_wide_ref not built (make -C oracle ref)")
@pytest.mark.parametrize("n", [1, 2, 4, 37, 1000])
def test_bvh4_wide_matches_reference_binary(orc, n):
    # tests/test.cpp:106-196 (compiled from the reference checkout, run here)
    rng = np.random.default_rng(n)
    tris = (rng.random((n, 9), dtype=np.float32) * 2 - 1).astype(np.float32)
    bvh2 = orc.build_lbvh2(tris)
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "BVH2.bin"), os.path.join(d, "BVH4_wide.bin")
        bvh2.tofile(a)
        subprocess.check_call([REF_BIN, a, b], stdout=subprocess.DEVNULL)
        ref = np.fromfile(b, dtype=np.uint32)
    assert np.array_equal(orc.bvh4_wide(bvh2), ref)


def test_bvh4_wide_tetra_known_answer(orc, golden_js):
    # SURVEY.md section 8c known answer: root kids [3,4,5,6]; nodes 1,2 keep [3,4,-,-]/[5,6,-,-]
    case = [c for c in golden_js["collapse"] if c["name"] == "tetra_survey"][0]
    w = orc.bvh4_wide(_u32(case["bvh2"]))
    assert w[0] == 7
    inv = 0xFFFFFFFF
    assert list(w[1 + 3: 1 + 7]) == [3, 4, 5, 6]
    assert list(w[1 + 8 + 3: 1 + 8 + 7]) == [3, 4, inv, inv]
    assert list(w[1 + 16 + 3: 1 + 16 + 7]) == [5, 6, inv, inv]
    assert list(w[1:4]) == [0xBC01BC01, 0x3C01BC01, 0x3C013C01]
