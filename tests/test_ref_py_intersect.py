"""The oracle's slab test, Moller-Trumbore and f16 decode against vectors produced by the REFERENCE's own Python statement
of that arithmetic (tests/test.py:20-27, 64-76, 82-99 of the reference, run by tests/golden/gen_ref_py_intersect.py in the
build container; the fixture holds inputs and outputs only).  This pins SURVEY 8 rows A4 (decode), A5 and A7 to code the
reference holds.  Traversal order, ray generation and shade() have no reference-held statement and stay unpinned.

Tolerance: the slab test is the same sequence of IEEE f32 operations on both sides -> bit for bit.  The triangle test goes
through numpy's `dot` and `cross`, whose summation order is not pinned -> 1e-6 relative on t where the case is well
conditioned, and the f32 forward-error bound of the formula (`mt_error_bound`: rounding of the two cross products and three
dot products, amplified by 1/det) where cancellation makes it larger; cases whose |det| lies within 1e-6 relative of the
epsilon (or whose u, v, u + v lie that close to an edge) may legitimately differ in the hit flag."""
import json
import os

import numpy as np
import pytest

import orc as orc_mod

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_py_intersect.json")


@pytest.fixture(scope="module")
def fix():
    return json.load(open(FIX))


@pytest.fixture(scope="module")
def orc():
    return orc_mod.load()


def f32(bits):
    return np.asarray(bits, np.uint32).view(np.float32)


def mt_error_bound(o, d, v0, v1, v2):
    """Generous bound on |t_f32 - t_exact| for ONE f32 evaluation of tests/test.py:82-99 in any summation order (computed in f64)."""
    u = 2.0 ** -24
    O, D, A, B, Cc = (x.astype(np.float64) for x in (o, d, v0, v1, v2))
    e1, e2, s_ = B - A, Cc - A, O - A
    de1, de2, ds = u * np.abs(e1), u * np.abs(e2), u * np.abs(s_)                 # one rounding each

    def cross_err(a, b, da, db):                                              # |a x b| component-wise error
        aa, ab = np.abs(a), np.abs(b)
        m = np.array([aa[1] * ab[2] + aa[2] * ab[1], aa[2] * ab[0] + aa[0] * ab[2], aa[0] * ab[1] + aa[1] * ab[0]])
        dm = np.array([da[1] * ab[2] + da[2] * ab[1] + aa[1] * db[2] + aa[2] * db[1], da[2] * ab[0] + da[0] * ab[2] + aa[2] * db[0] + aa[0] * db[2],
                       da[0] * ab[1] + da[1] * ab[0] + aa[0] * db[1] + aa[1] * db[0]])
        return 4 * u * m + dm

    p, q = np.cross(D, e2), np.cross(s_, e1)
    dp, dq = cross_err(D, e2, 0 * D, de2), cross_err(s_, e1, ds, de1)
    det, num = float(e1 @ p), float(e2 @ q)
    ddet = 4 * u * float(np.abs(e1) @ np.abs(p)) + float(np.abs(e1) @ dp) + float(de1 @ np.abs(p))
    dnum = 4 * u * float(np.abs(e2) @ np.abs(q)) + float(np.abs(e2) @ dq) + float(de2 @ np.abs(q))
    if det == 0.0:
        return float("inf")
    return dnum / abs(det) + abs(num) * ddet / (det * det) + 4 * u * abs(num / det)


def test_slab_matches_the_reference_python_bit_for_bit(fix, orc):
    s = fix["slab"]
    inf = f32([fix["inf_bits"]])[0]
    o, inv, mn, mx = (f32(s[k]).reshape(-1, 3) for k in ("o", "inv", "mn", "mx"))
    ret = f32(s["ret"])
    assert len(ret) >= 1000 and (ret != inf).sum() > 100 and (ret == inf).sum() > 100
    for i in range(len(ret)):
        hit, tmin = orc.slab(o[i], inv[i], mn[i], mx[i])
        # tests/test.py:72-76: tmin when tmax >= max(tmin, 0), else INF
        if ret[i] == inf:
            assert not hit or tmin >= inf, "case %d: reference misses, oracle hits with tmin %r" % (i, tmin)
        else:
            assert hit, "case %d: reference hits (tmin %r), oracle misses" % (i, ret[i])
            assert np.float32(tmin).view(np.uint32) == np.float32(ret[i]).view(np.uint32), "case %d: tmin %r vs reference %r" % (i, tmin, ret[i])


def test_safe_inv_dir_matches_the_generator_inputs(fix, orc):
    """The reciprocals the fixture was generated with are the oracle's safeInvDir (renderer.wgsl:74-80; tests/test.py:151-154)
    of unit directions: spot-check the 1e30 sentinel and the sign handling on the axis-parallel cases."""
    inv = f32(fix["slab"]["inv"]).reshape(-1, 3)
    big = inv[np.isclose(np.abs(inv), 1e30).any(axis=1)]
    assert len(big) > 50
    for row in big[:50]:
        d = np.where(np.isclose(np.abs(row), 1e30), 0.0, 1.0 / row).astype(np.float32)
        assert np.array_equal(orc.safe_inv_dir(d).view(np.uint32), row.view(np.uint32))


def test_moller_trumbore_matches_the_reference_python(fix, orc):
    m = fix["moller_trumbore"]
    inf = f32([fix["inf_bits"]])[0]
    o, d, v0, v1, v2 = (f32(m[k]).reshape(-1, 3) for k in ("o", "d", "v0", "v1", "v2"))
    ret = f32(m["ret"])
    assert len(ret) >= 1000 and (ret != inf).sum() > 100
    eps, band = 1e-7, 1e-6
    compared, borderline, tight = 0, 0, 0
    for i in range(len(ret)):
        hit, t = orc.moller_trumbore(o[i], d[i], v0[i], v1[i], v2[i])
        ref_hit = ret[i] != inf
        if hit == ref_hit:
            if hit:
                tol = max(band * abs(float(ret[i])), 2.0 * mt_error_bound(o[i], d[i], v0[i], v1[i], v2[i]))      # two f32 evaluations, each within the bound
                assert abs(float(t) - float(ret[i])) <= tol, "case %d: t %r vs reference %r (tolerance %g)" % (i, t, ret[i], tol)
                tight += abs(float(t) - float(ret[i])) <= band * abs(float(ret[i]))
            compared += 1
            continue
        # a differing hit flag is only acceptable next to one of the routine's thresholds (evaluated in f64 here)
        O, D, A, B, Cc = (x.astype(np.float64) for x in (o[i], d[i], v0[i], v1[i], v2[i]))
        e1, e2 = B - A, Cc - A
        p = np.cross(D, e2); det = float(e1 @ p)
        near = abs(abs(det) - eps) <= band * eps * 10
        if not near and abs(det) > 0:
            s_ = O - A; u = float(s_ @ p) / det; q = np.cross(s_, e1); v = float(D @ q) / det; tt = float(e2 @ q) / det
            near = min(abs(u), abs(u - 1), abs(v), abs(u + v - 1)) <= 1e-5 or abs(tt - eps) <= 1e-6
        assert near, "case %d: hit flags differ (oracle %s, reference %s) away from every threshold" % (i, hit, ref_hit)
        borderline += 1
    assert compared >= 0.99 * len(ret), "too many borderline cases (%d) for the comparison to mean anything" % borderline
    assert tight >= 0.9 * (ret != inf).sum(), "only %d of %d hits agree to 1e-6 relative" % (tight, (ret != inf).sum())


def test_f16_pair_decode_matches_the_reference_python(fix, orc):
    u = fix["unpack2x16float"]
    words = np.asarray(u["word"], np.uint32)
    want = np.asarray(u["lo_hi"], np.uint32).reshape(-1, 2)
    for w, (lo, hi) in zip(words, want):
        got_lo = np.float32(orc.lib.orc_f16_to_f32(int(w) & 0xFFFF)).view(np.uint32)
        got_hi = np.float32(orc.lib.orc_f16_to_f32(int(w) >> 16)).view(np.uint32)
        nan_lo, nan_hi = (int(w) & 0x7C00) == 0x7C00 and (int(w) & 0x3FF), ((int(w) >> 16) & 0x7C00) == 0x7C00 and ((int(w) >> 16) & 0x3FF)
        assert nan_lo or got_lo == lo, "word %08x low half" % w
        assert nan_hi or got_hi == hi, "word %08x high half" % w
