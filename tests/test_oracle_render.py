"""Hand-derived known answers for the parts of the oracle that no reference-produced vector
pins (renderer.wgsl and BVHBuilder.wgsl cannot execute here): see DESIGN.md section 9."""
import math

import numpy as np

import orc as orc_mod
from scenes import TETRA, random_soup, quat_yaw_pitch


def one_triangle():
    # big triangle in the plane z = 0 facing +z, covering the view centre
    return np.array([-1.0, -1.0, 0.0, 1.0, -1.0, 0.0, 0.0, 1.0, 0.0], np.float32)


def test_shade_known_answer_and_miss_colour(orc):
    tris = one_triangle()
    _, bvh4 = orc.build_bvh4(tris)
    w, h = 64, 36
    for mode in (orc_mod.MODE_PACKET, orc_mod.MODE_SINGLE):
        img, ids, _ = orc.render(orc.make_params(w, h, 1, mode=mode), tris, bvh4, want_tri_ids=True)
        ndl = 1.0 / math.sqrt(1 + 2.25 + 1)                   # n=(0,0,1) . normalize(1,1.5,1), renderer.wgsl:349-351
        want = np.array([0.9, 0.7, 0.3]) * (0.15 + ndl)
        assert ids[h // 2, w // 2] == 0
        assert np.allclose(img[h // 2, w // 2, :3], want, rtol=2e-7)
        assert img[h // 2, w // 2, 3] == 1.0
        assert ids[0, 0] == 0xFFFFFFFF
        assert np.all(img[0, 0, :3] == np.float32(0.01))          # renderer.wgsl:410


def test_row_zero_looks_down(orc):
    # py = 0 <-> p.y = -1 (no flip in the compute pass; tonemapper.wgsl:20,28 flips for display)
    tris = np.array([-0.2, -0.9, 0.0, 0.2, -0.9, 0.0, 0.0, -0.6, 0.0], np.float32)   # small triangle low in the scene
    _, bvh4 = orc.build_bvh4(tris)
    img, ids, _ = orc.render(orc.make_params(64, 64, 1, mode=orc_mod.MODE_SINGLE), tris, bvh4, want_tri_ids=True)
    rows = np.where((ids == 0).any(axis=1))[0]
    assert len(rows) > 0 and rows.max() < 32
    tm = orc.tonemap(img)
    hit_rows_display = np.where((tm[..., 0] > 100).any(axis=1))[0]
    assert hit_rows_display.min() >= 32                            # flipped for display


def test_packet_and_single_ray_agree_except_ties(orc):
    tris = random_soup(2000, 5)
    _, bvh4 = orc.build_bvh4(tris)
    w, h = 160, 90
    for cam, quat in [((0, 0, 2.5), (0, 0, 0, 1)), ((0.3, 0.2, 0.1), quat_yaw_pitch(1.0, 0.3))]:
        a, ia, _ = orc.render(orc.make_params(w, h, 2000, cam, quat, mode=orc_mod.MODE_PACKET), tris, bvh4, want_tri_ids=True)
        b, ib, _ = orc.render(orc.make_params(w, h, 2000, cam, quat, mode=orc_mod.MODE_SINGLE), tris, bvh4, want_tri_ids=True)
        assert np.array_equal(ia == 0xFFFFFFFF, ib == 0xFFFFFFFF)  # hit/miss never depends on packet order
        assert (ia != ib).mean() < 1e-3                            # only exact-t ties may pick the other triangle


def test_trace_ray_brute_force(orc):
    # closest hit from the BVH == brute force over all triangles (same Moller-Trumbore in float64 tolerance)
    tris = random_soup(500, 9, size=0.3)
    _, bvh4 = orc.build_bvh4(tris)
    rng = np.random.default_rng(1)
    T = tris.reshape(-1, 3, 3).astype(np.float64)
    for _ in range(200):
        o = rng.uniform(-1.5, 1.5, 3); d = rng.normal(size=3); d /= np.linalg.norm(d)
        hit, t, n, tri = orc.trace_ray(tris, bvh4, o, d)
        o32, d32 = o.astype(np.float32).astype(np.float64), d.astype(np.float32).astype(np.float64)
        e1, e2 = T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]
        p = np.cross(d32, e2); det = (e1 * p).sum(1)
        ok = np.abs(det) > 1e-7
        inv = np.where(ok, 1.0 / np.where(ok, det, 1), 0)
        s = o32 - T[:, 0]; u = inv * (s * p).sum(1); q = np.cross(s, e1); v = inv * (q * d32).sum(1); tt = inv * (e2 * q).sum(1)
        good = ok & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (tt > 1e-7)
        if good.any():
            assert hit and abs(t - tt[good].min()) < 1e-4
            any_hit, _, _, _ = orc.trace_ray(tris, bvh4, o, d, anyhit=True)
            assert any_hit
        else:
            assert not hit


def test_lbvh2_structure(orc):
    for n, seed in [(1, 0), (2, 1), (3, 2), (9, 3), (1000, 4)]:
        tris = random_soup(n, seed)
        b = orc.build_lbvh2(tris)
        nn = 2 * n - 1
        assert b[0] == nn and b.size == 1 + 6 * nn
        dec = lambda w, hi: np.array([(w >> (16 if hi else 0)) & 0xFFFF], np.uint16).view(np.float16)[0].astype(np.float64)
        def box(i):
            w = b[1 + 6 * i: 4 + 6 * i]
            return (np.array([dec(w[0], 0), dec(w[0], 1), dec(w[1], 0)]), np.array([dec(w[1], 1), dec(w[2], 0), dec(w[2], 1)]))
        seen_leaves = []
        stack = [0]
        while stack:
            i = stack.pop()
            meta = int(b[1 + 6 * i + 5])
            mn, mx = box(i)
            if meta & 0x80000000:
                ti = meta & 0x7FFFFFFF
                seen_leaves.append(ti)
                v = tris.reshape(-1, 3, 3)[ti].astype(np.float64)
                assert (mn < v.min(0)).all() and (mx > v.max(0)).all()      # widened by one f16 ULP: strictly outside
                assert i >= n - 1
            else:
                assert i < n - 1
                l, r = int(b[1 + 6 * i + 3]), int(b[1 + 6 * i + 4])
                for c in (l, r):
                    cmn, cmx = box(c)
                    assert (mn < cmn).all() and (mx > cmx).all()            # grows one ULP per level (BVHBuilder.wgsl:86-97)
                    stack.append(c)
        assert sorted(seen_leaves) == list(range(n))


def test_rng_and_sampling(orc):
    L = orc.lib
    import ctypes as C
    us = np.array([L.orc_rnd(1, p, s, b, d) for p in range(50) for s in range(4) for b in range(3) for d in range(5)])
    assert us.min() >= 0.0 and us.max() < 1.0 and abs(us.mean() - 0.5) < 0.02
    assert len(np.unique(us)) > 0.99 * us.size
    c, s = C.c_float(), C.c_float()
    worst = 0.0
    for u in np.linspace(0, 1, 2001, endpoint=False):
        L.orc_sincos_2pi(C.c_float(float(np.float32(u))), C.byref(c), C.byref(s))
        worst = max(worst, abs(c.value - math.cos(2 * math.pi * float(np.float32(u)))), abs(s.value - math.sin(2 * math.pi * float(np.float32(u)))))
    assert worst < 2e-6
    rng = np.random.default_rng(0)
    acc = np.zeros(3)
    for _ in range(3000):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        n32 = (C.c_float * 3)(*n.astype(np.float32)); out = (C.c_float * 3)()
        u1, u2 = float(np.float32(rng.random())), float(np.float32(rng.random()))
        L.orc_cosine_dir(n32, C.c_float(u1), C.c_float(u2), out)
        d = np.array(list(out))
        assert abs(np.linalg.norm(d) - 1) < 1e-5 and np.dot(d, n) >= -1e-6
        assert abs(np.dot(d, n) - math.sqrt(max(0.0, 1 - u1))) < 1e-5     # cos(theta) = sqrt(1-u1): cosine-weighted
    

def test_path_mode_direct_light_only(orc):
    # max_bounces = 0: unoccluded flat triangle facing the light -> L = base * ndl on hits, 0.01 on misses
    tris = one_triangle()
    _, bvh4 = orc.build_bvh4(tris)
    img, _, st = orc.render(orc.make_params(64, 36, 1, mode=orc_mod.MODE_PATH, spp=3, max_bounces=0, seed=4), tris, bvh4)
    ndl = 1.0 / math.sqrt(4.25)
    assert np.allclose(img[18, 32, :3], np.array([0.9, 0.7, 0.3]) * ndl, rtol=1e-6)
    assert np.allclose(img[0, 0, :3], 0.01, rtol=1e-6)
    assert st["samples"] == 64 * 36 * 3 and st["rays_shadow"] > 0


def test_path_mode_shadowing_and_bounce_energy(orc):
    # a floor below a blocker: the blocker's shadow (towards -Ldir) removes the direct term
    floor = [-1, -0.5, -1, 1, -0.5, -1, 1, -0.5, 1, -1, -0.5, -1, 1, -0.5, 1, -1, -0.5, 1]
    floor = np.array(floor, np.float32).reshape(2, 3, 3)[:, ::-1, :].reshape(-1)   # wind so the normal is +y
    blocker = np.array([-0.3, 0.2, -0.3, 0.3, 0.2, 0.3, 0.3, 0.2, -0.3, -0.3, 0.2, -0.3, -0.3, 0.2, 0.3, 0.3, 0.2, 0.3], np.float32)
    tris = np.concatenate([floor, blocker])
    _, bvh4 = orc.build_bvh4(tris)
    cam, quat = (0, 1.5, 0), quat_yaw_pitch(0.0, -math.pi / 2 + 1e-3)      # looking straight down
    p0 = orc.make_params(96, 96, 4, cam, quat, mode=orc_mod.MODE_PATH, spp=8, max_bounces=0, seed=2)
    direct, _, _ = orc.render(p0, tris, bvh4)
    p3 = orc.make_params(96, 96, 4, cam, quat, mode=orc_mod.MODE_PATH, spp=8, max_bounces=3, seed=2)
    multi, _, _ = orc.render(p3, tris, bvh4)
    lit = np.array([0.9, 0.7, 0.3]) * (1.5 / math.sqrt(4.25))
    floor_lit = np.isclose(direct[..., 0], lit[0], rtol=1e-4)
    floor_dark = direct[..., 0] == 0.0
    assert floor_lit.sum() > 500 and floor_dark.sum() > 20                  # both regions exist
    assert (multi[..., :3] >= direct[..., :3] - 1e-6).all()                 # bounces only add light
    assert multi[floor_dark][:, 0].mean() > 0.05                            # sky light reaches the shadow


def test_pinned_gamma_curve_is_accurate_enough_for_eight_bits(orc):
    """pow_1_2_2 (the one f32 evaluation of x^(1/2.2) both the oracle and the HIP tonemap kernel use, so that the tonemapped bytes are
    bit-identical) against float64 pow: relative error below 5e-6 everywhere, the 8-bit result equal except next to a rounding boundary."""
    import ctypes as C
    orc.lib.orc_pow_1_2_2.restype = C.c_float; orc.lib.orc_pow_1_2_2.argtypes = [C.c_float]
    xs = np.concatenate([np.linspace(0, 1, 20001, dtype=np.float32), (np.float32(10.0) ** np.linspace(-30, 0, 500)).astype(np.float32)])
    got = np.array([orc.lib.orc_pow_1_2_2(float(x)) for x in xs], np.float64)
    ref = np.minimum(np.where(xs > 1.2e-38, np.power(xs.astype(np.float64), 1 / 2.2), 0.0), 1.0)
    big = xs > 1e-30
    assert (np.abs(got - ref)[big] / ref[big]).max() < 5e-6
    q = lambda g: np.floor(np.clip(g, 0, 1) * 255 + 0.5)
    differ = q(got) != q(ref)
    assert differ.sum() <= 5 and np.all(np.abs(ref[differ] * 255 + 0.5 - np.round(ref[differ] * 255 + 0.5)) < 1e-3)
    assert orc.lib.orc_pow_1_2_2(0.0) == 0.0 and orc.lib.orc_pow_1_2_2(1.0) == 1.0 and orc.lib.orc_pow_1_2_2(-1.0) == 0.0
