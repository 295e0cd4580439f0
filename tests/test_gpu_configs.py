"""BASELINE.json's configurations run AS CONFIGURED (C2 .. C5 at full size, every pixel or a fixed pixel grid against the
CPU oracle, counters included), and the traversal corner cases of renderer.wgsl that ordinary scenes never reach:
the 64-entry stack cap with a non-zero drop count (:337) and the children every ray skips (:288-291).

Every test makes its own context, so a failure cannot leak state (batch size, caller-owned buffers) into the next one."""
import ctypes as C

import numpy as np
import pytest

import orc as orc_mod
from scenes import comb_bvh4, spoil_bvh4, random_soup, quat_yaw_pitch, pack_box, INVALID, LEAF

pytestmark = pytest.mark.gpu

COUNTERS = ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "stack_drops", "max_stack", "samples")
OMODE = {0: orc_mod.MODE_PACKET, 1: orc_mod.MODE_SINGLE, 2: orc_mod.MODE_PATH}


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.fixture
def ctx(rt):
    c = rt.Context(0)
    yield c
    c.close()


def render_shares(rt, ctx, count, make_params, stats=False):
    """One frame as `count` interleaved tile shares (rank = (tx+ty) % count, pt_render's tile_rank / tile_count), each into
    its slice of one gathered device buffer -- what the RCCL gather leaves on rank 0 -- then pt_deinterleave.
    Returns (image, summed counters or None)."""
    hip = C.CDLL("libamdhip64.so")
    p0 = make_params(0)
    w, h = p0.width, p0.height
    stride = max(rt.tile_layout(w, h, r, count)[1] for r in range(count))
    gathered = C.c_void_p()
    assert hip.hipMalloc(C.byref(gathered), C.c_size_t(stride * 4 * count)) == 0
    total = None
    try:
        for r in range(count):
            ctx.set_compact_buffer(gathered.value + r * stride * 4, stride)
            ctx.render(make_params(r))
            if stats:
                st = ctx.stats()
                total = st if total is None else {k: (max(total[k], st[k]) if k == "max_stack" else total[k] + st[k]) for k in st}
        ctx.synchronize()
        ctx.set_compact_buffer(0, 0)
        ctx.deinterleave(gathered.value, stride, w, h, count)
        img = ctx.read_radiance(w, h).copy()
    finally:
        hip.hipFree(gathered)
    return img, total


# ---------------------------------------------------------------------------------------------------------------
# renderer.wgsl:337 -- pushes beyond 64 entries are silently dropped, the nearest child first
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("levels,seed,all_hit", [(30, 1, True), (90, 2, False), (64, 3, True)])
def test_stack_overflow_drops_match_the_reference_semantics(rt, orc, ctx, levels, seed, all_hit):
    tris, bvh4 = comb_bvh4(levels, seed, all_hit)
    n = tris.size // 9
    ctx.set_triangles(tris)
    ctx.set_bvh4(bvh4)
    assert np.array_equal(ctx.read_bvh4(), bvh4)
    w, h = 96, 64
    cams = [((0, 0, 2.5), (0, 0, 0, 1)), ((0.3, -0.2, 2.2), quat_yaw_pitch(0.12, 0.08))]
    for cam, quat in cams:
        for mode in (0, 1, 2):
            kw = dict(mode=mode, spp=3, max_bounces=4, seed=5, frame=2)
            ref, _, ost = orc.render(orc.make_params(w, h, n, cam, quat, mode=OMODE[mode], spp=3, max_bounces=4, seed=5, frame=2), tris, bvh4)
            assert ost["stack_drops"] > 0 and ost["max_stack"] == 64          # the scene really overruns the stack
            ctx.render(ctx.make_params(w, h, cam, quat, stats=True, **kw))
            assert same_bits(ctx.read_radiance(), ref), (mode, cam)
            st = ctx.stats()                    # mode 0: the packet's shared stack drops the same pushes as the reference's (one count per packet)
            for k in COUNTERS:
                assert st[k] == ost[k], (k, mode, cam)
        # the un-instrumented megakernel and the one-pixel-per-lane kernel drop the same pushes
        ref, _, ost = orc.render(orc.make_params(w, h, n, cam, quat, mode=orc_mod.MODE_PATH, spp=3, max_bounces=4, seed=5, frame=2), tris, bvh4)
        ctx.render(ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=3, max_bounces=4, seed=5, frame=2))
        assert same_bits(ctx.read_radiance(), ref)
        ctx.render(ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=3, max_bounces=4, seed=5, frame=2, simple_kernel=True, stats=True))
        assert same_bits(ctx.read_radiance(), ref)
        st = ctx.stats()
        for k in COUNTERS:
            assert st[k] == ost[k], (k, "simple", cam)


# ---------------------------------------------------------------------------------------------------------------
# renderer.wgsl:288-291 -- children no ray enters: index >= numNodes (never fetched), degenerate box (fetched, rejected)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["soup", "dragon"])
def test_skipped_children_image_and_counters(rt, orc, ctx, kind):
    tris = random_soup(3000, 11) if kind == "soup" else rt.procedural_scene(0, 20000)
    n = tris.size // 9
    ctx.set_triangles(tris)
    ctx.build_bvh()
    clean = ctx.read_bvh4()
    bvh4, n_oob, n_deg = spoil_bvh4(clean, 7, count=60)
    assert n_oob >= 10 and n_deg >= 10
    ctx.set_bvh4(bvh4)
    w, h = 160, 96
    cam, quat = ((0, 0, 2.5), (0, 0, 0, 1)) if kind == "dragon" else ((0, 0, 0), quat_yaw_pitch(2.0, 0.4))
    changed = False
    for mode in (0, 1, 2):
        op = orc.make_params(w, h, n, cam, quat, mode=OMODE[mode], spp=2, max_bounces=4, seed=9)
        ref, _, ost = orc.render(op, tris, bvh4)
        base, _, bst = orc.render(op, tris, clean)
        changed = changed or ost["nodes_examined"] != bst["nodes_examined"]
        ctx.render(ctx.make_params(w, h, cam, quat, mode=mode, spp=2, max_bounces=4, seed=9, stats=True))
        assert same_bits(ctx.read_radiance(), ref), mode
        st = ctx.stats()
        for k in COUNTERS:
            assert st[k] == ost[k], (k, mode)
    assert changed                                                            # the planted children are on rays' paths
    ctx.render(ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=2, max_bounces=4, seed=9, simple_kernel=True, stats=True))
    ref, _, ost = orc.render(orc.make_params(w, h, n, cam, quat, mode=orc_mod.MODE_PATH, spp=2, max_bounces=4, seed=9), tris, bvh4)
    assert same_bits(ctx.read_radiance(), ref)
    assert ctx.stats()["nodes_examined"] == ost["nodes_examined"]


def test_degenerate_root_is_fetched_once_and_misses(rt, orc, ctx):
    # renderer.wgsl:240-244: the root record is popped (one fetch), found degenerate, and the ray ends
    tris = np.array([-1, -1, 0, 1, -1, 0, 0, 1, 0, -1, -1, -0.5, 1, -1, -0.5, 0, 1, -0.5], np.float32)
    box = pack_box((-1, -1, -0.6), (1, 1, 0.1))
    bad = [(box[0] & 0xFFFF0000) | (box[1] >> 16), (box[1] & 0xFFFF) | ((box[0] & 0xFFFF) << 16), box[2]]      # min.x <-> max.x
    leaf0 = pack_box((-1, -1, -0.01), (1, 1, 0.01)) + [INVALID] * 4 + [LEAF | 0]
    leaf1 = pack_box((-1, -1, -0.51), (1, 1, -0.49)) + [INVALID] * 4 + [LEAF | 1]
    for root_box, expect_hits in ((box, True), (bad, False)):
        bvh4 = np.array([3] + root_box + [1, 2, INVALID, INVALID, 0] + leaf0 + leaf1, np.uint32)
        ctx.set_triangles(tris)
        ctx.set_bvh4(bvh4)
        for mode in (0, 1, 2):
            ref, _, ost = orc.render(orc.make_params(48, 32, 2, mode=OMODE[mode], spp=2, max_bounces=2), tris, bvh4)
            ctx.render(ctx.make_params(48, 32, mode=mode, spp=2, max_bounces=2, stats=True))
            img = ctx.read_radiance()
            assert same_bits(img, ref)
            assert (img[..., 0] > 0.011).any() == expect_hits
            st = ctx.stats()
            for k in COUNTERS:
                assert st[k] == ost[k], (k, mode, expect_hits)
            if not expect_hits and mode != 0:
                assert st["nodes_examined"] == st["rays_closest"]          # exactly the root, once per ray (mode 0: once per 2x2 packet)


# ---------------------------------------------------------------------------------------------------------------
# C3 as configured: BVH4_wide input (tests/test.cpp:106-196), path mode, 16 spp, 8 bounces, 8-way pixel tiles
# ---------------------------------------------------------------------------------------------------------------
def test_c3_miniature_wide_bvh_path_mode_eight_tile_shares(rt, orc, ctx):
    tris = rt.procedural_scene(0, 20000)
    n = tris.size // 9
    ctx.set_triangles(tris)
    ctx.build_bvh()
    wide = rt.bvh2_to_bvh4_wide(ctx.read_bvh2())
    assert wide[0] == 2 * n - 1
    assert np.array_equal(wide, orc.bvh4_wide(ctx.read_bvh2()))
    ctx.set_bvh4(wide)                                  # nodes 1, 2, ... keep their records but are unreachable from the promoted root
    w, h = 160, 96
    kw = dict(mode=rt.PT_MODE_PATH, spp=16, max_bounces=8, seed=1)
    ref, _, ost = orc.render(orc.make_params(w, h, n, mode=orc_mod.MODE_PATH, spp=16, max_bounces=8, seed=1), tris, wide)
    img, st = render_shares(rt, ctx, 8, lambda r: ctx.make_params(w, h, tile_rank=r, tile_count=8, stats=True, **kw), stats=True)
    assert same_bits(img, ref)
    for k in COUNTERS:
        assert st[k] == ost[k], k
    img2, _ = render_shares(rt, ctx, 8, lambda r: ctx.make_params(w, h, tile_rank=r, tile_count=8, **kw))      # un-instrumented kernel
    assert same_bits(img2, ref)
    ctx.set_batch(4)                                    # the shares of four consecutive frames per launch, as bench.py submits them
    for f in range(4):
        ctx.render(ctx.make_params(w, h, tile_rank=3, tile_count=8, frame=f, **kw))
    ctx.synchronize()
    ctx.set_batch(1)


@pytest.fixture(scope="module")
def dragon_full(rt):
    c = rt.Context(0)
    tris = rt.procedural_scene(0, 871414)
    c.set_triangles(tris)
    c.build_bvh()
    c._tris = tris
    yield c
    c.close()


def test_c2_full_size_every_pixel_and_counters(rt, orc, dragon_full):
    """C2 exactly: 871,414 triangles, native LBVH2 -> BVH4, 1920x1080, 4 spp, 8 bounces.  EVERY pixel of the frame and every
    counter against the oracle (16 host threads)."""
    c = dragon_full
    bvh4 = c.read_bvh4()
    # the LBVH2 of all 871,414 triangles incl. the bottom-up refit of the internal bounds (pt_read_bvh2), word for word
    morton, tri_index = rt.morton_sort(c._tris)
    assert np.array_equal(c.read_bvh2(), orc.build_lbvh2(c._tris, morton, tri_index))
    w, h = 1920, 1080
    ref, ost = orc.render_mt(orc.make_params(w, h, 871414, mode=orc_mod.MODE_PATH, spp=4, max_bounces=8, seed=1), c._tris, bvh4)
    c.render(c.make_params(w, h, mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=1, stats=True))
    assert same_bits(c.read_radiance(), ref)
    st = c.stats()
    for k in COUNTERS:
        assert st[k] == ost[k], k
    c.render(c.make_params(w, h, mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=1))
    assert same_bits(c.read_radiance(), ref)


def test_c3_full_size_as_configured(rt, orc, dragon_full):
    """C3 exactly, on one GPU standing in for the eight: the BVH4_wide file of the dragon-class BVH2 (1,742,827 nodes),
    1920x1080, 16 spp, 8 bounces, rendered as 8 tile shares + pt_deinterleave.  Every pixel against the oracle traversing the
    same wide buffer; the shares' counters add up to the oracle's."""
    c = dragon_full
    collapsed = c.read_bvh4().copy()
    wide = rt.bvh2_to_bvh4_wide(c.read_bvh2())
    assert wide[0] == 2 * 871414 - 1
    c.set_bvh4(wide)
    try:
        w, h = 1920, 1080
        kw = dict(mode=rt.PT_MODE_PATH, spp=16, max_bounces=8, seed=1)
        ref, ost = orc.render_mt(orc.make_params(w, h, 871414, mode=orc_mod.MODE_PATH, spp=16, max_bounces=8, seed=1), c._tris, wide)
        img, st = render_shares(rt, c, 8, lambda r: c.make_params(w, h, tile_rank=r, tile_count=8, stats=True, **kw), stats=True)
        assert same_bits(img, ref)
        for k in COUNTERS:
            assert st[k] == ost[k], k
        assert st["samples"] == w * h * 16 and st["stack_drops"] == 0
        img2, _ = render_shares(rt, c, 8, lambda r: c.make_params(w, h, tile_rank=r, tile_count=8, **kw))
        assert same_bits(img2, ref)
    finally:
        c.set_bvh4(collapsed)


def test_c5_full_size_4k_progressive_accumulation(rt, orc, dragon_full):
    """C5 exactly, on one GPU: 3840x2160, 64 spp as 16 accumulated frames x 4 spp, 16 bounces, as 8 tile shares per frame (each
    share keeps its own running sum); the final image against the oracle on every 4th pixel in x and y (518,400 pixels x 64 spp)."""
    c = dragon_full
    bvh4 = c.read_bvh4()
    w, h, frames = 3840, 2160, 16
    hip = C.CDLL("libamdhip64.so")
    count = 8
    stride = max(rt.tile_layout(w, h, r, count)[1] for r in range(count))
    gathered = C.c_void_p()
    assert hip.hipMalloc(C.byref(gathered), C.c_size_t(stride * 4 * count)) == 0
    try:
        c.set_batch(frames)
        for r in range(count):                           # a rank's 16 frames: one accumulating sequence, one launch
            c.set_compact_buffer(gathered.value + r * stride * 4, stride)
            for f in range(frames):
                c.render(c.make_params(w, h, mode=rt.PT_MODE_PATH, spp=4, max_bounces=16, seed=1, frame=f, accumulate=True, tile_rank=r, tile_count=count))
            c.flush()
        c.synchronize()
        c.set_batch(1)
        c.set_compact_buffer(0, 0)
        c.deinterleave(gathered.value, stride, w, h, count)
        img = c.read_radiance(w, h)
    finally:
        hip.hipFree(gathered)
    ref, _ = orc.render_mt(orc.make_params(w, h, 871414, mode=orc_mod.MODE_PATH, spp=4, max_bounces=16, seed=1, frame=0, accum_frames=frames, step=(4, 4)), c._tris, bvh4)
    assert same_bits(img[::4, ::4], ref[::4, ::4])
    assert np.isfinite(img).all() and (img[..., 0] > 0.011).mean() > 0.08


def test_c4_full_size_sponza_class(rt, orc):
    """C4 exactly: sponza-class interior, 262,144 triangles, camera inside, 1920x1080, 4 spp, 8 bounces.  Every pixel and every
    counter against the oracle (deep LBVH: ~120 node records per ray, so this is the slow one: ~10 s on 16 host threads)."""
    c = rt.Context(0)
    try:
        tris = rt.procedural_scene(1, 262144)
        c.set_triangles(tris)
        c.build_bvh()
        bvh4 = c.read_bvh4()
        cam, quat = (0.55, -0.05, 0.05), quat_yaw_pitch(1.45, 0.05)
        w, h = 1920, 1080
        c.render(c.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=3, stats=True))
        img = c.read_radiance().copy(); st = c.stats()
        c.render(c.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=3))
        assert same_bits(c.read_radiance(), img)
        ref, ost = orc.render_mt(orc.make_params(w, h, 262144, cam, quat, mode=orc_mod.MODE_PATH, spp=4, max_bounces=8, seed=3), tris, bvh4)
        assert same_bits(img, ref)
        for k in COUNTERS:
            assert st[k] == ost[k], k
        assert st["rays_closest"] > 4 * st["samples"]                             # interior: long paths
    finally:
        c.close()
