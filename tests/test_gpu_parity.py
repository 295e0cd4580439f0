"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Integer work (BVH buffers) and f32 radiance are both compared BIT-EXACTLY: the
arithmetic contract of DESIGN.md section 3 makes the two paths the same sequence of IEEE
operations; the north-star tolerance (1e-4 relative L2) is asserted as well."""
import numpy as np
import pytest

import orc as orc_mod
from scenes import TETRA, random_soup, quat_yaw_pitch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


@pytest.mark.parametrize("n,seed", [(1, 0), (2, 1), (3, 2), (4, 3), (100, 4), (5000, 5), (120000, 6)])
def test_lbvh2_kernels_bit_exact(rt, orc, gpu_ctx, n, seed):
    tris = TETRA if n == 4 else random_soup(n, seed)
    m, t = rt.morton_sort(tris)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_lbvh2(m, t)
    got = gpu_ctx.read_bvh2()
    want = orc.build_lbvh2(tris, m, t)
    assert np.array_equal(got, want)


def test_lbvh2_tetra_known_answer(rt, gpu_ctx):
    # leaf bounds of the default tetrahedron = [-1,1]^3 widened by one f16 ULP
    m, t = rt.morton_sort(TETRA)
    gpu_ctx.set_triangles(TETRA)
    gpu_ctx.build_lbvh2(m, t)
    b = gpu_ctx.read_bvh2()
    assert b[0] == 7
    for node in range(3, 7):
        assert list(b[1 + 6 * node: 4 + 6 * node]) == [0xBC01BC01, 0x3C01BC01, 0x3C013C01]
    assert [int(b[1 + 6 * node + 5]) for node in range(3, 7)] == [0x80000003, 0x80000000, 0x80000002, 0x80000001]


@pytest.mark.parametrize("n,seed", [(1, 9), (2, 8), (3, 7), (4, 0), (5, 6), (64, 5), (65, 4), (777, 1), (30000, 2), (120000, 3)])
def test_build_bvh_end_to_end(rt, orc, gpu_ctx, n, seed):
    """pt_build_bvh runs Morton + sort, LBVH2, the collapse and the re-layout on the device: BVH2 and BVH4 equal the oracle's."""
    tris = TETRA if n == 4 else random_soup(n, seed)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh2, bvh4 = orc.build_bvh4(tris)
    assert np.array_equal(gpu_ctx.read_bvh2(), bvh2)
    assert np.array_equal(gpu_ctx.read_bvh4(), bvh4)


@pytest.mark.parametrize("kind", ["identical", "coplanar", "two_clusters", "tiny", "signed_zero", "mixed_zero"])
def test_build_bvh_degenerate_inputs(rt, orc, gpu_ctx, kind):
    """Equal Morton codes (index tie-break), zero extent on an axis (1e-20 floor), f16-subnormal bounds, -0 coordinates."""
    rng = np.random.default_rng(3)
    if kind == "identical":
        tris = np.tile(np.array([0.1, 0.2, 0.3, 0.4, 0.2, 0.3, 0.1, 0.5, 0.3], np.float32), 300)
    elif kind == "coplanar":
        tris = rng.uniform(-1, 1, (500, 3, 3)).astype(np.float32); tris[:, :, 2] = 0.25; tris = tris.reshape(-1)
    elif kind == "two_clusters":
        a = rng.uniform(-1e-3, 1e-3, (200, 3, 3)).astype(np.float32) + np.float32(-0.9)
        b = rng.uniform(-1e-3, 1e-3, (200, 3, 3)).astype(np.float32) + np.float32(0.9)
        tris = np.concatenate([a, b]).reshape(-1)
    elif kind == "tiny":
        tris = rng.uniform(-3e-6, 3e-6, (400, 3, 3)).astype(np.float32).reshape(-1)     # below the f16 normal range
    elif kind == "mixed_zero":
        # +0, -0 and +-1e-9 (all +-0 in f16) mixed inside triangles and between siblings: min / max of (-0, +0) decides whether incrementF16
        # (BVHBuilder.wgsl:63-82) lands on 0 or on the smallest subnormal -- pinned as -0 < +0 (oracle/pt_oracle.cpp::min_oz)
        tris = rng.uniform(-1, 1, (600, 3, 3)).astype(np.float32)
        zeros = np.array([0.0, -0.0, 1e-9, -1e-9, 4e-9, -4e-9], np.float32)
        for axis in range(3):
            pick = rng.random((600, 3)) < 0.5
            tris[:, :, axis][pick] = zeros[rng.integers(0, 6, int(pick.sum()))]
        tris = tris.reshape(-1)
    else:
        tris = rng.uniform(-1, 1, (256, 3, 3)).astype(np.float32)
        tris[::3, :, 0] = np.float32(-0.0); tris[1::3, :, 1] = np.float32(0.0); tris = tris.reshape(-1)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh2, bvh4 = orc.build_bvh4(tris)
    assert np.array_equal(gpu_ctx.read_bvh2(), bvh2)
    assert np.array_equal(gpu_ctx.read_bvh4(), bvh4)
    # the device layouts built from it render like the oracle
    p = gpu_ctx.make_params(64, 48, (0, 0, 2.5), (0, 0, 0, 1), mode=rt.PT_MODE_REFERENCE)
    gpu_ctx.render(p)
    want, _, _ = orc.render(orc.make_params(64, 48, len(tris) // 9, (0, 0, 2.5), (0, 0, 0, 1), mode=orc_mod.MODE_SINGLE), tris, bvh4)
    assert same_bits(gpu_ctx.read_radiance(), want)


CAMS = [((0, 0, 2.5), (0, 0, 0, 1)), ((0.4, 0.3, 1.7), quat_yaw_pitch(0.2, -0.15)), ((0, 0, 0), quat_yaw_pitch(2.0, 0.4))]


def _scene(rt, orc, gpu_ctx, kind):
    if kind == "tetra":
        tris = TETRA
    elif kind == "soup":
        tris = random_soup(3000, 11)
    else:
        tris = rt.procedural_scene(0, 20000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    return tris, gpu_ctx.read_bvh4()


@pytest.mark.parametrize("kind", ["tetra", "soup", "dragon"])
def test_reference_mode_single_ray_bit_exact(rt, orc, gpu_ctx, kind):
    tris, bvh4 = _scene(rt, orc, gpu_ctx, kind)
    w, h = 200, 120
    for cam, quat in CAMS:
        p = gpu_ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_REFERENCE, stats=True)
        gpu_ctx.render(p)
        img = gpu_ctx.read_radiance()
        st = gpu_ctx.stats()
        op = orc.make_params(w, h, tris.size // 9, cam, quat, mode=orc_mod.MODE_SINGLE)
        ref, _, ost = orc.render(op, tris, bvh4)
        assert rel_l2(img, ref) <= 1e-4
        assert same_bits(img, ref)
        for k in ("rays_closest", "nodes_examined", "tris_tested", "stack_drops", "max_stack", "samples"):
            assert st[k] == ost[k], k
        # the reference's frame runs on the persistent megakernel by default (instrumented above); the uninstrumented build and
        # the one-pixel-per-lane kernel (PT_FLAG_SIMPLE_KERNEL, counters included) give the same bits
        p.flags = 0
        gpu_ctx.render(p)
        assert same_bits(gpu_ctx.read_radiance(), ref)
        p.flags = rt.PT_FLAG_SIMPLE_KERNEL | rt.PT_FLAG_STATS
        gpu_ctx.render(p)
        assert same_bits(gpu_ctx.read_radiance(), ref)
        st2 = gpu_ctx.stats()
        for k in ("rays_closest", "nodes_examined", "tris_tested", "stack_drops", "max_stack", "samples"):
            assert st2[k] == ost[k], k
    # a batch of reference-mode frames with different cameras in one launch
    gpu_ctx.set_batch(len(CAMS))
    outs = []
    for cam, quat in CAMS:
        gpu_ctx.render(gpu_ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_REFERENCE))
    gpu_ctx.set_batch(1)
    last = gpu_ctx.read_radiance()
    want, _, _ = orc.render(orc.make_params(w, h, tris.size // 9, CAMS[-1][0], CAMS[-1][1], mode=orc_mod.MODE_SINGLE), tris, bvh4)
    assert same_bits(last, want)


@pytest.mark.parametrize("kind", ["tetra", "soup", "dragon"])
def test_literal_packet_mode_bit_exact(rt, orc, gpu_ctx, kind):
    tris, bvh4 = _scene(rt, orc, gpu_ctx, kind)
    for (w, h) in [(200, 120), (33, 17)]:      # odd sizes: partially filled packets
        cam, quat = CAMS[1]
        gpu_ctx.render(gpu_ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_REFERENCE_PACKET))
        img = gpu_ctx.read_radiance()
        ref, _, ost = orc.render(orc.make_params(w, h, tris.size // 9, cam, quat, mode=orc_mod.MODE_PACKET), tris, bvh4)
        assert same_bits(img, ref)
        # the instrumented form: the same image, and the oracle's packet-mode counters -- incl. the reference's double node fetch
        # (getBVHNode4 once per pop and once per child slot, renderer.wgsl:240, 292), which only this mode performs
        import ctypes as C
        gpu_ctx.render(gpu_ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_REFERENCE_PACKET, stats=True))
        assert same_bits(gpu_ctx.read_radiance(), ref)
        st = gpu_ctx.stats()
        for k in ("rays_closest", "nodes_examined", "tris_tested", "stack_drops", "max_stack", "samples"):
            assert st[k] == ost[k], (k, kind, w, h)
        dbg = np.zeros(24, np.uint64); rt.lib.pt_debug_counters(gpu_ctx.h, dbg.ctypes.data_as(C.c_void_p))
        assert int(dbg[7]) == ost["node_fetches_ref"]


@pytest.mark.parametrize("kind,spp,bounces", [("tetra", 2, 3), ("soup", 3, 4), ("dragon", 4, 8)])
def test_path_mode_bit_exact(rt, orc, gpu_ctx, kind, spp, bounces):
    tris, bvh4 = _scene(rt, orc, gpu_ctx, kind)
    w, h = 160, 96
    cam, quat = CAMS[0] if kind != "soup" else CAMS[2]
    p = gpu_ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=spp, max_bounces=bounces, seed=7, frame=3, stats=True)
    gpu_ctx.render(p)
    img = gpu_ctx.read_radiance()
    st = gpu_ctx.stats()
    op = orc.make_params(w, h, tris.size // 9, cam, quat, mode=orc_mod.MODE_PATH, spp=spp, max_bounces=bounces, seed=7, frame=3)
    ref, _, ost = orc.render(op, tris, bvh4)
    assert rel_l2(img, ref) <= 1e-4
    assert same_bits(img, ref)
    for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples"):
        assert st[k] == ost[k], k
    # the uninstrumented kernel gives the same image
    p.flags = 0
    gpu_ctx.render(p)
    assert same_bits(gpu_ctx.read_radiance(), img)
    # ... and so does the one-pixel-per-lane kernel (second, independent HIP implementation), counters included
    p.flags = rt.PT_FLAG_SIMPLE_KERNEL | rt.PT_FLAG_STATS
    gpu_ctx.render(p)
    assert same_bits(gpu_ctx.read_radiance(), img)
    st2 = gpu_ctx.stats()
    for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples", "max_stack"):
        assert st2[k] == st[k], k


def test_bvh4_wide_input_gives_same_hits(rt, orc, gpu_ctx):
    # config C3: the BVH4_wide file of tests/test.cpp is a valid BVH input; closest hits agree
    tris = rt.procedural_scene(0, 20000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh2 = gpu_ctx.read_bvh2()
    wide = rt.bvh2_to_bvh4_wide(bvh2)
    gpu_ctx.set_bvh4(wide)
    cam, quat = CAMS[0]
    gpu_ctx.render(gpu_ctx.make_params(160, 96, cam, quat, mode=rt.PT_MODE_REFERENCE))
    img = gpu_ctx.read_radiance()
    ref, _, _ = orc.render(orc.make_params(160, 96, tris.size // 9, cam, quat, mode=orc_mod.MODE_SINGLE), tris, wide)
    assert same_bits(img, ref)


def test_tile_sharding_matches_whole_frame(rt, gpu_ctx):
    import ctypes as C
    tris = rt.procedural_scene(0, 20000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    w, h = 200, 120
    kw = dict(mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, seed=5)
    gpu_ctx.render(gpu_ctx.make_params(w, h, simple_kernel=True, **kw))
    simple = gpu_ctx.read_radiance().copy()
    gpu_ctx.render(gpu_ctx.make_params(w, h, **kw))
    assert same_bits(gpu_ctx.read_radiance(), simple)
    full = gpu_ctx.read_radiance().copy()
    hip = C.CDLL("libamdhip64.so")
    for count in (2, 3, 8):
        stride = max(rt.tile_layout(w, h, r, count)[1] for r in range(count))
        gathered = C.c_void_p()
        assert hip.hipMalloc(C.byref(gathered), C.c_size_t(stride * 4 * count)) == 0
        for r in range(count):
            gpu_ctx.render(gpu_ctx.make_params(w, h, tile_rank=r, tile_count=count, **kw))
            ptr, floats = gpu_ctx.compact_radiance()
            gpu_ctx.synchronize()
            assert hip.hipMemcpy(C.c_void_p(gathered.value + r * stride * 4), C.c_void_p(ptr), C.c_size_t(floats * 4), 3) == 0
        gpu_ctx.deinterleave(gathered.value, stride, w, h, count)
        got = gpu_ctx.read_radiance()
        hip.hipFree(gathered)
        assert same_bits(got, full), count


def test_progressive_accumulation_bit_exact(rt, orc, gpu_ctx):
    # config C5 in miniature: F frames x spp accumulated into one image
    tris = rt.procedural_scene(0, 20000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh4 = gpu_ctx.read_bvh4()
    w, h, spp, frames = 120, 72, 2, 3
    for simple in (False, True):
        for f in range(frames):
            gpu_ctx.render(gpu_ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=spp, max_bounces=4, seed=11, frame=5 + f, accumulate=True, simple_kernel=simple))
        img = gpu_ctx.read_radiance()
        ref, _, _ = orc.render(orc.make_params(w, h, tris.size // 9, mode=orc_mod.MODE_PATH, spp=spp, max_bounces=4, seed=11, frame=5, accum_frames=frames), tris, bvh4)
        assert same_bits(img, ref), simple
        # a non-accumulating render restarts the running sum
        gpu_ctx.render(gpu_ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=spp, max_bounces=4, seed=11, frame=5, simple_kernel=simple))
        one, _, _ = orc.render(orc.make_params(w, h, tris.size // 9, mode=orc_mod.MODE_PATH, spp=spp, max_bounces=4, seed=11, frame=5), tris, bvh4)
        assert same_bits(gpu_ctx.read_radiance(), one)


def test_accumulation_checkpoint_resumes_bit_for_bit(rt, orc, gpu_ctx):
    """pt_read_accum / pt_set_accum (checkpoint row; config C5): 8 accumulated frames, dump, a NEW context, restore, 8 more --
    equal to 16 frames straight and to the oracle's 16, bit for bit; the same for a compact tile share; the dump itself is the raw
    per-pixel sums with the sample count in w."""
    tris = rt.procedural_scene(0, 20000)
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    bvh4 = gpu_ctx.read_bvh4()
    w, h, spp = 104, 64, 2
    kw = dict(mode=rt.PT_MODE_PATH, spp=spp, max_bounces=4, seed=21, accumulate=True)
    assert gpu_ctx.accum_info().floats == 0
    with pytest.raises(rt.PtError):
        gpu_ctx.read_accum()
    for f in range(16):
        gpu_ctx.render(gpu_ctx.make_params(w, h, frame=f, **kw))
    straight = gpu_ctx.read_radiance().copy()
    ref, _, _ = orc.render(orc.make_params(w, h, tris.size // 9, mode=orc_mod.MODE_PATH, spp=spp, max_bounces=4, seed=21, frame=0, accum_frames=16), tris, bvh4)
    assert same_bits(straight, ref)
    gpu_ctx.render(gpu_ctx.make_params(w, h, frame=99, mode=rt.PT_MODE_PATH, spp=spp, max_bounces=4, seed=21))       # ends the sequence
    for f in range(8):
        gpu_ctx.render(gpu_ctx.make_params(w, h, frame=f, **kw))
    info, dump = gpu_ctx.read_accum()
    assert (info.width, info.height, info.compact, info.samples, info.floats) == (w, h, 0, 8 * spp, w * h * 4)
    d = dump.reshape(h, w, 4)
    assert np.all(d[..., 3] == np.float32(8 * spp))
    assert same_bits(d[..., :3] * np.float32(1.0 / (8 * spp)), gpu_ctx.read_radiance()[..., :3])                       # the image is sum * (1 / count)
    fresh = rt.Context(0)
    try:
        fresh.set_triangles(tris); fresh.build_bvh()
        fresh.set_accum(info, dump)
        for f in range(8, 16):
            fresh.render(fresh.make_params(w, h, frame=f, **kw))
        assert same_bits(fresh.read_radiance(), straight)
        assert fresh.accum_info().samples == 16 * spp
        with pytest.raises(rt.PtError):
            bad = rt.PtAccumInfo(); bad.width = w; bad.height = h; bad.samples = 4; bad.floats = 12
            fresh.set_accum(bad, np.zeros(12, np.float32))
        # a tile share (compact, tile-major): rank 1 of 3
        sk = dict(kw, tile_rank=1, tile_count=3)
        for f in range(6):
            gpu_ctx.render(gpu_ctx.make_params(w, h, frame=f, **sk))
        gpu_ctx.synchronize()
        sinfo, sdump = gpu_ctx.read_accum()
        assert sinfo.compact == 1 and sinfo.tile_rank == 1 and sinfo.tile_count == 3 and sinfo.samples == 6 * spp
        assert sinfo.floats == rt.tile_layout(w, h, 1, 3)[0] * 256
        for f in range(6, 9):
            gpu_ctx.render(gpu_ctx.make_params(w, h, frame=f, **sk))
        gpu_ctx.synchronize()
        _, want9 = gpu_ctx.read_accum()
        fresh.set_accum(sinfo, sdump)
        for f in range(6, 9):
            fresh.render(fresh.make_params(w, h, frame=f, **sk))
        fresh.synchronize()
        _, got9 = fresh.read_accum()
        assert same_bits(got9, want9)
    finally:
        fresh.close()


@pytest.mark.parametrize("w,h", [(100, 60), (97, 64), (104, 61)])
def test_accumulation_checkpoint_at_ragged_resolutions(rt, gpu_ctx, w, h):
    """Checkpoints where width or height is not a multiple of 8: the edge tiles of a compact (tile-major) dump carry pixels outside the image that
    no kernel writes -- the dump zeroes them (two dumps of one state are the same bytes) and pt_set_accum validates the sample count on pixels
    inside the image.  Whole-frame COMPACT dump, a 1/3 share, and a share that owns no tile at all (more shares than tiles)."""
    tris = rt.procedural_scene(0, 8000)
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    fresh = rt.Context(0)
    try:
        fresh.set_triangles(tris); fresh.build_bvh()
        spp = 2
        for share in (dict(flags=rt.PT_FLAG_COMPACT), dict(tile_rank=1, tile_count=3), dict(tile_rank=2, tile_count=3)):
            flags = share.pop("flags", 0)
            def params(ctx, f):
                p = ctx.make_params(w, h, frame=f, mode=rt.PT_MODE_PATH, spp=spp, max_bounces=3, seed=5, accumulate=True, **share)
                p.flags |= flags
                return p
            for f in range(3):
                gpu_ctx.render(params(gpu_ctx, f))
            gpu_ctx.synchronize()
            info, dump = gpu_ctx.read_accum()
            rank, count = share.get("tile_rank", 0), share.get("tile_count", 1)
            ids = rt.tile_ids(w, h, rank, count)
            assert info.compact == 1 and info.samples == 3 * spp and info.floats == len(ids) * 256
            d = dump.reshape(len(ids), 8, 8, 4)
            tiles_x = (w + 7) // 8
            inside = np.zeros((len(ids), 8, 8), bool)
            for s, t in enumerate(ids):
                x0, y0 = (int(t) % tiles_x) * 8, (int(t) // tiles_x) * 8
                inside[s] = ((y0 + np.arange(8))[:, None] < h) & ((x0 + np.arange(8))[None, :] < w)
            assert not inside.all()                                             # the case under test: some edge tile sticks out
            assert np.all(d[inside][:, 3] == np.float32(3 * spp)) and np.all(d[~inside] == 0)
            _, again = gpu_ctx.read_accum()
            assert same_bits(again, dump)
            for f in range(3, 5):
                gpu_ctx.render(params(gpu_ctx, f))
            gpu_ctx.synchronize()
            _, want = gpu_ctx.read_accum()
            fresh.set_accum(info, dump)                                         # was refused: the last float of the dump is outside the image
            for f in range(3, 5):
                fresh.render(params(fresh, f))
            fresh.synchronize()
            _, got = fresh.read_accum()
            assert same_bits(got, want)
            bad = dump.copy(); bad[(len(ids) - 1) * 256 + 3] = np.float32(1)    # the last tile's first pixel IS inside: still validated
            with pytest.raises(rt.PtError):
                fresh.set_accum(info, bad)
        # a share with no tile: 4095 shares of a 13 x 8-tile frame, rank 4000 -- zero floats, accepted without touching the (empty) array
        empty = rt.PtAccumInfo(); empty.width, empty.height, empty.tile_rank, empty.tile_count, empty.compact, empty.samples, empty.floats = w, h, 4000, 4095, 1, 4, 0
        assert len(rt.tile_ids(w, h, 4000, 4095)) == 0
        fresh.set_accum(empty, np.zeros(0, np.float32))
        assert fresh.accum_info().samples == 4 and fresh.accum_info().floats == 0
        einfo, edump = fresh.read_accum()                                       # ... and reads back as what it is: nothing
        assert einfo.samples == 4 and edump.size == 0
    finally:
        fresh.close()


def test_sponza_class_interior_bit_exact(rt, orc, gpu_ctx):
    # config C4 in miniature: camera inside, every ray hits, long thin triangles -> deep LBVH
    tris = rt.procedural_scene(1, 30000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh4 = gpu_ctx.read_bvh4()
    cam, quat = (0.55, -0.05, 0.05), quat_yaw_pitch(1.45, 0.05)
    w, h = 128, 72
    gpu_ctx.render(gpu_ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=2, max_bounces=8, seed=3, stats=True))
    img = gpu_ctx.read_radiance(); st = gpu_ctx.stats()
    ref, _, ost = orc.render(orc.make_params(w, h, tris.size // 9, cam, quat, mode=orc_mod.MODE_PATH, spp=2, max_bounces=8, seed=3), tris, bvh4)
    assert same_bits(img, ref)
    assert st["rays_closest"] == ost["rays_closest"] and st["nodes_examined"] == ost["nodes_examined"] and st["max_stack"] == ost["max_stack"]
    gpu_ctx.render(gpu_ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_REFERENCE))
    ref1, ids, _ = orc.render(orc.make_params(w, h, tris.size // 9, cam, quat, mode=orc_mod.MODE_SINGLE), tris, bvh4, want_tri_ids=True)
    assert same_bits(gpu_ctx.read_radiance(), ref1)
    assert (ids != 0xFFFFFFFF).mean() > 0.9        # interior: nearly every camera ray hits


def test_short_reciprocal_forms_bit_exact(rt, gpu_ctx, orc):
    """The megakernel exists in two variants: one takes every reciprocal and square root by the compiler's IEEE sequences, the other by short
    correctly rounded forms that are only valid for operands of ordinary magnitude (pt_device.h::rcp_normal / sqrt_normal; tools/probes/rcp_exact.hip
    checks them over every f32 bit pattern of their range).  The host picks the short variant where it can bound the operands
    (pt_api.cpp::arith_is_bounded).  Both are the oracle's image bit for bit, and a camera whose quaternion is far from unit length (ray directions
    a million times too long) takes the general variant by itself."""
    import ctypes as C
    tris, bvh4 = _scene(rt, orc, gpu_ctx, "dragon")
    w, h = 160, 96
    cam, quat = CAMS[0]
    dbg = np.zeros(24, np.uint64)
    op = orc.make_params(w, h, tris.size // 9, cam, quat, mode=orc_mod.MODE_PATH, spp=3, max_bounces=5, seed=11, frame=2)
    ref, _, ost = orc.render(op, tris, bvh4)
    for knob, variant in ((None, 1), (0, 0)):
        gpu_ctx.debug_set_tune("BOUNDED", knob)
        gpu_ctx.render(gpu_ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=3, max_bounces=5, seed=11, frame=2, stats=True))
        img = gpu_ctx.read_radiance(); st = gpu_ctx.stats()
        rt.lib.pt_debug_counters(gpu_ctx.h, dbg.ctypes.data_as(C.c_void_p))
        assert int(dbg[7]) & 1 == variant, (knob, int(dbg[7]))
        assert 0 < (int(dbg[7]) & ((1 << 63) - 1)) >> 1 <= st["rays_closest"] + st["rays_shadow"]      # rays that entered the root box
        assert same_bits(img, ref), knob
        for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples"):
            assert st[k] == ost[k], (knob, k, st[k], ost[k])
    gpu_ctx.debug_set_tune("BOUNDED")
    cam, quat = CAMS[2]                                            # inside the closed mesh: whatever the directions become, they hit
    big = tuple(1100.0 * c for c in quat)                         # |q|^2 = 1.21e6 >= 2^20: the rotation terms of renderer.wgsl:66-72 scaled by that much
    op = orc.make_params(w, h, tris.size // 9, cam, big, mode=orc_mod.MODE_PATH, spp=2, max_bounces=3, seed=11, frame=2)
    ref, _, ost = orc.render(op, tris, bvh4)
    gpu_ctx.render(gpu_ctx.make_params(w, h, cam, big, mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, seed=11, frame=2, stats=True))
    img = gpu_ctx.read_radiance()
    rt.lib.pt_debug_counters(gpu_ctx.h, dbg.ctypes.data_as(C.c_void_p))
    assert int(dbg[7]) & 1 == 0                                    # the general variant, chosen by the host
    assert same_bits(img, ref)
    assert (img[..., 0] != np.float32(0.01)).any()                 # the frame is not empty
    # just INSIDE the gate, worst case for 1 / det: coordinates near the top of what f16 boxes can bound (|x| <= 3e4, edge components up to 6e4)
    # and |q|^2 = 260,100 < 2^18, so camera directions are up to 1 + 3 |q|^2 = 7.8e5 long -- the host must pick the short forms by itself
    # and they must still be the oracle's IEEE quotients bit for bit
    rng = np.random.default_rng(77)
    big_tris = rng.uniform(-3e4, 3e4, (1500, 3, 3)).astype(np.float32).reshape(-1)
    gpu_ctx.set_triangles(big_tris); gpu_ctx.build_bvh()
    bvh4b = gpu_ctx.read_bvh4()
    q510 = tuple(510.0 * c for c in quat_yaw_pitch(0.7, -0.3))
    cam0 = (100.0, -50.0, 25.0)
    op = orc.make_params(w, h, big_tris.size // 9, cam0, q510, mode=orc_mod.MODE_PATH, spp=2, max_bounces=4, seed=5, frame=1)
    ref, _, ost = orc.render(op, big_tris, bvh4b)
    gpu_ctx.render(gpu_ctx.make_params(w, h, cam0, q510, mode=rt.PT_MODE_PATH, spp=2, max_bounces=4, seed=5, frame=1, stats=True))
    img = gpu_ctx.read_radiance(); st = gpu_ctx.stats()
    rt.lib.pt_debug_counters(gpu_ctx.h, dbg.ctypes.data_as(C.c_void_p))
    assert int(dbg[7]) & 1 == 1                                    # the short forms, chosen by the host
    assert same_bits(img, ref)
    assert st["tris_tested"] == ost["tris_tested"] > 10000 and st["nodes_examined"] == ost["nodes_examined"]
    # and just outside (|q|^2 = 2^18 exactly): the general variant
    q512 = tuple(512.0 * c for c in (0.0, 0.0, 0.0, 1.0))
    gpu_ctx.render(gpu_ctx.make_params(w, h, cam0, q512, mode=rt.PT_MODE_PATH, spp=1, max_bounces=1, seed=5, stats=True))
    gpu_ctx.synchronize()
    rt.lib.pt_debug_counters(gpu_ctx.h, dbg.ctypes.data_as(C.c_void_p))
    assert int(dbg[7]) & 1 == 0


def test_quad_mode_drain_bit_exact(rt, gpu_ctx, orc):
    """Quad mode (a wavefront that has nothing left to start and holds at most 16 paths re-seats them one per quad of lanes: one child
    box per lane, Moller-Trumbore over three lanes, four stack entries per pop round) changes how a ray is executed, not what it does:
    image and every counter are the same with it and without it, in single launches and batches, whatever the number of paths a
    wavefront re-seats at -- and wavefronts do re-seat (the instrumented launch counts them).  (Deep stacks in quad mode: the comb
    BVHs of test_stack_overflow_drops_match_the_reference_semantics run with the default setting.)"""
    import ctypes as C
    tris = rt.procedural_scene(0, 60000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    kw = dict(mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=2)
    dbg = np.zeros(24, np.uint64)
    gpu_ctx.debug_set_tune("QUAD", 0)
    gpu_ctx.render(gpu_ctx.make_params(640, 360, stats=True, **kw))
    want = gpu_ctx.read_radiance().copy(); st0 = gpu_ctx.stats()
    rt.lib.pt_debug_counters(gpu_ctx.h, dbg.ctypes.data_as(C.c_void_p))
    assert dbg[16] == 0                                                                   # off: no wavefront re-seats
    for live in (16, 5, 1):
        gpu_ctx.debug_set_tune("QUAD", live)
        gpu_ctx.render(gpu_ctx.make_params(640, 360, stats=True, **kw))
        got = gpu_ctx.read_radiance().copy(); st = gpu_ctx.stats()
        assert same_bits(got, want), live
        for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples", "max_stack", "stack_drops"):
            assert st[k] == st0[k], (k, live, st[k], st0[k])
        rt.lib.pt_debug_counters(gpu_ctx.h, dbg.ctypes.data_as(C.c_void_p))
        assert dbg[16] > 0, live                                                          # wavefronts went on with one ray per quad
        gpu_ctx.render(gpu_ctx.make_params(640, 360, **kw))                               # the production kernel
        assert same_bits(gpu_ctx.read_radiance(), want), live
    # batches with quad mode on, against the same frames without it
    gpu_ctx.set_batch(3)
    outs = {}
    for q in (0, 16):
        gpu_ctx.debug_set_tune("QUAD", q)
        for f in range(3):
            gpu_ctx.render(gpu_ctx.make_params(320, 200, frame=f, **kw))
        outs[q] = gpu_ctx.read_radiance().copy()
    assert same_bits(outs[0], outs[16])
    gpu_ctx.set_batch(1)
    gpu_ctx.debug_set_tune("QUAD")


def test_forked_shadow_rays_bit_exact(rt, gpu_ctx, orc):
    """Quad mode hands the shadow ray of a path that goes on to an idle quad of the wavefront and starts the next bounce at once
    (pt_megakernel_loop.inc): the path's radiance terms are still added in bounce order (a path waits for its shadow ray's result at its
    next event that adds one), so image and counters are those of the run without it and of the oracle -- and shadow rays ARE handed
    over, found waiting for, and sometimes have no idle quad (the instrumented launch counts all three)."""
    import ctypes as C
    tris = rt.procedural_scene(0, 60000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    kw = dict(mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=5)
    dbg = np.zeros(24, np.uint64)
    res, res_forks = {}, {}
    for fork in (0, 1, 2):            # off / idle quads only (quad mode) / idle lanes too, as soon as the wavefront has nothing left to start (the default)
        gpu_ctx.debug_set_tune("FORK", fork)
        gpu_ctx.render(gpu_ctx.make_params(640, 360, stats=True, **kw))
        res[fork] = (gpu_ctx.read_radiance().copy(), gpu_ctx.stats())
        rt.lib.pt_debug_counters(gpu_ctx.h, dbg.ctypes.data_as(C.c_void_p))
        res_forks[fork] = int(dbg[19])
        assert int(dbg[7]) >> 63 == 0, fork                                      # no forked shadow ray was lost (the shade pass's own check)
        if fork:
            assert dbg[19] > 0 and dbg[21] > 0, (dbg[19], dbg[20], dbg[21])      # handed over / paths that waited for theirs
        else:
            assert dbg[19] == 0 and dbg[21] == 0
        gpu_ctx.render(gpu_ctx.make_params(640, 360, **kw))                     # the production kernel
        assert same_bits(gpu_ctx.read_radiance(), res[fork][0]), fork
    assert dbg[19] > res_forks[1]                                               # more shadow rays are handed over when idle lanes take them too
    for fork in (1, 2):
        assert same_bits(res[0][0], res[fork][0]), fork
        for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples", "max_stack", "stack_drops"):
            assert res[0][1][k] == res[fork][1][k], (fork, k, res[0][1][k], res[fork][1][k])
    # against the oracle on a frame it finishes in seconds
    bvh4 = gpu_ctx.read_bvh4()
    cam, quat = CAMS[0]
    gpu_ctx.render(gpu_ctx.make_params(160, 96, cam, quat, stats=True, **kw))
    got = gpu_ctx.read_radiance().copy(); st = gpu_ctx.stats()
    want, _, ost = orc.render(orc.make_params(160, 96, tris.size // 9, cam, quat, mode=orc_mod.MODE_PATH, spp=4, max_bounces=8, seed=5), tris, bvh4)
    assert same_bits(got, want)
    for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples"):
        assert st[k] == ost[k], k
    # quads from the first ray on (every shadow ray of a continuing path can be handed over) is an A/B build, not a knob: covered by tools/ab/kvariants.sh
    gpu_ctx.debug_set_tune("FORK")


def test_batched_launch_equals_frame_by_frame(rt, gpu_ctx):
    """pt_set_batch: several frames traced by one persistent launch give exactly the per-frame results
    (each into the output target that was current at submission), including an accumulating sequence."""
    import ctypes as C
    tris = rt.procedural_scene(0, 20000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    w, h, count = 200, 120, 2
    hip = C.CDLL("libamdhip64.so")
    nt, floats = rt.tile_layout(w, h, 0, count)
    cams = [((0, 0, 2.5), (0, 0, 0, 1)), ((0.3, 0.1, 2.2), quat_yaw_pitch(0.1, 0.05)), ((-0.2, 0.2, 2.4), quat_yaw_pitch(-0.1, 0.0))]
    def params(i, **kw):
        return gpu_ctx.make_params(w, h, cams[i][0], cams[i][1], mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, seed=5, frame=10 + i, **kw)
    # reference: one launch per frame
    want = []
    for i in range(3):
        gpu_ctx.render(params(i, tile_rank=0, tile_count=count))
        ptr, fl = gpu_ctx.compact_radiance(); gpu_ctx.synchronize()
        host = np.zeros(fl, np.float32)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(fl * 4), 2) == 0
        want.append(host)
    # batched: three frames, three caller-owned compact buffers, one launch
    bufs = []
    for i in range(3):
        p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(floats * 4)) == 0; bufs.append(p)
    gpu_ctx.set_batch(3)
    for i in range(3):
        gpu_ctx.set_compact_buffer(bufs[i].value, floats)
        gpu_ctx.render(params(i, tile_rank=0, tile_count=count))
    gpu_ctx.synchronize()
    for i in range(3):
        host = np.zeros(floats, np.float32)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), bufs[i], C.c_size_t(floats * 4), 2) == 0
        assert same_bits(host, want[i]), i
    # a partial batch is launched by the read-back; accumulation order is the submission order
    gpu_ctx.set_compact_buffer(0, 0)
    gpu_ctx.set_batch(1)
    for i in range(3):
        gpu_ctx.render(params(0, accumulate=True) if i == 0 else gpu_ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, seed=5, frame=10 + i, accumulate=True))
    ref = gpu_ctx.read_radiance().copy()
    gpu_ctx.render(gpu_ctx.make_params(w, h, mode=rt.PT_MODE_REFERENCE))      # ends the accumulating sequence
    gpu_ctx.set_batch(4)
    for i in range(3):
        gpu_ctx.render(params(0, accumulate=True) if i == 0 else gpu_ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, seed=5, frame=10 + i, accumulate=True))
    got = gpu_ctx.read_radiance().copy()                                       # flushes the open batch of 3
    gpu_ctx.set_batch(1)
    for b in bufs:
        hip.hipFree(b)
    assert same_bits(got, ref)


def test_whole_frame_output_buffers_in_a_batch(rt, gpu_ctx):
    """pt_set_output_buffer: each frame of a batched launch lands in the caller's buffer that was current at submission;
    frames that share a target leave the last submitted frame's result (what one launch per frame would leave)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    tris = rt.procedural_scene(0, 15000)
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    w, h, nf = 150, 90, 5
    def params(f):
        return gpu_ctx.make_params(w, h, (0.1 * f, 0, 2.5), (0, 0, 0, 1), mode=rt.PT_MODE_PATH, spp=5, max_bounces=3, seed=8, frame=f)
    want = []
    for f in range(nf):
        gpu_ctx.render(params(f)); want.append(gpu_ctx.read_radiance().copy())
    floats = w * h * 4
    bufs = []
    for f in range(nf):
        p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), C.c_size_t(floats * 4)) == 0; bufs.append(p)
    gpu_ctx.set_batch(nf)
    for f in range(nf):
        gpu_ctx.set_output_buffer(bufs[f].value, floats)
        gpu_ctx.render(params(f))
    last = gpu_ctx.read_radiance().copy()                      # launches the batch; reads the last frame's target
    assert same_bits(last, want[nf - 1])
    for f in range(nf):
        host = np.zeros((h, w, 4), np.float32)
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), bufs[f], C.c_size_t(floats * 4), 2) == 0
        assert same_bits(host, want[f]), f
    # all frames into ONE caller-owned buffer, then into the context's own: the last frame's result stays
    gpu_ctx.set_output_buffer(bufs[0].value, floats)
    for f in range(nf):
        gpu_ctx.render(params(f))
    assert same_bits(gpu_ctx.read_radiance(), want[nf - 1])
    gpu_ctx.set_output_buffer(0, 0)
    for f in range(nf):
        gpu_ctx.render(params(f))
    assert same_bits(gpu_ctx.read_radiance(), want[nf - 1])
    with pytest.raises(rt.PtError):
        gpu_ctx.set_output_buffer(bufs[0].value, floats - 4)
        gpu_ctx.render(params(0))
    gpu_ctx.set_output_buffer(0, 0)
    gpu_ctx.set_batch(1)
    for b in bufs:
        hip.hipFree(b)


def test_largest_batch_accumulates_like_single_launches(rt, gpu_ctx):
    """32 frames in one launch, accumulated: the running sum is the one 32 launches give; 70 independent frames in one launch
    (more than two upload chunks of per-frame parameters) land in their own targets; more than 256 per launch is refused."""
    tris = rt.procedural_scene(0, 12000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    w, h = 96, 64
    def run(batch):
        gpu_ctx.render(gpu_ctx.make_params(w, h, mode=rt.PT_MODE_REFERENCE))  # ends any accumulating sequence
        gpu_ctx.set_batch(batch)
        for f in range(32):
            gpu_ctx.render(gpu_ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=1, max_bounces=4, seed=9, frame=f, accumulate=True))
        out = gpu_ctx.read_radiance().copy()
        gpu_ctx.set_batch(1)
        return out
    a, b = run(1), run(32)
    assert same_bits(a, b)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    nf, floats = 70, w * h * 4
    buf = C.c_void_p(); assert hip.hipMalloc(C.byref(buf), C.c_size_t(floats * 4 * nf)) == 0
    def frame(f):
        return gpu_ctx.make_params(w, h, (0.01 * f, 0, 2.5), (0, 0, 0, 1), mode=rt.PT_MODE_PATH, spp=1, max_bounces=3, seed=9, frame=f)
    gpu_ctx.render(gpu_ctx.make_params(w, h, mode=rt.PT_MODE_REFERENCE))
    gpu_ctx.set_batch(nf)
    for f in range(nf):
        gpu_ctx.set_output_buffer(buf.value + f * floats * 4, floats)
        gpu_ctx.render(frame(f))
    gpu_ctx.synchronize()
    gpu_ctx.set_output_buffer(0, 0); gpu_ctx.set_batch(1)
    got = np.zeros((nf, h, w, 4), np.float32)
    assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), buf, C.c_size_t(floats * 4 * nf), 2) == 0
    hip.hipFree(buf)
    for f in (0, 31, 32, 63, 64, 69):
        gpu_ctx.render(frame(f))
        assert same_bits(got[f], gpu_ctx.read_radiance()), f
    with pytest.raises(rt.PtError):
        gpu_ctx.set_batch(257)


def test_too_many_samples_per_launch_is_rejected(rt, gpu_ctx):
    gpu_ctx.set_triangles(rt.procedural_scene(0, 12000)); gpu_ctx.build_bvh()
    with pytest.raises(rt.PtError) as e:
        gpu_ctx.render(gpu_ctx.make_params(4096, 4096, mode=rt.PT_MODE_PATH, spp=512, max_bounces=1))     # 2^33 samples
    assert e.value.code == 1 and "2^32" in str(e.value)
    gpu_ctx.render(gpu_ctx.make_params(64, 64, mode=rt.PT_MODE_PATH, spp=2, max_bounces=1))                # the context is still usable
    assert np.isfinite(gpu_ctx.read_radiance()).all()


def test_error_paths(rt, gpu_ctx):
    fresh = rt.Context(0)
    with pytest.raises(rt.PtError) as e:
        fresh.num_tris = 0
        fresh.render(fresh.make_params(8, 8))
    assert e.value.code == 4                      # PT_ERR_NO_SCENE
    fresh.set_triangles(TETRA)
    fresh.build_bvh()
    with pytest.raises(rt.PtError):
        fresh.render(fresh.make_params(8, 8, num_tris=99))   # more than uploaded
    bad = fresh.read_bvh4().copy()
    bad[1 + 4] = 0                                # second child of the root points at the root: cycle
    with pytest.raises(rt.PtError) as e2:
        fresh.set_bvh4(bad)
    assert e2.value.code == 5
    # a tile share count beyond what a running accumulation can remember (12 + 12 bits, as pt_set_accum) is refused by pt_render too
    with pytest.raises(rt.PtError) as e3:
        fresh.render(fresh.make_params(64, 64, mode=rt.PT_MODE_PATH, tile_rank=0, tile_count=4096))
    assert e3.value.code == 1 and "4095" in str(e3.value)                 # PT_ERR_INVALID_ARG
    fresh.render(fresh.make_params(64, 64, mode=rt.PT_MODE_PATH, tile_rank=3, tile_count=4095))       # the largest count is fine
    # a checkpoint whose `samples` disagrees with the sample count stored in its own data is refused (first / last pixel's w)
    for f in range(3):
        fresh.render(fresh.make_params(40, 24, mode=rt.PT_MODE_PATH, spp=2, max_bounces=2, frame=f, accumulate=True))
    info, dump = fresh.read_accum()
    assert info.samples == 6 and dump[3] == np.float32(6) and dump[-1] == np.float32(6)
    info.samples = 4
    with pytest.raises(rt.PtError) as e4:
        fresh.set_accum(info, dump)
    assert e4.value.code == 1 and "samples" in str(e4.value)
    info.samples = 6
    tampered = dump.copy(); tampered[-1] = np.float32(5)
    with pytest.raises(rt.PtError):
        fresh.set_accum(info, tampered)
    fresh.set_accum(info, dump)                                           # the untouched dump is accepted
    fresh.close()
