"""Edge cases the reference handles explicitly (empty scene, single triangle, odd sizes) and the
full-size configurations through size-independent properties."""
import numpy as np
import pytest

import orc as orc_mod
from scenes import TETRA, random_soup

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def test_empty_scene_renders_background(rt, gpu_ctx):
    # numTris == 0: BVH[0] = 0, every ray misses (renderer.wgsl:224, PathTracer.js:701-707)
    gpu_ctx.set_triangles(np.zeros(0, np.float32))
    gpu_ctx.build_bvh()
    assert gpu_ctx.scene_info() == {"numTris": 0, "numNodes2": 0, "numNodes4": 0}
    assert np.array_equal(gpu_ctx.read_bvh4(), np.array([0], np.uint32))
    for mode in (rt.PT_MODE_REFERENCE_PACKET, rt.PT_MODE_REFERENCE, rt.PT_MODE_PATH):
        gpu_ctx.render(gpu_ctx.make_params(40, 24, mode=mode, spp=2, max_bounces=2))
        img = gpu_ctx.read_radiance()
        assert np.all(img[..., :3] == np.float32(0.01)) and np.all(img[..., 3] == 1.0)


def test_single_triangle_root_leaf(rt, orc, gpu_ctx):
    tris = np.array([-1, -1, 0, 1, -1, 0, 0, 1, 0], np.float32)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh4 = gpu_ctx.read_bvh4()
    assert bvh4[0] == 1 and (bvh4[8] & 0x80000000)        # the root is the only node and it is a leaf
    for mode, omode in ((rt.PT_MODE_REFERENCE_PACKET, orc_mod.MODE_PACKET), (rt.PT_MODE_REFERENCE, orc_mod.MODE_SINGLE), (rt.PT_MODE_PATH, orc_mod.MODE_PATH)):
        gpu_ctx.render(gpu_ctx.make_params(50, 30, mode=mode, spp=2, max_bounces=2, seed=2))
        ref, _, _ = orc.render(orc.make_params(50, 30, 1, mode=omode, spp=2, max_bounces=2, seed=2), tris, bvh4)
        assert same_bits(gpu_ctx.read_radiance(), ref)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (9, 17), (257, 131)])
def test_odd_resolutions(rt, orc, gpu_ctx, w, h):
    tris = random_soup(600, 4)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh4 = gpu_ctx.read_bvh4()
    for mode, omode in ((rt.PT_MODE_REFERENCE_PACKET, orc_mod.MODE_PACKET), (rt.PT_MODE_REFERENCE, orc_mod.MODE_SINGLE), (rt.PT_MODE_PATH, orc_mod.MODE_PATH)):
        gpu_ctx.render(gpu_ctx.make_params(w, h, mode=mode, spp=3, max_bounces=2, seed=8))
        ref, _, _ = orc.render(orc.make_params(w, h, 600, mode=omode, spp=3, max_bounces=2, seed=8), tris, bvh4)
        assert same_bits(gpu_ctx.read_radiance(w, h), ref), (mode, w, h)


def test_num_tris_smaller_than_uploaded(rt, orc, gpu_ctx):
    # the UBO's numTris gates leaf tests (`ti < numTris`, renderer.wgsl:267): triangles past it are invisible
    tris = random_soup(500, 6)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh4 = gpu_ctx.read_bvh4()
    gpu_ctx.render(gpu_ctx.make_params(96, 64, mode=rt.PT_MODE_REFERENCE, num_tris=200))
    ref, _, _ = orc.render(orc.make_params(96, 64, 200, mode=orc_mod.MODE_SINGLE), tris, bvh4)
    assert same_bits(gpu_ctx.read_radiance(), ref)


def test_num_tris_gate_and_out_of_range_leaves_in_path_mode(rt, orc, gpu_ctx):
    """`ti < numTris` (renderer.wgsl:267) in the megakernel: (a) the UBO's numTris smaller than the uploaded count hides the
    triangles past it, (b) a leaf of a supplied BVH4 whose triangle index is out of range is entered and tests nothing (the
    wide layout points it at a never-hit record).  Image and counters equal the oracle's, instrumented kernel or not."""
    tris = random_soup(700, 9)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh4 = gpu_ctx.read_bvh4().copy()
    kw = dict(mode=rt.PT_MODE_PATH, spp=2, max_bounces=4, seed=5, frame=2)
    cam, quat = (0.1, 0.2, 2.6), (0, 0, 0, 1)

    def check(num_tris, tree, what):
        ref, _, ost = orc.render(orc.make_params(112, 72, num_tris, cam, quat, mode=orc_mod.MODE_PATH, spp=2, max_bounces=4, seed=5, frame=2), tris, tree)
        p = gpu_ctx.make_params(112, 72, cam, quat, num_tris=num_tris, stats=True, **kw)
        gpu_ctx.render(p)
        assert same_bits(gpu_ctx.read_radiance(), ref), what
        st = gpu_ctx.stats()
        for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples", "max_stack"):
            assert st[k] == ost[k], (what, k)
        p.flags = 0
        gpu_ctx.render(p)
        assert same_bits(gpu_ctx.read_radiance(), ref), what + " (uninstrumented)"
        p.flags = rt.PT_FLAG_SIMPLE_KERNEL
        gpu_ctx.render(p)
        assert same_bits(gpu_ctx.read_radiance(), ref), what + " (one pixel per lane)"

    check(250, bvh4, "numTris 250 of 700")
    # every fifth leaf gets a triangle index past the end
    m = int(bvh4[0]); rec = bvh4[1:1 + 8 * m].reshape(m, 8)
    leaves = np.nonzero(rec[:, 7] & 0x80000000)[0][::5]
    rec[leaves, 7] = 0x80000000 | (700 + (leaves % 97))
    gpu_ctx.set_bvh4(bvh4)
    check(700, bvh4, "out-of-range leaves")
    check(300, bvh4, "out-of-range leaves and numTris 300")


def test_rgba8_and_tonemap_match_oracle(rt, orc, gpu_ctx):
    tris = rt.procedural_scene(0, 20000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    gpu_ctx.render(gpu_ctx.make_params(160, 90, mode=rt.PT_MODE_REFERENCE))
    img = gpu_ctx.read_radiance()
    want8 = np.floor(np.clip(img, 0, 1) * 255 + 0.5).astype(np.uint8)             # rgba8unorm store
    assert np.array_equal(gpu_ctx.read_rgba8(), want8)
    tm = gpu_ctx.read_tonemapped(True).astype(np.int32)
    ref = orc.tonemap(img, quantize=True).astype(np.int32)                          # tonemapper.wgsl + vertical flip
    assert np.array_equal(tm, ref)                                                  # x^(1/2.2) is one pinned f32 evaluation on both sides (pow_1_2_2): bit for bit
    tm2 = gpu_ctx.read_tonemapped(False).astype(np.int32)
    assert np.array_equal(tm2, orc.tonemap(img, quantize=False).astype(np.int32))


@pytest.fixture(scope="module")
def full_scene(rt, gpu_ctx_full):
    return gpu_ctx_full


@pytest.fixture(scope="module")
def gpu_ctx_full(rt):
    ctx = rt.Context(0)
    tris = rt.procedural_scene(0, 871414)
    ctx.set_triangles(tris)
    ctx.build_bvh()
    ctx._tris = tris
    yield ctx
    ctx.close()


def test_full_size_c2_properties(rt, orc, gpu_ctx_full):
    """BASELINE config C2 at full size: determinism, kernel A/B equality, oracle agreement on a pixel
    subsample, counter consistency, BVH invariants."""
    ctx = gpu_ctx_full
    tris = ctx._tris
    info = ctx.scene_info()
    assert info["numTris"] == 871414 and info["numNodes2"] == 2 * 871414 - 1
    bvh2, bvh4 = ctx.read_bvh2(), ctx.read_bvh4()
    assert bvh4[0] == info["numNodes4"] and 871414 < info["numNodes4"] <= 2 * 871414 - 1
    leaves = bvh4[8::8] & 0x80000000 != 0
    assert leaves.sum() == 871414                                                  # one leaf per triangle
    assert np.array_equal(np.sort(bvh4[8::8][leaves] & 0x7FFFFFFF), np.arange(871414, dtype=np.uint32))
    w, h = 1920, 1080
    kw = dict(mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=1)
    ctx.render(ctx.make_params(w, h, stats=True, **kw)); a = ctx.read_radiance().copy(); st = ctx.stats()
    ctx.render(ctx.make_params(w, h, **kw)); b = ctx.read_radiance().copy()
    ctx.render(ctx.make_params(w, h, simple_kernel=True, **kw)); c = ctx.read_radiance().copy()
    assert same_bits(a, b) and same_bits(a, c)                                      # deterministic; both kernels agree
    assert st["samples"] == w * h * 4 and st["rays_closest"] >= st["samples"] and st["stack_drops"] == 0
    # oracle on every 16th pixel in x and y (8100 pixels x 4 spp): bit-exact
    ref, _, ost = orc.render(orc.make_params(w, h, 871414, mode=orc_mod.MODE_PATH, spp=4, max_bounces=8, seed=1, step=(16, 16)), tris, bvh4)
    assert same_bits(a[::16, ::16], ref[::16, ::16])
    assert np.isfinite(a).all() and a[..., :3].min() >= 0.0


def test_full_size_device_build_equals_host_build(rt, gpu_ctx_full):
    """871,414 triangles: pt_build_bvh (all on the device) against the host entry points that mirror the reference's
    JavaScript steps (pt_morton_sort = buildMortonAndSort, pt_collapse_lbvh2_to_bvh4 = collapseLBVH2ToBVH4)."""
    ctx = gpu_ctx_full
    bvh2, bvh4 = ctx.read_bvh2(), ctx.read_bvh4()
    morton, tri_index = rt.morton_sort(ctx._tris)
    leaf_tris = bvh2[1 + 6 * (871414 - 1) + 5::6] & 0x7FFFFFFF              # leaf i holds sorted triangle i
    assert np.array_equal(leaf_tris, tri_index)
    want4, n4 = rt.collapse_lbvh2_to_bvh4(bvh2, 871414)
    assert n4 == bvh4[0] and np.array_equal(bvh4, want4[: 1 + 8 * n4])


def test_full_size_c3_bvh4_wide(rt, orc, gpu_ctx_full):
    # config C3's input: BVH4_wide of the same BVH2 (M = 2N-1 nodes): closest hits identical to the collapsed BVH4
    ctx = gpu_ctx_full
    bvh2 = ctx.read_bvh2()
    collapsed = ctx.read_bvh4().copy()
    ctx.render(ctx.make_params(960, 540, mode=rt.PT_MODE_REFERENCE)); ref = ctx.read_radiance().copy()
    wide = rt.bvh2_to_bvh4_wide(bvh2)
    assert wide[0] == 2 * 871414 - 1
    ctx.set_bvh4(wide)
    ctx.render(ctx.make_params(960, 540, mode=rt.PT_MODE_REFERENCE)); got = ctx.read_radiance().copy()
    ctx.set_bvh4(collapsed)
    diff = (got.view(np.uint32) != ref.view(np.uint32)).any(axis=2).mean()
    assert diff < 1e-4                                                              # same closest hit except exact-t ties


def test_full_size_c5_4k_accumulate_smoke(rt, gpu_ctx_full):
    # config C5 shape: 3840x2160, 4 spp per frame, 16 bounces, accumulated over frames (2 here)
    ctx = gpu_ctx_full
    for f in range(2):
        ctx.render(ctx.make_params(3840, 2160, mode=rt.PT_MODE_PATH, spp=4, max_bounces=16, seed=1, frame=f, accumulate=True))
    img = ctx.read_radiance()
    assert img.shape == (2160, 3840, 4) and np.isfinite(img).all()
    assert (img[..., 0] > 0.011).mean() > 0.08                                     # the object is there
    ms = ctx.last_render_ms()
    assert ms > 0


@pytest.mark.parametrize("gpus,steps,fixed_batch", [(2, 40, None), (2, 23, None), (2, 23, "8"), (4, 20, None)])
def test_bench_two_ranks_gloo_rehearsal_verifies_gathered_frame(gpus, steps, fixed_batch):
    """bench.py's N > 1 path end to end on the one-GPU box, started the way the driver starts it (`python bench.py --gpus 2`, no
    launcher): bench.py spawns its two ranks itself, they share cuda:0, the gather is staged through gloo
    (PT_BENCH_BACKEND=gloo), rank 0 de-interleaves every launch with one call and the last timed frame is checked against the CPU oracle.
    40 steps = four launches of 10 = full batches; 23 steps = 6, 6, 6, 5 with a 1 / 1 / 1 warm-up: launches SHORTER than the batch, so
    pt_flush and a gather of fewer frames than the buffers hold are on the path; PT_BENCH_BATCH=8 = fixed batches 8, 8, 7.  The last case
    is the driver's command shape (20 steps: 5, 5, 5, 5) with FOUR ranks on the one GPU (within the box's limit of six GPU processes)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "PT_BENCH_BATCH", "PT_BENCH_SCHEDULE")}
    env.update(PT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    if fixed_batch:
        env["PT_BENCH_BATCH"] = fixed_batch
    out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(gpus), "--steps", str(steps), "--warmup", "3" if gpus == 2 else "5",
                                   "--width", "640", "--height", "360"], env=env, cwd=root, text=True, stderr=subprocess.STDOUT, timeout=600)
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == gpus and res["verified"] is True and res["value"] > 0 and res["scaling"] == "strong"


@pytest.mark.parametrize("fail", ["1:1", "0:0"])
def test_bench_rank_failure_ends_the_run_nonzero_and_fast(fail):
    """`python bench.py --gpus 2` (gloo staging, both ranks on cuda:0) with one rank raising in the middle of the timed region (PT_BENCH_FAIL =
    "rank:launch"): the failing rank reports and exits through bench.abort_rank, torch.distributed.run stops the other one, bench.py returns
    non-zero with no result line -- in seconds, not at the collective timeout and never at the driver's limit."""
    import os, subprocess, sys, time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "PT_BENCH_BATCH", "PT_BENCH_SCHEDULE")}
    env.update(PT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", PT_BENCH_FAIL=fail, PT_BENCH_TIMEOUT="60")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--width", "320", "--height", "180"],
                       env=env, cwd=root, text=True, capture_output=True, timeout=300)
    dt = time.time() - t0
    assert r.returncode != 0
    assert "injected failure (PT_BENCH_FAIL) on rank %s" % fail.split(":")[0] in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]          # no result line from a run that lost a rank
    assert dt < 150, dt


def test_bench_single_gpu_contract_line():
    """bench.py at N = 1 (small frame, few steps): ONE JSON line with the contract's keys, the roofline and CPU-baseline
    objects, a frame verified against the oracle, kernel busy time consistent with the step time, and the figures for the
    reference's own call shape (one render() per frame)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.check_output([sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "3", "--width", "480", "--height", "272",
                                   "--no-cpu-baseline"], cwd=root, text=True, stderr=subprocess.DEVNULL, timeout=600)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["steps"] == 12 and r["warmup"] == 3 and r["unit"] == "Msamples/s" and r["higher_is_better"] is True
    assert r["verified"] is True and r["vs_baseline"] is None and r["dtype"] == "f32" and "workload" in r["config"]
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_busy_ms", "algorithmic"):
        assert k in rf, k
    assert rf["frames_per_launch"] == 12.0 and rf["launches"] == 1
    # the timed region is repeated; the line carries the median repetition and the spread
    assert r["reps"] == 9 and len(r["ms_per_step_all"]) == 9 and r["ms_per_step_min_max"][0] <= r["ms_per_step"] <= r["ms_per_step_min_max"][1]
    assert sorted(r["ms_per_step_all"])[4] == r["ms_per_step"] and sorted(r["ms_per_step_all"][:5])[2] == r["ms_per_step_first5"]
    # ... and is followed by a second or so of back-to-back launches that end on the verified frame, bit for bit
    assert r["sustained"]["last_frame_identical_to_verified"] is True and r["sustained"]["frames"] >= 256 and r["sustained"]["seconds"] >= 1.0
    assert set(rf["peaks"]) == {"l1_gather_requests", "valu_issue", "salu_issue", "l2_bandwidth", "hbm_fabric"}
    assert {k for k, v in rf["peaks"].items() if v["source"] == "guide"} == {"valu_issue", "l2_bandwidth", "hbm_fabric"}
    assert rf["bound"] in ("valu", "l2", "hbm", "unmeasured")                         # the headline is taken among the guide peaks only
    assert rf["kernel"].startswith("trace_paths_kernel<0, true>")                    # the variant that ran: this scene and camera are bounded
    side = rf["lane_utilisation_by_side"]
    assert side["frames"] == 12 and 0 < side["lanes_per_step"]["leaf_side"] < side["lanes_per_step"]["node_side"] < side["lanes_per_step"]["traversing"] <= 64
    assert abs(side["lanes_per_step"]["node_side"] + side["lanes_per_step"]["leaf_side"] - side["lanes_per_step"]["traversing"]) < 0.02
    # the trace kernel is busy for most of the timed region and never longer than it
    assert 0.3 * r["ms_per_step"] * 12 < rf["kernel_busy_ms"] <= r["ms_per_step"] * 12 * 1.02
    assert rf["frac"] is None or 0.0 < rf["frac"] <= 1.0
    assert abs(r["value"] - 480 * 272 * 4 / (r["ms_per_step"] * 1e-3) / 1e6) / r["value"] < 1e-2
    shape = r["reference_call_shape"]
    assert shape["ms_per_frame_solo"] >= shape["ms_per_frame_pipelined"] * 0.9 > 0


def test_prebuilt_bvh_files_roundtrip(rt, orc, gpu_ctx, tmp_path):
    """configs C2/C3 name data/BVH2.bin and data/BVH4_wide.bin: dump, reload into a fresh context, render the same."""
    tris = rt.procedural_scene(0, 20000)
    gpu_ctx.set_triangles(tris)
    gpu_ctx.build_bvh()
    bvh2, bvh4 = gpu_ctx.read_bvh2(), gpu_ctx.read_bvh4()
    p2, pw = str(tmp_path / "BVH2.bin"), str(tmp_path / "BVH4_wide.bin")
    rt.write_u32_file(p2, bvh2)
    rt.write_u32_file(pw, rt.bvh2_to_bvh4_wide(rt.read_u32_file(p2)))
    gpu_ctx.render(gpu_ctx.make_params(160, 90, mode=rt.PT_MODE_PATH, spp=2, max_bounces=3)); want = gpu_ctx.read_radiance().copy()
    fresh = rt.Context(0)
    fresh.set_triangles(tris)
    fresh.set_bvh2(rt.read_u32_file(p2))                       # collapse on load, like buildBVH after its read-back
    assert np.array_equal(fresh.read_bvh4(), bvh4) and np.array_equal(fresh.read_bvh2(), bvh2)
    fresh.render(fresh.make_params(160, 90, mode=rt.PT_MODE_PATH, spp=2, max_bounces=3))
    assert same_bits(fresh.read_radiance(), want)
    fresh.set_bvh4(rt.read_u32_file(pw))                       # the offline converter's output is a valid BVH input as well
    fresh.render(fresh.make_params(160, 90, mode=rt.PT_MODE_REFERENCE)); a = fresh.read_radiance().copy()
    gpu_ctx.render(gpu_ctx.make_params(160, 90, mode=rt.PT_MODE_REFERENCE)); b = gpu_ctx.read_radiance()
    assert ((a.view(np.uint32) != b.view(np.uint32)).any(axis=2)).mean() < 1e-3
    with pytest.raises(rt.PtError):
        fresh.set_bvh2(bvh2[:-5])                              # truncated file
    fresh.close()


def test_timing_ring_survives_a_launch_that_traces_nothing(rt, gpu_ctx):
    """The root-box cull can remove every owned tile (scene in front of the eye but off-screen): the launch then has no items, and
    the timing pair of the ring slot it consumed must still be recorded -- pt_timing_collect used to fail on never-recorded events
    and leave the ring broken."""
    tris = (random_soup(300, 2).reshape(-1, 3) * np.float32(0.05) + np.array([30.0, 0.0, -3.0], np.float32)).astype(np.float32).reshape(-1)   # far off to the right
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    p = gpu_ctx.make_params(96, 64, cam_pos=(0, 0, 2.5), mode=rt.PT_MODE_PATH, spp=2, max_bounces=2)
    gpu_ctx.timing_begin(3)
    gpu_ctx.render(p); gpu_ctx.render(p)
    ms = gpu_ctx.timing_collect(3)
    assert len(ms) == 2 and np.all(ms >= 0.0) and np.all(ms < 5.0)
    img = gpu_ctx.read_radiance()
    assert np.all(img[..., :3] == np.float32(0.01))
    gpu_ctx.timing_begin(2)                      # and the ring is usable afterwards
    gpu_ctx.render(p)
    assert len(gpu_ctx.timing_collect(2)) == 1


def test_buffer_busy_tracks_queued_and_delivered_frames(rt, gpu_ctx):
    """pt_buffer_busy: a caller-owned target is busy while a frame queued by pt_set_batch (or still in flight) points into it, and
    free once everything has been delivered; pt_set_output_buffer(NULL) launches what is queued AND waits for it."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    tris = random_soup(400, 5)
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    w, h = 64, 40
    nbytes = w * h * 16
    bufs = []
    for _ in range(2):
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), C.c_size_t(nbytes)) == 0
        bufs.append(ptr.value)
    try:
        gpu_ctx.set_batch(4)
        p = gpu_ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=2, max_bounces=2)
        gpu_ctx.set_output_buffer(bufs[0], w * h * 4); gpu_ctx.render(p)
        gpu_ctx.set_output_buffer(bufs[1], w * h * 4); p.frame = 1; gpu_ctx.render(p)
        assert gpu_ctx.buffer_busy(bufs[0], nbytes) and gpu_ctx.buffer_busy(bufs[1], nbytes)      # queued, not launched yet
        assert gpu_ctx.buffer_busy(bufs[0] + 16, 16) and gpu_ctx.buffer_busy(bufs[0] + nbytes - 4, 64)    # any overlap with a frame's bytes counts
        assert not gpu_ctx.buffer_busy(bufs[0] + nbytes, 64) or bufs[1] == bufs[0] + nbytes        # just behind the frame: free
        gpu_ctx.set_output_buffer(0, 0)                                                         # launches the partial batch and waits
        assert not gpu_ctx.buffer_busy(bufs[0], nbytes) and not gpu_ctx.buffer_busy(bufs[1], nbytes)
        out = np.zeros((h, w, 4), np.float32)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(bufs[1]), C.c_size_t(nbytes), C.c_int(2)) == 0
        gpu_ctx.set_batch(1)
        gpu_ctx.render(p)
        assert same_bits(out, gpu_ctx.read_radiance())
        # More launches in flight than there are frame slots: the first launch targets bufs[0], two dozen heavier ones
        # into bufs[1] follow without a host wait; bufs[0] must read busy until ITS launch has delivered, whatever slot is reused meanwhile,
        # and free afterwards without anybody having waited for the later ones explicitly.
        tris2 = rt.procedural_scene(0, 20000)
        gpu_ctx.set_triangles(tris2); gpu_ctx.build_bvh()
        big_w, big_h = 512, 288
        big = []
        for _ in range(2):
            ptr = C.c_void_p()
            assert hip.hipMalloc(C.byref(ptr), C.c_size_t(big_w * big_h * 16)) == 0
            big.append(ptr.value)
        bufs.extend(big)
        q = gpu_ctx.make_params(big_w, big_h, mode=rt.PT_MODE_PATH, spp=8, max_bounces=8)
        gpu_ctx.synchronize()
        gpu_ctx.set_output_buffer(big[0], big_w * big_h * 4); gpu_ctx.render(q)
        gpu_ctx.set_output_buffer(big[1], big_w * big_h * 4)
        seen_busy_first = gpu_ctx.buffer_busy(big[0], 16)
        for i in range(24):
            q.frame = i + 1; gpu_ctx.render(q)
        assert gpu_ctx.buffer_busy(big[1], big_w * big_h * 16)                                     # 24 launches queued behind each other
        assert seen_busy_first                                                                       # the launch had just been submitted
        r = gpu_ctx.make_params(big_w, big_h, mode=rt.PT_MODE_PATH, spp=8, max_bounces=8, simple_kernel=True)       # the one-pixel-per-lane kernel is tracked as well
        gpu_ctx.set_output_buffer(big[0], big_w * big_h * 4); gpu_ctx.render(r)
        assert gpu_ctx.buffer_busy(big[0] + 1024, 4)
        gpu_ctx.synchronize()
        assert not gpu_ctx.buffer_busy(big[0], big_w * big_h * 16) and not gpu_ctx.buffer_busy(big[1], big_w * big_h * 16)
        gpu_ctx.set_output_buffer(0, 0)
    finally:
        for b in bufs:
            hip.hipFree(C.c_void_p(b))


def test_deinterleave_batch_scatters_every_frame_of_a_gathered_batch(rt, gpu_ctx):
    """pt_deinterleave_batch: three frames, three tile shares rendered into one [rank][frame][stride] buffer (the layout a gather of
    a batch leaves on the root), scattered by ONE launch into three row-major frames; each equals the whole-frame render."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    tris = random_soup(500, 7)
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    w, h, world, frames = 100, 52, 3, 3
    stride = max(rt.tile_layout(w, h, r, world)[1] for r in range(world))
    g_ptr, f_ptr = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(g_ptr), C.c_size_t(world * frames * stride * 4)) == 0
    assert hip.hipMalloc(C.byref(f_ptr), C.c_size_t(frames * w * h * 16)) == 0
    try:
        for r in range(world):
            for j in range(frames):
                gpu_ctx.set_compact_buffer(g_ptr.value + (r * frames + j) * stride * 4, stride)
                gpu_ctx.render(gpu_ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, seed=3, frame=j, tile_rank=r, tile_count=world))
        gpu_ctx.synchronize()
        gpu_ctx.deinterleave_batch(g_ptr.value, frames * stride, stride, frames, w, h, world, f_ptr.value, w * h * 4)
        last = gpu_ctx.read_radiance(w, h).copy()                       # the read-backs see the last frame of the batch
        got = np.zeros((frames, h, w, 4), np.float32)
        gpu_ctx.synchronize()
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), f_ptr, C.c_size_t(got.nbytes), C.c_int(2)) == 0
        gpu_ctx.set_compact_buffer(0, 0)
        for j in range(frames):
            gpu_ctx.render(gpu_ctx.make_params(w, h, mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, seed=3, frame=j))
            assert same_bits(got[j], gpu_ctx.read_radiance()), j
        assert same_bits(last, got[frames - 1])
        # the context's own frame buffer as the target: only the last frame is scattered
        gpu_ctx.deinterleave_batch(g_ptr.value, frames * stride, stride, frames, w, h, world)
        assert same_bits(gpu_ctx.read_radiance(w, h), got[frames - 1])
        with pytest.raises(rt.PtError):
            gpu_ctx.deinterleave_batch(g_ptr.value, stride, stride, frames, w, h, world, f_ptr.value, w * h * 4)     # rank stride too small for three frames
    finally:
        hip.hipFree(g_ptr); hip.hipFree(f_ptr)


@pytest.mark.parametrize("w,h,world,cam,spp", [(200, 120, 3, (0, 0, 2.5), 2), (97, 61, 4, (0.2, -0.1, 4.0), 3), (160, 96, 2, (0, 0, 0.2), 1), (96, 64, 8, (30.0, 0, 2.5), 2)])
def test_packed_shares_rebuild_every_frame(rt, gpu_ctx, w, h, world, cam, spp):
    """pt_traced_tile_rect / pt_pack_shares / pt_unpack_batch: each rank's frames packed (its tiles inside the traced rectangle, 12 bytes per
    pixel), laid out as one gather would leave them on the root, rebuilt by ONE launch -- every frame equals the whole-frame render bit for
    bit: a rectangle inside the image, the whole image (camera inside the scene), an EMPTY rectangle (scene off-screen: nothing travels),
    odd sizes, 2 / 3 / 4 / 8 ranks; the rectangle is the one the launches trace; pt_packed_tile_ids is the buffer's order."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    tris = (random_soup(500, 7).reshape(-1, 3) * np.float32(0.6)).astype(np.float32).reshape(-1)
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    frames = 3
    kw = dict(mode=rt.PT_MODE_PATH, spp=spp, max_bounces=3, seed=3)
    rect = gpu_ctx.traced_tile_rect(gpu_ctx.make_params(w, h, cam, **kw))
    tiles_x, tiles_y = (w + 7) // 8, (h + 7) // 8
    assert 0 <= rect[0] <= rect[2] <= tiles_x and 0 <= rect[1] <= rect[3] <= tiles_y
    if cam[2] < 1.0: assert rect == (0, 0, tiles_x, tiles_y)                 # camera inside the root box: nothing can be left out
    if cam[0] > 10.0: assert (rect[2] - rect[0]) * (rect[3] - rect[1]) == 0  # scene far off-screen: nothing to trace, nothing to ship
    stride = max(rt.tile_layout(w, h, r, world)[1] for r in range(world))
    max_tiles, pstride = rt.packed_layout(w, h, world, rect)
    assert pstride == max_tiles * 192 and pstride <= stride * 3 // 4
    for r in range(world):
        ids = rt.packed_tile_ids(w, h, r, world, rect).tolist()
        assert ids == [t for t in rt.tile_ids(w, h, r, world).tolist() if rect[0] <= t % tiles_x < rect[2] and rect[1] <= t // tiles_x < rect[3]] and len(ids) <= max_tiles
    c_ptr, p_ptr, f_ptr = C.c_void_p(), C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(c_ptr), C.c_size_t(frames * stride * 4)) == 0
    assert hip.hipMalloc(C.byref(p_ptr), C.c_size_t(max(world * frames * pstride * 4, 64))) == 0
    assert hip.hipMalloc(C.byref(f_ptr), C.c_size_t(frames * w * h * 16)) == 0
    try:
        for r in range(world):
            for j in range(frames):
                gpu_ctx.set_compact_buffer(c_ptr.value + j * stride * 4, stride)
                gpu_ctx.render(gpu_ctx.make_params(w, h, cam, frame=j, tile_rank=r, tile_count=world, **kw))
            if pstride:        # rank r's packed frames where a gather would put them: [rank][frame][pstride]
                gpu_ctx.pack_shares(c_ptr.value, stride, frames, w, h, r, world, rect, p_ptr.value + r * frames * pstride * 4, pstride)
            gpu_ctx.synchronize()
        gpu_ctx.unpack_batch(p_ptr.value, frames * pstride, pstride, frames, w, h, world, rect, spp, f_ptr.value, w * h * 4)
        last = gpu_ctx.read_radiance(w, h).copy()
        got = np.zeros((frames, h, w, 4), np.float32)
        gpu_ctx.synchronize()
        assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), f_ptr, C.c_size_t(got.nbytes), C.c_int(2)) == 0
        gpu_ctx.set_compact_buffer(0, 0)
        for j in range(frames):
            gpu_ctx.render(gpu_ctx.make_params(w, h, cam, frame=j, **kw))
            assert same_bits(got[j], gpu_ctx.read_radiance()), j
        assert same_bits(last, got[frames - 1])
        gpu_ctx.unpack_batch(p_ptr.value, frames * pstride, pstride, frames, w, h, world, rect, spp)          # own frame buffer: the last frame
        assert same_bits(gpu_ctx.read_radiance(w, h), got[frames - 1])
        with pytest.raises(rt.PtError):
            gpu_ctx.unpack_batch(p_ptr.value, frames * pstride, pstride, frames, w, h, world, (0, 0, tiles_x + 1, tiles_y), spp)     # rectangle outside the image
    finally:
        hip.hipFree(c_ptr); hip.hipFree(p_ptr); hip.hipFree(f_ptr)


def test_timeline_variant_same_bits_and_a_sane_record(rt, gpu_ctx, orc):
    """The TIMELINE variant of the megakernel (knob TIMELINE; tools/wave_timeline.py, bench.py's lane_utilisation_by_side) is the production kernel plus
    wave-uniform bookkeeping: the image is the production variant's (and the oracle's) bit for bit -- single launches, a batched launch, a tile share --,
    and every wavefront leaves a record that adds up: begin <= queue-dry <= end, lanes at a leaf <= traversing lanes <= 64 x steps, the snapshot taken when
    the queue ran dry <= the totals.  With the knob off again no record is written."""
    import ctypes as C
    import orc as orc_mod
    tris = rt.procedural_scene(0, 60000)
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    bvh4 = gpu_ctx.read_bvh4()
    w, h = 640, 360
    kw = dict(mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=2)
    gpu_ctx.render(gpu_ctx.make_params(w, h, frame=3, **kw))
    want = gpu_ctx.read_radiance().copy()
    ref, _, ost = orc.render(orc.make_params(w, h, tris.size // 9, mode=orc_mod.MODE_PATH, spp=4, max_bounces=8, seed=2, frame=3), tris, bvh4)
    assert np.array_equal(want.view(np.uint32), ref.view(np.uint32))

    def record():
        buf = np.zeros((8192, 24), np.uint64); n = C.c_uint32()
        rt.lib.pt_debug_wave_times(gpu_ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_uint32(8192), C.byref(n))
        return buf[: n.value].astype(np.int64)

    gpu_ctx.debug_set_tune("TIMELINE", 1)
    gpu_ctx.render(gpu_ctx.make_params(w, h, frame=3, **kw))
    assert np.array_equal(gpu_ctx.read_radiance().view(np.uint32), want.view(np.uint32))
    r = record()
    assert len(r) >= 1024 and (r[:, 20] == 2).all()                              # every wavefront of the grid wrote its record, as the TIMELINE variant
    ran = r[:, 3] > 0
    assert ran.any()
    assert (r[:, 0] <= r[:, 2]).all() and (r[ran, 0] <= r[ran, 1]).all() and (r[ran, 1] <= r[ran, 2]).all()        # begin <= queue dry <= end
    assert (r[:, 9] <= r[:, 7]).all() and (r[:, 7] <= 64 * r[:, 3]).all()                                           # leaf lanes <= traversing lanes <= 64 x steps
    assert (r[:, 6] <= r[:, 3]).all() and (r[:, 8] <= r[:, 7]).all() and (r[:, 15] <= r[:, 4]).all()                # the queue-dry snapshot <= the totals
    assert (r[:, 10:14] == 0).all()                                                                                # cycle shares: COUNTERS variant only
    steps_c, lanes_c = r[:, 3].sum(), r[:, 7].sum()
    # the counts are the algorithm's, not the variant's: the COUNTERS variant of the same frame steps the same lanes (wavefront-steps vary with scheduling, lane-steps do not)
    gpu_ctx.debug_set_tune("TIMELINE")
    gpu_ctx.render(gpu_ctx.make_params(w, h, frame=3, stats=True, **kw))
    st = gpu_ctx.stats()
    rc = record()
    assert (rc[:, 20] == 1).all() and rc[:, 7].sum() == lanes_c and abs(int(rc[:, 3].sum()) - int(steps_c)) < 0.05 * steps_c
    for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested"):
        assert st[k] == ost[k]
    # a batched launch and a tile share on the TIMELINE variant: the same bits
    gpu_ctx.debug_set_tune("TIMELINE", 1)
    gpu_ctx.set_batch(3)
    for f in (1, 2, 3):
        gpu_ctx.render(gpu_ctx.make_params(w, h, frame=f, **kw))
    assert np.array_equal(gpu_ctx.read_radiance().view(np.uint32), want.view(np.uint32))
    gpu_ctx.set_batch(1)
    gpu_ctx.debug_set_tune("TIMELINE")
