#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

One *step* = one complete frame of the C2 workload: dragon-class procedural mesh (871,414
triangles, the reference's dragon.glb is absent), native LBVH2 -> collapsed BVH4, 1920x1080,
4 samples per pixel, 8 bounces, camera (0,0,2.5) / identity quaternion / FOV 70 deg
(src/main.js:12, PathTracer.js:761).  value = Msamples/s = W*H*spp*K / wall time of the K
timed steps (max over ranks), scene already resident in HBM.

N > 1: one process per GPU.  Under torch.distributed.run the ranks come from the environment; started
plainly (`python bench.py --gpus N`) this process launches the N ranks itself -- as children, before anything has
touched a GPU -- and exits with their status.  The frame is sharded by interleaved 8x8 pixel tiles
(rank = (tx+ty) % N); each rank renders its tiles into a compact buffer, RCCL gathers the buffers on rank 0 over
xGMI (torch.distributed backend "nccl"), rank 0 de-interleaves them into the full frame.  The gather of launch b
overlaps the trace of launch b+1.  Total work per step is fixed -> "scaling": "strong".

Every run checks one TIMED frame against the CPU oracle on a fixed pixel grid (every 4th pixel in x and y, all samples,
bit for bit) and prints "verified": true; a mismatch ends the run with a non-zero status and no result line.

The timed region (exactly K steps between barriers + synchronisations, max over ranks) is repeated --reps times (default 9;
SURVEY 8d asks for at least 3) after the warm-up; `value` / `ms_per_step` are the MEDIAN repetition, `ms_per_step_min_max` the spread, `ms_per_step_all`
every repetition in order.  (Nine since round 5: behind a 5-step warm-up the chip is still ramping -- twelve repetitions of the driver's command read 0.777, 0.727, 0.721,
0.708, 0.717, 0.714, 0.705, 0.707, 0.705, 0.711, 0.703, 0.705 ms per step -- and the median of five sat on the ramp.)

The JSON line's `roofline`: `fractions` holds, for every resource a gather-and-compute kernel can be bound by (vector issue, scalar issue, L2 bandwidth, HBM-side
traffic, the vector L1's request rate), per-frame counter total / kernel busy time / peak; `peaks` says where each peak comes from -- "guide" =
/opt/skills/guides/MI355X_MICROARCH.md, "self-measured" = a probe under tools/probes/ -- and `bound` / `frac` are the largest fraction AMONG THE GUIDE PEAKS, so the
headline can be recomputed from profiles/r06_pmc_bench.json and the guide alone.  The kernel's busy time is measured live (union of the per-launch hipEvent intervals on
the launch streams, median repetition); the per-frame counter totals come from the rocprofv3 --pmc passes of THIS command committed under profiles/ (tools/pmc_bench.sh).
`kernel` names the variant that actually ran (the host picks the reciprocal forms per scene and camera).  `lane_utilisation_by_side` decomposes the kernel's lane
utilisation: wavefront-steps, traversing lanes and lanes at a leaf from one launch on the TIMELINE variant (production speed), against the static vector instructions
of the step's sides.  `responds_to` carries what probe builds of the production step showed the kernel responds to.  The
SURVEY 8d algorithmic byte rate is reported next to it, not as a fraction of a roof it does not touch: the scene is cache resident.
`sustained` (N = 1): after the timed region the GPU renders for about 11 s in 32-frame launches -- the driver's 20 steps are 15 ms of GPU time,
too short for any outside observer to see, and a utilisation sampler with a 5 s period needs two periods to be sure to land inside -- and the last of
those frames, the one verified against the oracle, must come out bit-identical; its rate is reported next to `value`, never instead of it.
`reference_call_shape`: one render() per frame as src/main.js:70-74 calls it -- pipelined (64 and 128 frames without a host wait: the steady rate is
their difference, `drain_ms` what the 64-frame run took beyond it) and awaited (wall time per frame next to the GPU time hipEvents measured around the
frame's kernels, so host latency and kernel time can be told apart).
`configs` adds the whole-frame rate of C4 (sponza-class interior: no empty space, every camera ray hits) and rays / node tests per
second for both, so the figure that does not lean on empty space travels with the line.
"""
import argparse
import hashlib
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")   # before the HIP runtime initialises (overlapped frame slots)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WIDTH, HEIGHT, SPP, BOUNCES, SEED = 1920, 1080, 4, 8, 1
NUM_TRIS, SCENE_SEED = 871414, 20260109
BYTES_NODE, BYTES_TRI, BYTES_SAMPLE = 32, 36, 16
# MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, a wave64 VALU instruction occupies its SIMD for 2 cycles, 2.4 GHz max clock;
# L2 aggregate ~34.5 TB/s (128-byte lines); HBM3E 8 TB/s
SIMDS, VALU_CYCLES_PER_WAVE_INST, CLOCK_HZ = 1024, 2.0, 2.4e9
VALU_PEAK_GINST = SIMDS * CLOCK_HZ / VALU_CYCLES_PER_WAVE_INST / 1e9
L2_PEAK_GBS, L2_LINE, HBM_PEAK_GBS = 34500.0, 128, 8000.0
# the vector L1's request rate for this access pattern (every lane gathers its own 64-byte record with 4 x dwordx4): 217.7 G records/s x 4
# requests, measured chip-wide by tools/probes/gather64.hip (profiles/r03_gather64_probe.txt, V0) -- 1.42 requests per cycle and CU
L1_GATHER_PEAK_GREQ = 217.7 * 4
# the scalar unit: ONE per CU (guide glossary); its issue rate is not in the guide -- tools/probes/salu_rate.hip measures 586 G scalar instructions / s chip-wide with
# 8 ... 32 resident wavefronts per CU (profiles/r06_salu_rate_probe.txt) = 0.953 of one instruction per CU and cycle at 2.4 GHz (the probe's own loop overhead is the rest);
# a single wavefront issues one every ~4.5 cycles.  Peak used: 1 per CU and cycle.
CUS = 256
SALU_PEAK_GINST = CUS * CLOCK_HZ / 1e9
# static vector instructions of one traversal step of the dense instance, by the lanes that execute them (compile-only census of this kernel build,
# pinned by tests/test_kernel_resources.py::test_step_census_matches_bench_constants): node side = four slab tests + child order + the three stack stores,
# leaf side = Moller-Trumbore, pop = the stack pop (its loop body again per skipped entry), common = fetch address + exit test
STEP_VALU = {"node_side": 110, "leaf_side": 59, "pop": 14, "pop_per_entry": 5, "common": 6}
PMC_FILE = os.path.join(ROOT, "profiles", "r06_pmc_bench.json")
KERNEL_SOURCES = ["pt_megakernel.hip", "pt_megakernel_loop.inc", "pt_device.h", "pt_kernels.h"]
VERIFY_STEP = 4         # the timed frame is checked on every 4th pixel in x and y (1/16 of the frame, ~0.6 s of oracle time)
SHARD_PIECES = 3        # launches a sharded run is cut into at least (tools/ab/pieces_sweep.sh -> profiles/r05_m3_shard_pieces.txt: [7, 7, 6] beats [5, 5, 5, 5] by 2.5 .. 5.5 % at N = 2 / 4 / 8 once the last gather is added)


def _normalised_source(path):
    """Kernel source without // comments, trailing blanks and empty lines: a comment edit must not make the counter file look stale."""
    out = []
    for line in open(path, "r", encoding="utf-8", errors="replace"):
        code = line.split("//", 1)[0].rstrip()
        if code:
            out.append(code)
    return "\n".join(out).encode()


def source_tag():
    """Identifies the kernel build the PMC summary was taken from."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(_normalised_source(os.path.join(ROOT, "raytracer-public_amd", "csrc", f)))
    return h.hexdigest()[:16]


def algorithmic_bytes(st):
    return BYTES_NODE * st["nodes_examined"] + BYTES_TRI * st["tris_tested"] + BYTES_SAMPLE * st["samples"]


def load_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc as orc_mod
    return orc_mod, orc_mod.load()


def cpu_baseline(tris, bvh4):
    """The CPU oracle (a port of the same loop) on a bounded sample of the same workload: the whole frame, single thread."""
    orc_mod, orc = load_oracle()
    p = orc.make_params(WIDTH, HEIGHT, NUM_TRIS, mode=orc_mod.MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, step=(1, 1))
    t0 = time.time()
    _, _, st = orc.render(p, tris, bvh4)
    dt = time.time() - t0
    return {"value": round(st["samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "every pixel of the same 1920x1080/4spp/8-bounce frame (%d samples, %.1f s), oracle/pt_oracle.cpp single thread, %d host cores present"
                      % (st["samples"], dt, os.cpu_count() or 0)}


def cpu_baseline_threads(tris, bvh4):
    """The same oracle over row bands on several host threads, for context: the whole frame again.
    Thread count = the CPU share of a one-GPU box (16) or fewer."""
    orc_mod, orc = load_oracle()
    threads = max(1, min(16, os.cpu_count() or 1))
    p = orc.make_params(WIDTH, HEIGHT, NUM_TRIS, mode=orc_mod.MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED)
    t0 = time.time()
    _, st = orc.render_mt(p, tris, bvh4, threads=threads)
    dt = time.time() - t0
    return {"value": round(st["samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "every pixel of the same frame in 8-row bands over %d host threads (%d samples, %.1f s), oracle/pt_oracle.cpp" % (threads, st["samples"], dt)}


def cpu_baseline_node(tris, bvh4):
    """The same loop as a single-thread Node/JS program (oracle/js/pt_oracle.js, bit-identical to the C++
    oracle): every 2nd pixel in x and y of the same frame."""
    import shutil, tempfile
    import numpy as np
    node = shutil.which("node")
    if node is None:
        return None
    orc_mod, _ = load_oracle()
    focal, aspect = orc_mod.focal_aspect(WIDTH, HEIGHT)
    with tempfile.TemporaryDirectory() as d:
        np.ascontiguousarray(tris, np.float32).tofile(os.path.join(d, "t")); np.ascontiguousarray(bvh4, np.uint32).tofile(os.path.join(d, "b"))
        P = dict(width=WIDTH, height=HEIGHT, focal=float(focal), aspect=float(aspect), camPos=[0, 0, 2.5], camQuat=[0, 0, 0, 1], frame=0, mode=2,
                 spp=SPP, maxBounces=BOUNCES, seed=SEED, numTris=NUM_TRIS, stepX=2, stepY=2)
        json.dump(P, open(os.path.join(d, "p"), "w"))
        info = json.loads(subprocess.check_output([node, os.path.join(ROOT, "oracle", "js", "pt_oracle.js"), os.path.join(d, "t"), os.path.join(d, "b"),
                                                   os.path.join(d, "p"), os.path.join(d, "o")], text=True))
    return {"value": round(info["stats"]["samples"] / info["seconds"] / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "every 2nd pixel in x and y of the same frame (%d samples, %.1f s), oracle/js/pt_oracle.js on Node %s, single thread"
                      % (info["stats"]["samples"], info["seconds"], info["node"])}


def schedule(n_steps, batch, world, fixed_batch):
    """[(first step, frames)] of the launches that cover n_steps.

    One GPU: full batches (a run of up to 32 frames of work is ONE launch: its sparse tail is paid once).
    Sharded (world > 1): the gather of launch b travels while launch b+1 traces, and the LAST launch's gather and its drain are
    exposed, so a run is cut into SHARD_PIECES launches of about equal size at least, never more than `batch` frames each, the last
    ones no larger than the ones before -- chosen from the one-GPU replay of every rank's submission sequence
    (tools/shard_schedule_sim.py, tools/ab/pieces_sweep.sh -> profiles/r05_m3_shard_pieces.txt).  PT_BENCH_SCHEDULE="10,10" overrides (replays only)."""
    forced = os.environ.get("PT_BENCH_SCHEDULE")
    if forced and world > 1 and not fixed_batch:
        sizes = [int(x) for x in forced.split(",") if x.strip()]
        if sum(sizes) == n_steps and all(0 < x <= batch for x in sizes):
            out, done = [], 0
            for b in sizes:
                out.append((done, b)); done += b
            return out
    out, done = [], 0
    if world > 1 and not fixed_batch:
        pieces = max(SHARD_PIECES, -(-n_steps // batch))
        pieces = max(1, min(pieces, n_steps))
        base, extra = divmod(n_steps, pieces)
        for i in range(pieces):                      # the larger pieces first
            b = base + (1 if i < extra else 0)
            out.append((done, b)); done += b
        return out
    while done < n_steps:
        b = min(batch, n_steps - done)
        out.append((done, b)); done += b
    return out


def submit_launches(ctx, p, launches, first_frame, batch, set_target, after_launch):
    """The submission sequence of a run: for every launch its frames (pt_render with the target that is current), a pt_flush for a
    partial batch, then `after_launch(k, nf)`.  bench.py and tools/shard_schedule_sim.py both submit through here."""
    for k, (first, nf) in enumerate(launches):
        for j in range(nf):
            p.frame = first_frame + first + j
            set_target(k, j, p.frame)
            ctx.render(p)                # the library launches a full batch with its last frame ...
        if nf < batch:
            ctx.flush()                  # ... and a shorter one here (the slots stay sized for `batch` frames)
        after_launch(k, nf)


def busy_ms(starts, durs):
    """Length of the union of the [start, start + dur] intervals."""
    iv = sorted((float(s), float(s) + float(d)) for s, d in zip(starts, durs))
    total, cur_a, cur_b = 0.0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                total += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        total += cur_b - cur_a
    return total


COLLECTIVE_TIMEOUT_S = float(os.environ.get("PT_BENCH_TIMEOUT", "120"))     # every collective (and the rendezvous) gives up after this long: well under the driver's 600 s limit


def init_distributed(dist, backend, rank, world, device=None, init_method=None):
    """The process group of a sharded run, with a timeout on every collective: a rank that never arrives (died, hung) makes the others raise after
    COLLECTIVE_TIMEOUT_S instead of sitting in dist.gather until somebody kills the job (torch's default is 10 minutes -- the driver's whole limit)."""
    from datetime import timedelta
    kw = dict(backend=backend, rank=rank, world_size=world, timeout=timedelta(seconds=COLLECTIVE_TIMEOUT_S))
    if init_method:
        kw["init_method"] = init_method
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(**kw)


def abort_rank(dist, rank, exc, code=4):
    """A rank that raised after the process group came up: say what happened, take the group down best-effort (a watchdog ends the process if
    the teardown itself blocks on a collective the dead peer will never join), and EXIT non-zero -- a plain exit, never an exec.  Under
    torch.distributed.run the agent then stops the other ranks at once; started any other way they run into the collective timeout above."""
    import threading, traceback
    print("bench.py: rank %d failed: %s" % (rank, "".join(traceback.format_exception(type(exc), exc, exc.__traceback__)).rstrip()), file=sys.stderr, flush=True)
    threading.Timer(10.0, lambda: os._exit(code)).start()
    try:
        if dist is not None and dist.is_initialized():
            dist.destroy_process_group()
    except BaseException:
        pass
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(code)


class GatherPipeline:
    """The collective side of a sharded run: launch b's packed shares are gathered to rank 0 while launch b+1 traces; `finish` of launch b-1 runs
    after launch b has been submitted.  One instance per run; bench.py and tests/test_bench_multirank_gloo.py (8 gloo ranks, CPU tensors) both drive it.

    packed[slot]   [batch, floats] this rank's shares of the launch in flight in `slot` (two slots, used alternately)
    gathered[slot] rank 0: [world, batch, floats]; None elsewhere
    on_finish(slot, nf): rank 0, after the gather of a launch has landed (bench.py: pt_unpack_batch into the frames)
    host_stage: gather through host memory, synchronously (gloo rehearsals); before_host_copy() makes the producer's work visible first"""

    def __init__(self, dist, torch, rank, world, packed, gathered, on_finish, host_stage=False, before_host_copy=None):
        self.dist, self.torch, self.rank, self.world = dist, torch, rank, world
        self.packed, self.gathered, self.on_finish = packed, gathered, on_finish
        self.host_stage, self.before_host_copy = host_stage, before_host_copy
        self.pending = None
        self.gathers = 0

    def _host_gather(self, buf):
        if self.before_host_copy:
            self.before_host_copy()
        src = buf.cpu()
        lst = [self.torch.empty_like(src) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(src, lst, dst=0)
        return lst

    def ship(self, slot, nf):
        """Launch in packed[slot][:nf] has been submitted (and packed): start its gather, then finish the previous launch's."""
        if self.host_stage:
            work = ("host", self._host_gather(self.packed[slot][:nf]))
        else:
            glist = [self.gathered[slot][r][:nf] for r in range(self.world)] if self.rank == 0 else None
            work = ("async", self.dist.gather(self.packed[slot][:nf], glist, dst=0, async_op=True))
        self.gathers += 1
        self.finish()                      # gather(b-1) has had the whole launch b to complete
        self.pending = (work, slot, nf)

    def finish(self):
        if self.pending is None:
            return
        (kind, work), slot, nf = self.pending
        self.pending = None
        if kind == "async":
            work.wait()
        elif self.rank == 0:
            self.gathered[slot][:, :nf].copy_(self.torch.stack(work))
        if self.rank == 0:
            self.on_finish(slot, nf)


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as children of a process that has not
    touched the GPU (no exec of an initialised process) and hand their exit status on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--reps", type=int, default=9, help="repetitions of the timed region (value = the median one)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C4 leg of the `configs` block")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--verify", action="store_true", help="kept for compatibility: the check against the oracle always runs")
    ap.add_argument("--no-reference-shape", action="store_true", help="skip the one-render()-per-frame figures")
    ap.add_argument("--sustained-seconds", type=float, default=11.0, help="N = 1: after the timed region, render for about this long in 32-frame launches and compare the "
                                                                          "last frame with the verified one, bit for bit (0 = skip)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world            # launched under torch.distributed.run: the launcher's rank count wins

    import torch                     # first: its HIP runtime is the one libmi355pt binds to
    import torch.distributed as dist
    import numpy as np
    rt = importlib.import_module("raytracer-public_amd")

    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    device = local_rank % max(n_dev, 1)
    torch.cuda.set_device(device)
    if world > 1:
        init_distributed(dist, backend, rank, world, device=torch.device("cuda", device))
    try:
        run_bench(args, rank, world, device, backend, torch, dist, np, rt)
    except SystemExit:
        raise
    except BaseException as exc:             # a rank that fails after the rendezvous must not leave the others waiting in a collective
        if world > 1:
            abort_rank(dist, rank, exc)
        raise


def run_bench(args, rank, world, device, backend, torch, dist, np, rt):

    width, height = args.width, args.height
    tris = rt.procedural_scene(rt.SCENE_DRAGON_CLASS, NUM_TRIS, SCENE_SEED)
    ctx = rt.Context(device)
    # the context's own non-blocking stream, made visible to torch so that the RCCL gather (which orders
    # itself against torch's current stream) follows the render / resolve kernels
    stream = torch.cuda.ExternalStream(ctx.get_stream(), device=device)
    ctx.set_triangles(tris)
    ctx.build_bvh()                  # Morton+sort, LBVH2 kernels, collapse: data/BVH2.bin equivalent

    # Every step is a NEW frame of the same camera (frame index i: a fresh sample set, as in a progressive render), so the
    # frames that share a batched launch do not trace identical rays.
    def params(frame=0, stats=False, whole=False):
        return ctx.make_params(width, height, mode=rt.PT_MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, frame=frame,
                               tile_rank=0 if whole else rank, tile_count=1 if whole else world, stats=stats)

    sharded = world > 1
    # Frames are submitted in batches: the library traces `batch` consecutive frames with one persistent
    # launch (pt_set_batch), which amortises the sparse tail of a frame -- essential for the small per-GPU
    # shares of a sharded run.  The RCCL gather then moves one batch at a time (fewer, larger collectives).
    fixed_batch = bool(os.environ.get("PT_BENCH_BATCH"))
    # A run of up to 32 frames of work is ONE launch (its sparse tail is paid once); longer runs use launches of 32 frames of work,
    # which overlap on the context's side streams.
    batch = int(os.environ.get("PT_BENCH_BATCH", "0")) or args.steps
    batch = max(1, min(32 * world, 256, batch))
    if world > 1 and not fixed_batch:       # a sharded run is cut into pieces (schedule()): buffers and pt_set_batch are sized for the largest launch
        batch = max([nf for _, nf in schedule(args.steps, batch, world, False)] + [nf for _, nf in schedule(max(args.warmup, 1), batch, world, False)])
    launch_log = []                  # (tag, frames) of every un-instrumented megakernel launch, in submission order

    ctx.set_batch(1)
    host_stage = False
    if sharded:
        stride = max(rt.tile_layout(width, height, r, world)[1] for r in range(world))
        compact = [torch.zeros(batch, stride, dtype=torch.float32, device="cuda") for _ in range(2)]
        # What travels are PACKED shares (mi355pt.h, pt_pack_shares): the rank's tiles inside the rectangle of tiles in which a camera ray can
        # reach the scene at all, 12 bytes per pixel; rank 0 fills the rest of the frame with the camera-miss value.  Same camera every step,
        # so one rectangle for the run -- a quarter of the compact buffers' bytes for this frame.
        rect = ctx.traced_tile_rect(ctx.make_params(width, height, mode=rt.PT_MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED))
        pstride = rt.packed_layout(width, height, world, rect)[1]
        packed = [torch.zeros(batch, max(pstride, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.zeros(world, batch, max(pstride, 4), dtype=torch.float32, device="cuda") for _ in range(2)] if rank == 0 else [None, None]
        frames_full = torch.zeros(batch, height * width * 4, dtype=torch.float32, device="cuda") if rank == 0 else None     # the de-interleaved frames of one launch
        host_stage = backend != "nccl"   # rehearsal path (gloo): stage through host memory
        torch.cuda.synchronize()         # the zero fills ran on torch's default stream; everything below uses the context's
        if not host_stage:               # bring the RCCL communicator up before anything is timed (even with --warmup 0)
            probe = torch.zeros(8, device="cuda")
            dist.gather(probe, [torch.zeros(8, device="cuda") for _ in range(world)] if rank == 0 else None, dst=0)
            torch.cuda.synchronize()
    else:
        # whole frames: every frame of a launch is delivered into its own device buffer (two launches' worth, used alternately),
        # so no frame's result is overwritten by a later frame of the same launch
        frames_out = torch.zeros(2 * batch, height * width * 4, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()

    # exact traversal counters of this rank's share of every timed frame (deterministic per frame index): one instrumented
    # launch per frame, before the timed region
    my_stats = None
    frame_bytes = []
    import ctypes as C
    dbg = np.zeros(24, np.uint64)
    rays_entering = 0                # rays that enter the root box at all (camera rays that miss it are most of C2's "rays")
    rcp_short_variant = True
    with torch.cuda.stream(stream):
        if sharded:
            ctx.set_compact_buffer(compact[0][0].data_ptr(), stride)
        for i in range(args.steps):
            ctx.render(params(i, stats=True))
            st = ctx.stats()
            rt.lib.pt_debug_counters(ctx.h, dbg.ctypes.data_as(C.c_void_p)); rays_entering += (int(dbg[7]) & ((1 << 63) - 1)) >> 1
            rcp_short_variant = rcp_short_variant and bool(int(dbg[7]) & 1)      # which variant the host picks for this scene / camera (the same decision for the timed launches)
            frame_bytes.append(algorithmic_bytes(st))
            my_stats = st if my_stats is None else {k: my_stats[k] + st[k] for k in st}
    my_bytes = float(sum(frame_bytes)) / max(len(frame_bytes), 1)       # mean per frame
    ctx.set_batch(batch)

    last_frame = {}                  # where the last timed frame ended up (whole-frame runs)
    fail_at = tuple(int(x) for x in os.environ["PT_BENCH_FAIL"].split(":")) if os.environ.get("PT_BENCH_FAIL") else None
    pipe = None
    if sharded:
        def unpack(slot, nf):            # rank r's packed share of frame j sits at gathered[slot][r][j]: all nf frames are rebuilt by ONE launch, each into its own row-major frame
            ctx.unpack_batch(gathered[slot].data_ptr(), batch * max(pstride, 4), max(pstride, 4), nf, width, height, world, rect, SPP, frames_full.data_ptr(), height * width * 4)
            last_frame["buf"] = nf - 1
        pipe = GatherPipeline(dist, torch, rank, world, packed, gathered, unpack, host_stage=host_stage, before_host_copy=stream.synchronize)

    def ship(slot, nf):              # the launch in compact[slot][:nf] has been submitted: pack it behind its resolve, gather it, finish the previous one
        if pstride:
            ctx.pack_shares(compact[slot].data_ptr(), stride, nf, width, height, rank, world, rect, packed[slot].data_ptr(), max(pstride, 4))
        pipe.ship(slot, nf)

    def run(n_steps, first_frame, p, tag):    # called with `stream` current: n_steps frames as the launches of schedule()
        def set_target(k, j, frame):
            if sharded:
                ctx.set_compact_buffer(compact[k & 1][j].data_ptr(), stride)
            else:
                ctx.set_output_buffer(frames_out[(k & 1) * batch + j].data_ptr(), height * width * 4)
                last_frame["index"], last_frame["buf"] = frame, (k & 1) * batch + j

        def after_launch(k, nf):
            launch_log.append((tag, nf))
            if fail_at and tag == "timed" and (rank, k) == fail_at:      # PT_BENCH_FAIL="rank:launch" (tests): this rank dies in the middle of the timed region
                raise RuntimeError("injected failure (PT_BENCH_FAIL) on rank %d at launch %d" % (rank, k))
            if sharded:
                ship(k & 1, nf)

        submit_launches(ctx, p, schedule(n_steps, batch, world, fixed_batch), first_frame, batch, set_target, after_launch)
        if sharded:
            pipe.finish()

    p = params()
    with torch.cuda.stream(stream):
        run(args.warmup, args.steps, p, "warmup")
    # ---- the timed region, `reps` times: exactly K steps between barriers + synchronisations, max over ranks; the median repetition is the result
    reps = max(1, args.reps)
    rep_elapsed, rep_spans = [], []
    for rep in range(reps):
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.timing_begin(args.steps)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            run(args.steps, 0, p, "timed" if rep == 0 else "timed-rep%d" % rep)      # the same K frame indices every repetition: the same work
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        e = time.perf_counter() - t0
        rep_spans.append(ctx.timing_collect_spans(args.steps))
        if sharded:
            t = torch.tensor([e], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        rep_elapsed.append(e)
    order = sorted(range(reps), key=lambda i: rep_elapsed[i])
    med = order[(reps - 1) // 2]                     # the median repetition (the lower one of an even count)
    elapsed = rep_elapsed[med]
    k_start, k_ms = rep_spans[med]

    # ---- the last timed frame against the CPU oracle (fixed pixel grid, every sample, bit for bit) -----------------------------
    verified = None
    bvh4 = None
    if rank == 0:
        orc_mod, orc = load_oracle()
        bvh4 = ctx.read_bvh4()
        with torch.cuda.stream(stream):
            if sharded:
                ctx.synchronize()
                got = frames_full[last_frame["buf"]].reshape(height, width, 4).cpu().numpy()     # the last de-interleaved frame
            else:
                ctx.synchronize()
                got = frames_out[last_frame["buf"]].reshape(height, width, 4).cpu().numpy()
        want, _, _ = orc.render(orc.make_params(width, height, NUM_TRIS, mode=orc_mod.MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED,
                                                frame=args.steps - 1, step=(VERIFY_STEP, VERIFY_STEP)), tris, bvh4)
        a, b = got[::VERIFY_STEP, ::VERIFY_STEP], want[::VERIFY_STEP, ::VERIFY_STEP]
        verified = bool(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32)))
        if not verified:
            print("bench.py: timed frame %d differs from the CPU oracle on the %d-pixel grid (%d of %d pixels)"
                  % (args.steps - 1, VERIFY_STEP, int((a != b).any(axis=2).sum()), a.shape[0] * a.shape[1]), file=sys.stderr, flush=True)
    if sharded:
        ok = torch.tensor([1 if (verified or rank != 0) else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.broadcast(ok, src=0)
        if int(ok.item()) == 0:
            ctx.close(); dist.destroy_process_group()
            sys.exit(3)
    elif not verified:
        ctx.close()
        sys.exit(3)

    # ---- where the lanes are: one more launch of the timed shape on the TIMELINE variant of the kernel (production registers and speed; wave-uniform
    #      counters in scalar registers): wavefront-steps, traversing lanes, lanes at a leaf -- the decomposition of the counter files' lane utilisation
    lanes_by_side = None
    if world == 1 and rank == 0:
        import ctypes as C2
        ctx.debug_set_tune("TIMELINE", 1)
        with torch.cuda.stream(stream):
            n_tl = min(batch, args.steps)
            for j in range(n_tl):
                p.frame = j
                ctx.set_output_buffer(frames_out[j].data_ptr(), height * width * 4)
                ctx.render(p)
            if n_tl < batch:
                ctx.flush()
            ctx.synchronize()
            tl_ms = ctx.last_render_ms()
        ctx.debug_set_tune("TIMELINE")
        wt = np.zeros((16384, 24), np.uint64); nw = C2.c_uint32()
        rt.lib.pt_debug_wave_times(ctx.h, wt.ctypes.data_as(C2.c_void_p), C2.c_uint32(16384), C2.byref(nw))
        wt = wt[: nw.value].astype(np.float64)
        if len(wt) and wt[:, 3].sum() > 0 and int(wt[0, 20]) == 2:
            steps_w, lanes, leaf = wt[:, 3].sum(), wt[:, 7].sum(), wt[:, 9].sum()
            node = lanes - leaf
            V = STEP_VALU
            # the pop runs for the lanes that left a leaf or found no child (every leaf lane, plus the node lanes without a hit): bounded below by the leaf lanes
            model = (V["node_side"] * node + V["leaf_side"] * leaf + V["pop"] * leaf + V["common"] * lanes) / (64.0 * steps_w * (V["node_side"] + V["leaf_side"] + V["pop"] + V["common"]))
            lanes_by_side = {
                "frames": n_tl, "wavefront_steps_per_frame": round(steps_w / n_tl, 1), "lane_steps_per_frame": round(lanes / n_tl, 1),
                "lanes_per_step": {"traversing": round(lanes / steps_w, 2), "node_side": round(node / steps_w, 2), "leaf_side": round(leaf / steps_w, 2)},
                "share_of_64_lanes": {"traversing": round(lanes / steps_w / 64.0, 4), "node_side": round(node / steps_w / 64.0, 4), "leaf_side": round(leaf / steps_w / 64.0, 4)},
                "shade_passes_per_frame": round(wt[:, 4].sum() / n_tl, 1), "refill_passes_per_frame": round(wt[:, 5].sum() / n_tl, 1),
                "static_valu_per_step": V,
                "modelled_step_lane_utilisation": round(model, 4),
                "launch_ms": round(tl_ms, 4),
                "how": "one launch of the timed shape on trace_paths_kernel<2, *> (TIMELINE: the production kernel + wave-uniform counters in scalar registers, no scratch, "
                       "same registers and occupancy; knob TIMELINE); node side = traversing lanes not at a leaf; model = sum(static vector instructions of a side x lanes "
                       "executing it) / (64 x all of them): the step's share of the kernel-wide lane_utilisation, which also holds the shade and refill passes",
            }

    # ---- the reference's own call shape: one render() per frame, no batching (src/main.js:70-74) ------------------------------
    ref_shape = None
    if world == 1 and not args.no_reference_shape:
        ctx.set_output_buffer(0, 0)
        ctx.set_batch(1)
        n_ref, n_solo = 64, max(8, min(args.steps, 32))
        pr = params(whole=True)
        with torch.cuda.stream(stream):
            for i in range(8):
                pr.frame = 5000 + i; ctx.render(pr); launch_log.append(("reference-shape warmup", 1))
            ctx.synchronize()
            walls = []
            for n in (n_ref, 2 * n_ref):         # frames follow each other like requestAnimationFrame callbacks that do not wait; the run ends with a wait
                t1 = time.perf_counter()
                for i in range(n):
                    pr.frame = 6000 + i; ctx.render(pr); launch_log.append(("reference-shape pipelined", 1))
                ctx.synchronize()
                walls.append(time.perf_counter() - t1)
            steady = (walls[1] - walls[0]) / n_ref                  # what a frame costs while frames keep coming
            drain = walls[0] - n_ref * steady                       # what the last frames' drain adds to a run that ends
            t1 = time.perf_counter(); gpu_ms = []
            for i in range(n_solo):              # `await render()` + read-back before the next frame
                pr.frame = 7000 + i; ctx.render(pr); ctx.synchronize(); launch_log.append(("reference-shape solo", 1))
                gpu_ms.append(ctx.last_render_ms())                 # hipEvents on the context's stream around this frame's kernels (trace on its side stream + resolve)
            solo = (time.perf_counter() - t1) / n_solo
        ref_shape = {"frames_pipelined": [n_ref, 2 * n_ref], "ms_per_frame_pipelined": round(steady * 1e3, 4), "msamples_pipelined": round(width * height * SPP / steady / 1e6, 1),
                     "ms_per_frame_pipelined_64_with_drain": round(walls[0] / n_ref * 1e3, 4), "drain_ms": round(drain * 1e3, 4),
                     "frames_solo": n_solo, "ms_per_frame_solo": round(solo * 1e3, 4), "msamples_solo": round(width * height * SPP / solo / 1e6, 1),
                     "gpu_ms_per_frame_solo": round(float(np.median(gpu_ms)), 4), "host_ms_per_frame_solo": round(solo * 1e3 - float(np.median(gpu_ms)), 4),
                     "note": "pt_set_batch(1): one pt_render per frame as PathTracer.render() is called (src/main.js:70-74); pipelined = no host wait between frames (steady rate = "
                             "(128-frame run - 64-frame run) / 64, drain_ms = the 64-frame run beyond 64 x that), solo = pt_synchronize after every frame (gpu = hipEvents around the "
                             "frame's kernels, host = the rest of the wall time); `value` above uses pt_set_batch, an extension the reference API does not have"}

    # ---- sustained: a second or two of back-to-back 32-frame launches (the driver's 20 steps are 15 ms of GPU time: too short for any outside
    #      observer -- a utilisation sampler, a power reading -- to see), ending on the frame that was verified above: it must come out bit-identical
    sustained = None
    if world == 1 and args.sustained_seconds > 0 and rank == 0:
        ctx.set_output_buffer(0, 0)
        ctx.set_batch(32)
        ps = params(whole=True)
        with torch.cuda.stream(stream):
            for j in range(32):                  # one untimed launch: the frame slots are re-sized for 32-frame launches here
                ps.frame = j % args.steps; ctx.render(ps)
            launch_log.append(("sustained warmup", 32)); ctx.synchronize()
        n_sus, t1 = 0, time.perf_counter()
        with torch.cuda.stream(stream):
            while True:
                for j in range(32):
                    ps.frame = args.steps - 1 if j == 31 else (n_sus + j) % args.steps      # the timed frames over and over; every launch ends on the verified one
                    ctx.render(ps)
                launch_log.append(("sustained", 32)); n_sus += 32
                if n_sus % 256 == 0:
                    ctx.synchronize()
                    if time.perf_counter() - t1 >= args.sustained_seconds: break
            ctx.synchronize()
            dt = time.perf_counter() - t1
            again = ctx.read_radiance().reshape(height, width, 4)
        same = bool(np.array_equal(np.ascontiguousarray(again[::VERIFY_STEP, ::VERIFY_STEP]).view(np.uint32), np.ascontiguousarray(got[::VERIFY_STEP, ::VERIFY_STEP]).view(np.uint32)))
        sustained = {"frames": n_sus, "seconds": round(dt, 3), "ms_per_frame": round(dt / n_sus * 1e3, 4), "msamples": round(width * height * SPP * n_sus / dt / 1e6, 1),
                     "last_frame_identical_to_verified": same}
        ctx.set_batch(1)
        if not same:
            print("bench.py: the verified frame came out differently at the end of the sustained run", file=sys.stderr, flush=True)
            ctx.close(); sys.exit(3)

    # ---- the other single-GPU configuration: C4, a sponza-class interior (every camera ray hits, long paths: no empty space to lean on) --------
    configs = None
    if world == 1 and not args.no_configs and (width, height) == (WIDTH, HEIGHT):
        def rates(st, seconds_per_frame, entering):
            return {"rays_per_s": round((st["rays_closest"] + st["rays_shadow"]) / seconds_per_frame, 1), "rays_entering_root_per_s": round(entering / seconds_per_frame, 1), "node_tests_per_s": round(st["nodes_examined"] / seconds_per_frame, 1),
                    "tri_tests_per_s": round(st["tris_tested"] / seconds_per_frame, 1)}
        c2_frame = {k: my_stats[k] / args.steps for k in my_stats}
        configs = {"C2": {"ms_per_frame": round(elapsed / args.steps * 1e3, 4), "msamples": round(width * height * SPP * args.steps / elapsed / 1e6, 1),
                          "rays_per_frame": int(c2_frame["rays_closest"] + c2_frame["rays_shadow"]), "rays_entering_root_per_frame": int(rays_entering / args.steps),
                          **rates(c2_frame, elapsed / args.steps, rays_entering / args.steps)}}
        c4 = rt.Context(device)
        try:
            c4_tris = rt.procedural_scene(rt.SCENE_SPONZA_CLASS, 262144, SCENE_SEED)
            c4.set_triangles(c4_tris); c4.build_bvh()
            cam4, quat4 = (0.55, -0.05, 0.05), (0.0, 0.6630, 0.0, 0.7486)
            q = c4.make_params(width, height, cam4, quat4, mode=rt.PT_MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, stats=True)
            c4.render(q); st4 = c4.stats()
            rt.lib.pt_debug_counters(c4.h, dbg.ctypes.data_as(C.c_void_p)); enter4 = (int(dbg[7]) & ((1 << 63) - 1)) >> 1
            q = c4.make_params(width, height, cam4, quat4, mode=rt.PT_MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED)
            c4.set_batch(8)
            for i in range(8):
                q.frame = 100 + i; c4.render(q)
            c4.synchronize()
            times = []
            for rep in range(3):
                t1 = time.perf_counter()
                for i in range(8):
                    q.frame = 200 + 8 * rep + i; c4.render(q)
                c4.synchronize()
                times.append((time.perf_counter() - t1) / 8)
            t4 = sorted(times)[1]
            configs["C4"] = {"workload": "sponza-class procedural interior, 262,144 triangles, camera inside, %dx%d, %d spp, %d bounces, 8 frames per launch, median of 3 launches" % (width, height, SPP, BOUNCES),
                             "ms_per_frame": round(t4 * 1e3, 3), "msamples": round(width * height * SPP / t4 / 1e6, 1), "ms_per_frame_min_max": [round(min(times) * 1e3, 3), round(max(times) * 1e3, 3)],
                             "rays_per_frame": int(st4["rays_closest"] + st4["rays_shadow"]), "rays_entering_root_per_frame": enter4, **rates(st4, t4, enter4),
                             "algorithmic_GBps": round(algorithmic_bytes(st4) / t4 / 1e9, 1)}
        finally:
            c4.close()

    if rank == 0:
        log_path = os.environ.get("PT_BENCH_LAUNCH_LOG")
        if log_path:
            json.dump({"launches": launch_log, "steps": args.steps, "warmup": args.warmup, "batch": batch, "source_tag": source_tag()}, open(log_path, "w"))
        samples_per_step = width * height * SPP
        value = samples_per_step * args.steps / elapsed / 1e6
        n_launch = max(len(k_ms), 1)
        frames_per_launch = args.steps / n_launch
        busy = busy_ms(k_start, k_ms) if len(k_ms) else float("nan")            # ms the GPU spent in trace_paths_kernel (overlaps counted once)
        busy_per_frame = busy / args.steps
        algorithmic_gbs = my_bytes / (busy_per_frame * 1e-3) / 1e9
        # per-frame counter totals of the timed launches of this very command, from the committed rocprofv3 --pmc passes
        pmc, pmc_info = None, {"file": os.path.relpath(PMC_FILE, ROOT), "present": os.path.exists(PMC_FILE)}
        if os.path.exists(PMC_FILE) and world == 1 and (width, height) == (WIDTH, HEIGHT):
            try:
                pmc = json.load(open(PMC_FILE))
                pmc_info.update({"source_tag": pmc.get("source_tag"), "stale": pmc.get("source_tag") != source_tag(),
                                 "command": pmc.get("command"), "frames_per_launch": pmc.get("frames_per_launch")})
            except Exception as e:       # a broken summary must not take the throughput line with it
                pmc, pmc_info["error"] = None, str(e)
        fractions, fractions_builder, traffic, lane_util = {}, {}, None, None
        # The counters are only used when they describe THIS kernel build and THIS command: a stale summary (kernel sources edited since,
        # another --steps / --warmup, another launch shape) yields "unmeasured" -- never a fraction of somebody else's counters.
        pmc_usable = bool(pmc) and not pmc_info.get("stale") and abs(float(pmc.get("frames_per_launch") or 0) - frames_per_launch) < 1e-9 and \
            ("--steps %d " % args.steps) in (str(pmc.get("command")) + " ") and ("--warmup %d " % args.warmup) in (str(pmc.get("command")) + " ")
        pmc_info["used"] = pmc_usable
        if pmc and not pmc_usable:
            pmc_info["why_unused"] = "source_tag differs (kernel edited since the passes: re-run tools/pmc_bench.sh)" if pmc_info.get("stale") else "the passes were taken over another command / launch shape"
        # Every resource the kernel can be bound by: (counter expression per frame, peak, unit, where the peak comes from).  `bound` / `frac` are taken among the
        # resources whose peak is a GUIDE figure (/opt/skills/guides/MI355X_MICROARCH.md); a peak this repository measured itself (tools/probes/) is reported,
        # labelled, and never the headline.
        GUIDE, PROBE = "guide", "self-measured"
        resources = {
            "valu_issue": ("valu", lambda c: c["SQ_INSTS_VALU"], VALU_PEAK_GINST, "Ginst/s", GUIDE,
                           "SIMD-32: 2 cycles per wave64 instruction, 1,024 SIMDs, 2.4 GHz; plain f32 FMA / mul / min-max chains sustain 1,068-1,081 G/s = 0.87-0.88 of it with this kernel's six "
                           "wavefronts per SIMD, 1,097 with eight (tools/probes/valu_rate.hip, profiles/r06_valu_rate_probe.txt)"),
            "l2_bandwidth": ("l2", lambda c: (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) * L2_LINE, L2_PEAK_GBS, "GB/s", GUIDE, "34.5 TB/s aggregate, 128-byte lines"),
            "hbm_fabric": ("hbm", lambda c: (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, HBM_PEAK_GBS, "GB/s", GUIDE, "8 TB/s (FETCH_SIZE includes Infinity-Cache hits)"),
            "salu_issue": ("salu", lambda c: c["SQ_INSTS_SALU"], SALU_PEAK_GINST, "Ginst/s", PROBE,
                           "one scalar unit per CU (guide glossary) x one instruction per cycle (tools/probes/salu_rate.hip: 586 G/s of 614.4 measured, profiles/r06_salu_rate_probe.txt), 2.4 GHz"),
            "l1_gather_requests": ("l1", lambda c: c["TCP_TOTAL_CACHE_ACCESSES_sum"], L1_GATHER_PEAK_GREQ, "Greq/s", PROBE,
                                   "tools/probes/gather64.hip, profiles/r03_gather64_probe.txt: 871 G requests/s of a pure 64-byte-record gather"),
        }
        res_rates = {}
        if pmc_usable:
            c = pmc["per_frame"]
            s = busy_per_frame * 1e-3

            def fracs(sec):
                out_f = {}
                for name, (_, expr, peak, _, _, _) in resources.items():
                    try:
                        out_f[name] = expr(c) / sec / 1e9 / peak
                    except KeyError:          # a counter the passes did not collect
                        pass
                return out_f

            fractions = fracs(s)
            res_rates = {name: resources[name][1](c) / s / 1e9 for name in fractions}
            # the same counters over the kernel time of the session they were taken in (tools/pmc_bench.sh stores it): a 10 % difference
            # between that box and this one shows up as a difference between the two sets instead of hiding in `frac`
            bms = pmc.get("builder_kernel_busy_ms_per_frame")
            if bms:
                fractions_builder = fracs(float(bms) * 1e-3)
                pmc_info["builder_kernel_busy_ms_per_frame"] = bms
            lane_util = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
            traffic = int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 * frames_per_launch)
            guide_fracs = {k: v for k, v in fractions.items() if resources[k][4] == GUIDE}
            bound = max(guide_fracs, key=guide_fracs.get)
            roof = (resources[bound][0], res_rates[bound], resources[bound][2], resources[bound][3])
        else:
            roof = ("unmeasured", None, None, None)
        variant = "trace_paths_kernel<0, %s>" % ("true" if rcp_short_variant else "false")
        roofline = {
            "bound": roof[0], "achieved": None if roof[1] is None else round(roof[1], 2), "peak": roof[2], "unit": roof[3],
            "frac": None if roof[1] is None else round(roof[1] / roof[2], 5), "traffic": traffic,
            "bound_among": "resources whose peak is a guide figure (valu_issue, l2_bandwidth, hbm_fabric); self-measured peaks are listed in `fractions`, never the headline",
            "kernel": "%s (persistent megakernel; INSTR 0 = no counters; BOUNDED %s = %s reciprocal / square-root forms, chosen by the host for this scene and these cameras: stats word 7 bit 0 of the instrumented launches)"
                      % (variant, "true" if rcp_short_variant else "false", "short" if rcp_short_variant else "general"),
            "kernel_busy_ms": round(busy, 4), "kernel_busy_ms_per_frame": round(busy_per_frame, 5), "launches": int(len(k_ms)),
            "kernel_avg_launch_ms": round(float(np.mean(k_ms)), 4) if len(k_ms) else None, "frames_per_launch": frames_per_launch,
            "fractions": {k: round(v, 5) for k, v in fractions.items()},
            "rates": {k: round(v, 2) for k, v in res_rates.items()},
            "peaks": {k: {"peak": resources[k][2], "unit": resources[k][3], "source": resources[k][4], "what": resources[k][5]} for k in resources},
            "fractions_over_builder_time": {k: round(v, 5) for k, v in fractions_builder.items()},
            "lane_utilisation": None if lane_util is None else round(lane_util, 4),
            "lane_utilisation_by_side": lanes_by_side,
            "responds_to": {"resource": "valu_issue", "evidence": "probe builds of the production step, same-session A/Bs: profiles/r04_p1_step_sensitivity_probe.txt (each added instruction costs 0.1-0.25 %), "
                            "r04_g1_fetch_pieces_ab.txt (20 % fewer L1 requests: 2.5-3.6 % SLOWER), r05_d1_dense_micro_ab.txt (fewer vector instructions per step: faster by as much)"},
            "pmc": pmc_info,
            "definition": "frac = (per-frame counter total of the timed launches, rocprofv3 --pmc of this command, profiles/) / (kernel busy time per frame, hipEvents of this run) / peak; "
                          "valu_issue: SQ_INSTS_VALU x 2 cycles over 1024 SIMDs x 2.4 GHz; l2_bandwidth: (TCC_HIT+TCC_MISS) x 128 B over 34.5 TB/s; hbm_fabric: (FETCH_SIZE+WRITE_SIZE) x 1024 over 8 TB/s; "
                          "salu_issue: SQ_INSTS_SALU over 256 CUs x 2.4 GHz; l1_gather_requests: TCP_TOTAL_CACHE_ACCESSES over 871 G/s; bound = the largest fraction among guide peaks; "
                          "fractions_over_builder_time = the same counters over the kernel time of the session the counters were taken in (another box: the two sets differ by the boxes' speed difference)",
            "algorithmic": {"GBps": round(algorithmic_gbs, 2), "bytes_per_frame": int(my_bytes), "bytes_per_frame_min_max": [int(min(frame_bytes)), int(max(frame_bytes))],
                            "over_hbm_peak": round(algorithmic_gbs / HBM_PEAK_GBS, 4),
                            "note": "SURVEY 8d: 32 B x node records examined + 36 B x triangles tested + 16 B x samples, over the kernel busy time; the working set (67 MB) is "
                                    "L2 / Infinity-Cache resident, so this rate is served on-die and is NOT a fraction of the HBM roof (it can exceed 1); north_star's '>= 40 % "
                                    "of HBM roofline' is not met by measured HBM-side traffic (hbm_fabric above) and cannot be for a cache-resident scene",
                            "counters_all_timed_frames": {k: my_stats[k] for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples")}},
        }
        out = {
            "metric": "Msamples/sec @1920x1080 Stanford-Dragon-class, 4 spp, 8 bounces",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "reps": reps, "ms_per_step_min_max": [round(min(rep_elapsed) / args.steps * 1e3, 4), round(max(rep_elapsed) / args.steps * 1e3, 4)],
            "ms_per_step_all": [round(e / args.steps * 1e3, 4) for e in rep_elapsed],
            # the statistic rounds 1-4 reported (median of the first five repetitions), kept so that earlier BENCH_rNN.json lines stay comparable like for like
            "ms_per_step_first5": round(sorted(rep_elapsed[:5])[(min(reps, 5) - 1) // 2] / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": verified,
            "verified_how": "timed frame %d, every %dth pixel in x and y (all %d samples each), bit for bit against oracle/pt_oracle.cpp" % (args.steps - 1, VERIFY_STEP, SPP),
            "config": {"workload": "C2: dragon-class procedural closed mesh (%d tris, seed %d; dragon.glb absent), native LBVH2->BVH4, %dx%d, %d spp, %d bounces, camera (0,0,2.5) identity quat FOV 70, a new frame index (sample set) every step"
                                   % (NUM_TRIS, SCENE_SEED, width, height, SPP, BOUNCES),
                       "triangles": NUM_TRIS, "bvh4_nodes": ctx.scene_info()["numNodes4"], "width": width, "height": height,
                       "spp": SPP, "max_bounces": BOUNCES, "seed": SEED, "frames_per_launch": batch,
                       "sharding": ("interleaved 8x8 tiles over %d GPUs, RCCL gather of packed shares (tiles inside the traced rectangle %s, 12 B per pixel: %d of %d floats per frame and rank) to rank 0"
                                    % (world, str(rect), pstride, stride)) if sharded else "single GPU, whole frame"},
            "roofline": roofline,
            "reference_call_shape": ref_shape,
            "sustained": sustained,
            "configs": configs,
        }
        if world == 1 and not args.no_cpu_baseline and (width, height) == (WIDTH, HEIGHT):
            out["cpu_baseline"] = cpu_baseline(tris, bvh4)
            out["cpu_baseline_threads"] = cpu_baseline_threads(tris, bvh4)
            out["cpu_baseline_node"] = cpu_baseline_node(tris, bvh4)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
