#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

One *step* = one complete frame of the C2 workload: dragon-class procedural mesh (871,414
triangles, the reference's dragon.glb is absent), native LBVH2 -> collapsed BVH4, 1920x1080,
4 samples per pixel, 8 bounces, camera (0,0,2.5) / identity quaternion / FOV 70 deg
(src/main.js:12, PathTracer.js:761).  value = Msamples/s = W*H*spp*K / wall time of the K
timed steps (max over ranks), scene already resident in HBM.

N > 1 (one process per GPU, launched by torch.distributed.run): the frame is sharded by
interleaved 8x8 pixel tiles (rank = (tx+ty) % N); each rank renders its tiles into a compact
buffer, RCCL gathers the buffers on rank 0 over xGMI (torch.distributed backend "nccl"), rank 0
de-interleaves them into the full frame.  The gather of frame i overlaps the render of frame
i+1 (double-buffered).  Total work per step is fixed -> "scaling": "strong".
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")   # before the HIP runtime initialises (overlapped frame slots)

WIDTH, HEIGHT, SPP, BOUNCES, SEED = 1920, 1080, 4, 8, 1
NUM_TRIS, SCENE_SEED = 871414, 20260109
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
BYTES_NODE, BYTES_TRI, BYTES_SAMPLE = 32, 36, 16


def algorithmic_bytes(st):
    return BYTES_NODE * st["nodes_examined"] + BYTES_TRI * st["tris_tested"] + BYTES_SAMPLE * st["samples"]


def cpu_baseline(tris, bvh4):
    """The CPU oracle (a port of the same loop) on a bounded sample of the same workload:
    the whole frame, single thread."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc as orc_mod
    orc = orc_mod.load()
    p = orc.make_params(WIDTH, HEIGHT, NUM_TRIS, mode=orc_mod.MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, step=(1, 1))
    t0 = time.time()
    _, _, st = orc.render(p, tris, bvh4)
    dt = time.time() - t0
    return {"value": round(st["samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": "every pixel of the same 1920x1080/4spp/8-bounce frame (%d samples, %.1f s), oracle/pt_oracle.cpp single thread, %d host cores present"
                      % (st["samples"], dt, os.cpu_count() or 0)}, st


def cpu_baseline_threads(tris, bvh4):
    """The same oracle over row bands on several host threads (ctypes releases the GIL; bands write disjoint rows), for
    context: the whole frame again.  Thread count = the CPU share of a one-GPU box (16) or fewer."""
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc as orc_mod
    orc = orc_mod.load()
    threads = max(1, min(16, os.cpu_count() or 1))
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1); bvh4 = np.ascontiguousarray(bvh4, np.uint32)
    img = np.zeros((HEIGHT, WIDTH, 4), np.float32)
    bands = [(y, min(y + 8, HEIGHT)) for y in range(0, HEIGHT, 8)]          # 8-row bands, handed out dynamically

    def work(band):
        p = orc.make_params(WIDTH, HEIGHT, NUM_TRIS, mode=orc_mod.MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, rect=(0, band[0], WIDTH, band[1]))
        st = orc_mod.Stats()
        rc = orc.lib.orc_render(C.byref(p), tris.ctypes.data_as(C.POINTER(C.c_float)), bvh4.ctypes.data_as(C.POINTER(C.c_uint32)),
                                img.ctypes.data_as(C.POINTER(C.c_float)), None, C.byref(st))
        assert rc == 0
        return st.samples

    t0 = time.time()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        samples = sum(ex.map(work, bands))
    dt = time.time() - t0
    return {"value": round(samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": "every pixel of the same frame in 8-row bands over %d host threads (%d samples, %.1f s), oracle/pt_oracle.cpp" % (threads, samples, dt)}


def cpu_baseline_node(tris, bvh4):
    """The same loop as a single-thread Node/JS program (oracle/js/pt_oracle.js, bit-identical to the C++
    oracle): every 2nd pixel in x and y of the same frame."""
    import shutil, subprocess, tempfile
    import numpy as np
    node = shutil.which("node")
    if node is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc as orc_mod
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--verify", action="store_true", help="rank 0: check the gathered frame bit-for-bit against a whole-frame render")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch                     # first: its HIP runtime is the one libmi355pt binds to
    import torch.distributed as dist
    import numpy as np
    rt = importlib.import_module("raytracer-public_amd")

    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    device = local_rank % max(n_dev, 1)
    torch.cuda.set_device(device)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

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
    def params(frame=0, stats=False):
        return ctx.make_params(width, height, mode=rt.PT_MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, frame=frame,
                               tile_rank=rank, tile_count=world, stats=stats)

    sharded = world > 1
    # Frames are submitted in batches: the library traces `batch` consecutive frames with one persistent
    # launch (pt_set_batch), which amortises the sparse tail of a frame -- essential for the small per-GPU
    # shares of a sharded run.  The RCCL gather then moves one batch at a time (fewer, larger collectives).
    # At least two launches per timed region so that consecutive launches overlap (measured: tools/sweeps/tune16.sh); up to 32
    # frames of work per launch (256 frames = pt_set_batch's maximum).  A sharded run gathers batch b while batch b+1 traces, so
    # its LAST gather is exposed: its launches shrink towards the end (128, 64, 32, 16, 8, 8 for 256 steps on 8 GPUs).
    batch = int(os.environ.get("PT_BENCH_BATCH", "0")) or (args.steps + 1) // 2
    batch = max(1, min(32 * world, 256, batch))

    def schedule(n_steps):
        """[(first step, frames)] of the launches that cover n_steps."""
        out, done = [], 0
        while done < n_steps:
            left = n_steps - done
            b = min(batch, left)
            if world > 1 and not os.environ.get("PT_BENCH_BATCH"):
                b = min(b, max(8, left // 2), left)
            out.append((done, b)); done += b
        return out
    ctx.set_batch(1)
    if sharded:
        stride = max(rt.tile_layout(width, height, r, world)[1] for r in range(world))
        compact = [torch.zeros(batch, stride, dtype=torch.float32, device="cuda") for _ in range(2)]
        gathered = [torch.zeros(world, batch, stride, dtype=torch.float32, device="cuda") for _ in range(2)] if rank == 0 else [None, None]
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
    with torch.cuda.stream(stream):
        if sharded:
            ctx.set_compact_buffer(compact[0][0].data_ptr(), stride)
        for i in range(args.steps):
            ctx.render(params(i, stats=True))
            st = ctx.stats()
            frame_bytes.append(algorithmic_bytes(st))
            my_stats = st if my_stats is None else {k: my_stats[k] + st[k] for k in st}
    my_bytes = float(sum(frame_bytes)) / max(len(frame_bytes), 1)       # mean per frame
    ctx.set_batch(batch)

    pending = [None]

    def finish(prev):
        if prev is None:
            return
        work, slot, nf = prev
        work.wait()
        if rank == 0:
            if host_stage:
                gathered[slot][:, :nf].copy_(torch.stack(work.cpu_list))
            for j in range(nf):          # rank r's buffer of frame j sits at gathered[slot][r][j]
                ctx.deinterleave(gathered[slot].data_ptr() + j * stride * 4, batch * stride, width, height, world)

    class _HostWork:                 # gloo rehearsal: synchronous host gather
        def __init__(self, buf):
            stream.synchronize()
            src = buf.cpu()
            self.cpu_list = [torch.empty_like(src) for _ in range(world)] if rank == 0 else None
            dist.gather(src, self.cpu_list, dst=0)

        def wait(self):
            pass

    def ship(slot, nf):              # the launch in compact[slot][:nf] has been submitted: gather it, finish the previous one
        if host_stage:
            work = _HostWork(compact[slot][:nf])
        else:
            glist = [gathered[slot][r][:nf] for r in range(world)] if rank == 0 else None
            work = dist.gather(compact[slot][:nf], glist, dst=0, async_op=True)
        finish(pending[0])               # gather(b-1) has had the whole launch b to complete
        pending[0] = (work, slot, nf)

    def run(n_steps, first_frame, p):    # called with `stream` current: n_steps frames as the launches of schedule()
        for k, (first, nf) in enumerate(schedule(n_steps)):
            for j in range(nf):
                p.frame = first_frame + first + j
                if sharded:
                    ctx.set_compact_buffer(compact[k & 1][j].data_ptr(), stride)
                else:
                    ctx.set_output_buffer(frames_out[(k & 1) * batch + j].data_ptr(), height * width * 4)
                ctx.render(p)                # the library launches a full batch with its last frame ...
            if nf < batch:
                ctx.flush()                  # ... and a shorter one here (the slots stay sized for `batch` frames)
            if sharded:
                ship(k & 1, nf)
        if sharded:
            finish(pending[0])
            pending[0] = None

    p = params()
    with torch.cuda.stream(stream):
        run(args.warmup, args.steps, p)
    if sharded:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.timing_begin(args.steps)
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        run(args.steps, 0, p)
    torch.cuda.synchronize()
    if sharded:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ctx.timing_collect(args.steps)

    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    verified = None
    if args.verify and rank == 0:
        with torch.cuda.stream(stream):
            got = ctx.read_radiance(width, height).copy()           # last de-interleaved (or whole) frame
            ctx.render(ctx.make_params(width, height, mode=rt.PT_MODE_PATH, spp=SPP, max_bounces=BOUNCES, seed=SEED, frame=args.steps - 1))
            want = ctx.read_radiance(width, height)
        verified = bool(np.array_equal(got.view(np.uint32), want.view(np.uint32)))
        if not verified:
            raise SystemExit("bench.py --verify: gathered frame differs from the whole-frame render")

    if rank == 0:
        samples_per_step = width * height * SPP
        value = samples_per_step * args.steps / elapsed / 1e6
        # one event pair per launch; a launch traces up to `batch` frames
        n_launch = max(len(kernel_ms), 1)
        k_avg_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        frames_per_launch = args.steps / n_launch
        achieved = my_bytes * frames_per_launch / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # written from a rocprofv3 --pmc pass
        if os.path.exists(pmc):
            try:
                per_frame = json.load(open(pmc)).get("bytes_per_launch_n%d" % world)     # measured per frame of work
                traffic = int(per_frame * (args.steps / max(len(kernel_ms), 1))) if per_frame else None
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/sec @1920x1080 Stanford-Dragon-class, 4 spp, 8 bounces",
            "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "verified": verified,
            "config": {"workload": "C2: dragon-class procedural closed mesh (%d tris, seed %d; dragon.glb absent), native LBVH2->BVH4, %dx%d, %d spp, %d bounces, camera (0,0,2.5) identity quat FOV 70, a new frame index (sample set) every step"
                                   % (NUM_TRIS, SCENE_SEED, width, height, SPP, BOUNCES),
                       "triangles": NUM_TRIS, "bvh4_nodes": ctx.scene_info()["numNodes4"], "width": width, "height": height,
                       "spp": SPP, "max_bounces": BOUNCES, "seed": SEED,
                       "sharding": ("interleaved 8x8 tiles over %d GPUs, RCCL gather to rank 0" % world) if sharded else "single GPU, whole frame"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": "trace_paths_kernel (persistent megakernel)", "kernel_avg_ms": round(k_avg_ms, 4),
                         "frames_per_launch": frames_per_launch,
                         "note": "algorithmic bytes (reference record sizes x records examined) over the per-launch duration by hipEvents on the launch stream; "
                                 "a launch traces up to %d frames and consecutive launches overlap on side streams; the scene is cache resident, so the algorithmic rate "
                                 "may exceed the HBM peak -- see traffic" % batch,
                         "achieved_from_throughput": round(my_bytes * args.steps / elapsed / 1e9, 2),
                         "algorithmic_bytes_per_frame": int(my_bytes), "algorithmic_bytes_per_frame_min_max": [int(min(frame_bytes)), int(max(frame_bytes))],
                         "algorithmic_bytes_per_launch": int(my_bytes * frames_per_launch),
                         "counters_all_timed_frames": {k: my_stats[k] for k in ("rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "samples")}},
        }
        if world == 1 and not args.no_cpu_baseline and (width, height) == (WIDTH, HEIGHT):
            bvh4 = ctx.read_bvh4()
            base, ost = cpu_baseline(tris, bvh4)
            out["cpu_baseline"] = base
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
