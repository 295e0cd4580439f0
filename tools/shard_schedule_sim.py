#!/usr/bin/env python3
"""Compute side of a sharded bench.py run, replayed on ONE GPU: for (N, steps, warmup) play rank r's exact submission
sequence -- tile_rank = r, tile_count = N, the same pt_set_batch / pt_set_compact_buffer / pt_render / pt_flush calls in the
same order, through bench.submit_launches with bench.schedule's launches -- without the collective, and print ms per step
for every r.  What a rank's GPU needs for its share bounds the N-GPU step time from below (the gather of the last launch
and rank 0's de-interleave come on top; the gathers before it travel while the next launch traces).

  python3 tools/shard_schedule_sim.py --gpus 8 --steps 20 --warmup 5 [--schedule 10,10] [--ranks 0,3]

--schedule overrides bench.schedule for the timed steps (to compare candidates in one session)."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--schedule", type=str, default="")
    ap.add_argument("--ranks", type=str, default="")
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()

    import torch
    import bench
    rt = importlib.import_module("raytracer-public_amd")
    world = args.gpus
    W, H = bench.WIDTH, bench.HEIGHT
    tris = rt.procedural_scene(rt.SCENE_DRAGON_CLASS, bench.NUM_TRIS, bench.SCENE_SEED)
    ctx = rt.Context(0)
    stream = torch.cuda.ExternalStream(ctx.get_stream(), device=0)
    ctx.set_triangles(tris); ctx.build_bvh()

    cap = max(1, min(32 * world, 256, args.steps))
    if args.schedule:
        os.environ["PT_BENCH_SCHEDULE"] = args.schedule
    timed = bench.schedule(args.steps, cap, world, False)
    os.environ.pop("PT_BENCH_SCHEDULE", None)
    warm = bench.schedule(max(args.warmup, 1), cap, world, False)
    batch = max([nf for _, nf in timed] + [nf for _, nf in warm]) if world > 1 else cap
    stride = max(rt.tile_layout(W, H, r, world)[1] for r in range(world))
    compact = [torch.zeros(batch, stride, dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    ranks = [int(x) for x in args.ranks.split(",")] if args.ranks else list(range(world))
    print("N=%d steps=%d warmup=%d: launches %s (pt_set_batch %d), %d share-frames of %.4f frames of work each"
          % (world, args.steps, args.warmup, [nf for _, nf in timed], batch, args.steps, 1.0 / world))
    worst = 0.0
    for r in ranks:
        p = ctx.make_params(W, H, mode=rt.PT_MODE_PATH, spp=bench.SPP, max_bounces=bench.BOUNCES, seed=bench.SEED, tile_rank=r, tile_count=world)
        ctx.set_batch(batch)

        def set_target(k, j, frame):
            ctx.set_compact_buffer(compact[k & 1][j].data_ptr(), stride)

        best = None
        with torch.cuda.stream(stream):
            for rep in range(args.reps):
                if args.warmup:
                    bench.submit_launches(ctx, p, warm, args.steps, batch, set_target, lambda k, nf: None)
                ctx.synchronize()
                t0 = time.perf_counter()
                bench.submit_launches(ctx, p, timed, 1000 * rep, batch, set_target, lambda k, nf: None)
                ctx.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            ctx.set_compact_buffer(0, 0)
        ms = best / args.steps * 1e3
        worst = max(worst, ms)
        print("  rank %d: %.4f ms per step (%.3f ms for the %d steps; best of %d)" % (r, ms, best * 1e3, args.steps, args.reps))
    whole = None
    if not args.ranks:
        # the same steps as whole frames on this GPU, for the ratio
        p = ctx.make_params(W, H, mode=rt.PT_MODE_PATH, spp=bench.SPP, max_bounces=bench.BOUNCES, seed=bench.SEED)
        b1 = max(1, min(32, args.steps)); ctx.set_batch(b1)
        for rep in range(2):
            ctx.synchronize(); t0 = time.perf_counter()
            for i in range(args.steps):
                p.frame = 5000 + i; ctx.render(p)
            ctx.synchronize(); whole = (time.perf_counter() - t0) / args.steps * 1e3
    print("  slowest rank %.4f ms per step%s" % (worst, "" if whole is None else "; whole frames on this GPU %.4f ms per step -> compute side divides by %.2f at N=%d" % (whole, whole / worst, world)))
    ctx.close()


if __name__ == "__main__":
    main()
