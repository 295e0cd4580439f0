#!/usr/bin/env python3
"""The reference's own workload (README.md:94-98: 1080p, one primary ray per pixel, N.L shade): frames per second on the
dragon-class scene in PT_MODE_REFERENCE -- on the persistent megakernel (one render() per launch; launches of 8 and 32 frames; a
lone frame with a host wait) and on the one-pixel-per-lane kernel (PT_FLAG_SIMPLE_KERNEL) -- and in PT_MODE_REFERENCE_PACKET
(literal 2x2 packets)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
ctx = rt.Context(0); ctx.set_triangles(rt.procedural_scene(0, 871414)); ctx.build_bvh()


def run(name, mode, batch=1, simple=False, lone=False, n=480):
    p = ctx.make_params(1920, 1080, mode=mode, simple_kernel=simple)
    ctx.set_batch(batch)
    for _ in range(2 * max(batch, 10)): ctx.render(p)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render(p)
        if lone: ctx.synchronize()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ctx.set_batch(1)
    print("%-58s %.3f ms/frame, %6.0f FPS, %6.0f M primary samples/s" % (name + ":", dt / n * 1e3, n / dt, 1920 * 1080 * n / dt / 1e6), flush=True)


run("PT_MODE_REFERENCE, megakernel, one render() per launch", rt.PT_MODE_REFERENCE)
if os.environ.get("PT_TUNE_SLOTS"): sys.exit(0)      # tools/ab/ref_sweep.sh only wants the first line
run("PT_MODE_REFERENCE, megakernel, 8 frames per launch", rt.PT_MODE_REFERENCE, batch=8)
run("PT_MODE_REFERENCE, megakernel, 32 frames per launch", rt.PT_MODE_REFERENCE, batch=32)
run("PT_MODE_REFERENCE, megakernel, lone frames (host waits)", rt.PT_MODE_REFERENCE, lone=True, n=200)
run("PT_MODE_REFERENCE, one pixel per lane", rt.PT_MODE_REFERENCE, simple=True)
run("PT_MODE_REFERENCE_PACKET (literal 2x2 packets)", rt.PT_MODE_REFERENCE_PACKET, n=200)
