#!/usr/bin/env python3
"""The reference's own workload (README.md:94-98: 1080p, one primary ray per pixel, N.L shade): frames per second
in PT_MODE_REFERENCE (one ray per lane) and PT_MODE_REFERENCE_PACKET (literal 2x2 packets) on the dragon-class scene."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
ctx = rt.Context(0); ctx.set_triangles(rt.procedural_scene(0, 871414)); ctx.build_bvh()
for name, mode in (("PT_MODE_REFERENCE", rt.PT_MODE_REFERENCE), ("PT_MODE_REFERENCE_PACKET", rt.PT_MODE_REFERENCE_PACKET)):
    p = ctx.make_params(1920, 1080, mode=mode)
    for _ in range(20): ctx.render(p)
    ctx.synchronize()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n): ctx.render(p)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print("%s: %.3f ms/frame, %.0f FPS, %.0f M primary samples/s" % (name, dt / n * 1e3, n / dt, 1920 * 1080 * n / dt / 1e6))
