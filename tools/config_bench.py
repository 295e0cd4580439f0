#!/usr/bin/env python3
"""Throughput of the BASELINE.json configurations on ONE MI355X (C3 and C5 are 8-GPU configurations: this is one GPU's
whole-frame rate, the sharded rate is bench.py's business):
  C2 dragon-class, BVH2 -> collapsed BVH4, 1080p, 4 spp, 8 bounces
  C3 dragon-class, BVH4_wide input (tests/test.cpp format), 1080p, 16 spp, 8 bounces
  C4 sponza-class interior, 1080p, 4 spp, 8 bounces
  C5 dragon-class, 3840x2160, 64 spp as 16 accumulated frames of 4 spp, 16 bounces"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
ctx = rt.Context(0)


def run(name, w, h, spp, bounces, frames, cam=(0, 0, 2.5), quat=(0, 0, 0, 1), accumulate=False, batch=8):
    ctx.set_batch(batch)
    def loop(n, base):
        for i in range(n):
            ctx.render(ctx.make_params(w, h, cam, quat, mode=rt.PT_MODE_PATH, spp=spp, max_bounces=bounces, seed=1, frame=base + i, accumulate=accumulate))
        ctx.synchronize()
    loop(batch * 2, 1000)
    t0 = time.perf_counter(); loop(frames, 0); dt = time.perf_counter() - t0
    ctx.set_batch(1)
    print("%-4s %4dx%-4d %2d spp %2d bounces%s: %8.3f ms/frame, %7.0f Msamples/s  (%d frames, %d per launch)"
          % (name, w, h, spp, bounces, " accumulate" if accumulate else "", dt / frames * 1e3, w * h * spp * frames / dt / 1e6, frames, batch), flush=True)


dragon = rt.procedural_scene(rt.SCENE_DRAGON_CLASS, 871414, 20260109)
ctx.set_triangles(dragon); ctx.build_bvh()
run("C2", 1920, 1080, 4, 8, 64, batch=32)
wide = rt.bvh2_to_bvh4_wide(ctx.read_bvh2())
bvh4 = ctx.read_bvh4()
ctx.set_bvh4(wide)
run("C3", 1920, 1080, 16, 8, 16, batch=8)
ctx.set_bvh4(bvh4)
run("C5", 3840, 2160, 4, 16, 32, accumulate=True, batch=16)
sponza = rt.procedural_scene(rt.SCENE_SPONZA_CLASS, 262144)
ctx.set_triangles(sponza); ctx.build_bvh()
run("C4", 1920, 1080, 4, 8, 16, cam=(0.55, -0.05, 0.05), quat=(0.0, 0.6630, 0.0, 0.7486), batch=8)
