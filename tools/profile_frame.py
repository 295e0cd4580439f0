#!/usr/bin/env python3
"""Render the C2 frame a few times without torch (for rocprofv3 --pmc / --kernel-trace passes).
usage: python tools/profile_frame.py [frames] [spp] [bounces] [mode]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bounces = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mode = int(sys.argv[4]) if len(sys.argv) > 4 else rt.PT_MODE_PATH
tris = rt.procedural_scene(rt.SCENE_DRAGON_CLASS, 871414)
ctx = rt.Context(0)
ctx.set_triangles(tris)
ctx.build_bvh()
p = ctx.make_params(1920, 1080, mode=mode, spp=spp, max_bounces=bounces, seed=1)
ms = []
for _ in range(frames):
    ctx.render(p)
    ms.append(ctx.last_render_ms())
print("kernel ms:", " ".join("%.3f" % m for m in ms))
ctx.close()
