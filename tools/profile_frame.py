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
scene = os.environ.get("PF_SCENE", "dragon")
if scene == "sponza":      # config C4: sponza-class interior, camera inside looking down the hall
    tris = rt.procedural_scene(rt.SCENE_SPONZA_CLASS, 262144)
    cam, quat = (0.55, -0.05, 0.05), (0.0, 0.6630, 0.0, 0.7486)
else:
    tris = rt.procedural_scene(rt.SCENE_DRAGON_CLASS, 871414)
    cam, quat = (0, 0, 2.5), (0, 0, 0, 1)
ctx = rt.Context(0)
ctx.set_triangles(tris)
ctx.build_bvh()
p = ctx.make_params(1920, 1080, cam, quat, mode=mode, spp=spp, max_bounces=bounces, seed=1)
ms = []
for _ in range(frames):
    ctx.render(p)
    ms.append(ctx.last_render_ms())
print("kernel ms:", " ".join("%.3f" % m for m in ms))
if os.environ.get("PF_STATS"):
    p.flags |= rt.PT_FLAG_STATS; ctx.render(p); st = ctx.stats(); print(st, "algorithmic GB", (32 * st["nodes_examined"] + 36 * st["tris_tested"] + 16 * st["samples"]) / 1e9)
if os.environ.get("PF_OUT"):
    import numpy as np; np.save(os.environ["PF_OUT"], ctx.read_tonemapped()[::2, ::2, :3])
ctx.close()
