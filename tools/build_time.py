#!/usr/bin/env python3
"""Scene build time (pt_set_triangles + pt_build_bvh) for the C2 / C4 stand-in scenes."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
ctx = rt.Context(0)
for kind, n in ((0, 871414), (1, 262144), (0, 100000)):
    tris = rt.procedural_scene(kind, n)
    for rep in range(3):
        t0 = time.perf_counter(); ctx.set_triangles(tris); t1 = time.perf_counter(); ctx.build_bvh(); t2 = time.perf_counter()
    print("kind %d, %7d tris: set_triangles %.1f ms, build_bvh %.1f ms, numNodes4 %d" % (kind, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ctx.scene_info()["numNodes4"]))
