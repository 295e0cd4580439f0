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
    # the first pt_read_bvh2 after a build also runs the bottom-up walk that gives the internal BVH2 nodes their bounds (device time of
    # that kernel alone: rocprofv3 --kernel-trace --stats, lbvh2_leaves_kernel<false, true>: 0.12 ms for 871,414 triangles)
    t3 = time.perf_counter(); ctx.read_bvh2(); t4 = time.perf_counter(); ctx.read_bvh2(); t5 = time.perf_counter()
    print("kind %d, %7d tris: set_triangles %.1f ms, build_bvh %.1f ms, numNodes4 %d; readBVH2 (walk + the copy to the host) %.1f ms, again (copy alone) %.1f ms"
          % (kind, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, ctx.scene_info()["numNodes4"], (t4 - t3) * 1e3, (t5 - t4) * 1e3))
