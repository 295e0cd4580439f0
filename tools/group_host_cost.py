#!/usr/bin/env python3
"""Host time per pt_group_render: 8 copy-transport members on ONE GPU, a tiny frame (so the GPU is never what the host waits for), at
batch 1 / 32 / 256.  What is timed is the submitting thread: per frame 8 x (pt_set_compact_buffer + pt_render), per batch the members'
launches, 8 peer copies (in place of the one ncclGather per member) and ONE de-interleave launch on rank 0.  Compare with the GPU time of
a 1/8 share of a 1080p frame (0.09 ms): the group must not be host-bound at the batch sizes it is used with."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
members = int(os.environ.get("GH_MEMBERS", "8"))
w, h = (int(x) for x in os.environ.get("GH_SIZE", "64x40").split("x"))
tris = rt.procedural_scene(0, 2000)
g = rt.Group([0] * members, rt.PT_GROUP_TRANSPORT_COPY)
g.set_triangles(tris); g.build_bvh()
print("%d members on one GPU, %dx%d frame, copy transport" % (members, w, h))
for batch in (1, 32, 256):
    g.set_batch(batch)
    n = max(2 * batch, 256)
    p = g.make_params(w, h, mode=rt.PT_MODE_PATH, spp=1, max_bounces=1)
    for i in range(batch): p.frame = i; g.render(p)          # buffers, slots, first launches
    g.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        p.frame = 1000 + i
        g.render(p)
    t_submit = time.perf_counter() - t0                      # the host side only: nothing here waits for the GPU
    g.synchronize()
    t_all = time.perf_counter() - t0
    print("batch %3d: host %.1f us per pt_group_render (%.1f us per member call), %.1f us per frame incl. the final wait; %d frames" %
          (batch, t_submit / n * 1e6, t_submit / n / members * 1e6, t_all / n * 1e6, n))
g.close()
