#!/usr/bin/env python3
"""Per-wavefront census of an instrumented launch of the two-slot megakernel (pt_trace2.hip): how many node steps, leaf steps and
service passes a wavefront ran and how many lanes took part in each (the kernel is bound by vector-instruction issue, so the
density of each kind of step is what decides its speed).  PF_SCENE=sponza for config C4; PT_TUNE_SHADE / PT_TUNE_LEAF thresholds."""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
sponza = os.environ.get("PF_SCENE") == "sponza"
tris = rt.procedural_scene(1, 262144) if sponza else rt.procedural_scene(0, 871414)
cam, quat = ((0.55, -0.05, 0.05), (0.0, 0.6630, 0.0, 0.7486)) if sponza else ((0, 0, 2.5), (0, 0, 0, 1))
ctx = rt.Context(0); ctx.set_triangles(tris); ctx.build_bvh()
p = ctx.make_params(1920, 1080, cam, quat, mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, stats=True)
ctx.render(p); ctx.render(p)
print("ms (stats build):", ctx.last_render_ms())
buf = np.zeros((8192, 16), np.uint64); n = C.c_uint32()
rt.lib.pt_debug_wave_times(ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_uint32(8192), C.byref(n))
w = buf[: n.value].astype(np.float64)
t0 = w[:, 0].min()
beg, qe, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0
print("waves", n.value)
for name, v in (("queue-empty", qe), ("end", end)):
    print("%-12s us: min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % ((name,) + tuple(np.percentile(v, [0, 10, 50, 90, 99, 100]))))
names = ("node", "leaf", "hit", "ret", "cam")
its = [w[:, 3 + k].sum() for k in range(5)]
lanes = [w[:, 9 + k].sum() for k in range(5)]
it_node_q, l_node_q = w[:, 8].sum(), w[:, 14].sum()
cost = {"node": 180, "leaf": 100, "hit": 330, "ret": 110, "cam": 260}
for k, nm in enumerate(names):
    print("%-5s %8.3fM iterations, %5.1f lanes each%s" % (nm, its[k] / 1e6, lanes[k] / max(its[k], 1),
          (" (%.1f before the queue ran dry: %.3fM iterations)" % (l_node_q / max(it_node_q, 1), it_node_q / 1e6)) if nm == "node" else ""))
print("estimated VALU: %.0fM" % (sum(its[k] * cost[nm] for k, nm in enumerate(names)) / 1e6))
st = ctx.stats(); print(st)
dbg = np.zeros(16, np.uint64); rt.lib.pt_debug_counters(ctx.h, dbg.ctypes.data_as(C.c_void_p))
print('pushes %d, at depth>=8 %.2f%%, >=12 %.2f%%, spilled %.3f%%' % (dbg[8], 100.0 * dbg[9] / dbg[8], 100.0 * dbg[10] / dbg[8], 100.0 * dbg[11] / dbg[8]))
