#!/usr/bin/env python3
"""Per-wavefront timeline of a megakernel launch: when did each wavefront see the queue empty, when did it re-seat its paths, when did it exit,
how many traversal steps / shade passes / refills did it run, at what lane utilisation.

  WT_VARIANT=timeline (default)  the TIMELINE variant of the kernel (knob TIMELINE = 1): the production kernel + wave-uniform bookkeeping in scalar
                                 registers -- production registers, no scratch, production speed; timestamps, step / pass counts, lane sums
  WT_VARIANT=counters            the COUNTERS variant (PT_FLAG_STATS): per-lane counters the compiler spills -- 2.5 x slower; adds the cycle shares,
                                 path lengths and the fork / re-seat counters.  Its TIMES describe the slow kernel, not the production one.
  WT_BATCH=n                     an n-frame launch;  PF_SCENE=sponza  config C4 instead of C2"""
import ctypes as C, importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
sponza = os.environ.get("PF_SCENE") == "sponza"      # config C4 instead of C2
tris = rt.procedural_scene(1, 262144) if sponza else rt.procedural_scene(0, 871414)
cam, quat = ((0.55, -0.05, 0.05), (0.0, 0.6630, 0.0, 0.7486)) if sponza else ((0, 0, 2.5), (0, 0, 0, 1))
ctx = rt.Context(0); ctx.set_triangles(tris); ctx.build_bvh()
counters = os.environ.get("WT_VARIANT", "timeline") == "counters"
p = ctx.make_params(1920, 1080, cam, quat, mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, stats=counters)
B = int(os.environ.get("WT_BATCH", "1"))            # frames in the instrumented launch (diagnostics knob STATSBATCH)
if B > 1:
    if counters: rt.lib.pt_debug_set_tune(ctx.h, b"STATSBATCH", C.c_uint32(1))
    ctx.set_batch(B)
for rep in range(3):                                # production launch first (its time is the yardstick), then the instrumented one twice
    if rep == 1 and not counters: rt.lib.pt_debug_set_tune(ctx.h, b"TIMELINE", C.c_uint32(1))
    if rep == 0 and counters: p.flags &= ~rt.PT_FLAG_STATS
    for k in range(3 if rep == 0 else 1):
        for i in range(B):
            p.frame = 100 * rep + i; ctx.render(p)
        ctx.synchronize()
    if rep == 0:
        print("frames in the launch: %d; ms (production kernel): %.3f" % (B, ctx.last_render_ms()))
        if counters: p.flags |= rt.PT_FLAG_STATS
print("ms (%s variant): %.3f" % ("COUNTERS" if counters else "TIMELINE", ctx.last_render_ms()))
buf = np.zeros((8192, 24), np.uint64); n = C.c_uint32()
rt.lib.pt_debug_wave_times(ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_uint32(8192), C.byref(n))
w = buf[: n.value].astype(np.float64)
t0 = w[:, 0].min()
beg, qe, end = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0   # microseconds
print("waves", n.value)
for name, v in (("begin", beg), ("queue-empty", qe), ("end", end)):
    print("%-12s us: min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % ((name,) + tuple(np.percentile(v, [0, 10, 50, 90, 99, 100]))))
print("iterations per wave: p50 %d max %d ; shade passes p50 %d ; refills p50 %d" % (np.median(w[:, 3]), w[:, 3].max(), np.median(w[:, 4]), np.median(w[:, 5])))
print("iter time us (p50):", np.median((end - beg) / np.maximum(w[:, 3], 1)))
live = w[:, 1] > 0
pre_it, post_it = w[live, 6], w[live, 3] - w[live, 6]
pre_t, post_t = (qe[live] - beg[live]), (end[live] - qe[live])
print("pre-exhaustion : iters p50 %d, us/iter p50 %.2f, lane util %.1f%%" % (np.median(pre_it), np.median(pre_t / np.maximum(pre_it, 1)), 100 * w[live, 8].sum() / (64 * pre_it.sum())))
print("post-exhaustion: iters p50 %d, us/iter p50 %.2f, lane util %.1f%%" % (np.median(post_it), np.median(post_t / np.maximum(post_it, 1)), 100 * (w[live, 7] - w[live, 8]).sum() / (64 * np.maximum(post_it.sum(), 1))))
print("leaf-lane share of traversing lanes: %.1f%%" % (100 * w[:, 9].sum() / w[:, 7].sum()))
print("total wave-iterations %.3fM, lane-steps %.1fM" % (w[:, 3].sum() / 1e6, w[:, 7].sum() / 1e6))
tot = (w[:, 2] - w[:, 0]).sum() * 24.0   # 100 MHz ticks -> ~2.4 GHz cycles
if counters:
  print('cycle shares of wave lifetime (s_memtime): shade %.1f%%, refill %.1f%%, traversal step %.1f%% (pre-exhaustion %.1f%%)' % tuple(100 * w[:, k].sum() / tot for k in (10, 11, 12, 13)))
if counters:
  print('cycles per shade pass %.0f, per refill %.0f, per step pre %.0f, post %.0f' % (w[:, 10].sum() / w[:, 4].sum(), w[:, 11].sum() / w[:, 5].sum(), w[live, 13].sum() / pre_it.sum(), (w[live, 12] - w[live, 13]).sum() / post_it.sum()))
rs = w[:, 16] > 0
if rs.any():
    t_rs = (w[rs, 16] - t0) / 100.0
    it_1 = w[rs, 17] - w[rs, 6]; it_q = w[rs, 3] - w[rs, 17]              # iterations after the queue ran dry: one ray per lane / per quad
    print("re-seated (one ray per quad): %d of %d wavefronts; at us: p10 %.0f p50 %.0f p90 %.0f; after queue-empty by us p50 %.0f" % (rs.sum(), n.value, *np.percentile(t_rs, [10, 50, 90]), np.median(t_rs - qe[rs])))
    print("  iterations between queue-empty and re-seating p50 %d (us/iter p50 %.2f), in quad mode p50 %d max %d (us/iter p50 %.2f); rays per quad-mode wavefront-iteration %.2f" % (
        np.median(it_1), np.median((t_rs - qe[rs]) / np.maximum(it_1, 1)), np.median(it_q), it_q.max(), np.median((end[rs] - t_rs) / np.maximum(it_q, 1)), (w[rs, 7] - w[rs, 18]).sum() / 4.0 / max(it_q.sum(), 1)))
    if counters: print("  cycles per step: one ray per lane after queue-empty %.0f, quad mode %.0f" % ((w[rs, 19] - w[rs, 13]).sum() / max(it_1.sum(), 1), (w[rs, 12] - w[rs, 19]).sum() / max(it_q.sum(), 1)))
order = np.argsort(end)[::-1][:12]
print("slowest waves: end us | queue-empty us | iterations after | shade passes after" + (" | longest path that ended after (steps)" if counters else ""))
for k in order:
    print("  %7.0f | %7.0f | %5d | %4d" % (end[k], qe[k], w[k, 3] - w[k, 6], w[k, 4] - w[k, 15]) + (" | %5d" % w[k, 14] if counters else ""))
post = w[live, 3] - w[live, 6]
print("iterations after queue-empty: p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(post, [50, 90, 99, 100])))
print("us per iteration after queue-empty, the 12 slowest wavefronts: " + ", ".join("%.2f" % ((end[k] - qe[k]) / max(w[k, 3] - w[k, 6], 1)) for k in order))
print('wavefronts still running at us after the first began: ' + ', '.join('%d: %d' % (t, int((end > t).sum())) for t in np.percentile(end, [10, 30, 50, 70, 90, 97]).astype(int)))
if counters:
    st = ctx.stats(); print(st)
    print("longest path ended after queue-empty (steps): p50 %d p90 %d p99 %d max %d" % tuple(np.percentile(w[live, 14], [50, 90, 99, 100])))
    dbg = np.zeros(24, np.uint64); rt.lib.pt_debug_counters(ctx.h, dbg.ctypes.data_as(C.c_void_p))
    print('pushes %d, at depth>=8 %.2f%%, >=12 %.2f%%, spilled(>=kShort) %.3f%%' % (dbg[8], 100.0 * dbg[9] / dbg[8], 100.0 * dbg[10] / dbg[8], 100.0 * dbg[11] / dbg[8]))
    print('wavefronts that re-seated their paths (one ray per quad of lanes): %d' % int(dbg[16]))
    print('longest path %d traversal steps, longest ray %d; paths >= 512 steps: %d, >= 1024: %d' % (dbg[12], dbg[13], dbg[14], dbg[15]))
    print('paths >= 512 steps: %.1f %% of their traversal steps belong to shadow rays (%d of %d)' % (100.0 * dbg[17] / max(int(dbg[18]), 1), dbg[17], dbg[18]))
    print('quad mode: shadow rays handed to an idle quad %d, no idle quad %d, paths that waited for their shadow ray %d; shadow rays in all %d' % (dbg[19], dbg[20], dbg[21], st['rays_shadow']))
    print('after a wavefront had nothing left to start: %.1f %% of its (path x iteration) slots were paths waiting for their forked shadow ray (%d of %d)' % (100.0 * dbg[22] / max(int(dbg[23]), 1), dbg[22], dbg[23]))
