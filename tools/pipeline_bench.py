#!/usr/bin/env python3
"""Back-to-back frames without host syncs (what bench.py times): ms per frame vs frame slots."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if os.environ.get("PB_TORCH"):
    import torch
# PB_BASE=1: the library of an earlier commit (tools/ab/raytracer_base: its __init__.py + libmi355pt.so, built by tools/ab/README) for same-session A/B
if os.environ.get("PB_BASE"): sys.path.insert(0, os.path.join(ROOT, "tools", "ab")); rt = importlib.import_module("raytracer_base")
else: rt = importlib.import_module("raytracer-public_amd")
sponza = os.environ.get("PF_SCENE") == "sponza"      # config C4 instead of C2
tris = rt.procedural_scene(1, 262144) if sponza else rt.procedural_scene(0, 871414)
cam, quat = ((0.55, -0.05, 0.05), (0.0, 0.6630, 0.0, 0.7486)) if sponza else ((0, 0, 2.5), (0, 0, 0, 1))
ctx = rt.Context(0)
if os.environ.get("PB_TORCH") == "2":
    st = torch.cuda.Stream(); ctx.set_stream(st.cuda_stream)
ctx.set_triangles(tris); ctx.build_bvh()
tc = int(os.environ.get('PB_TILES', '1'))
p = ctx.make_params(1920, 1080, cam, quat, mode=rt.PT_MODE_PATH, spp=int(os.environ.get('PB_SPP', '4')), max_bounces=int(os.environ.get('PB_BOUNCES', '8')), tile_rank=0, tile_count=tc)
B = int(os.environ.get('PB_BATCH', '1')); ctx.set_batch(B)
for _ in range(8 * B): ctx.render(p)
ctx.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if os.environ.get("PB_RING"): ctx.timing_begin(n)
t0 = time.perf_counter()
vary = bool(os.environ.get("PB_VARY"))        # a new frame index (new sample set) every frame, like a progressive render
solo = bool(os.environ.get("PB_SOLO"))        # wait for every frame before the next one is submitted (latency of a single frame)
for i in range(n):
    if vary: p.frame = 1000 + i
    ctx.render(p)
    if solo and (i + 1) % B == 0: ctx.synchronize()      # wait for every launch (B frames) before the next is submitted
ctx.synchronize()
dt = time.perf_counter() - t0
if os.environ.get("PB_RING"):
    ms = ctx.timing_collect(n); print("ring: kernel avg %.3f ms" % ms.mean())
print("tiles 1/%d batch=%d " % (tc, B), end="")
print("slots=%s: %.3f ms/frame, %.0f Msamples/s" % (os.environ.get("PT_TUNE_SLOTS", "default"), dt / n * 1e3, 1920 * 1080 * p.spp * n / dt / 1e6))
