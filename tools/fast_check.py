#!/usr/bin/env python3
"""Acceptance check of the FAST kernel variant (knob FAST): full-size frames of C2 and C4 against the exact kernel (which is the
oracle's image bit for bit): relative L2 of the path-traced frame (BASELINE's tolerance: 1e-4), differing pixels, and -- as the stand-in for "identical
primary-hit triangle ids" -- differing pixels of the REFERENCE-mode frame, whose colour is a function of the primary hit's triangle normal alone."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
W, H = 1920, 1080
for name, kind, n, cam, quat in (("C2 dragon-class", 0, 871414, (0, 0, 2.5), (0, 0, 0, 1)), ("C4 sponza-class", 1, 262144, (0.55, -0.05, 0.05), (0.0, 0.6630, 0.0, 0.7486))):
    ctx = rt.Context(0); ctx.set_triangles(rt.procedural_scene(kind, n)); ctx.build_bvh()
    out = {}
    for fast in (0, 1):
        ctx.debug_set_tune("FAST", fast)
        ctx.render(ctx.make_params(W, H, cam, quat, mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, seed=1, frame=3)); path = ctx.read_radiance().copy()
        ctx.render(ctx.make_params(W, H, cam, quat, mode=rt.PT_MODE_REFERENCE)); ref = ctx.read_radiance().copy()
        out[fast] = (path, ref)
    a, b = out[0][0][..., :3].astype(np.float64), out[1][0][..., :3].astype(np.float64)
    rel = float(np.sqrt(((a - b) ** 2).sum() / (a ** 2).sum()))
    px_rel = np.sqrt(((a - b) ** 2).sum(axis=2)) / np.maximum(np.sqrt((a ** 2).sum(axis=2)), 1e-30)
    diff_path = int((out[0][0].view(np.uint32) != out[1][0].view(np.uint32)).any(axis=2).sum())
    diff_ref = int((out[0][1].view(np.uint32) != out[1][1].view(np.uint32)).any(axis=2).sum())
    print("%s: path-traced frame FAST vs exact: relative L2 %.3e (whole image), worst pixel %.3e, %d of %d pixels differ in some bit; reference-mode frame (primary hit's normal): %d pixels differ"
          % (name, rel, float(px_rel.max()), diff_path, W * H, diff_ref))
    ctx.close()
