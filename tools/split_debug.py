#!/usr/bin/env python3
"""Diagnostics for ray splitting: mismatching pixels against the un-split run for a few variants of PT_TUNE_SPLIT (bit 0 on, bit 1 helpers do not give, bit 2 any-hit rays are not split)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
tris = rt.procedural_scene(0, 60000)
ctx = rt.Context(0); ctx.set_triangles(tris); ctx.build_bvh()
ctx.debug_set_tune("CONSOLIDATE", 0)
w, h = 640, 360
for bounces in (0, 1, 8):
    kw = dict(mode=rt.PT_MODE_PATH, spp=4, max_bounces=bounces, seed=3)
    ctx.debug_set_tune("SPLIT", 0)
    ctx.render(ctx.make_params(w, h, **kw)); want = ctx.read_radiance().copy()
    for v in (1, 3):
        ctx.debug_set_tune("SPLIT", v)
        bad = []
        for rep in range(3):
            ctx.render(ctx.make_params(w, h, **kw)); got = ctx.read_radiance()
            d = (got.view(np.uint32) != want.view(np.uint32)).any(axis=2)
            bad.append(int(d.sum()))
        ys, xs = np.nonzero(d)
        print("bounces %d split=%d: mismatching pixels %s  max |diff| %.4g  first %s" % (bounces, v, bad, float(np.abs(got - want).max()), list(zip(xs[:4].tolist(), ys[:4].tolist()))))
