#!/usr/bin/env python3
"""Randomised differential soak of the hot path: the persistent megakernel under random scenes, cameras,
resolutions, spp / bounce counts, tile shares, batch sizes, accumulation and quad-mode thresholds against the one-pixel-per-lane
kernel (a second, independent HIP implementation that the parity tests pin to the oracle), bit for bit.

usage: python tools/soak.py [seconds] [seed]        (exit code 1 and the failing configuration on a mismatch)"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("raytracer-public_amd")
hip = C.CDLL("libamdhip64.so")


def quat(yaw, pitch):
    cy, sy, cp, sp = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2)
    return (float(cy * sp), float(sy * cp), float(-sy * sp), float(cy * cp))


def same(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    ctx = rt.Context(0)
    t_end = time.time() + seconds
    rounds = frames = 0
    last_print = time.time()
    while time.time() < t_end:
        kind = int(rng.integers(0, 3))
        if kind == 2:
            n = int(rng.integers(1, 4000)); tris = rng.uniform(-1, 1, n * 9).astype(np.float32)          # triangle soup: deep, bad BVHs
        else:
            n = int(rng.integers(12000, 120000)); tris = rt.procedural_scene(kind, n, int(rng.integers(1, 1000)))
        ctx.set_triangles(tris); ctx.build_bvh()
        for _ in range(int(rng.integers(2, 6))):
            w, h = int(rng.integers(1, 420)), int(rng.integers(1, 300))
            spp, bounces = int(rng.integers(1, 7)), int(rng.integers(0, 11))
            inside = kind == 1 or rng.random() < 0.2
            cams = []
            nf = int(rng.integers(1, 13)) if rng.random() < 0.8 else int(rng.integers(13, 41))       # sometimes more than one full launch
            for f in range(nf):
                pos = rng.uniform(-0.6, 0.6, 3) if inside else rng.uniform(-1, 1, 3) * 0.5 + np.array([0, 0, 2.4])
                cams.append((tuple(float(v) for v in pos), quat(rng.uniform(-3.2, 3.2) if inside else rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5))))
            accumulate = bool(rng.random() < 0.35)
            count = int(rng.choice([1, 1, 2, 3, 4, 8])) if not accumulate else 1
            batch = int(rng.integers(1, 13)) if rng.random() < 0.8 else int(rng.integers(13, 33))
            sd = int(rng.integers(0, 2 ** 31))
            # the drain's quad mode at its default and at odd thresholds (paths a wavefront may hold when it re-seats them; 0 = never)
            quad = int(rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 16, 9, 4, 1, 0]))
            ctx.debug_set_tune("QUAD", quad)
            fork = int(rng.choice([0xFFFFFFFF, 0xFFFFFFFF, 2, 1, 0]))       # forked shadow rays: default (idle lanes and quads), the same, idle quads only, off
            ctx.debug_set_tune("FORK", fork)
            cfg = dict(kind=kind, n=n, w=w, h=h, spp=spp, bounces=bounces, nf=nf, accumulate=accumulate, count=count, batch=batch, seed=sd, quad=quad, fork=fork)
            kw = dict(mode=rt.PT_MODE_PATH, spp=spp, max_bounces=bounces, seed=sd)

            def params(f, **extra):
                return ctx.make_params(w, h, cams[f][0], cams[f][1], frame=f, accumulate=accumulate, **kw, **extra)

            # expected: one launch per frame with the one-pixel-per-lane kernel
            ctx.set_batch(1)
            ctx.render(ctx.make_params(w, h, mode=rt.PT_MODE_REFERENCE))       # ends any accumulating sequence
            want = []
            for f in range(nf):
                ctx.render(params(f, simple_kernel=True)); want.append(ctx.read_radiance().copy())
            ctx.render(ctx.make_params(w, h, mode=rt.PT_MODE_REFERENCE))
            # megakernel, batched
            ctx.set_batch(batch)
            if count == 1:
                if accumulate:
                    for f in range(nf):
                        ctx.render(params(f))
                    got = [None] * (nf - 1) + [ctx.read_radiance().copy()]       # the running result after the last frame
                else:
                    got = []
                    for f in range(nf):                                            # read-backs flush partial batches at random points
                        ctx.render(params(f))
                        if rng.random() < 0.4 or f == nf - 1:
                            got += [None] * (f - len(got)) + [ctx.read_radiance().copy()]
                for f in range(nf):
                    if got[f] is not None and not same(got[f], want[f]):
                        print("MISMATCH whole frame", cfg, "frame", f); sys.exit(1)
            else:
                stride = max(rt.tile_layout(w, h, r, count)[1] for r in range(count))
                bufs = C.c_void_p(); assert hip.hipMalloc(C.byref(bufs), C.c_size_t(stride * 4 * count * nf)) == 0
                assert hip.hipMemset(bufs, 0, C.c_size_t(stride * 4 * count * nf)) == 0
                for r in range(count):                                             # rank r's frames as one batched sequence
                    for f in range(nf):
                        ctx.set_compact_buffer(bufs.value + (f * count + r) * stride * 4, stride)
                        ctx.render(params(f, tile_rank=r, tile_count=count))
                    if rng.random() < 0.5: ctx.flush()                              # otherwise the next rank's first frame launches the open batch
                ctx.synchronize()
                ctx.set_compact_buffer(0, 0)
                for f in range(nf):
                    ctx.deinterleave(bufs.value + f * count * stride * 4, stride, w, h, count)
                    if not same(ctx.read_radiance(), want[f]):
                        print("MISMATCH sharded", cfg, "frame", f); sys.exit(1)
                hip.hipFree(bufs)
            ctx.set_batch(1)
            rounds += 1; frames += nf
            if time.time() - last_print > 20:
                print("soak: %d configurations, %d frames compared, all bit-identical" % (rounds, frames), flush=True); last_print = time.time()
    print("soak ok: %d configurations, %d frames compared (seed %d), all bit-identical" % (rounds, frames, seed))


if __name__ == "__main__":
    main()
