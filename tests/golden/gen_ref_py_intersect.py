#!/usr/bin/env python3
"""Golden vectors for the slab test and Moller-Trumbore from code the REFERENCE holds: tests/test.py:64-99.

Runs in the build container only (the reference checkout is absent on the GPU box).  tests/test.py cannot be imported
(its module level needs pygltflib and the absent data files), so the three functions are taken out of its source with
`ast` -- `unpack2x16float` (:20-27), `intersect_aabb` (:64-76), `intersect_triangle` (:82-99) -- and compiled on their
own: no stub for the missing module, nothing else of the file runs.  Their free names are `np`, the constant `INF`
(the file's own `np.float32(1e30)`, :5, evaluated from its AST node) and the counter `NODES_INTERSECTED`.

The functions are fed seeded f32 inputs; inputs and outputs go to tests/golden/ref_py_intersect.json as u32 bit
patterns (data only: no text of the reference is stored).  tests/test_ref_py_intersect.py checks the CPU oracle's
single-lane probes (orc_slab, orc_moller_trumbore, orc_f16_to_f32) against them.

usage: python3 tests/golden/gen_ref_py_intersect.py [/root/reference]
"""
import ast
import json
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(REF, "tests", "test.py")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_py_intersect.json")
WANT = ("unpack2x16float", "intersect_aabb", "intersect_triangle")


def load_reference_functions():
    tree = ast.parse(open(SRC, "r", encoding="utf-8").read(), SRC)
    ns = {"np": np, "NODES_INTERSECTED": 0}
    picked = []
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and node.targets[0].id == "INF":
            ns["INF"] = eval(compile(ast.Expression(node.value), SRC, "eval"), {"np": np})
        if isinstance(node, ast.FunctionDef) and node.name in WANT:
            picked.append(node)
    assert "INF" in ns and sorted(n.name for n in picked) == sorted(WANT), "tests/test.py no longer holds the expected definitions"
    exec(compile(ast.Module(body=picked, type_ignores=[]), SRC, "exec"), ns)
    return ns


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32).ravel().tolist()


def f16_grid(rng, n, lo, hi):
    """f32 values that are exactly representable in f16 (what a decoded BVH bound is)."""
    return rng.uniform(lo, hi, n).astype(np.float16).astype(np.float32)


def safe_inv(d):
    """renderer.wgsl:74-80 / tests/test.py:151-154 in f32."""
    out = np.empty(3, np.float32)
    for i in range(3):
        out[i] = np.float32(1.0) / d[i] if abs(d[i]) > 1e-8 else np.float32(1e30)
    return out


def main():
    ns = load_reference_functions()
    aabb, tri, unpack, INF = ns["intersect_aabb"], ns["intersect_triangle"], ns["unpack2x16float"], ns["INF"]
    rng = np.random.default_rng(20260109)

    # ---- slab test: f16-representable boxes in [-1, 1]^3, rays from outside / inside / along an axis -------------------------
    slab = {"o": [], "inv": [], "mn": [], "mx": [], "ret": []}
    for k in range(1500):
        a, b = f16_grid(rng, 3, -1.0, 1.0), f16_grid(rng, 3, -1.0, 1.0)
        mn, mx = np.minimum(a, b), np.maximum(a, b)
        kind = k % 6
        if kind == 0:    # the reference's camera position, aimed somewhere near the box
            o = np.array([0.0, 0.0, 2.5], np.float32)
            d = ((mn + mx) * 0.5 + rng.normal(0, 0.3, 3).astype(np.float32)) - o
        elif kind == 1:  # origin inside the box
            o = (mn + (mx - mn) * rng.uniform(0, 1, 3).astype(np.float32)).astype(np.float32)
            d = rng.normal(0, 1, 3)
        elif kind == 2:  # axis-parallel ray (two zero direction components: the 1e30 reciprocal)
            o = rng.uniform(-2, 2, 3).astype(np.float32)
            d = np.zeros(3); d[rng.integers(3)] = rng.choice([-1.0, 1.0])
        elif kind == 3:  # origin on a face plane of the box
            o = rng.uniform(-2, 2, 3).astype(np.float32); ax = rng.integers(3); o[ax] = mn[ax] if rng.integers(2) else mx[ax]
            d = rng.normal(0, 1, 3)
        else:            # anything
            o = rng.uniform(-3, 3, 3).astype(np.float32)
            d = rng.normal(0, 1, 3)
        d = np.asarray(d, np.float32)
        n = np.float32(np.sqrt(np.float32((d * d).sum())))
        d = (d / n).astype(np.float32) if n > 0 else np.array([0, 0, -1], np.float32)
        inv = safe_inv(d)
        r = aabb(o.astype(np.float32), inv, mn, mx)
        slab["o"] += bits(o); slab["inv"] += bits(inv); slab["mn"] += bits(mn); slab["mx"] += bits(mx); slab["ret"] += bits(np.float32(r))

    # ---- Moller-Trumbore: rays aimed at / near / away from random triangles; parallel and behind-the-origin cases ------------
    mt = {"o": [], "d": [], "v0": [], "v1": [], "v2": [], "ret": []}
    for k in range(1500):
        scale = np.float32(10.0 ** rng.uniform(-2.5, 0.0))
        c = rng.uniform(-1, 1, 3).astype(np.float32)
        v0, v1, v2 = (c + rng.normal(0, 1, 3).astype(np.float32) * scale for _ in range(3))
        v0, v1, v2 = v0.astype(np.float32), v1.astype(np.float32), v2.astype(np.float32)
        o = np.array([0.0, 0.0, 2.5], np.float32) if k % 3 == 0 else rng.uniform(-2, 2, 3).astype(np.float32)
        kind = k % 5
        if kind in (0, 1):   # through a point of the triangle's plane, inside or just outside
            w = rng.uniform(-0.2, 1.0, 2)
            target = v0 + (v1 - v0) * np.float32(w[0]) + (v2 - v0) * np.float32(w[1]) * np.float32(1.0 if kind else 0.6)
            d = target - o
        elif kind == 2:      # away from it
            d = o - (v0 + v1 + v2) / np.float32(3.0)
        elif kind == 3:      # in the triangle's plane (det ~ 0)
            d = (v1 - v0) * np.float32(rng.uniform(-1, 1)) + (v2 - v0) * np.float32(rng.uniform(-1, 1))
        else:
            d = rng.normal(0, 1, 3)
        d = np.asarray(d, np.float32)
        n = np.float32(np.sqrt(np.float32((d * d).sum())))
        d = (d / n).astype(np.float32) if n > 0 else np.array([0, 0, -1], np.float32)
        r = tri(o, d, v0, v1, v2)
        mt["o"] += bits(o); mt["d"] += bits(d); mt["v0"] += bits(v0); mt["v1"] += bits(v1); mt["v2"] += bits(v2); mt["ret"] += bits(np.float32(r))

    # ---- f16 pair decode (getBVHNode4's unpack2x16float) ------------------------------------------------------------------------
    words = rng.integers(0, 2 ** 32, 1000, dtype=np.uint64).astype(np.uint32)
    words[:8] = [0, 0x80000000, 0x3C00BC00, 0x7BFF0001, 0x03FF8400, 0xFBFF7BFF, 0x00010001, 0x7C00FC00]
    lo_hi = []
    for w in words:
        f0, f1 = unpack(int(w))
        lo_hi += bits(np.array([f0, f1], np.float32))

    json.dump({"source": "tests/test.py:20-27, 64-76, 82-99 of the reference, extracted with ast and run under numpy %s" % np.__version__,
               "inf_bits": bits(np.float32(INF))[0], "slab": slab, "moller_trumbore": mt,
               "unpack2x16float": {"word": [int(w) for w in words], "lo_hi": lo_hi}},
              open(OUT, "w"), separators=(",", ":"))
    hits_s = sum(1 for r in slab["ret"] if r != bits(np.float32(INF))[0])
    hits_t = sum(1 for r in mt["ret"] if r != bits(np.float32(INF))[0])
    print("wrote %s: %d slab cases (%d hits), %d triangle cases (%d hits), %d f16 words" % (OUT, len(slab["ret"]), hits_s, len(mt["ret"]), hits_t, len(words)))


if __name__ == "__main__":
    main()
