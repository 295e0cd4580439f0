"""The single-thread Node/JS form of the oracle (oracle/js/pt_oracle.js) against the C++ oracle:
bit-identical images and identical traversal counters."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import orc as orc_mod
from scenes import TETRA, random_soup, quat_yaw_pitch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NODE = shutil.which("node")
JS = os.path.join(ROOT, "oracle", "js", "pt_oracle.js")

pytestmark = pytest.mark.skipif(NODE is None, reason="node missing")


def run_js(tmp_path, tris, bvh4, w, h, cam, quat, mode, spp=1, bounces=0, seed=1, frame=0, step=(1, 1)):
    focal, aspect = orc_mod.focal_aspect(w, h)
    tris.astype(np.float32).tofile(str(tmp_path / "t.f32")); bvh4.astype(np.uint32).tofile(str(tmp_path / "b.u32"))
    P = dict(width=w, height=h, focal=float(focal), aspect=float(aspect), camPos=list(map(float, cam)), camQuat=list(map(float, quat)),
             frame=frame, mode=mode, spp=spp, maxBounces=bounces, seed=seed, numTris=tris.size // 9, stepX=step[0], stepY=step[1])
    (tmp_path / "p.json").write_text(json.dumps(P))
    info = json.loads(subprocess.check_output([NODE, JS, str(tmp_path / "t.f32"), str(tmp_path / "b.u32"), str(tmp_path / "p.json"), str(tmp_path / "o.f32")], text=True))
    return np.fromfile(str(tmp_path / "o.f32"), np.float32).reshape(h, w, 4), info


@pytest.mark.parametrize("kind", ["tetra", "soup"])
def test_js_oracle_matches_cpp_oracle(orc, tmp_path, kind):
    tris = TETRA if kind == "tetra" else random_soup(1500, 21)
    _, bvh4 = orc.build_bvh4(tris)
    n = tris.size // 9
    cam, quat = ((0.2, 0.1, 2.3), quat_yaw_pitch(0.1, -0.05))
    w, h = 96, 60
    img, _ = run_js(tmp_path, tris, bvh4, w, h, cam, quat, 1)
    ref, _, _ = orc.render(orc.make_params(w, h, n, cam, quat, mode=orc_mod.MODE_SINGLE), tris, bvh4)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    img, info = run_js(tmp_path, tris, bvh4, w, h, cam, quat, 2, spp=3, bounces=5, seed=9, frame=2)
    ref, _, st = orc.render(orc.make_params(w, h, n, cam, quat, mode=orc_mod.MODE_PATH, spp=3, max_bounces=5, seed=9, frame=2), tris, bvh4)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    js = info["stats"]
    assert (js["raysClosest"], js["raysShadow"], js["nodesExamined"], js["trisTested"], js["samples"]) == (
        st["rays_closest"], st["rays_shadow"], st["nodes_examined"], st["tris_tested"], st["samples"])


def test_js_fmaf_is_correctly_rounded():
    # the JS fmaf repairs double rounding: check against numpy float64->float32 of the exact value on hard cases
    script = "const {fmaf}=require(%r);const c=JSON.parse(process.argv[1]);console.log(JSON.stringify(c.map(x=>fmaf(x[0],x[1],x[2]))));" % JS
    rng = np.random.default_rng(3)
    cases = []
    for _ in range(2000):
        a, b = np.float32(rng.normal()), np.float32(rng.normal())
        p = float(a) * float(b)                                    # exact in double
        target = np.float32(rng.normal())
        # craft c so that a*b + c lands next to a float32 rounding boundary
        mid = (float(target) + float(np.nextafter(target, np.float32(np.inf)))) / 2
        c = np.float32(mid - p)
        cases.append([float(a), float(b), float(c)])
    got = json.loads(subprocess.check_output([NODE, "-e", script, json.dumps(cases)], text=True))
    from fractions import Fraction
    for (a, b, c), g in zip(cases, got):
        exact = Fraction(a) * Fraction(b) + Fraction(c)
        lo = np.float32(float(exact))
        cand = sorted({float(lo), float(np.nextafter(lo, np.float32(-np.inf))), float(np.nextafter(lo, np.float32(np.inf)))})
        best = min(cand, key=lambda v: (abs(Fraction(v) - exact), int(np.float32(v).view(np.uint32)) & 1))
        assert g == best, (a, b, c, g, best)
