"""BASELINE's C4 and C2 through the ingest their wording names, at full size, through the Node boundary (VERDICT r5 "next" 1):

  C4  "Sponza GLB (~260k tris) via loadGLB(), 1920x1080, 4 spp, 8 bounces": a realistic 262,144-triangle GLB (tests/glb_writer.py: several meshes
      and primitives, TRS + matrix nodes, u16 and u32 index buffers, interleaved views) -> `node js/main.mjs --glb ...` = Scene.loadGLB
      (normalize: true, mode "cube", src/main.js:20-23) -> setScene -> BVH2 dump -> render.  The triangles the loader produced equal the independent
      numpy glTF rules' (f32 rounding), data/BVH2.bin equals the oracle's LBVH2 of them word for word, and the frame equals the oracle's on every pixel.
  C2  the same at the reference's own scene size: an 871,414-triangle GLB through loadGLB, from the reference's camera; and
      "Stanford Dragon (data/BVH2.bin)": the 871,414-triangle run's data/BVH2.bin, loaded by a FRESH process through setBVH2 (file -> collapse ->
      render), gives the build path's frame bit for bit (and the oracle's on a pixel grid).

  C3 / C5  as worded through js/main.js with eight member contexts: BVH4_wide + 16 spp; 4K, 64 spp accumulated over 16 frames, 16 bounces.

Plus the failure behaviour of the drivers: a GLB that exists and cannot be read ends the process non-zero; only an absent one falls back."""
import os
import re
import shutil
import subprocess
import sys
import time

import numpy as np
import pytest

import orc as orc_mod
from scenes import quat_yaw_pitch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NODE = shutil.which("node")
MAIN_MJS = os.path.join(ROOT, "raytracer-public_amd", "js", "main.mjs")
MAIN_JS = os.path.join(ROOT, "raytracer-public_amd", "js", "main.js")
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(NODE is None, reason="node is not installed")]


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def run_node(args, cwd, timeout=600):
    r = subprocess.run([NODE] + args, capture_output=True, text=True, timeout=timeout, cwd=cwd)
    return r.returncode, r.stdout, r.stderr


def test_c4_sponza_class_glb_through_loadglb_full_size(tmp_path, rt, orc):
    from glb_writer import write_realistic_glb, normalize_cube
    n = 262144
    soup = rt.procedural_scene(rt.SCENE_SPONZA_CLASS, n)
    glb = str(tmp_path / "sponza_class.glb")
    expect, stats = write_realistic_glb(glb, soup, seed=4)
    assert stats["u16"] >= 2 and stats["u32"] >= 1 and stats["interleaved"] >= 2 and stats["nonindexed"] == 1
    want_tris, center, scale = normalize_cube(expect)                 # what loadGLB(..., {normalize: true, mode: "cube"}) must hand to setScene
    cam = tuple(float(np.float32(v)) for v in (np.array([0.55, -0.05, 0.05]) - center) * scale)      # the C4 camera, moved with the mesh
    quat = tuple(float(np.float32(v)) for v in quat_yaw_pitch(1.45, 0.05))
    w, h = 1920, 1080
    rc, out, err = run_node([MAIN_MJS, "--glb", glb, "--mode", "2", "--spp", "4", "--bounces", "8", "--seed", "3", "--frames", "1", "--width", str(w), "--height", str(h),
                             "--cam", ",".join(repr(v) for v in cam), "--quat", ",".join(repr(v) for v in quat),
                             "--dump", str(tmp_path / "data" / "BVH2.bin"), "--radiance", str(tmp_path / "img.f32"), "--triangles", str(tmp_path / "tris.f32")], str(tmp_path))
    assert rc == 0, err
    m = re.search(r"Loaded .* -> (\d+) triangles in (\d+) ms", out)
    assert m and int(m.group(1)) == n, out
    print("Scene.loadGLB: %d triangles (%d-byte GLB) in %s ms" % (n, os.path.getsize(glb), m.group(2)))
    tris = np.fromfile(str(tmp_path / "tris.f32"), np.float32)
    assert tris.size == n * 9
    assert np.allclose(tris.reshape(-1, 3, 3), want_tris.astype(np.float32), rtol=2e-6, atol=2e-6)
    # data/BVH2.bin is the oracle's LBVH2 of those triangles, word for word
    morton, tri_index = rt.morton_sort(tris)
    bvh2 = np.fromfile(str(tmp_path / "data" / "BVH2.bin"), np.uint32)
    assert np.array_equal(bvh2, orc.build_lbvh2(tris, morton, tri_index))
    bvh4, _ = orc.collapse_bvh4(bvh2, n)
    # the frame main.mjs rendered last (setFrameCount(1)): every pixel against the oracle
    ref, ost = orc.render_mt(orc.make_params(w, h, n, cam, quat, mode=orc_mod.MODE_PATH, spp=4, max_bounces=8, seed=3, frame=1), tris, bvh4)
    img = np.fromfile(str(tmp_path / "img.f32"), np.float32).reshape(h, w, 4)
    assert same_bits(img, ref)
    assert ost["rays_closest"] > 4 * ost["samples"]                   # an interior: long paths, nearly every camera ray hits


def test_c2_dragon_class_glb_through_loadglb_full_size(tmp_path, rt, orc):
    """src/main.js:18-46 as written, at the size of the reference's own scene: an 871,414-triangle GLB (the dragon-class mesh through tests/glb_writer.py:
    14.6 MB, u16 and u32 index buffers, interleaved views, TRS / matrix / mirrored nodes) -> loadGLB(normalize: true, mode "cube") -> setScene -> BVH2
    dump -> render from the reference's camera (0, 0, 2.5).  Triangles against the numpy glTF rules, data/BVH2.bin against the oracle's LBVH2 word for
    word, the 1080p / 4 spp / 8-bounce frame against the oracle on every 4th pixel in x and y (all samples)."""
    from glb_writer import write_realistic_glb, normalize_cube
    n = 871414
    soup = rt.procedural_scene(rt.SCENE_DRAGON_CLASS, n)
    glb = str(tmp_path / "dragon_class.glb")
    expect, stats = write_realistic_glb(glb, soup, seed=7)
    assert stats["u32"] >= 1 and stats["u16"] >= 2
    want_tris, _, _ = normalize_cube(expect)
    w, h = 1920, 1080
    rc, out, err = run_node([MAIN_MJS, "--glb", glb, "--mode", "2", "--spp", "4", "--bounces", "8", "--seed", "1", "--frames", "1", "--width", str(w), "--height", str(h),
                             "--dump", str(tmp_path / "data" / "BVH2.bin"), "--radiance", str(tmp_path / "img.f32"), "--triangles", str(tmp_path / "tris.f32")], str(tmp_path))
    assert rc == 0, err
    m = re.search(r"Loaded .* -> (\d+) triangles in (\d+) ms", out)
    assert m and int(m.group(1)) == n, out
    print("Scene.loadGLB: %d triangles (%d-byte GLB) in %s ms" % (n, os.path.getsize(glb), m.group(2)))
    tris = np.fromfile(str(tmp_path / "tris.f32"), np.float32)
    assert tris.size == n * 9
    assert np.allclose(tris.reshape(-1, 3, 3), want_tris.astype(np.float32), rtol=2e-6, atol=2e-6)
    morton, tri_index = rt.morton_sort(tris)
    bvh2 = np.fromfile(str(tmp_path / "data" / "BVH2.bin"), np.uint32)
    assert bvh2.size == 1 + 6 * (2 * n - 1)
    assert np.array_equal(bvh2, orc.build_lbvh2(tris, morton, tri_index))
    bvh4, _ = orc.collapse_bvh4(bvh2, n)
    ref, _, _ = orc.render(orc.make_params(w, h, n, mode=orc_mod.MODE_PATH, spp=4, max_bounces=8, seed=1, frame=1, step=(4, 4)), tris, bvh4)
    img = np.fromfile(str(tmp_path / "img.f32"), np.float32).reshape(h, w, 4)
    assert same_bits(img[::4, ::4], ref[::4, ::4])
    assert (img[..., 0] > 0.011).mean() > 0.08                        # the mesh is in front of the reference's camera


def test_c2_bvh2_bin_reloaded_by_a_fresh_process(tmp_path, rt, orc):
    n = 871414
    w, h = 1920, 1080
    common = ["--mode", "2", "--spp", "4", "--bounces", "8", "--seed", "1", "--frames", "2", "--width", str(w), "--height", str(h)]
    dump = str(tmp_path / "data" / "BVH2.bin")
    # run 1: the driver's own sequence (no dragon.glb: the procedural stand-in of the same triangle budget), data/BVH2.bin written, frame kept
    rc, out, err = run_node([MAIN_MJS] + common + ["--dump", dump, "--radiance", str(tmp_path / "built.f32"), "--triangles", str(tmp_path / "tris.f32")], str(tmp_path))
    assert rc == 0, err
    assert "GLB not available" in out and "BVH2 dump complete" in out
    assert os.path.getsize(dump) == 4 * (1 + 6 * (2 * n - 1))          # PathTracer.js:227: 1 + 6 (2N - 1) words
    # run 2: a fresh process installs the FILE through setBVH2 (collapse, then the renderer's BVH4) instead of building
    t0 = time.time()
    rc, out2, err = run_node([MAIN_MJS] + common + ["--bvh2", dump, "--dump", str(tmp_path / "again" / "BVH2.bin"), "--radiance", str(tmp_path / "loaded.f32")], str(tmp_path))
    assert rc == 0, err
    assert "Installed prebuilt BVH2" in out2 and "BVH Build Time" not in out2
    print("fresh process, data/BVH2.bin -> setBVH2 -> 2 frames: %.1f s wall" % (time.time() - t0))
    built = np.fromfile(str(tmp_path / "built.f32"), np.float32).reshape(h, w, 4)
    loaded = np.fromfile(str(tmp_path / "loaded.f32"), np.float32).reshape(h, w, 4)
    assert same_bits(built, loaded)
    # the file survives the round trip through the second process unchanged (readBVH2 after setBVH2)
    assert open(dump, "rb").read() == open(str(tmp_path / "again" / "BVH2.bin"), "rb").read()
    # and both are the oracle's frame (every 4th pixel in x and y, all samples) over the BVH4 collapsed from the FILE
    tris = np.fromfile(str(tmp_path / "tris.f32"), np.float32)
    bvh4, _ = orc.collapse_bvh4(np.fromfile(dump, np.uint32), n)
    ref, _, _ = orc.render(orc.make_params(w, h, n, mode=orc_mod.MODE_PATH, spp=4, max_bounces=8, seed=1, frame=2, step=(4, 4)), tris, bvh4)
    assert same_bits(loaded[::4, ::4], ref[::4, ::4])


EIGHT = ",".join(["0"] * 8)          # eight member contexts on the one GPU (copy transport: RCCL refuses members that share a device)


def test_c3_through_node_wide_bvh_sixteen_spp_eight_members(tmp_path, rt, orc):
    """C3 as worded -- "Stanford Dragon BVH4_wide, 1920x1080, 16 spp, 8 bounces, pixel-tiled across 8 GPUs" -- through the Node driver: the BVH2 of the
    871,414-triangle scene is promoted to BVH4_wide (the reference's bin/test), written as data/BVH4_wide.bin, installed with setBVH4 and traversed by a
    group of EIGHT member contexts (interleaved 8x8 tiles, one gather per frame; here on one GPU through the copy transport).  The file equals the oracle's
    promotion (itself pinned against the compiled reference), the frame equals the oracle traversing that wide buffer on every 4th pixel in x and y."""
    n, w, h = 871414, 1920, 1080
    rc, out, err = run_node([MAIN_JS, "--devices", EIGHT, "--transport", "copy", "--mode", "2", "--spp", "16", "--bounces", "8", "--seed", "1", "--frames", "1",
                             "--width", str(w), "--height", str(h), "--bvh4-wide", "--dump-wide", str(tmp_path / "data" / "BVH4_wide.bin"),
                             "--dump", str(tmp_path / "data" / "BVH2.bin"), "--radiance", str(tmp_path / "img.f32"), "--triangles", str(tmp_path / "tris.f32")], str(tmp_path))
    assert rc == 0, err
    assert "Rendering on 8 GPUs" in out and "Traversing BVH4_wide: %d nodes" % (2 * n - 1) in out
    tris = np.fromfile(str(tmp_path / "tris.f32"), np.float32)
    bvh2 = np.fromfile(str(tmp_path / "data" / "BVH2.bin"), np.uint32)
    wide = np.fromfile(str(tmp_path / "data" / "BVH4_wide.bin"), np.uint32)
    assert wide[0] == 2 * n - 1 and np.array_equal(wide, orc.bvh4_wide(bvh2))
    ref, _, _ = orc.render(orc.make_params(w, h, n, mode=orc_mod.MODE_PATH, spp=16, max_bounces=8, seed=1, frame=1, step=(4, 4)), tris, wide)
    img = np.fromfile(str(tmp_path / "img.f32"), np.float32).reshape(h, w, 4)
    assert same_bits(img[::4, ::4], ref[::4, ::4])


def test_c5_through_node_4k_progressive_sixteen_bounces_eight_members(tmp_path, rt, orc):
    """C5 as worded -- "3840x2160, 64 spp progressive accumulate, 16 bounces, 8 GPUs" -- through the Node driver: 16 accumulated frames of 4 spp in one
    batch per member, eight member contexts (each keeps the running sums of its own tiles; the gather happens when the image is asked for).  The image
    equals the oracle's 16-frame accumulation on every 8th pixel in x and y (all 64 samples each)."""
    n, w, h, frames = 871414, 3840, 2160, 16
    rc, out, err = run_node([MAIN_JS, "--devices", EIGHT, "--transport", "copy", "--mode", "2", "--spp", "4", "--bounces", "16", "--seed", "1", "--frames", str(frames), "--batch", str(frames),
                             "--accumulate", "--width", str(w), "--height", str(h), "--dump", str(tmp_path / "data" / "BVH2.bin"),
                             "--radiance", str(tmp_path / "img.f32"), "--triangles", str(tmp_path / "tris.f32")], str(tmp_path))
    assert rc == 0, err
    tris = np.fromfile(str(tmp_path / "tris.f32"), np.float32)
    bvh4, _ = orc.collapse_bvh4(np.fromfile(str(tmp_path / "data" / "BVH2.bin"), np.uint32), n)
    ref, _ = orc.render_mt(orc.make_params(w, h, n, mode=orc_mod.MODE_PATH, spp=4, max_bounces=16, seed=1, frame=1, accum_frames=frames, step=(8, 8)), tris, bvh4)
    img = np.fromfile(str(tmp_path / "img.f32"), np.float32).reshape(h, w, 4)
    assert same_bits(img[::8, ::8], ref[::8, ::8])
    assert np.isfinite(img).all() and (img[..., 0] > 0.011).mean() > 0.08


@pytest.mark.parametrize("driver", [MAIN_MJS, MAIN_JS])
def test_unreadable_glb_stops_the_driver_absent_glb_falls_back(tmp_path, driver):
    """src/main.js:20-23 has no catch and Scene.js:27-30 rejects: a GLB that cannot be read ends the app.  Here too -- non-zero status, the loader's
    message, no frame rendered; a file that is simply not there is the one case that falls back to the stand-in."""
    bad = tmp_path / "broken.glb"
    bad.write_bytes(b"glTF\x02\x00\x00\x00" + b"\x00" * 40)
    small = ["--width", "160", "--height", "96", "--frames", "2", "--tris", "5000", "--dump", str(tmp_path / "d" / "BVH2.bin")]
    rc, out, err = run_node([driver, "--glb", str(bad)] + small, str(tmp_path), timeout=120)
    assert rc != 0 and "GLB" in err and "FPS" not in out and "stand-in" not in out
    rc, out, err = run_node([driver, "--glb", str(tmp_path / "absent.glb")] + small, str(tmp_path), timeout=120)
    assert rc == 0, err
    assert "GLB not available" in out and "FPS" in out
    # a .gltf whose external buffer is missing is an unreadable file, not an absent one ... its error is ENOENT too, but for ANOTHER path:
    # the stand-in must not silently replace a scene whose .gltf is there
    import json
    gltf = tmp_path / "scene.gltf"
    gltf.write_text(json.dumps({"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}], "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}],
                                "buffers": [{"byteLength": 36, "uri": "gone.bin"}], "bufferViews": [{"buffer": 0, "byteLength": 36}],
                                "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}]}))
    rc, out, err = run_node([driver, "--glb", str(gltf)] + small, str(tmp_path), timeout=120)
    assert rc != 0 and "stand-in" not in out


def test_present_step_through_the_driver_ppm_equals_the_tonemapper(tmp_path, rt, orc):
    """NEXT-4 through the driver: `js/main.js --out frame.ppm` writes what the reference's tonemapper pass puts on the canvas (tonemapper.wgsl:24-41: Reinhard,
    gamma 1/2.2, vertical flip, applied to the rgba8unorm output texture) -- equal to the oracle's tonemapper over the oracle's frame, byte for byte."""
    w, h, n = 320, 180, 20000
    ppm = tmp_path / "frame.ppm"
    rc, out, err = run_node([MAIN_JS, "--tris", str(n), "--mode", "1", "--frames", "2", "--width", str(w), "--height", str(h), "--dump", str(tmp_path / "d" / "BVH2.bin"),
                             "--out", str(ppm), "--radiance", str(tmp_path / "img.f32"), "--triangles", str(tmp_path / "tris.f32")], str(tmp_path), timeout=120)
    assert rc == 0, err
    raw = ppm.read_bytes()
    head = ("P6\n%d %d\n255\n" % (w, h)).encode()
    assert raw.startswith(head) and len(raw) == len(head) + w * h * 3
    got = np.frombuffer(raw[len(head):], np.uint8).reshape(h, w, 3)
    tris = np.fromfile(str(tmp_path / "tris.f32"), np.float32)
    bvh4, _ = orc.collapse_bvh4(np.fromfile(str(tmp_path / "d" / "BVH2.bin"), np.uint32), n)
    ref, _, _ = orc.render(orc.make_params(w, h, n, mode=orc_mod.MODE_SINGLE, frame=2), tris, bvh4)
    assert same_bits(np.fromfile(str(tmp_path / "img.f32"), np.float32).reshape(h, w, 4), ref)
    assert np.array_equal(got, orc.tonemap(ref, quantize=True)[..., :3])
