"""Scene.loadGLB against the reference's OWN loader on what it loads beyond the two bundled GLBs (VERDICT r5 "missing" 1): triangle strips and
fans, sparse accessors, .gltf JSON with an external .bin and with data: URIs, several buffers, integer (quantised) positions, several scenes,
multi-primitive meshes on nodes with children.  The files come from tests/golden/synth_gltf.js (deterministic); the expected triangles are the
bit patterns three's GLTFLoader.load + the reference's Scene.js statements produced for the same bytes
(tests/golden/gen_golden_glb.js -> tests/golden/gltf_synth_golden.json).  Bit for bit, raw and normalised."""
import json
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def test_synthetic_gltf_files_match_three_gltfloader_bit_for_bit(tmp_path):
    got = json.loads(subprocess.check_output([NODE, os.path.join(HERE, "js_gltf_synth_check.js"), str(tmp_path)], text=True, timeout=120))
    gold = json.load(open(os.path.join(HERE, "golden", "gltf_synth_golden.json")))
    assert set(got) == set(gold) and len(gold) >= 8
    assert {"strips.glb", "fans.glb", "sparse.glb", "external.gltf", "datauri.gltf", "quantized.glb"} <= set(gold)
    for f, g in gold.items():
        assert got[f]["numTris"] == g["numTris"] > 0, f
        assert got[f]["world_bits"] == g["world_f32_bits"], f
        assert got[f]["normalized_bits"] == g["normalized_cube_f32_bits"], f
    # the external .bin really is a separate file (the .gltf alone does not load)
    os.remove(os.path.join(str(tmp_path), "external_0.bin"))
    r = subprocess.run([NODE, os.path.join(HERE, "js_glb_dump.js"), os.path.join(str(tmp_path), "external.gltf"), os.path.join(str(tmp_path), "o")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "ENOENT" in r.stderr


def test_load_rejections_tell_absent_from_unreadable(tmp_path):
    """Scene.js:27-30 rejects on every failed load.  The drivers (js/main.js, js/main.mjs) fall back to the procedural stand-in only for a file that
    is not there (err.code === "ENOENT"); anything else propagates -- so the rejection has to carry the difference."""
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"glTF\x02\x00\x00\x00 definitely truncated")
    script = ("const {Scene}=require(%r);console.error=()=>{};const out={};"
              "new Scene().loadGLB(%r).then(()=>{out.a='loaded';},e=>{out.a=String(e.code);}).then(()=>new Scene().loadGLB(%r)).then(()=>{out.b='loaded';},e=>{out.b=String(e.code)+'|'+e.message;})"
              ".then(()=>console.log(JSON.stringify(out)));") % (os.path.join(ROOT, "raytracer-public_amd", "js", "Scene.js"), str(tmp_path / "absent.glb"), str(bad))
    out = json.loads(subprocess.check_output([NODE, "-e", script], text=True, timeout=60))
    assert out["a"] == "ENOENT"
    assert out["b"].startswith("undefined|") and "GLB" in out["b"]
    # and the drivers' catch lets only ENOENT through to the stand-in
    for drv in ("main.js", "main.mjs"):
        src = open(os.path.join(ROOT, "raytracer-public_amd", "js", drv)).read()
        assert 'if (!e || e.code !== "ENOENT") throw e;' in src, drv


def test_scene_sort_triangles_matches_the_references_order():
    """Scene.sortTriangles (Scene.js:169-224: a 30-bit Morton order of the centroids whose bit spreading starts from the UNtruncated coordinate x 1024):
    the permutation equals the one the reference's own method body produced for the same deterministic sets (tests/golden/gen_golden_scene_sort.js cuts
    the method out of the reference at generation time and stores only the resulting order) -- a zero-extent axis, tight clusters with many equal codes
    and exact duplicates (stability of the sort) included."""
    script = ("const {Scene}=require(%r);const {inputs}=require(%r);console.log=()=>{};const out={};"
              "for (const n of Object.keys(inputs)) { const s=new Scene(); s.triangles=inputs[n]().map((t,k)=>Object.assign(t,{id:k})); s.sortTriangles(); out[n]=s.triangles.map(t=>t.id); }"
              "process.stdout.write(JSON.stringify(out));") % (os.path.join(ROOT, "raytracer-public_amd", "js", "Scene.js"), os.path.join(HERE, "golden", "scene_sort_inputs.js"))
    got = json.loads(subprocess.check_output([NODE, "-e", script], text=True, timeout=60))
    gold = json.load(open(os.path.join(HERE, "golden", "scene_sort_golden.json")))
    assert set(got) == set(gold) and len(gold) >= 5
    for name, order in gold.items():
        assert got[name] == order, name
        assert sorted(order) == list(range(len(order)))
    assert gold["uniform_2000"] != list(range(2000))                     # the sort does something
