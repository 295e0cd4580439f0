"""The Node host side (raytracer-public_amd/js + the N-API addon over the C ABI)."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NODE = shutil.which("node")
ADDON = os.path.join(ROOT, "raytracer-public_amd", "napi", "mi355pt.node")

pytestmark = pytest.mark.skipif(NODE is None or not os.path.exists(ADDON), reason="node or the built addon is missing")


def test_addon_host_functions_match_reference_vectors():
    # same golden vectors as tests/test_host_build.py, through PathTracer.js -> N-API -> C ABI
    out = subprocess.check_output([NODE, os.path.join(HERE, "js_host_check.js"), os.path.join(HERE, "golden", "pathtracer_js_golden.json")], text=True)
    assert out.startswith("ok ")


def test_esm_face_runs_the_integration_sequence():
    """The reference's driver binds with ES-module imports (src/main.js:1-2).  libs/PathTracer.js and libs/Scene.js (ES modules by
    libs/package.json; .mjs aliases next to them) are that face: the INTEGRATION.md section-1 sequence up to buildMortonAndSort /
    collapseLBVH2ToBVH4 runs through them under Node, against the reference-produced golden vectors."""
    out = subprocess.check_output([NODE, os.path.join(HERE, "js_esm_check.mjs"), os.path.join(HERE, "golden", "pathtracer_js_golden.json")], text=True)
    assert out.startswith("ok ")
    # the ESM driver starts with the reference's two import lines as written
    head = open(os.path.join(ROOT, "raytracer-public_amd", "js", "main.mjs")).read()
    assert 'import * as PT from "./libs/PathTracer.js";\nimport * as PTScene from "./libs/Scene.js";' in head


@pytest.mark.gpu
def test_esm_face_renders_on_the_gpu(tmp_path, orc):
    """src/main.js's sequence through the ES-module imports up to render(): the frame equals the oracle's bit for bit."""
    import orc as orc_mod
    out = subprocess.check_output([NODE, os.path.join(HERE, "js_esm_render.mjs"), os.path.join(HERE, "golden", "steve.glb"), str(tmp_path)], text=True)
    assert out.strip().splitlines()[-1] == "ok 72"
    tris = np.fromfile(str(tmp_path / "tris.bin"), np.float32)
    bvh4, _ = orc.collapse_bvh4(orc.build_lbvh2(tris), 72)
    ref, _, _ = orc.render(orc.make_params(160, 96, 72, (0.3, 0.2, 2.5), (0, 0, 0, 1), mode=orc_mod.MODE_SINGLE, frame=3), tris, bvh4)
    img = np.fromfile(str(tmp_path / "img.bin"), np.float32).reshape(96, 160, 4)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))


def test_scene_js_matches_three_gltfloader():
    # Scene.loadGLB / parseGLTF / normalizeMesh / getTrianglesFloat32 vs the numbers three's GLTFLoader +
    # the reference's Scene.js arithmetic produced for the reference's two bundled GLBs
    got = json.loads(subprocess.check_output([NODE, os.path.join(HERE, "js_scene_check.js")], text=True))
    gold = json.load(open(os.path.join(HERE, "golden", "glb_golden.json")))
    assert set(got) == set(gold) == {"dodecahedron.glb", "steve.glb"}
    for f, g in gold.items():
        assert got[f]["numTris"] == g["numTris"]
        assert got[f]["world_bits"] == g["world_f32_bits"]
        assert got[f]["normalized_bits"] == g["normalized_cube_f32_bits"]
        assert got[f]["first"] == g["first_world_vertex_f64"]          # even the f64 intermediates agree


def test_scene_js_rejects_garbage(tmp_path):
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"not a glb at all, definitely")
    script = "const {Scene}=require(%r);const l=console.error;console.error=()=>{};new Scene().loadGLB(%r).then(()=>process.exit(1),e=>{console.log('rejected:'+e.message);});" % (
        os.path.join(ROOT, "raytracer-public_amd", "js", "Scene.js"), str(bad))
    out = subprocess.check_output([NODE, "-e", script], text=True)
    assert "rejected:" in out and "GLB" in out


@pytest.mark.gpu
def test_node_driver_end_to_end(tmp_path, orc):
    """src/main.js call sequence under Node on the GPU: GLB -> setScene -> BVH2 dump -> render; the dumped
    BVH2 and the rendered reference-mode frame equal the oracle's bit for bit."""
    import orc as orc_mod
    glb = os.path.join(HERE, "golden", "steve.glb")
    dump = str(tmp_path / "data" / "BVH2.bin")
    script = r"""
const path=require('path'); const PT=require(%r); const {Scene}=require(%r);
(async()=>{ const log=console.log; console.log=()=>{};
 const pt=new PT.PathTracer({width:160,height:96},{mode:PT.MODE_REFERENCE});
 await pt.initialize(); const s=new Scene(); await s.loadGLB(%r,{normalize:true,mode:'cube'}); await pt.setScene(s);
 const n=(pt.trianglesData.length/9)|0; const b2=await pt.readBVH2(pt.computeBVH2Sizing(n).bytes);
 PT.native().writeU32File(%r,b2);
 pt.setCameraPosition(0.3,0.2,2.5); pt.setCameraQuaternion(0,0,0,1); pt.setFrameCount(7); await pt.render();
 const img=pt.readRadiance(); const px=pt.readRGBA8();
 require('fs').writeFileSync(%r, Buffer.from(img.buffer)); require('fs').writeFileSync(%r, Buffer.from(pt.trianglesData.buffer));
 pt.options.mode=PT.MODE_PATH; pt.options.spp=2; pt.options.maxBounces=3; await pt.render(); const img2=pt.readRadiance();
 require('fs').writeFileSync(%r, Buffer.from(img2.buffer));
 log(JSON.stringify({n:n, rgba0:[px[0],px[1],px[2],px[3]]})); pt.destroy(); })().catch(e=>{console.error(e);process.exit(1);});
""" % (os.path.join(ROOT, "raytracer-public_amd", "js", "PathTracer.js"), os.path.join(ROOT, "raytracer-public_amd", "js", "Scene.js"),
       glb, dump, str(tmp_path / "img.bin"), str(tmp_path / "tris.bin"), str(tmp_path / "img2.bin"))
    os.makedirs(os.path.dirname(dump), exist_ok=True)
    info = json.loads(subprocess.check_output([NODE, "-e", script], text=True).strip().splitlines()[-1])
    tris = np.fromfile(str(tmp_path / "tris.bin"), np.float32)
    assert info["n"] == 72 and tris.size == 72 * 9
    bvh2 = orc.build_lbvh2(tris)
    assert np.array_equal(np.fromfile(dump, np.uint32), bvh2)                  # data/BVH2.bin == oracle LBVH2
    bvh4, _ = orc.collapse_bvh4(bvh2, 72)
    ref, _, _ = orc.render(orc.make_params(160, 96, 72, (0.3, 0.2, 2.5), (0, 0, 0, 1), mode=orc_mod.MODE_SINGLE, frame=7), tris, bvh4)
    img = np.fromfile(str(tmp_path / "img.bin"), np.float32).reshape(96, 160, 4)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    ref2, _, _ = orc.render(orc.make_params(160, 96, 72, (0.3, 0.2, 2.5), (0, 0, 0, 1), mode=orc_mod.MODE_PATH, spp=2, max_bounces=3, seed=1, frame=7), tris, bvh4)
    img2 = np.fromfile(str(tmp_path / "img2.bin"), np.float32).reshape(96, 160, 4)
    assert np.array_equal(img2.view(np.uint32), ref2.view(np.uint32))
    assert info["rgba0"][3] == 255


@pytest.mark.gpu
def test_node_group_one_image_from_three_members(tmp_path, orc):
    """js/PathTracer.js with `devices: [0,0,0]` drives a pt_group (three member contexts sharing cuda:0 through the copy transport --
    RCCL refuses members on one GPU): same call sequence, one image per render(), bit-identical to the oracle; js/main.js --devices too."""
    import orc as orc_mod
    glb = os.path.join(HERE, "golden", "dodecahedron.glb")
    script = r"""
const PT=require(%r); const {Scene}=require(%r);
(async()=>{ const log=console.log; console.log=()=>{};
 const pt=new PT.PathTracer({width:144,height:88},{mode:PT.MODE_PATH,spp:3,maxBounces:4,seed:6,devices:[0,0,0],transport:'copy'});
 await pt.initialize(); const s=new Scene(); await s.loadGLB(%r,{normalize:true,mode:'cube'}); await pt.setScene(s);
 const n=(pt.trianglesData.length/9)|0; const b2=await pt.readBVH2(pt.computeBVH2Sizing(n).bytes);
 pt.setCameraPosition(0.2,-0.1,2.4); pt.setCameraQuaternion(0,0,0,1);
 pt.setBatch(2); for (let f=1; f<=3; f++) { pt.setFrameCount(f); await pt.render(); }
 const img=pt.readRadiance();
 require('fs').writeFileSync(%r, Buffer.from(img.buffer)); require('fs').writeFileSync(%r, Buffer.from(pt.trianglesData.buffer)); require('fs').writeFileSync(%r, Buffer.from(b2.buffer));
 log(JSON.stringify({n:n, gpus:pt.gpuCount()})); pt.destroy(); })().catch(e=>{console.error(e);process.exit(1);});
""" % (os.path.join(ROOT, "raytracer-public_amd", "js", "PathTracer.js"), os.path.join(ROOT, "raytracer-public_amd", "js", "Scene.js"),
       glb, str(tmp_path / "img.bin"), str(tmp_path / "tris.bin"), str(tmp_path / "bvh2.bin"))
    info = json.loads(subprocess.check_output([NODE, "-e", script], text=True).strip().splitlines()[-1])
    assert info == {"n": 36, "gpus": 3}
    tris = np.fromfile(str(tmp_path / "tris.bin"), np.float32)
    bvh2 = orc.build_lbvh2(tris)
    assert np.array_equal(np.fromfile(str(tmp_path / "bvh2.bin"), np.uint32), bvh2)
    bvh4, _ = orc.collapse_bvh4(bvh2, 36)
    ref, _, _ = orc.render(orc.make_params(144, 88, 36, (0.2, -0.1, 2.4), (0, 0, 0, 1), mode=orc_mod.MODE_PATH, spp=3, max_bounces=4, seed=6, frame=3), tris, bvh4)
    img = np.fromfile(str(tmp_path / "img.bin"), np.float32).reshape(88, 144, 4)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    out = subprocess.check_output([NODE, os.path.join(ROOT, "raytracer-public_amd", "js", "main.js"), "--devices", "0,0", "--transport", "copy", "--frames", "4", "--mode", "2",
                                   "--width", "320", "--height", "180", "--tris", "20000", "--dump", str(tmp_path / "d" / "BVH2.bin")], text=True)
    assert "Rendering on 2 GPUs" in out and "Msamples/s" in out


@pytest.mark.gpu
def test_node_accumulation_checkpoint_resumes(tmp_path):
    """PathTracer.readAccumulation() / restoreAccumulation() (pt_read_accum / pt_set_accum through the addon): 3 accumulated frames, dump,
    a NEW PathTracer with the same scene, restore, 3 more -- equal to 6 straight, bit for bit."""
    glb = os.path.join(HERE, "golden", "steve.glb")
    script = r"""
const PT=require(%r); const {Scene}=require(%r);
(async()=>{ const log=console.log; console.log=()=>{};
 const opt={mode:PT.MODE_PATH,spp:2,maxBounces:3,seed:4,accumulate:true};
 const s=new Scene(); await s.loadGLB(%r,{normalize:true,mode:'cube'});
 const mk=async()=>{ const pt=new PT.PathTracer({width:120,height:72},opt); await pt.initialize(); await pt.setScene(s); pt.setCameraPosition(0.2,0.1,2.6); return pt; };
 const a=await mk(); for (let f=0; f<6; f++) { a.setFrameCount(f); await a.render(); } const straight=a.readRadiance(); a.destroy();
 const b=await mk(); for (let f=0; f<3; f++) { b.setFrameCount(f); await b.render(); } const dump=b.readAccumulation(); b.destroy();
 const c=await mk(); c.restoreAccumulation(dump); for (let f=3; f<6; f++) { c.setFrameCount(f); await c.render(); } const resumed=c.readRadiance();
 let threw=false; try { c.restoreAccumulation({width:120,height:72,samples:2,data:new Float32Array(8)}); } catch (e) { threw=/libmi355pt error 1/.test(e.message); }
 c.destroy();
 const u=new Uint32Array(straight.buffer), v=new Uint32Array(resumed.buffer); let same=u.length===v.length; for (let i=0; same && i<u.length; i++) same=u[i]===v[i];
 log(JSON.stringify({same:same, samples:dump.samples, floats:dump.data.length, compact:dump.compact, w3:dump.data[3], threw:threw})); })().catch(e=>{console.error(e);process.exit(1);});
""" % (os.path.join(ROOT, "raytracer-public_amd", "js", "PathTracer.js"), os.path.join(ROOT, "raytracer-public_amd", "js", "Scene.js"), glb)
    info = json.loads(subprocess.check_output([NODE, "-e", script], text=True).strip().splitlines()[-1])
    assert info == {"same": True, "samples": 6, "floats": 120 * 72 * 4, "compact": 0, "w3": 6, "threw": True}

