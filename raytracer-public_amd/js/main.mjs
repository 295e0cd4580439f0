// main.mjs -- the reference's src/main.js as an ES module under Node: its first two lines are src/main.js:1-2 AS WRITTEN (they resolve against
// libs/PathTracer.js and libs/Scene.js, the ES-module faces of this package), and what follows is the reference's sequence statement by statement
// (src/main.js:5-76) without the DOM: a {width, height} object for the <canvas>, a fixed camera for the FPSCamera, a fixed number of frames for
// requestAnimationFrame, the BVH2 dump written to data/BVH2.bin instead of POSTed to /api/write.  (No top-level await: Node 12.)
//   node raytracer-public_amd/js/main.mjs [--frames N] [--width W --height H] [--glb path] [--tris N] [--mode 0|1|2 --spp S --bounces B]
import * as PT from "./libs/PathTracer.js";
import * as PTScene from "./libs/Scene.js";
import fs from "fs";
import path from "path";

function arg(name, dflt) { const i = process.argv.indexOf("--" + name); return i >= 0 && i + 1 < process.argv.length ? process.argv[i + 1] : dflt; }

async function main() {
  const canvas = { width: Number(arg("width", 1920)), height: Number(arg("height", 1080)) };        // index.html:10
  const pathTracer = new PT.PathTracer(canvas, { mode: Number(arg("mode", PT.MODE_REFERENCE)), spp: Number(arg("spp", 4)), maxBounces: Number(arg("bounces", 8)) });   // src/main.js:8 (options: this build's extension)
  const camera = { position: [0, 0, 2.5], rotation: [0, 0, 0, 1] };                                  // src/main.js:10-14

  await pathTracer.initialize();                                                                     // src/main.js:16

  // ---------- Scene ----------  src/main.js:18-24
  const scene = new PTScene.Scene();
  const glb = arg("glb", "/assets/dragon.glb");
  try {
    await scene.loadGLB(glb, { normalize: true, mode: "cube" });
  } catch (e) {
    const n = Number(arg("tris", 871414));
    console.log("GLB not available (" + glb + "): using the procedural dragon-class stand-in,", n, "triangles");
    scene.getTrianglesFloat32 = () => PT.native().proceduralScene(0, n, 20260109);
  }
  await pathTracer.setScene(scene);

  // ---------- BVH Dump (ONCE) ----------  src/main.js:26-46
  const numTris = (pathTracer.trianglesData.length / 9) | 0;
  const { bytes: bvh2Bytes } = pathTracer.computeBVH2Sizing(numTris);
  const bvh2U32 = await pathTracer.readBVH2(bvh2Bytes);
  const dump = arg("dump", path.join("data", "BVH2.bin"));
  console.log("Uploading BVH2:", bvh2U32.length * 4, "bytes");
  fs.mkdirSync(path.dirname(dump), { recursive: true });
  PT.native().writeU32File(dump, bvh2U32);
  console.log("BVH2 dump complete");

  // ---------- Render Loop ----------  src/main.js:48-76
  const frames = Number(arg("frames", 30));
  let frameIndex = 0;
  const t0 = Date.now();
  for (let f = 0; f < frames; f++) {
    frameIndex++;
    pathTracer.setCameraPosition(camera.position[0], camera.position[1], camera.position[2]);
    pathTracer.setCameraQuaternion(camera.rotation[0], camera.rotation[1], camera.rotation[2], camera.rotation[3]);
    pathTracer.setFrameCount(frameIndex);
    await pathTracer.render();
  }
  pathTracer.synchronize();
  console.log((frames / ((Date.now() - t0) / 1000)).toFixed(1) + " FPS (" + frames + " frames)");
  pathTracer.destroy();
}

main().catch((e) => { console.error(e); process.exit(1); });
