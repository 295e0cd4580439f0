// node tests/js_scene_check.js  -> prints JSON {file: {world_bits, normalized_bits}} for the fixture GLBs
"use strict";
const path = require("path");
const { Scene } = require(path.join(__dirname, "..", "raytracer-public_amd", "js", "Scene.js"));
const bits = (a) => Array.from(new Uint32Array(a.buffer.slice(a.byteOffset, a.byteOffset + a.byteLength)));
(async () => {
  const out = {};
  const log = console.log; console.log = () => {};
  for (const f of ["dodecahedron.glb", "steve.glb"]) {
    const s = new Scene();
    await s.loadGLB(path.join(__dirname, "golden", f), { normalize: false });
    const world = s.getTrianglesFloat32();
    const s2 = new Scene();
    await s2.loadGLB(path.join(__dirname, "golden", f), { normalize: true, mode: "cube" });
    out[f] = { numTris: world.length / 9, world_bits: bits(world), normalized_bits: bits(s2.getTrianglesFloat32()), first: s.getTriangles()[0].v0 };
  }
  console.log = log;
  console.log(JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
