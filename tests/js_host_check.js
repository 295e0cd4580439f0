// node tests/js_host_check.js <golden.json> : host-only entry points of the N-API addon (no GPU)
"use strict";
const path = require("path"), fs = require("fs");
const PT = require(path.join(__dirname, "..", "raytracer-public_amd", "js", "PathTracer.js"));
const g = JSON.parse(fs.readFileSync(process.argv[2], "utf8"));
const pt = new PT.PathTracer({ width: 4, height: 4 });
const eq = (a, b) => a.length === b.length && a.every((v, i) => (v >>> 0) === (b[i] >>> 0));
let checks = 0;
function must(c, what) { if (!c) { console.error("FAIL", what); process.exit(1); } checks++; }
for (const s of g.sizing) {
  must(JSON.stringify(pt.computeBVH2Sizing(s.numTris)) === JSON.stringify(s.bvh2), "computeBVH2Sizing " + s.numTris);
  must(pt.computeBVH4Sizing(s.bvh2.numNodes2).bytes === s.bvh4_of_numNodes2.bytes, "computeBVH4Sizing");
}
for (const c of g.morton) {
  const tris = new Float32Array(new Uint32Array(c.tris_f32_bits).buffer);
  const r = pt.buildMortonAndSort(tris);
  must(r.mortonSorted instanceof Uint32Array && eq(r.mortonSorted, c.mortonSorted) && eq(r.triIndexSorted, c.triIndexSorted), "buildMortonAndSort " + c.name);
}
for (const c of g.collapse) {
  const r = pt.collapseLBVH2ToBVH4(new Uint32Array(c.bvh2), c.numTris);
  must(r.numNodes4 === c.numNodes4 && eq(r.bvh4U32, c.bvh4), "collapseLBVH2ToBVH4 " + c.name);
}
// default mesh and fields of the reference constructor (PathTracer.js:60-95)
must(pt.trianglesData.length === 36 && pt.cameraPosition[2] === 3.5 && pt.frameCount === 0, "constructor defaults");
// no device -> render()/buildBVH() are silent no-ops like the reference (:673, :757)
pt.render().then(() => pt.buildBVH(pt.trianglesData)).then(() => {
  let threw = false;
  try { PT.native().collapse(new Uint32Array([3, 0, 0, 0, 7, 9, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0]), 2); } catch (e) { threw = /libmi355pt error 5/.test(e.message); }
  must(threw, "malformed BVH2 raises");
  console.log("ok", checks, PT.native().version());
});
