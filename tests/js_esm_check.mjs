// node tests/js_esm_check.mjs <golden.json> : the INTEGRATION.md section-1 sequence up to the host-side BVH steps, through the ES-module face of
// the package -- the first two imports are the reference's src/main.js:1-2 with the path of this package in front (no GPU needed).
import * as PT from "../raytracer-public_amd/js/libs/PathTracer.js";
import * as PTScene from "../raytracer-public_amd/js/libs/Scene.js";
import * as PTm from "../raytracer-public_amd/js/libs/PathTracer.mjs";
import * as PTScenem from "../raytracer-public_amd/js/libs/Scene.mjs";
import fs from "fs";

const g = JSON.parse(fs.readFileSync(process.argv[2], "utf8"));
let checks = 0;
function must(c, what) { if (!c) { console.error("FAIL", what); process.exit(1); } checks++; }
const eq = (a, b) => a.length === b.length && a.every((v, i) => (v >>> 0) === (b[i] >>> 0));

must(typeof PT.PathTracer === "function" && typeof PTScene.Scene === "function", "named exports");
must(PTm.PathTracer === PT.PathTracer && PTScenem.Scene === PTScene.Scene, ".mjs faces are the same classes");
must(PT.MODE_REFERENCE === 1 && PT.MODE_PATH === 2 && PT.MODE_REFERENCE_PACKET === 0, "mode constants");
const pathTracer = new PT.PathTracer({ width: 64, height: 40 });                    // src/main.js:8 with a {width, height} canvas
must(pathTracer.trianglesData.length === 36 && pathTracer.cameraPosition[2] === 3.5, "constructor defaults (PathTracer.js:67, :79-84)");
const scene = new PTScene.Scene();                                                   // src/main.js:19
scene.setTrianglesFloat32(pathTracer.trianglesData);
must(scene.getTriangles().length === 4, "Scene holds the tetrahedron");
for (const c of g.morton) {                                                          // PathTracer.js:427
  const tris = new Float32Array(new Uint32Array(c.tris_f32_bits).buffer);
  const r = pathTracer.buildMortonAndSort(tris);
  must(eq(r.mortonSorted, c.mortonSorted) && eq(r.triIndexSorted, c.triIndexSorted), "buildMortonAndSort " + c.name);
}
for (const c of g.collapse) {                                                        // PathTracer.js:506
  const r = pathTracer.collapseLBVH2ToBVH4(new Uint32Array(c.bvh2), c.numTris);
  must(r.numNodes4 === c.numNodes4 && eq(r.bvh4U32, c.bvh4), "collapseLBVH2ToBVH4 " + c.name);
}
const numTris = (pathTracer.trianglesData.length / 9) | 0;                           // src/main.js:27-28
const { bytes: bvh2Bytes } = pathTracer.computeBVH2Sizing(numTris);
must(bvh2Bytes === 4 + 24 * (2 * numTris - 1), "computeBVH2Sizing");
pathTracer.render().then(() => { console.log("ok", checks, PT.native().version()); });   // no device: a silent no-op like the reference (PathTracer.js:757)
