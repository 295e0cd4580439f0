// gen_golden_glmatrix.js -- generates tests/golden/glmatrix_camera_golden.json by REQUIRING the gl-matrix the reference ships
// (node_modules/gl-matrix, the dependency FPSCamera builds its orientation with: src/libs/controls/input-handler.js:99-110).
//
//   node tests/golden/gen_golden_glmatrix.js [/root/reference]        (build container only; the reference is absent on the GPU box)
//
// For seeded (yaw, pitch) pairs it records what the reference's own call sequence produces:
//   q = normalize(setAxisAngle(Y, yaw) * setAxisAngle(X, pitch))       input-handler.js:101-104
//   fwd / right / up = transformQuat((0,0,-1) / (1,0,0) / (0,1,0), q)   input-handler.js:108-110
// as f64 values (gl-matrix computes in Float32Array by default; setMatrixArrayType(Array) keeps doubles, so the fixture pins the
// CONVENTION -- multiplication order, handedness, xyzw layout -- to 1e-6 rather than one library's f32 rounding).
// Only inputs and outputs are stored; nothing of gl-matrix or the reference is written into this repository.
"use strict";
const fs = require("fs");
const path = require("path");
const REF = process.argv[2] || "/root/reference";
const glm = require(path.join(REF, "node_modules", "gl-matrix"));
glm.glMatrix.setMatrixArrayType(Array);
const { quat, vec3 } = glm;

let s = 0x2545F491 >>> 0;
function lcg() { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s; }
function frand() { return (lcg() >>> 8) / 16777216; }

const Y_AXIS = [0, 1, 0], X_AXIS = [1, 0, 0];
const cases = [];
const fixed = [[0, 0], [Math.PI / 2, 0], [0, Math.PI / 4], [Math.PI, -0.5], [-2.2, 1.2], [0.7, -1.3]];
for (let i = 0; i < 64; i++) {
  const yaw = i < fixed.length ? fixed[i][0] : (frand() * 2 - 1) * Math.PI;
  const pitch = i < fixed.length ? fixed[i][1] : (frand() * 2 - 1) * 1.5;
  const qYaw = quat.setAxisAngle(quat.create(), Y_AXIS, yaw);
  const qPitch = quat.setAxisAngle(quat.create(), X_AXIS, pitch);
  const q = quat.create();
  quat.multiply(q, qYaw, qPitch);
  quat.normalize(q, q);
  const fwd = vec3.transformQuat(vec3.create(), vec3.fromValues(0, 0, -1), q);
  const right = vec3.transformQuat(vec3.create(), vec3.fromValues(1, 0, 0), q);
  const up = vec3.transformQuat(vec3.create(), vec3.fromValues(0, 1, 0), q);
  // a few arbitrary vectors through the same rotation
  const probes = [];
  for (let k = 0; k < 3; k++) {
    const v = [frand() * 2 - 1, frand() * 2 - 1, frand() * 2 - 1];
    probes.push({ v: v, out: Array.from(vec3.transformQuat(vec3.create(), v, q)) });
  }
  cases.push({ yaw: yaw, pitch: pitch, q: Array.from(q), fwd: Array.from(fwd), right: Array.from(right), up: Array.from(up), probes: probes });
}
const out = { generator: "tests/golden/gen_golden_glmatrix.js", source: "gl-matrix " + require(path.join(REF, "node_modules", "gl-matrix", "package.json")).version + " as shipped in the reference's node_modules", cases: cases };
fs.writeFileSync(path.join(__dirname, "glmatrix_camera_golden.json"), JSON.stringify(out));
console.log("wrote", cases.length, "cases");
