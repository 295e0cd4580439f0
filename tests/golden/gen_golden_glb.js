// gen_golden_glb.js -- golden triangle arrays for the reference's two bundled GLBs
// (public/assets/dodecahedron.glb, steve.glb), produced by the reference's own path:
// three's GLTFLoader.parse + the parseGLTF / normalizeMesh / getTrianglesFloat32 logic of
// src/libs/Scene.js.  Scene.js itself does not parse under Node 12 (`??`), so this script
// evaluates its three methods' semantics through three's classes exactly as Scene.js does
// (scene.updateMatrixWorld(true); traverse; toNonIndexed; Vector3.applyMatrix4(matrixWorld)).
//
//   node tests/golden/gen_golden_glb.js [/root/reference]
//
// Second leg (round 6): the synthetic files of synth_gltf.js (strips, fans, sparse accessors, .gltf with external and data-URI
// buffers, quantised positions, several scenes ...) go through the reference's path AS Scene.js:15-42 WALKS IT -- GLTFLoader.load(url),
// i.e. FileLoader + fetch -- with browser-environment shims only (fetch / Request / Headers over fs and data: URIs; Node 12 has none),
// then through the same extract / normalise statements -> gltf_synth_golden.json (bit patterns only).
//
// Runs from a scratch directory that symlinks the reference's node_modules (nothing is copied
// into this repo except the two small .glb data files and the resulting numbers).
"use strict";
const fs = require("fs"), os = require("os"), path = require("path"), cp = require("child_process");
const REF = process.argv[2] || "/root/reference";
const OUT = path.join(__dirname, "glb_golden.json");
const scratch = fs.mkdtempSync(path.join(os.tmpdir(), "ptglb-"));
fs.symlinkSync(path.join(REF, "node_modules"), path.join(scratch, "node_modules"));
fs.writeFileSync(path.join(scratch, "package.json"), '{"type":"module"}');
fs.writeFileSync(path.join(scratch, "run.js"), `
globalThis.self = globalThis;
if (typeof AbortController === "undefined") globalThis.AbortController = class { constructor() { this.signal = {}; } abort() {} };
// browser environment for FileLoader (three.core.js: fetch(new Request(url, {headers: new Headers(...), signal})), then response.arrayBuffer() / .text())
globalThis.AbortSignal = globalThis.AbortSignal || {};
globalThis.Headers = globalThis.Headers || class { constructor(h) { this.h = h || {}; } get(k) { return this.h[k] || null; } };
globalThis.Request = globalThis.Request || class { constructor(url) { this.url = url; } };
globalThis.fetch = globalThis.fetch || (async (req) => {
  const url = typeof req === "string" ? req : req.url;
  let buf;
  const m = /^data:[^,]*?(;base64)?,(.*)$/.exec(url);
  if (m) buf = m[1] ? Buffer.from(m[2], "base64") : Buffer.from(decodeURIComponent(m[2]), "utf8");
  else { try { buf = fs.readFileSync(url); } catch (e) { return { status: 404, statusText: "Not Found", url: url, headers: new Headers() }; } }
  const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
  return { status: 200, url: url, headers: new Headers(), arrayBuffer: async () => ab, text: async () => buf.toString("utf8"), json: async () => JSON.parse(buf.toString("utf8")) };
});
import * as THREE from "three";
import { GLTFLoader } from "three/examples/jsm/loaders/GLTFLoader.js";
import fs from "fs";
const files = JSON.parse(process.argv[2]);
const out = {};
function extract(gltf) {   // Scene.js:47-99
  const scene = gltf.scene; scene.updateMatrixWorld(true);
  const tris = [];
  scene.traverse((obj) => {
    if (!obj.isMesh) return;
    const geom = obj.geometry; const worldMatrix = obj.matrixWorld.clone();
    const nonIndexed = geom.index !== null ? geom.toNonIndexed() : geom.clone();
    const pos = nonIndexed.getAttribute("position"); if (!pos) return;
    const array = pos.array, count = pos.count;
    for (let i = 0; i < count; i += 3) {
      const v = [0, 1, 2].map((k) => new THREE.Vector3(array[(i + k) * 3], array[(i + k) * 3 + 1], array[(i + k) * 3 + 2]).applyMatrix4(worldMatrix));
      tris.push({ v0: [v[0].x, v[0].y, v[0].z], v1: [v[1].x, v[1].y, v[1].z], v2: [v[2].x, v[2].y, v[2].z] });
    }
  });
  return tris;
}
function normalize(tris, mode) {   // Scene.js:104-165
  let min = [Infinity, Infinity, Infinity], max = [-Infinity, -Infinity, -Infinity];
  for (const t of tris) for (const v of [t.v0, t.v1, t.v2]) for (let k = 0; k < 3; k++) { min[k] = Math.min(min[k], v[k]); max[k] = Math.max(max[k], v[k]); }
  const center = [(min[0] + max[0]) * 0.5, (min[1] + max[1]) * 0.5, (min[2] + max[2]) * 0.5];
  const maxDim = Math.max(max[0] - min[0], max[1] - min[1], max[2] - min[2]);
  let scale = 2.0 / maxDim; if (mode === "sphere") scale = 1.0 / (maxDim * 0.5);
  for (const t of tris) for (const v of [t.v0, t.v1, t.v2]) for (let k = 0; k < 3; k++) v[k] = (v[k] - center[k]) * scale;
}
function f32(tris) { const a = new Float32Array(tris.length * 9); let o = 0; for (const t of tris) for (const v of [t.v0, t.v1, t.v2]) for (let k = 0; k < 3; k++) a[o++] = v[k]; return a; }
const bits = (a) => Array.from(new Uint32Array(a.buffer));
const synth = JSON.parse(process.argv[4] || "[]");
if (synth.length) {        // second leg: loader.load(url) as Scene.js:19-32
  const sout = {}; let left = synth.length;
  const warn = console.warn; console.warn = () => {};           // toNonIndexed on zero-triangle strips etc.
  for (const f of synth) {
    new GLTFLoader().load(f, (gltf) => {
      const raw = extract(gltf); const rawF32 = f32(raw);
      normalize(raw, "cube");
      sout[f.split("/").pop()] = { numTris: rawF32.length / 9, world_f32_bits: bits(rawF32), normalized_cube_f32_bits: bits(f32(raw)) };
      if (--left === 0) { console.warn = warn; fs.writeFileSync(process.argv[5], JSON.stringify(sout)); }
    }, undefined, (e) => { console.error("load failed", f, e); process.exit(1); });
  }
}
let pending = files.length;
for (const f of files) {
  const buf = fs.readFileSync(f); const ab = buf.buffer.slice(buf.byteOffset, buf.byteOffset + buf.byteLength);
  new GLTFLoader().parse(ab, "", (gltf) => {
    const raw = extract(gltf); const rawF32 = f32(raw);
    const first = raw.length ? raw[0].v0.slice() : null;
    normalize(raw, "cube");
    out[f.split("/").pop()] = { numTris: rawF32.length / 9, first_world_vertex_f64: first, world_f32_bits: bits(rawF32), normalized_cube_f32_bits: bits(f32(raw)) };
    if (--pending === 0) { fs.writeFileSync(process.argv[3], JSON.stringify(out)); }
  }, (e) => { console.error("parse failed", f, e); process.exit(1); });
}
`);
const files = ["dodecahedron.glb", "steve.glb"].map((f) => path.join(REF, "public/assets", f));
const OUT2 = path.join(__dirname, "gltf_synth_golden.json");
const synthDir = fs.mkdtempSync(path.join(os.tmpdir(), "ptsynth-"));
const synth = require("./synth_gltf.js").writeAll(synthDir).map((f) => path.join(synthDir, f));
try {
  cp.execFileSync(process.execPath, [path.join(scratch, "run.js"), JSON.stringify(files), OUT, JSON.stringify(synth), OUT2], { stdio: "inherit", cwd: scratch });
  console.log("wrote", OUT, fs.statSync(OUT).size, "bytes");
  console.log("wrote", OUT2, fs.statSync(OUT2).size, "bytes");
  for (const f of files) fs.copyFileSync(f, path.join(__dirname, path.basename(f)));   // the .glb inputs are data fixtures
} finally {
  for (const f of fs.readdirSync(synthDir)) fs.unlinkSync(path.join(synthDir, f));
  fs.rmdirSync(synthDir);
  for (const f of fs.readdirSync(scratch)) { const p = path.join(scratch, f); if (fs.lstatSync(p).isSymbolicLink()) fs.unlinkSync(p); else fs.unlinkSync(p); }
  fs.rmdirSync(scratch);
}
