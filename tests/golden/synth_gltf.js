// synth_gltf.js -- deterministic synthetic glTF 2.0 files that exercise what the reference's loader (three's GLTFLoader behind
// src/libs/Scene.js:15-99) accepts beyond the two bundled GLBs: triangle strips and fans (indexed and not), sparse accessors
// (over a buffer view and over zeros), .gltf JSON with an external .bin and with base64 data URIs, several buffers, interleaved
// views, normalised-integer positions, several scenes, multi-primitive meshes on nodes with children, matrix + TRS nodes.
//
//   node tests/golden/synth_gltf.js <out_dir>      writes every case's files, prints the list of entry files as JSON
//
// Used twice with the same bytes: by gen_golden_glb.js (pushes the files through the reference's own three GLTFLoader and records the
// triangles' bit patterns -> gltf_synth_golden.json) and by tests/test_js_gltf_golden.py (pushes them through js/Scene.js).
// Node-12-safe CommonJS; its own LCG so the bytes do not depend on the JS engine's Math.random.
"use strict";
const fs = require("fs");
const path = require("path");

function rng(seed) {                       // 32-bit LCG (Numerical Recipes constants), uniform in [0, 1)
  let s = seed >>> 0;
  return function () { s = (Math.imul(s, 1664525) + 1013904223) >>> 0; return s / 4294967296; };
}

class Builder {
  constructor() { this.buffers = [[]]; this.views = []; this.accessors = []; }
  addBuffer() { this.buffers.push([]); return this.buffers.length - 1; }
  view(bytes, opt) {                       // opt: { buffer, stride, pad }
    const o = opt || {}, b = o.buffer || 0, buf = this.buffers[b];
    while (buf.length % 4) buf.push(0);
    for (let i = 0; i < (o.pad || 0); i++) buf.push(0xAB);           // bytes in front of the view that nobody may read
    const v = { buffer: b, byteOffset: buf.length, byteLength: bytes.length };
    if (o.stride) v.byteStride = o.stride;
    for (let i = 0; i < bytes.length; i++) buf.push(bytes[i]);
    this.views.push(v);
    return this.views.length - 1;
  }
  accessor(view, componentType, count, type, opt) {
    const o = opt || {}, a = { componentType: componentType, count: count, type: type };
    if (view !== null) a.bufferView = view;
    if (o.offset) a.byteOffset = o.offset;
    if (o.normalized) a.normalized = true;
    if (o.sparse) a.sparse = o.sparse;
    this.accessors.push(a);
    return this.accessors.length - 1;
  }
  json(doc, bufferDefs) {
    return Object.assign({ asset: { version: "2.0", generator: "tests/golden/synth_gltf.js" } }, doc,
      { buffers: bufferDefs, bufferViews: this.views, accessors: this.accessors });
  }
  bufferBytes(i) { const b = Buffer.from(this.buffers[i]); const pad = (4 - (b.length % 4)) % 4; return Buffer.concat([b, Buffer.alloc(pad)]); }
  glb(doc) {                               // single-buffer binary container
    const bin = this.bufferBytes(0);
    let js = Buffer.from(JSON.stringify(this.json(doc, [{ byteLength: bin.length }])), "utf8");
    js = Buffer.concat([js, Buffer.alloc((4 - (js.length % 4)) % 4, 0x20)]);
    const head = Buffer.alloc(12); head.writeUInt32LE(0x46546c67, 0); head.writeUInt32LE(2, 4); head.writeUInt32LE(12 + 8 + js.length + 8 + bin.length, 8);
    const h1 = Buffer.alloc(8); h1.writeUInt32LE(js.length, 0); h1.writeUInt32LE(0x4e4f534a, 4);
    const h2 = Buffer.alloc(8); h2.writeUInt32LE(bin.length, 0); h2.writeUInt32LE(0x004e4942, 4);
    return Buffer.concat([head, h1, js, h2, bin]);
  }
}

function f32bytes(arr) { const f = new Float32Array(arr); return Buffer.from(f.buffer, f.byteOffset, f.byteLength); }
function typed(T, arr) { const t = new T(arr); return Buffer.from(t.buffer, t.byteOffset, t.byteLength); }
function positions(r, n, scale) { const a = []; for (let i = 0; i < n * 3; i++) a.push((r() * 2 - 1) * (scale || 2)); return a; }
function indices(r, n, count) { const a = []; for (let i = 0; i < count; i++) a.push(Math.floor(r() * n)); return a; }
function unitQuat(r) { const q = [r() - 0.5, r() - 0.5, r() - 0.5, r() - 0.5]; const l = Math.hypot(q[0], q[1], q[2], q[3]); return q.map((x) => x / l); }
function trsMatrix(t, q, s) {              // column-major, like a glTF `matrix`
  const x = q[0], y = q[1], z = q[2], w = q[3];
  return [(1 - 2 * (y * y + z * z)) * s[0], 2 * (x * y + z * w) * s[0], 2 * (x * z - y * w) * s[0], 0,
          2 * (x * y - z * w) * s[1], (1 - 2 * (x * x + z * z)) * s[1], 2 * (y * z + x * w) * s[1], 0,
          2 * (x * z + y * w) * s[2], 2 * (y * z - x * w) * s[2], (1 - 2 * (x * x + y * y)) * s[2], 0, t[0], t[1], t[2], 1];
}

// ---- the cases ------------------------------------------------------------------------------------------------------------------
const cases = {};

// triangle strips: u16-indexed, non-indexed, and one too short to hold a triangle (2 vertices -> nothing); TRS node over a matrix child
cases["strips.glb"] = function () {
  const r = rng(101), g = new Builder();
  const pA = positions(r, 14), aA = g.accessor(g.view(f32bytes(pA)), 5126, 14, "VEC3");
  const iA = indices(r, 14, 11), aiA = g.accessor(g.view(typed(Uint16Array, iA)), 5123, 11, "SCALAR");
  const pB = positions(r, 9), aB = g.accessor(g.view(f32bytes(pB)), 5126, 9, "VEC3");
  const pC = positions(r, 2), aC = g.accessor(g.view(f32bytes(pC)), 5126, 2, "VEC3");
  const doc = { scene: 0, scenes: [{ nodes: [0] }],
    nodes: [{ translation: [0.3, -0.2, 0.7], rotation: unitQuat(r), scale: [1.5, 0.75, 1.25], mesh: 0, children: [1] },
            { matrix: trsMatrix([-0.4, 0.1, 0.2], unitQuat(r), [0.8, 1.1, 0.9]), mesh: 1 }],
    meshes: [{ primitives: [{ attributes: { POSITION: aA }, indices: aiA, mode: 5 }, { attributes: { POSITION: aB }, mode: 5 }] },
             { primitives: [{ attributes: { POSITION: aC }, mode: 5 }, { attributes: { POSITION: aB }, mode: 5 }] }] };
  return { "strips.glb": g.glb(doc) };
};

// triangle fans: u8-indexed and non-indexed, next to a plain TRIANGLES primitive and a LINE_STRIP that is not a mesh
cases["fans.glb"] = function () {
  const r = rng(202), g = new Builder();
  const pA = positions(r, 10), aA = g.accessor(g.view(f32bytes(pA)), 5126, 10, "VEC3");
  const iA = indices(r, 10, 8), aiA = g.accessor(g.view(typed(Uint8Array, iA)), 5121, 8, "SCALAR");
  const pB = positions(r, 7), aB = g.accessor(g.view(f32bytes(pB)), 5126, 7, "VEC3");
  const iT = indices(r, 7, 12), aiT = g.accessor(g.view(typed(Uint32Array, iT)), 5125, 12, "SCALAR");
  const doc = { scenes: [{ nodes: [1, 0] }],                           // no `scene` member: scene 0
    nodes: [{ mesh: 0, scale: [-1, 1, 1] },                             // a mirroring scale (negative determinant)
            { matrix: trsMatrix([1, 2, 3], unitQuat(r), [2, 2, 2]), children: [2] }, { mesh: 0, rotation: unitQuat(r) }],
    meshes: [{ primitives: [{ attributes: { POSITION: aA }, indices: aiA, mode: 6 }, { attributes: { POSITION: aB }, mode: 6 },
                            { attributes: { POSITION: aB }, indices: aiT }, { attributes: { POSITION: aA }, mode: 3 }] }] };
  return { "fans.glb": g.glb(doc) };
};

// sparse accessors: over a buffer view (u16 sparse indices), over zeros (no buffer view, u8 sparse indices), and u32 sparse indices with
// byte offsets inside shared views; the triangle index accessor itself sparse as well
cases["sparse.glb"] = function () {
  const r = rng(303), g = new Builder();
  const base = positions(r, 12);
  const vBase = g.view(f32bytes(base));
  const sIdx = [1, 4, 5, 10], sVal = positions(r, 4, 3);
  const vSI = g.view(typed(Uint16Array, [9, 9].concat(sIdx))), vSV = g.view(f32bytes([7, 7, 7].concat(sVal)));
  const aA = g.accessor(vBase, 5126, 12, "VEC3", { sparse: { count: 4, indices: { bufferView: vSI, byteOffset: 4, componentType: 5123 }, values: { bufferView: vSV, byteOffset: 12 } } });
  const zIdx = [0, 2, 3, 6, 7], zVal = positions(r, 5, 1.5);
  const aZ = g.accessor(null, 5126, 8, "VEC3", { sparse: { count: 5, indices: { bufferView: g.view(typed(Uint8Array, zIdx)), componentType: 5121 }, values: { bufferView: g.view(f32bytes(zVal)) } } });
  const uIdx = [3, 11], uVal = positions(r, 2, 4);
  const aU = g.accessor(vBase, 5126, 12, "VEC3", { sparse: { count: 2, indices: { bufferView: g.view(typed(Uint32Array, uIdx)), componentType: 5125 }, values: { bufferView: g.view(f32bytes(uVal)) } } });
  const tri = indices(r, 12, 18);
  const aiS = g.accessor(g.view(typed(Uint16Array, tri)), 5123, 18, "SCALAR", { sparse: { count: 3, indices: { bufferView: g.view(typed(Uint8Array, [0, 7, 17])), componentType: 5121 },
                                                                               values: { bufferView: g.view(typed(Uint16Array, [11, 0, 5])) } } });
  const triZ = indices(r, 8, 9), aiZ = g.accessor(g.view(typed(Uint8Array, triZ)), 5121, 9, "SCALAR");
  const doc = { scene: 0, scenes: [{ nodes: [0, 1] }],
    nodes: [{ mesh: 0, translation: [0.1, 0.2, 0.3] }, { mesh: 1, rotation: unitQuat(r), scale: [1.2, 1.2, 0.6] }],
    meshes: [{ primitives: [{ attributes: { POSITION: aA }, indices: aiS }, { attributes: { POSITION: aZ }, indices: aiZ }] },
             { primitives: [{ attributes: { POSITION: aU } }, { attributes: { POSITION: aZ }, mode: 5 }] }] };
  return { "sparse.glb": g.glb(doc) };
};

// .gltf JSON: buffer 0 an external file next to it, buffer 1 a base64 data URI; a three-level hierarchy
cases["external.gltf"] = function () {
  const r = rng(404), g = new Builder();
  const b1 = g.addBuffer();
  const pA = positions(r, 20), aA = g.accessor(g.view(f32bytes(pA), { pad: 8 }), 5126, 20, "VEC3");
  const iA = indices(r, 20, 30), aiA = g.accessor(g.view(typed(Uint16Array, iA), { buffer: b1 }), 5123, 30, "SCALAR");
  const pB = positions(r, 6), aB = g.accessor(g.view(f32bytes(pB), { buffer: b1 }), 5126, 6, "VEC3");
  const doc = { scene: 0, scenes: [{ nodes: [0] }],
    nodes: [{ children: [1, 3], translation: [0, 1, 0] }, { mesh: 0, rotation: unitQuat(r), children: [2] }, { mesh: 1, scale: [0.5, 2, 1] },
            { mesh: 1, matrix: trsMatrix([2, 0, -1], unitQuat(r), [1, 1, 1]) }],
    meshes: [{ primitives: [{ attributes: { POSITION: aA }, indices: aiA }] }, { primitives: [{ attributes: { POSITION: aB } }, { attributes: { POSITION: aB }, mode: 6 }] }] };
  const bin0 = g.bufferBytes(0), bin1 = g.bufferBytes(1);
  const js = g.json(doc, [{ byteLength: bin0.length, uri: "external_0.bin" }, { byteLength: bin1.length, uri: "data:application/octet-stream;base64," + bin1.toString("base64") }]);
  return { "external.gltf": Buffer.from(JSON.stringify(js, null, 1), "utf8"), "external_0.bin": bin0 };
};

// .gltf JSON, every buffer a data URI (one of them with the older "application/gltf-buffer" media type); interleaved strip
cases["datauri.gltf"] = function () {
  const r = rng(505), g = new Builder();
  const b1 = g.addBuffer();
  const n = 11, inter = [];
  const p = positions(r, n);
  for (let i = 0; i < n; i++) inter.push(p[i * 3], p[i * 3 + 1], p[i * 3 + 2], 0, 0, 1, i / n, 1 - i / n);      // position + normal + uv, stride 32
  const vI = g.view(f32bytes(inter), { stride: 32 });
  const aP = g.accessor(vI, 5126, n, "VEC3"), aN = g.accessor(vI, 5126, n, "VEC3", { offset: 12 });
  const iS = indices(r, n, 9), aiS = g.accessor(g.view(typed(Uint16Array, iS), { buffer: b1 }), 5123, 9, "SCALAR");
  const iT = indices(r, n, 15), aiT = g.accessor(g.view(typed(Uint8Array, iT), { buffer: b1 }), 5121, 15, "SCALAR");
  const doc = { scene: 0, scenes: [{ nodes: [0] }], nodes: [{ mesh: 0, rotation: unitQuat(r), translation: [0.25, 0.5, -0.75] }],
    meshes: [{ primitives: [{ attributes: { POSITION: aP, NORMAL: aN }, indices: aiS, mode: 5 }, { attributes: { POSITION: aP, NORMAL: aN }, indices: aiT, mode: 4 },
                            { attributes: { POSITION: aP }, indices: aiS, mode: 6 }] }] };
  const bin0 = g.bufferBytes(0), bin1 = g.bufferBytes(1);
  const js = g.json(doc, [{ byteLength: bin0.length, uri: "data:application/gltf-buffer;base64," + bin0.toString("base64") },
                          { byteLength: bin1.length, uri: "data:application/octet-stream;base64," + bin1.toString("base64") }]);
  return { "datauri.gltf": Buffer.from(JSON.stringify(js), "utf8") };
};

// integer positions (KHR_mesh_quantization): normalised int16 / uint8 and plain int8 / uint16, indexed and interleaved.  The reference takes the
// attribute's RAW array (Scene.js:66-85 reads pos.array, not getX): a normalised integer arrives as the integer.
cases["quantized.glb"] = function () {
  const r = rng(606), g = new Builder();
  const q16 = []; for (let i = 0; i < 10; i++) q16.push(Math.floor(r() * 65535) - 32767, Math.floor(r() * 65535) - 32767, Math.floor(r() * 65535) - 32767, 0);
  q16[0] = -32768; q16[1] = 32767;
  const a16 = g.accessor(g.view(typed(Int16Array, q16), { stride: 8 }), 5122, 10, "VEC3", { normalized: true });
  const i16 = indices(r, 10, 12), ai16 = g.accessor(g.view(typed(Uint32Array, i16)), 5125, 12, "SCALAR");
  const q8 = []; for (let i = 0; i < 9; i++) q8.push(Math.floor(r() * 256), Math.floor(r() * 256), Math.floor(r() * 256), 0);
  const a8 = g.accessor(g.view(typed(Uint8Array, q8), { stride: 4 }), 5121, 9, "VEC3", { normalized: true });
  const i8 = indices(r, 9, 9), ai8 = g.accessor(g.view(typed(Uint8Array, i8)), 5121, 9, "SCALAR");
  const s8 = []; for (let i = 0; i < 6; i++) s8.push(Math.floor(r() * 255) - 127, Math.floor(r() * 255) - 127, Math.floor(r() * 255) - 127, 0);
  const aS8 = g.accessor(g.view(typed(Int8Array, s8), { stride: 4 }), 5120, 6, "VEC3");
  const iS8 = indices(r, 6, 6), aiS8 = g.accessor(g.view(typed(Uint16Array, iS8)), 5123, 6, "SCALAR");
  const u16 = []; for (let i = 0; i < 6; i++) u16.push(Math.floor(r() * 65536), Math.floor(r() * 65536), Math.floor(r() * 65536), 0);
  const aU16 = g.accessor(g.view(typed(Uint16Array, u16), { stride: 8 }), 5123, 6, "VEC3");
  const doc = { scene: 0, scenes: [{ nodes: [0, 1] }], extensionsUsed: ["KHR_mesh_quantization"], extensionsRequired: ["KHR_mesh_quantization"],
    nodes: [{ mesh: 0, scale: [1 / 32767, 1 / 32767, 1 / 32767], translation: [0.5, 0, 0] }, { mesh: 1, scale: [0.01, 0.01, 0.01] }],
    meshes: [{ primitives: [{ attributes: { POSITION: a16 }, indices: ai16 }, { attributes: { POSITION: a8 }, indices: ai8 }] },
             { primitives: [{ attributes: { POSITION: aS8 }, indices: aiS8 }, { attributes: { POSITION: aU16 }, indices: aiS8, mode: 5 }] }] };
  return { "quantized.glb": g.glb(doc) };
};

// two scenes (the second is the one shown), a multi-primitive mesh on a node WITH children (three: a Group whose first children are the
// primitives, then the node's children), a mesh used twice, a camera-only node, a node without anything
cases["scenes.glb"] = function () {
  const r = rng(707), g = new Builder();
  const pA = positions(r, 9), aA = g.accessor(g.view(f32bytes(pA)), 5126, 9, "VEC3");
  const pB = positions(r, 6), aB = g.accessor(g.view(f32bytes(pB)), 5126, 6, "VEC3");
  const pC = positions(r, 5), aC = g.accessor(g.view(f32bytes(pC)), 5126, 5, "VEC3");
  const iC = indices(r, 5, 6), aiC = g.accessor(g.view(typed(Uint16Array, iC)), 5123, 6, "SCALAR");
  const doc = { scene: 1, scenes: [{ nodes: [4] }, { nodes: [0, 5, 3] }],
    cameras: [{ type: "perspective", perspective: { yfov: 0.8, znear: 0.1 } }],
    nodes: [{ mesh: 0, translation: [0, 0, 1], children: [1, 2] }, { mesh: 1, rotation: unitQuat(r) }, { camera: 0, children: [6] },
            { mesh: 1, scale: [3, 3, 3] }, { mesh: 0 }, {}, { mesh: 2, translation: [-1, -1, -1] }],
    meshes: [{ primitives: [{ attributes: { POSITION: aA } }, { attributes: { POSITION: aB } }] },
             { primitives: [{ attributes: { POSITION: aC }, indices: aiC }] }, { primitives: [{ attributes: { POSITION: aB }, mode: 6 }] }] };
  return { "scenes.glb": g.glb(doc) };
};

// a strip and a fan over an INTERLEAVED view, non-indexed: three's draw-mode conversion gives them an index, so the reference's
// toNonIndexed() path de-interleaves them correctly
cases["interleaved_strip.glb"] = function () {
  const r = rng(808), g = new Builder();
  const n = 8, inter = [], p = positions(r, n);
  for (let i = 0; i < n; i++) inter.push(p[i * 3], p[i * 3 + 1], p[i * 3 + 2], 1, 0, 0);
  const vI = g.view(f32bytes(inter), { stride: 24 });
  const aP = g.accessor(vI, 5126, n, "VEC3"), aP2 = g.accessor(vI, 5126, n - 2, "VEC3", { offset: 48 });
  const doc = { scene: 0, scenes: [{ nodes: [0] }], nodes: [{ mesh: 0, matrix: trsMatrix([0, 0, 0], unitQuat(r), [1, -1, 1]) }],
    meshes: [{ primitives: [{ attributes: { POSITION: aP }, mode: 5 }, { attributes: { POSITION: aP2 }, mode: 6 }] }] };
  return { "interleaved_strip.glb": g.glb(doc) };
};

function writeAll(dir) {
  fs.mkdirSync(dir, { recursive: true });
  const entries = [];
  for (const name of Object.keys(cases)) {
    const files = cases[name]();
    for (const f of Object.keys(files)) fs.writeFileSync(path.join(dir, f), files[f]);
    entries.push(name);
  }
  return entries;
}

module.exports = { cases, writeAll, Builder, rng };
if (require.main === module) console.log(JSON.stringify(writeAll(process.argv[2] || ".")));
