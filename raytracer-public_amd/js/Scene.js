// Scene.js -- Node counterpart of the reference's src/libs/Scene.js: GLB -> world-space
// triangle soup (Float32Array, 9 floats per triangle), optional normalisation to [-1,1]^3.
// Same class / method names and option semantics (loadGLB(url, {normalize, mode}),
// parseGLTF, normalizeMesh, getTrianglesFloat32, getTriangles).  The reference parses with
// three's GLTFLoader; this file carries its own dependency-free GLB reader that follows the
// same arithmetic (all in doubles until the final Float32Array store):
//   * node transforms: a `matrix` node is decomposed to position/quaternion/scale and
//     re-composed, TRS nodes are composed directly -- what three's Object3D.applyMatrix4 /
//     updateMatrix do (GLTFLoader.js node loader), so world matrices carry the same rounding;
//   * world matrix = parent world x local, scene traversal in DFS pre-order, a node's own mesh
//     primitives before its children (Object3D.traverse order, Scene.js:53);
//   * indexed primitives are expanded (toNonIndexed, Scene.js:59-60); each vertex goes through
//     Vector3.applyMatrix4 including its perspective divide (Scene.js:69-85);
//   * triangle strips and fans become triangles in the vertex order of three's toTrianglesDrawMode (what GLTFLoader applies to
//     modes 5 / 6), sparse accessors are resolved as GLTFLoader.loadAccessor does, `.gltf` JSON files bring their buffers as
//     external files (relative to the .gltf) or data: URIs, a GLB may name further buffers next to its BIN chunk;
//   * the reference reads the position attribute's RAW array (Scene.js:66-85: `pos.array`, not getX): integer positions
//     (KHR_mesh_quantization) arrive as the integers, also when the accessor says `normalized` -- reproduced here.
// All of the above is pinned bit for bit by running three's own GLTFLoader on the same files (tests/golden/gen_golden_glb.js ->
// glb_golden.json, gltf_synth_golden.json).  One deliberate difference: a NON-indexed TRIANGLES primitive over an interleaved view
// is read with its stride here; the reference reads the interleaved array as if it were tightly packed (geometry.clone() keeps the
// InterleavedBufferAttribute, Scene.js:66 then indexes its .array by 3) and gets normals / UVs mixed into the positions.
// Draco- and (required) meshopt-compressed files are refused loudly -- the reference's GLTFLoader has no decoder set either (Scene.js:6).
// Node-12-safe CommonJS.
"use strict";
const fs = require("fs");
const path = require("path");

// ---- small column-major 4x4 helpers (element order = three's Matrix4.elements) ------------
function identity() { return [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]; }

function multiply(a, b) {           // a * b
  const t = new Array(16);
  for (let c = 0; c < 4; c++) {
    const b0 = b[c * 4], b1 = b[c * 4 + 1], b2 = b[c * 4 + 2], b3 = b[c * 4 + 3];
    for (let r = 0; r < 4; r++) t[c * 4 + r] = a[r] * b0 + a[4 + r] * b1 + a[8 + r] * b2 + a[12 + r] * b3;
  }
  return t;
}

function compose(p, q, s) {         // translation, quaternion xyzw, scale -> matrix
  const x = q[0], y = q[1], z = q[2], w = q[3];
  const x2 = x + x, y2 = y + y, z2 = z + z;
  const xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2;
  const wx = w * x2, wy = w * y2, wz = w * z2;
  return [
    (1 - (yy + zz)) * s[0], (xy + wz) * s[0], (xz - wy) * s[0], 0,
    (xy - wz) * s[1], (1 - (xx + zz)) * s[1], (yz + wx) * s[1], 0,
    (xz + wy) * s[2], (yz - wx) * s[2], (1 - (xx + yy)) * s[2], 0,
    p[0], p[1], p[2], 1,
  ];
}

function determinant(e) {
  const n11 = e[0], n12 = e[4], n13 = e[8], n14 = e[12];
  const n21 = e[1], n22 = e[5], n23 = e[9], n24 = e[13];
  const n31 = e[2], n32 = e[6], n33 = e[10], n34 = e[14];
  const n41 = e[3], n42 = e[7], n43 = e[11], n44 = e[15];
  return (
    n41 * (+n14 * n23 * n32 - n13 * n24 * n32 - n14 * n22 * n33 + n12 * n24 * n33 + n13 * n22 * n34 - n12 * n23 * n34) +
    n42 * (+n11 * n23 * n34 - n11 * n24 * n33 + n14 * n21 * n33 - n13 * n21 * n34 + n13 * n24 * n31 - n14 * n23 * n31) +
    n43 * (+n11 * n24 * n32 - n11 * n22 * n34 - n14 * n21 * n32 + n12 * n21 * n34 + n14 * n22 * n31 - n12 * n24 * n31) +
    n44 * (-n13 * n22 * n31 - n11 * n23 * n32 + n11 * n22 * n33 + n13 * n21 * n32 - n12 * n21 * n33 + n12 * n23 * n31)
  );
}

function quaternionFromRotation(m) {
  const m11 = m[0], m12 = m[4], m13 = m[8], m21 = m[1], m22 = m[5], m23 = m[9], m31 = m[2], m32 = m[6], m33 = m[10];
  const trace = m11 + m22 + m33;
  if (trace > 0) {
    const s = 0.5 / Math.sqrt(trace + 1.0);
    return [(m32 - m23) * s, (m13 - m31) * s, (m21 - m12) * s, 0.25 / s];
  } else if (m11 > m22 && m11 > m33) {
    const s = 2.0 * Math.sqrt(1.0 + m11 - m22 - m33);
    return [0.25 * s, (m12 + m21) / s, (m13 + m31) / s, (m32 - m23) / s];
  } else if (m22 > m33) {
    const s = 2.0 * Math.sqrt(1.0 + m22 - m11 - m33);
    return [(m12 + m21) / s, 0.25 * s, (m23 + m32) / s, (m13 - m31) / s];
  }
  const s = 2.0 * Math.sqrt(1.0 + m33 - m11 - m22);
  return [(m13 + m31) / s, (m23 + m32) / s, 0.25 * s, (m21 - m12) / s];
}

function decompose(e) {             // matrix -> { p, q, s }
  let sx = Math.sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
  const sy = Math.sqrt(e[4] * e[4] + e[5] * e[5] + e[6] * e[6]);
  const sz = Math.sqrt(e[8] * e[8] + e[9] * e[9] + e[10] * e[10]);
  if (determinant(e) < 0) sx = -sx;
  const r = e.slice();
  const ix = 1 / sx, iy = 1 / sy, iz = 1 / sz;
  r[0] *= ix; r[1] *= ix; r[2] *= ix; r[4] *= iy; r[5] *= iy; r[6] *= iy; r[8] *= iz; r[9] *= iz; r[10] *= iz;
  return { p: [e[12], e[13], e[14]], q: quaternionFromRotation(r), s: [sx, sy, sz] };
}

function applyMatrix4(v, e) {       // with perspective divide
  const x = v[0], y = v[1], z = v[2];
  const w = 1 / (e[3] * x + e[7] * y + e[11] * z + e[15]);
  return [(e[0] * x + e[4] * y + e[8] * z + e[12]) * w, (e[1] * x + e[5] * y + e[9] * z + e[13]) * w, (e[2] * x + e[6] * y + e[10] * z + e[14]) * w];
}

// ---- containers (GLB / .gltf), buffers, accessors ---------------------------------------------------------------------------------
function parseGLB(buf) {
  if (buf.length < 20 || buf.readUInt32LE(0) !== 0x46546c67) throw new Error("not a GLB file (bad magic)");
  const version = buf.readUInt32LE(4);
  if (version !== 2) throw new Error("unsupported glTF container version " + version);
  const total = Math.min(buf.readUInt32LE(8), buf.length);
  let off = 12, json = null, bin = null;
  while (off + 8 <= total) {
    const len = buf.readUInt32LE(off), type = buf.readUInt32LE(off + 4);
    const body = buf.slice(off + 8, off + 8 + len);
    if (type === 0x4e4f534a && json === null) json = JSON.parse(body.toString("utf8"));
    else if (type === 0x004e4942 && bin === null) bin = body;
    off += 8 + len + ((4 - (len % 4)) % 4);
  }
  if (!json) throw new Error("GLB has no JSON chunk");
  return { json: json, bin: bin };
}

// GLTFLoader.parse: a file that starts with the magic "glTF" is the binary container, anything else is JSON text.
// Buffers as GLTFParser.loadBuffer resolves them: no uri on buffer 0 = the GLB's BIN chunk; data: URIs decoded in place; anything
// else is a path relative to the file (LoaderUtils.resolveURL; absolute paths stay).
function parseFile(buf, baseDir) {
  let gltf;
  if (buf.length >= 4 && buf.toString("latin1", 0, 4) === "glTF") gltf = parseGLB(buf);
  else {
    let json;
    try { json = JSON.parse(buf.toString("utf8")); } catch (e) { throw new Error("not a GLB file (bad magic) and not glTF JSON: " + e.message); }
    gltf = { json: json, bin: null };
  }
  const json = gltf.json;
  if (!json.asset || !(Number(json.asset.version) >= 2)) throw new Error("unsupported glTF asset version (2.0 or later is required)");
  gltf.buffers = (json.buffers || []).map(function (def, i) {
    if (def.uri === undefined) {
      if (i === 0 && gltf.bin) return gltf.bin;
      throw new Error("buffer " + i + " has no uri and is not the GLB's binary chunk");
    }
    const m = /^data:[^,]*?(;base64)?,(.*)$/.exec(def.uri);
    if (m) return m[1] ? Buffer.from(m[2], "base64") : Buffer.from(decodeURIComponent(m[2]), "latin1");
    if (/^(https?|blob):/i.test(def.uri)) throw new Error("buffer " + i + ": remote uris are not supported (" + def.uri.slice(0, 40) + ")");
    const rel = decodeURIComponent(def.uri);
    try { return fs.readFileSync(path.isAbsolute(rel) ? rel : path.join(baseDir || ".", rel)); }
    catch (e) {      // the scene file IS there: its missing buffer is an unreadable scene, not an absent one (drivers fall back only on err.code === "ENOENT")
      const err = new Error("buffer " + i + " (" + def.uri + ") cannot be read: " + e.message); err.code = "PT_GLTF_BUFFER"; throw err;
    }
  });
  return gltf;
}

const COMPONENT = { 5120: [1, "Int8"], 5121: [1, "UInt8"], 5122: [2, "Int16LE"], 5123: [2, "UInt16LE"], 5125: [4, "UInt32LE"], 5126: [4, "FloatLE"] };
const TYPE_SIZE = { SCALAR: 1, VEC2: 2, VEC3: 3, VEC4: 4, MAT2: 4, MAT3: 9, MAT4: 16 };

function viewBytes(gltf, viewIndex, what) {
  const view = (gltf.json.bufferViews || [])[viewIndex];
  if (!view) throw new Error("missing bufferView " + viewIndex + " (" + what + ")");
  const ext = view.extensions || {};
  for (const name of ["EXT_meshopt_compression", "KHR_meshopt_compression"])
    if (ext[name] && (gltf.json.extensionsRequired || []).indexOf(name) >= 0) throw new Error("meshopt-compressed buffer views are not supported (" + name + " is required by this file)");
  const buffers = gltf.buffers || (gltf.bin ? [gltf.bin] : []);
  const data = buffers[view.buffer || 0];
  if (!data) throw new Error("bufferView " + viewIndex + " names buffer " + (view.buffer || 0) + ", which this file does not have");
  return { data: data, offset: view.byteOffset || 0, stride: view.byteStride };
}

// GLTFParser.loadAccessor: the RAW component values (no `normalized` scaling: see the header), `count` elements of TYPE_SIZE
// components; an accessor without a buffer view is zeros; `sparse` replaces the listed elements (values tightly packed).
function readAccessor(gltf, index) {
  const acc = (gltf.json.accessors || [])[index];
  if (!acc) throw new Error("missing accessor " + index);
  const comp = COMPONENT[acc.componentType];
  if (!comp) throw new Error("unsupported accessor component type " + acc.componentType);
  const n = TYPE_SIZE[acc.type], count = acc.count;
  if (!n) throw new Error("unsupported accessor type " + acc.type);
  const out = new Array(count * n);
  const rd = "read" + comp[1];
  if (acc.bufferView === undefined) { for (let i = 0; i < out.length; i++) out[i] = 0; }
  else {
    const v = viewBytes(gltf, acc.bufferView, "accessor " + index);
    const base = v.offset + (acc.byteOffset || 0);
    const stride = v.stride || comp[0] * n;
    if (count > 0 && base + (count - 1) * stride + comp[0] * n > v.data.length) throw new Error("accessor " + index + " reads beyond the end of its buffer");
    for (let i = 0; i < count; i++)
      for (let k = 0; k < n; k++) out[i * n + k] = v.data[rd](base + i * stride + k * comp[0]);
  }
  if (acc.sparse !== undefined) {
    const sp = acc.sparse, icomp = COMPONENT[sp.indices.componentType];
    if (!icomp || sp.indices.componentType === 5126) throw new Error("unsupported sparse index component type " + sp.indices.componentType);
    const vi = viewBytes(gltf, sp.indices.bufferView, "sparse indices of accessor " + index);
    const vv = viewBytes(gltf, sp.values.bufferView, "sparse values of accessor " + index);
    const ib = vi.offset + (sp.indices.byteOffset || 0), vb = vv.offset + (sp.values.byteOffset || 0);
    for (let i = 0; i < sp.count; i++) {
      const at = vi.data["read" + icomp[1]](ib + i * icomp[0]);
      if (at >= count) continue;                   // a typed-array store beyond the end is dropped (BufferAttribute.setX)
      for (let k = 0; k < n; k++) out[at * n + k] = vv.data[rd](vb + (i * n + k) * comp[0]);
    }
  }
  return { data: out, n: n, count: count };
}

class Scene {
  constructor() {
    this.loader = null;             // the reference keeps a GLTFLoader here (Scene.js:6)
    this.triangles = [];
    this._normalizeEnabled = false;
    this._normalizeMode = "cube";   // "cube" | "sphere"
  }

  // Scene.js:15-42.  `url` is a file path; "/assets/x.glb" style URLs resolve against
  // options.assetRoot, $PT_ASSET_ROOT or ./public like the reference's static server.
  async loadGLB(url, options) {
    const o = options || {};
    this._normalizeEnabled = (o.normalize !== undefined && o.normalize !== null) ? o.normalize : false;
    this._normalizeMode = (o.mode !== undefined && o.mode !== null) ? o.mode : "cube";
    let file = url;
    if (!fs.existsSync(file)) {
      const root = o.assetRoot || process.env.PT_ASSET_ROOT || "public";
      const alt = path.join(root, url.replace(/^\/+/, ""));
      if (fs.existsSync(alt)) file = alt;
    }
    let gltf;
    try { gltf = parseFile(fs.readFileSync(file), path.dirname(file)); }
    catch (err) { console.error("GLB load failed:", err); throw err; }      // Scene.js:27-30 (a missing file keeps fs's err.code === "ENOENT")
    this.parseGLTF(gltf);
    if (this._normalizeEnabled) {
      console.log("Normalizing mesh (" + this._normalizeMode + ")...");
      this.normalizeMesh();
    }
  }

  // Scene.js:47-99.  `gltf` = { json, bin } from parseGLB.
  parseGLTF(gltf) {
    const json = gltf.json;
    this.triangles.length = 0;
    const sceneIndex = json.scene !== undefined ? json.scene : 0;
    const scene = (json.scenes || [])[sceneIndex];
    if (!scene) return;
    const self = this;
    function localMatrix(node) {
      if (node.matrix !== undefined) {             // applyMatrix4 + updateMatrix: decompose, then compose
        const d = decompose(multiply(node.matrix.slice(), identity()));
        return compose(d.p, d.q, d.s);
      }
      return compose(node.translation || [0, 0, 0], node.rotation || [0, 0, 0, 1], node.scale || [1, 1, 1]);
    }
    function emitPrimitive(prim, world) {
      const mode = prim.mode === undefined ? 4 : prim.mode;
      if (mode > 6) throw new Error("THREE.GLTFLoader: Primitive mode unsupported: " + mode);     // what the reference's loader throws
      if (prim.extensions && prim.extensions.KHR_draco_mesh_compression) throw new Error("Draco-compressed meshes are not supported (the reference's GLTFLoader has no DRACOLoader either)");
      if (mode < 4) return;                        // points / lines are not meshes (Scene.js:54)
      if (prim.attributes.POSITION === undefined) return;
      const pos = readAccessor(gltf, prim.attributes.POSITION);
      if (pos.n !== 3) throw new Error("POSITION accessor is not VEC3");
      let index;
      if (prim.indices !== undefined) index = readAccessor(gltf, prim.indices).data;
      else { index = new Array(pos.count); for (let i = 0; i < pos.count; i++) index[i] = i; }
      let order = index;
      if (mode === 5 || mode === 6) {              // BufferGeometryUtils.toTrianglesDrawMode (GLTFLoader applies it to strips and fans)
        order = [];
        const nt = index.length - 2;
        if (mode === 6) for (let i = 1; i <= nt; i++) order.push(index[0], index[i], index[i + 1]);
        else for (let i = 0; i < nt; i++) { if (i % 2 === 0) order.push(index[i], index[i + 1], index[i + 2]); else order.push(index[i + 2], index[i + 1], index[i]); }
      }
      const nTri = Math.floor(order.length / 3);
      for (let t = 0; t < nTri; t++) {
        const v = [];
        for (let k = 0; k < 3; k++) {
          const i = order[t * 3 + k];
          if (!(i < pos.count)) throw new Error("triangle index " + i + " is outside its POSITION accessor (" + pos.count + " vertices)");
          // a float position attribute is a Float32Array in three: values pass through f32 (integers are exact either way)
          v.push(applyMatrix4([Math.fround(pos.data[i * 3]), Math.fround(pos.data[i * 3 + 1]), Math.fround(pos.data[i * 3 + 2])], world));
        }
        self.triangles.push({
          v0: v[0], v1: v[1], v2: v[2],
          centroid: [(v[0][0] + v[1][0] + v[2][0]) / 3, (v[0][1] + v[1][1] + v[2][1]) / 3, (v[0][2] + v[1][2] + v[2][2]) / 3],
        });
      }
    }
    function visit(nodeIndex, parentWorld, depth) {
      if (depth > 512) throw new Error("node hierarchy too deep (cycle?)");
      const node = json.nodes[nodeIndex];
      if (!node) return;
      const world = multiply(parentWorld, localMatrix(node));
      if (node.mesh !== undefined) {
        const mesh = json.meshes[node.mesh];
        for (const prim of (mesh.primitives || [])) emitPrimitive(prim, world);
      }
      for (const c of (node.children || [])) visit(c, world, depth + 1);
    }
    for (const n of (scene.nodes || [])) visit(n, identity(), 0);
  }

  // Scene.js:104-165
  normalizeMesh() {
    if (this.triangles.length === 0) return;
    const min = [Infinity, Infinity, Infinity], max = [-Infinity, -Infinity, -Infinity];
    for (const t of this.triangles)
      for (const v of [t.v0, t.v1, t.v2])
        for (let k = 0; k < 3; k++) { min[k] = Math.min(min[k], v[k]); max[k] = Math.max(max[k], v[k]); }
    const center = [(min[0] + max[0]) * 0.5, (min[1] + max[1]) * 0.5, (min[2] + max[2]) * 0.5];
    const maxDim = Math.max(max[0] - min[0], max[1] - min[1], max[2] - min[2]);
    let scale = 2.0 / maxDim;                                     // cube: [-1,1]
    if (this._normalizeMode === "sphere") scale = 1.0 / (maxDim * 0.5);   // same value, as in the reference (:139-144)
    for (const t of this.triangles) {
      for (const v of [t.v0, t.v1, t.v2]) for (let k = 0; k < 3; k++) v[k] = (v[k] - center[k]) * scale;
      t.centroid = [(t.v0[0] + t.v1[0] + t.v2[0]) / 3, (t.v0[1] + t.v1[1] + t.v2[1]) / 3, (t.v0[2] + t.v1[2] + t.v2[2]) / 3];
    }
    console.log("Mesh normalized.");
  }

  // Scene.js:169-224 (the reference keeps its call commented out, :41; the method is part of the class all the same).  Orders the triangles along
  // a 30-bit Morton curve of their centroids inside the centroids' bounding box.  Same arithmetic as the reference, key by key: the normalised coordinate
  // times 1024, clamped to [0, 1023] and NOT truncated before the bit spreading -- each `(v * k) & mask` step truncates its own product (ToInt32), so
  // the fraction takes part in the first step.  The reference hands a comparator that recomputes both codes to Array.prototype.sort; the codes are
  // computed once per triangle here and sorted with the same stable sort (V8's), which gives the same order.
  sortTriangles() {
    if (this.triangles.length === 0) return;
    console.log("Sorting " + this.triangles.length + " triangles spatially...");
    const lo = [Infinity, Infinity, Infinity], hi = [-Infinity, -Infinity, -Infinity];
    for (const t of this.triangles)
      for (let k = 0; k < 3; k++) { lo[k] = Math.min(lo[k], t.centroid[k]); hi[k] = Math.max(hi[k], t.centroid[k]); }
    const extent = [hi[0] - lo[0] || 1, hi[1] - lo[1] || 1, hi[2] - lo[2] || 1];
    const spread = function (v) {
      v = (v * 0x00010001) & 0xff0000ff; v = (v * 0x00000101) & 0x0f00f00f; v = (v * 0x00000011) & 0xc30c30c3; v = (v * 0x00000005) & 0x49249249;
      return v;
    };
    const axis = function (c, k) { return spread(Math.min(Math.max(((c - lo[k]) / extent[k]) * 1024, 0), 1023)); };
    const keyed = this.triangles.map(function (t) { return { t: t, code: axis(t.centroid[0], 0) | (axis(t.centroid[1], 1) << 1) | (axis(t.centroid[2], 2) << 2) }; });
    keyed.sort(function (a, b) { return a.code - b.code; });
    for (let i = 0; i < keyed.length; i++) this.triangles[i] = keyed[i].t;
    console.log("Sorting complete.");
  }

  // Scene.js:230-241
  getTrianglesFloat32() {
    const arr = new Float32Array(this.triangles.length * 9);
    let o = 0;
    for (const t of this.triangles) {
      arr[o++] = t.v0[0]; arr[o++] = t.v0[1]; arr[o++] = t.v0[2];
      arr[o++] = t.v1[0]; arr[o++] = t.v1[1]; arr[o++] = t.v1[2];
      arr[o++] = t.v2[0]; arr[o++] = t.v2[1]; arr[o++] = t.v2[2];
    }
    return arr;
  }

  getTriangles() { return this.triangles; }         // Scene.js:243

  // procedural stand-in when the reference's dragon.glb / Sponza are absent (SURVEY.md 0.3)
  setTrianglesFloat32(f32) {
    this.triangles = [];
    for (let i = 0; i + 8 < f32.length; i += 9)
      this.triangles.push({ v0: [f32[i], f32[i + 1], f32[i + 2]], v1: [f32[i + 3], f32[i + 4], f32[i + 5]], v2: [f32[i + 6], f32[i + 7], f32[i + 8]], centroid: [0, 0, 0] });
    this._raw = f32;
  }
}

module.exports = { Scene, parseGLB, parseFile };
