// gen_golden_js.js -- generates tests/golden/pathtracer_js_golden.json by IMPORTING the
// reference's own src/libs/PathTracer.js (CPU methods only) under Node.
//
//   node tests/golden/gen_golden_js.js [/root/reference]
//
// The reference's package.json has no "type":"module", so Node will not load its ES-module
// sources in place; this script makes a run-time scratch copy of PathTracer.js + io.js in
// os.tmpdir() next to a {"type":"module"} package.json, imports that, and deletes it.  Nothing
// of the reference is written into this repository: the committed fixture holds only inputs
// (as u32 bit patterns) and the outputs the reference code produced for them.
"use strict";
const fs = require("fs");
const os = require("os");
const path = require("path");

const REF = process.argv[2] || "/root/reference";
const OUT = path.join(__dirname, "pathtracer_js_golden.json");

// deterministic LCG so the fixture is reproducible
let lcgState = 0x12345678 >>> 0;
function lcg() { lcgState = (Math.imul(lcgState, 1664525) + 1013904223) >>> 0; return lcgState; }
function frand() { return (lcg() >>> 8) / 16777216; }

function f32bits(arr) { return Array.from(new Uint32Array(new Float32Array(arr).buffer)); }

function randomTriangles(n, opts) {
  const t = new Float32Array(n * 9);
  for (let i = 0; i < n; i++) {
    const cx = frand() * 2 - 1, cy = frand() * 2 - 1, cz = (opts.flat ? 0.25 : frand() * 2 - 1);
    const s = opts.size;
    for (let v = 0; v < 3; v++) {
      t[i * 9 + v * 3 + 0] = cx + (frand() - 0.5) * s;
      t[i * 9 + v * 3 + 1] = cy + (frand() - 0.5) * s;
      t[i * 9 + v * 3 + 2] = cz + (opts.flat ? 0 : (frand() - 0.5) * s);
    }
  }
  if (opts.dupes) { // exact duplicate triangles -> equal Morton codes -> index tie-break
    for (let i = 0; i < opts.dupes; i++) {
      const src = (lcg() % n), dst = (lcg() % n);
      for (let k = 0; k < 9; k++) t[dst * 9 + k] = t[src * 9 + k];
    }
  }
  return t;
}

// random LBVH2-shaped buffer: internal nodes 0..n-2 (root 0), leaves n-1..2n-2, 6 words/node,
// f16 bounds (incl. subnormals / negative values), leaf meta = LEAF|tri, leaf children = 0
function randomBVH2(n) {
  const LEAF = 0x80000000;
  const numNodes = 2 * n - 1;
  const u = new Uint32Array(1 + 6 * numNodes);
  u[0] = numNodes;
  const h = (x) => x & 0xffff;
  function randF16(lo) { // finite f16 bit pattern; some subnormals, some negatives
    const r = lcg() % 100;
    let e, m = lcg() & 0x3ff;
    if (r < 8) e = 0; else e = 8 + (lcg() % 8);      // subnormal or ~[2^-7, 2^0]
    if (r < 3) m = 0;                                 // exact +0 / -0 (Math.min(+0,-0) = -0 matters)
    const s = (lcg() & 1) ? 0x8000 : 0;
    return h(s | (e << 10) | m);
  }
  const f16val = (b) => { if ((b & 0x7fff) === 0) return (b & 0x8000) ? -1e-30 : 1e-30; const s = (b & 0x8000) ? -1 : 1, e = (b >> 10) & 31, m = b & 0x3ff; return e === 0 ? s * m * Math.pow(2, -24) : s * (1 + m / 1024) * Math.pow(2, e - 15); };
  const bounds = new Array(numNodes);
  let nextInternal = 0;
  function build(first, last) { // leaves [first,last]
    if (first === last) {
      const id = n - 1 + first;
      const b = [];
      for (let k = 0; k < 3; k++) { let a = randF16(), c = randF16(); if (f16val(a) > f16val(c)) { const t = a; a = c; c = t; } b.push([a, c]); }
      bounds[id] = b;
      const off = 1 + id * 6;
      u[off + 0] = (b[0][0] | (b[1][0] << 16)) >>> 0; u[off + 1] = (b[2][0] | (b[0][1] << 16)) >>> 0; u[off + 2] = (b[1][1] | (b[2][1] << 16)) >>> 0;
      u[off + 3] = 0; u[off + 4] = 0; u[off + 5] = (LEAF | (lcg() % n)) >>> 0;
      return id;
    }
    const id = nextInternal++;
    const split = first + (lcg() % (last - first));
    const l = build(first, split), r = build(split + 1, last);
    const b = [];
    for (let k = 0; k < 3; k++) {
      const lo = f16val(bounds[l][k][0]) <= f16val(bounds[r][k][0]) ? bounds[l][k][0] : bounds[r][k][0];
      const hi = f16val(bounds[l][k][1]) >= f16val(bounds[r][k][1]) ? bounds[l][k][1] : bounds[r][k][1];
      b.push([lo, hi]);
    }
    bounds[id] = b;
    const off = 1 + id * 6;
    u[off + 0] = (b[0][0] | (b[1][0] << 16)) >>> 0; u[off + 1] = (b[2][0] | (b[0][1] << 16)) >>> 0; u[off + 2] = (b[1][1] | (b[2][1] << 16)) >>> 0;
    u[off + 3] = l; u[off + 4] = r; u[off + 5] = 0;
    return id;
  }
  if (n === 1) { build(0, 0); } else { build(0, n - 1); }
  return u;
}

async function main() {
  const scratch = fs.mkdtempSync(path.join(os.tmpdir(), "ptref-"));
  try {
    for (const f of ["PathTracer.js", "io.js"]) fs.copyFileSync(path.join(REF, "src/libs", f), path.join(scratch, f));
    fs.writeFileSync(path.join(scratch, "package.json"), '{"type":"module"}');
    const mod = await import(path.join(scratch, "PathTracer.js"));
    const pt = new mod.PathTracer({ width: 4, height: 4 });

    const golden = { generator: "tests/golden/gen_golden_js.js", node: process.version, reference_files: ["src/libs/PathTracer.js:227-238,411-481,506-667"], sizing: [], morton: [], collapse: [] };

    for (const n of [0, 1, 4, 871414]) golden.sizing.push({ numTris: n, bvh2: pt.computeBVH2Sizing(n), bvh4_of_numNodes2: pt.computeBVH4Sizing(n > 0 ? 2 * n - 1 : 0) });

    const mortonCases = [
      { name: "default_tetrahedron", tris: pt.trianglesData },
      { name: "random_64", tris: randomTriangles(64, { size: 0.2 }) },
      { name: "random_700_dupes", tris: randomTriangles(700, { size: 0.05, dupes: 120 }) },
      { name: "flat_300", tris: randomTriangles(300, { size: 0.1, flat: true }) },
      { name: "single", tris: randomTriangles(1, { size: 0.5 }) },
    ];
    for (const c of mortonCases) {
      const r = pt.buildMortonAndSort(c.tris);
      golden.morton.push({ name: c.name, tris_f32_bits: f32bits(c.tris), mortonSorted: Array.from(r.mortonSorted), triIndexSorted: Array.from(r.triIndexSorted) });
    }

    // tetra BVH2 quoted in SURVEY.md section 8c
    const tet = new Uint32Array(1 + 6 * 7); tet[0] = 7;
    const B = [0xbc01bc01, 0x3c01bc01, 0x3c013c01];
    const topo = [[1, 2], [3, 4], [5, 6]];
    for (let i = 0; i < 7; i++) { const o = 1 + i * 6; tet[o] = B[0]; tet[o + 1] = B[1]; tet[o + 2] = B[2]; if (i < 3) { tet[o + 3] = topo[i][0]; tet[o + 4] = topo[i][1]; tet[o + 5] = 0; } else { tet[o + 3] = 0; tet[o + 4] = 0; tet[o + 5] = (0x80000000 | [3, 0, 2, 1][i - 3]) >>> 0; } }
    const collapseCases = [{ name: "tetra_survey", n: 4, bvh2: tet }];
    for (const n of [1, 2, 3, 5, 17, 100, 333]) collapseCases.push({ name: "random_" + n, n: n, bvh2: randomBVH2(n) });
    for (const c of collapseCases) {
      const r = pt.collapseLBVH2ToBVH4(c.bvh2, c.n);
      golden.collapse.push({ name: c.name, numTris: c.n, bvh2: Array.from(c.bvh2), numNodes4: r.numNodes4, bvh4: Array.from(r.bvh4U32) });
    }
    const r0 = pt.collapseLBVH2ToBVH4(new Uint32Array([0]), 0);
    golden.collapse.push({ name: "empty", numTris: 0, bvh2: [0], numNodes4: r0.numNodes4, bvh4: Array.from(r0.bvh4U32) });

    fs.writeFileSync(OUT, JSON.stringify(golden));
    console.log("wrote", OUT, fs.statSync(OUT).size, "bytes");
  } finally {
    for (const f of fs.readdirSync(scratch)) fs.unlinkSync(path.join(scratch, f));
    fs.rmdirSync(scratch);
  }
}
main().catch((e) => { console.error(e); process.exit(1); });
