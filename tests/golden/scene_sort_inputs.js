// Deterministic triangle sets for Scene.sortTriangles (shared by gen_golden_scene_sort.js and tests/test_js_gltf_golden.py): objects of the shape
// Scene.parseGLTF makes ({ v0, v1, v2, centroid }), from an own LCG.  Node-12-safe CommonJS.
"use strict";
const { rng } = require("./synth_gltf.js");
function soup(seed, n, scale, shift) {
  return function () {
    const r = rng(seed), out = [];
    for (let i = 0; i < n; i++) {
      const c = [(r() * 2 - 1) * scale[0] + shift[0], (r() * 2 - 1) * scale[1] + shift[1], (r() * 2 - 1) * scale[2] + shift[2]];
      const v = [0, 1, 2].map(() => [c[0] + (r() - 0.5) * 0.1, c[1] + (r() - 0.5) * 0.1, c[2] + (r() - 0.5) * 0.1]);
      out.push({ v0: v[0], v1: v[1], v2: v[2], centroid: [(v[0][0] + v[1][0] + v[2][0]) / 3, (v[0][1] + v[1][1] + v[2][1]) / 3, (v[0][2] + v[1][2] + v[2][2]) / 3] });
    }
    return out;
  };
}
const inputs = {
  uniform_2000: soup(11, 2000, [1, 1, 1], [0, 0, 0]),
  flat_z_500: function () { const t = soup(12, 500, [3, 0.5, 1], [10, -2, 0])(); for (const x of t) x.centroid[2] = 0.25; return t; },     // zero extent on an axis: `|| 1`
  clustered_1500: function () { const a = soup(13, 750, [0.01, 0.01, 0.01], [-5, -5, -5])(), b = soup(14, 750, [0.01, 0.01, 0.01], [5, 5, 5])(); return a.concat(b); },   // many equal codes: stability
  one: soup(15, 1, [1, 1, 1], [0, 0, 0]),
  duplicates_300: function () { const t = soup(16, 100, [1, 1, 1], [0, 0, 0])(); return t.concat(t.map((x) => Object.assign({}, x)), t.map((x) => Object.assign({}, x))); },
};
module.exports = { inputs };
