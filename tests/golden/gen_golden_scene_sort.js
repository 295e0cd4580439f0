// gen_golden_scene_sort.js -- golden triangle ORDER for Scene.sortTriangles (src/libs/Scene.js:169-224), produced by the reference's own method body.
// Scene.js cannot be imported under Node 12 (`??` in loadGLB), and importing it would pull in three; this script reads the file as text, cuts the
// sortTriangles method out AT GENERATION TIME, turns it into a function and runs it on deterministic triangle sets.  Nothing of the reference's text is
// stored: the output is the resulting permutation (indices) per input set -> scene_sort_golden.json.
//   node tests/golden/gen_golden_scene_sort.js [/root/reference]
"use strict";
const fs = require("fs"), path = require("path");
const REF = process.argv[2] || "/root/reference";
const src = fs.readFileSync(path.join(REF, "src/libs/Scene.js"), "utf8");
const start = src.indexOf("sortTriangles() {");
if (start < 0) throw new Error("sortTriangles not found");
let i = src.indexOf("{", start), depth = 0, end = -1;
for (; i < src.length; i++) { if (src[i] === "{") depth++; else if (src[i] === "}") { depth--; if (depth === 0) { end = i; break; } } }
const body = src.slice(src.indexOf("{", start) + 1, end);
const sortTriangles = new Function(body);            // `this.triangles` inside: called with a { triangles } receiver
const { inputs } = require("./scene_sort_inputs.js");
const out = {};
const log = console.log; console.log = () => {};
for (const name of Object.keys(inputs)) {
  const tris = inputs[name]().map((t, k) => Object.assign(t, { id: k }));
  const recv = { triangles: tris };
  sortTriangles.call(recv);
  out[name] = recv.triangles.map((t) => t.id);
}
console.log = log;
fs.writeFileSync(path.join(__dirname, "scene_sort_golden.json"), JSON.stringify(out));
console.log("wrote scene_sort_golden.json", Object.keys(out).map((k) => k + ":" + out[k].length).join(" "));
