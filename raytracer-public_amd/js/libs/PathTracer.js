// libs/PathTracer.js -- the ES-module face of ../PathTracer.js, at the path and under the name the reference's driver imports
// (src/main.js:1: `import * as PT from "./libs/PathTracer.js";`): with this package's js/ directory in the place of the reference's src/,
// that line resolves as written.  (libs/package.json makes the .js files of this directory ES modules; the implementation stays the
// CommonJS module next door, loaded once by Node whichever face is used.  Node >= 12.17.)
import cjs from "../PathTracer.js";
export const PathTracer = cjs.PathTracer;
export const MODE_REFERENCE_PACKET = cjs.MODE_REFERENCE_PACKET, MODE_REFERENCE = cjs.MODE_REFERENCE, MODE_PATH = cjs.MODE_PATH;
export const native = cjs.native;
export default cjs;
