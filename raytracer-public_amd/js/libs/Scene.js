// libs/Scene.js -- the ES-module face of ../Scene.js (src/main.js:2: `import * as PTScene from "./libs/Scene.js";`); see libs/PathTracer.js.
import cjs from "../Scene.js";
export const Scene = cjs.Scene;
export const parseGLB = cjs.parseGLB, parseFile = cjs.parseFile;
export default cjs;
