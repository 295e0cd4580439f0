// the same face for hosts that name ES modules by extension (import * as PT from ".../libs/PathTracer.mjs")
export * from "./PathTracer.js";
export { default } from "./PathTracer.js";
