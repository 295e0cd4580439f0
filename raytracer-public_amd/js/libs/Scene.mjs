export * from "./Scene.js";
export { default } from "./Scene.js";
