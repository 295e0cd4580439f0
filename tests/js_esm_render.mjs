// node tests/js_esm_render.mjs <glb> <out dir> : src/main.js's sequence through the ES-module face, on the GPU -- writes the triangles and the frame
import * as PT from "../raytracer-public_amd/js/libs/PathTracer.js";
import * as PTScene from "../raytracer-public_amd/js/libs/Scene.js";
import fs from "fs";
import path from "path";

async function main() {
  const log = console.log; console.log = () => {};
  const pathTracer = new PT.PathTracer({ width: 160, height: 96 });                 // default options: the reference's frame (one primary ray per pixel)
  await pathTracer.initialize();
  const scene = new PTScene.Scene();
  await scene.loadGLB(process.argv[2], { normalize: true, mode: "cube" });
  await pathTracer.setScene(scene);
  pathTracer.setCameraPosition(0.3, 0.2, 2.5);
  pathTracer.setCameraQuaternion(0, 0, 0, 1);
  pathTracer.setFrameCount(3);
  await pathTracer.render();
  fs.writeFileSync(path.join(process.argv[3], "img.bin"), Buffer.from(pathTracer.readRadiance().buffer));
  fs.writeFileSync(path.join(process.argv[3], "tris.bin"), Buffer.from(pathTracer.trianglesData.buffer));
  log("ok " + ((pathTracer.trianglesData.length / 9) | 0));
  pathTracer.destroy();
}
main().catch((e) => { console.error(e); process.exit(1); });
