// PathTracer.js -- Node host class with the API of the reference's src/libs/PathTracer.js,
// driving libmi355pt (HIP, MI355X) through the N-API addon instead of WebGPU.
// Node-12-safe CommonJS (no ??, ?., top-level await).  Method names, arity, return shapes and
// Promise-ness follow the reference (cited per method); the extension options (mode, spp,
// maxBounces, seed, accumulate) default to the reference's behaviour: one primary ray per pixel.
"use strict";
const path = require("path");

// overlapped frame slots need more than ROCm's default 4 hardware queues (read at HIP runtime init)
if (process.env.GPU_MAX_HW_QUEUES === undefined) process.env.GPU_MAX_HW_QUEUES = "12";
let addon = null;
function native() {
  if (!addon) addon = require(path.join(__dirname, "..", "napi", "mi355pt.node"));   // throws loudly when not built
  return addon;
}

const MODE_REFERENCE_PACKET = 0, MODE_REFERENCE = 1, MODE_PATH = 2;

class PathTracer {
  // reference: constructor(canvas), PathTracer.js:60-95 -- uses only canvas.width / canvas.height
  constructor(canvas, options) {
    this.canvas = canvas;
    this.device = null;                         // the reference's GPUDevice slot: here the native context handle
    this.cameraPosition = [0.0, 0.0, 3.5];      // :67
    this.cameraQuaternion = [0.0, 0.0, 0.0, 1.0];
    this.buffers = {};                          // kept for shape compatibility; device buffers live in the context
    this.frameCount = 0;
    this.trianglesData = new Float32Array([     // default tetrahedron, :79-84
      1, 1, 1, -1, -1, 1, -1, 1, -1,
      1, 1, 1, -1, 1, -1, 1, -1, -1,
      1, 1, 1, 1, -1, -1, -1, -1, 1,
      -1, -1, 1, 1, -1, -1, -1, 1, -1,
    ]);
    const o = options || {};
    this.options = {
      device: o.device === undefined ? -1 : o.device,
      mode: o.mode === undefined ? MODE_REFERENCE : o.mode,
      spp: o.spp || 1, maxBounces: o.maxBounces || 0, seed: o.seed === undefined ? 1 : o.seed,
      accumulate: !!o.accumulate, stats: !!o.stats, bruteForce: !!o.bruteForce,
    };
    // Several GPUs, still one image per render(): `gpus: N` (devices 0..N-1) or `devices: [..]` makes this PathTracer drive a
    // group of contexts -- pixel tiles interleaved over the GPUs, gathered on the first one over RCCL (pt_group_*, include/mi355pt.h).
    // `transport: "copy"` swaps the collective for peer copies (members may then share a GPU: rehearsals on a one-GPU machine).
    this.groupDevices = o.devices ? Int32Array.from(o.devices) : (o.gpus && o.gpus > 1 ? Int32Array.from({ length: o.gpus }, (_, i) => i) : null);
    this.groupTransport = o.transport === "copy" ? 1 : 0;
    this.group = null;
    this._hasBVH = false;
  }

  // :97-102 (adapter/device/shaders/buffers/pipelines) -> one native context per GPU
  async initialize() {
    if (this.groupDevices) { this.group = native().groupCreate(this.groupDevices, this.groupTransport); this.device = this.group; }
    else this.device = native().create(this.options.device);
  }

  computeBVH2Sizing(numTris) { return native().computeBVH2Sizing(numTris); }      // :227
  computeBVH4Sizing(numNodes4) { return native().computeBVH4Sizing(numNodes4); }  // :234

  buildMortonAndSort(trianglesData) {            // :427 -> { mortonSorted, triIndexSorted }
    return native().mortonSort(trianglesData);
  }

  async readBVH2(bytes) {                        // :485 -> fresh Uint32Array copy
    return this.group ? native().groupReadBVH2(this.group, bytes) : native().readBVH2(this.device, bytes);
  }

  collapseLBVH2ToBVH4(bvh2U32, numTris) {        // :506 -> { bvh4U32, numNodes4 }
    return native().collapse(bvh2U32, numTris);
  }

  async buildBVH(trianglesData) {                // :671-749
    if (!this.device) return;                    // `if (!device) return`, :673
    const t0 = Date.now();
    if (this.group) { native().groupSetTriangles(this.group, trianglesData); native().groupBuildBVH(this.group); }
    else { native().setTriangles(this.device, trianglesData); native().buildBVH(this.device); }
    this._hasBVH = true;
    console.log("BVH Build Time:", Date.now() - t0, "ms");   // :745-748 prints timings
  }

  async setScene(scene) {                        // :751-754
    this.trianglesData = scene.getTrianglesFloat32();
    await this.buildBVH(this.trianglesData);
  }

  // config C1 extension: brute-force scene = uploaded triangles + analytic spheres, no BVH
  setBruteForceScene(trianglesData, spheresXYZR) {
    if (this.group) throw new Error("brute-force scenes (config C1) render on one GPU");
    this.trianglesData = trianglesData;
    native().setTriangles(this.device, trianglesData);
    native().setSpheres(this.device, spheresXYZR);
    this.options.bruteForce = true; this._hasBVH = true;
  }

  // setScene for a scene whose BVH2 was built before (data/BVH2.bin, what src/main.js:27-46 dumps): upload the triangles, install the
  // BVH2 through the reference's own route (collapseLBVH2ToBVH4, then the BVH4 the renderer traverses) -- no rebuild
  async setSceneWithBVH2(scene, bvh2U32) {
    if (!this.device) return;
    this.trianglesData = scene.getTrianglesFloat32();
    if (this.group) native().groupSetTriangles(this.group, this.trianglesData); else native().setTriangles(this.device, this.trianglesData);
    this.setBVH2(bvh2U32);
  }

  // install a prebuilt BVH (data/BVH2.bin or data/BVH4_wide.bin) instead of rebuilding
  setBVH2(bvh2U32) { if (this.group) native().groupSetBVH2(this.group, bvh2U32); else native().setBVH2(this.device, bvh2U32); this._hasBVH = true; }
  setBVH4(bvh4U32) { if (this.group) native().groupSetBVH4(this.group, bvh4U32); else native().setBVH4(this.device, bvh4U32); this._hasBVH = true; }

  async render() {                               // :756-822
    if (!this._hasBVH) return;                   // `if (!this.buffers.BVH) return`, :757
    const numTriangles = (this.trianglesData.length / 9) | 0;
    const fov = (70.0 * Math.PI) / 180;          // :761
    const focal = 1.0 / Math.tan(0.5 * fov);
    const UBO = new Float32Array([               // :764-787, same 16 floats in the same order
      this.canvas.width, this.canvas.height, focal, this.canvas.width / this.canvas.height,
      this.cameraPosition[0], this.cameraPosition[1], this.cameraPosition[2], numTriangles,
      this.cameraQuaternion[0], this.cameraQuaternion[1], this.cameraQuaternion[2], this.cameraQuaternion[3],
      this.frameCount, 0, 0, 0,
    ]);
    if (this.group) native().groupRender(this.group, UBO, this.options);   // every GPU traces its tiles; the gather follows on their streams
    else native().render(this.device, UBO, this.options);                  // asynchronous on the GPU, like queue.submit (:821)
  }

  setCameraPosition(x, y, z) { this.cameraPosition = [x, y, z]; }           // :824
  setCameraQuaternion(x, y, z, w) { this.cameraQuaternion = [x, y, z, w]; } // :828
  setFrameCount(frameCount) { this.frameCount = frameCount; }               // :832

  // queue `n` (1..256) consecutive render() calls into one persistent GPU launch; read-backs flush a partial batch
  setBatch(n) { if (this.group) native().groupSetBatch(this.group, n); else native().setBatch(this.device, n); }
  flush() { if (this.group) native().groupFlush(this.group); else native().flush(this.device); }

  // ---- results (the reference presents to a canvas; a Node host reads them back) ----
  readRadiance() { const w = this.canvas.width, h = this.canvas.height; return this.group ? native().groupReadRadiance(this.group, w, h) : native().readRadiance(this.device, w, h); }
  readRGBA8() { const w = this.canvas.width, h = this.canvas.height; return this.group ? native().groupReadRGBA8(this.group, w, h) : native().readRGBA8(this.device, w, h); }      // outputTex equivalent, :163-172
  readTonemapped(fromRGBA8) {                                                                                                      // tonemapper.wgsl
    const w = this.canvas.width, h = this.canvas.height, q = fromRGBA8 !== false;
    return this.group ? native().groupReadTonemapped(this.group, w, h, q) : native().readTonemapped(this.device, w, h, q);
  }
  // Checkpoint / resume of a progressive accumulation (options.accumulate): the raw running sums (f32 RGB sums + sample count per pixel).
  // readAccumulation() -> { width, height, tileRank, tileCount, compact, samples, data: Float32Array }; restoreAccumulation(that object) on a
  // PathTracer with the same scene continues bit for bit with the next render() (frame counts go on where the dumped run stopped).
  readAccumulation() { if (this.group) throw new Error("readAccumulation: per-context state; on a group use the member contexts"); return native().readAccumulation(this.device); }
  restoreAccumulation(dump) { if (this.group) throw new Error("restoreAccumulation: per-context state; on a group use the member contexts"); native().restoreAccumulation(this.device, dump); }
  lastRenderMs() { if (this.group) throw new Error("lastRenderMs: per-context timing; not available on a group"); return native().lastRenderMs(this.device); }
  getStats() { if (this.group) throw new Error("getStats: per-context counters; not available on a group"); return native().getStats(this.device); }
  synchronize() { if (this.group) native().groupSynchronize(this.group); else native().synchronize(this.device); }
  gpuCount() { return this.group ? native().groupSize(this.group) : 1; }
  destroy() {
    if (this.group) { native().groupDestroy(this.group); this.group = null; this.device = null; }
    else if (this.device) { native().destroy(this.device); this.device = null; }
  }
}

module.exports = { PathTracer, MODE_REFERENCE_PACKET, MODE_REFERENCE, MODE_PATH, native };
