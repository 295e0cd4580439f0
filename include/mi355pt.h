/* mi355pt.h -- C ABI of libmi355pt.so, the MI355X-native drop-in for the per-pixel-sample
 * hot path of 31415Hacker/RayTracer-public (src/shaders/renderer.wgsl) and for the
 * scene-build steps that feed it.
 *
 * This is the boundary a reference-side binding attaches to (N-API addon:
 * raytracer-public_amd/napi/addon.c; ctypes: raytracer-public_amd/__init__.py; see
 * INTEGRATION.md).  Plain pointers and sizes only.  Every entry point names the reference
 * interface it replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - every function returns a PtStatus (0 = OK); pt_last_error(ctx) gives the message of
 *     the last failure on that context (pt_last_error(NULL): last context-less failure).
 *     The reference signals errors by JS exceptions / rejected Promises
 *     (src/libs/io.js:3, src/libs/Scene.js:27-30) -- the N-API layer turns a non-zero status
 *     into exactly that.
 *   - host arrays are copied during the call and never retained, like queue.writeBuffer
 *     (src/libs/PathTracer.js:679,692-699,740,789).
 *   - one context = one GPU = one caller thread at a time; work is issued in order on one HIP
 *     stream (the WebGPU queue of the reference is in-order as well).
 *   - buffer layouts are the reference's own:
 *       triangles  f32[9*N]            v0xyz v1xyz v2xyz               (renderer.wgsl:82-89)
 *       BVH2       u32[1 + 6*(2N-1)]   word0 = node count              (BVHBuilder.wgsl:5-7,83-132)
 *       BVH (BVH4) u32[1 + 8*M]        word0 = node count              (renderer.wgsl:91-111)
 *     radiance out: f32 RGBA, row py, column px, py = 0 <-> p.y = -1 (no flip; renderer.wgsl:387-411).
 */
#ifndef MI355PT_H
#define MI355PT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the entry points declared in this header are its whole dynamic symbol table
 * (tests/test_host_build.py checks both directions against `nm -D`). */
#if defined(__GNUC__)
#define PT_API __attribute__((visibility("default")))
#else
#define PT_API
#endif

typedef struct PtContext PtContext;

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID_ARG = 1,
    PT_ERR_NO_DEVICE = 2,     /* no HIP device / HIP runtime failure at create */
    PT_ERR_HIP = 3,           /* a HIP call failed (message in pt_last_error) */
    PT_ERR_NO_SCENE = 4,      /* render / readback before triangles + BVH were set */
    PT_ERR_BAD_BVH = 5,       /* malformed BVH buffer (size, child index, cycle) */
    PT_ERR_IO = 6,
    PT_ERR_OOM = 7
} PtStatus;

/* renderer modes */
enum {
    PT_MODE_REFERENCE_PACKET = 0, /* literal renderer.wgsl:355-413: 2x2 ray packets with one shared stack + lane masks (a quad of lanes per packet) */
    PT_MODE_REFERENCE        = 1, /* same image, one ray per lane (identical except on exact-t ties) */
    PT_MODE_PATH             = 2  /* build-defined extension: spp, bounces, NEE, Russian roulette */
};

/* Mirrors RendererUBO (renderer.wgsl:14-19, packed at PathTracer.js:764-787) plus the
 * build-defined extension fields. */
typedef struct PtRenderParams {
    uint32_t width, height;      /* resolution.xy  */
    float    focal, aspect;      /* resolution.zw: 1/tan(35 deg), W/H (PathTracer.js:761-769) */
    float    cam_pos[3];         /* camPosNumTris.xyz */
    uint32_t num_tris;           /* u32(camPosNumTris.w) (renderer.wgsl:398); must be <= uploaded count */
    float    cam_quat[4];        /* camQuat xyzw */
    uint32_t frame;              /* frameCounter.x (unused by the reference shader; sample index base in PT_MODE_PATH) */
    uint32_t mode;               /* PT_MODE_* */
    uint32_t spp;                /* PT_MODE_PATH: samples per pixel this frame (>= 1) */
    uint32_t max_bounces;        /* PT_MODE_PATH: indirect bounces after the primary hit */
    uint32_t seed;               /* PT_MODE_PATH: RNG seed */
    uint32_t accumulate;         /* PT_MODE_PATH: 0 = replace, 1 = add this frame to the running per-pixel sum */
    uint32_t tile_rank;          /* pixel-tile sharding: this context renders 8x8 tiles with (tx+ty) % tile_count == tile_rank */
    uint32_t tile_count;         /* 0 or 1 = whole frame, row-major output */
    uint32_t flags;              /* PT_FLAG_* */
} PtRenderParams;

enum {
    PT_FLAG_STATS = 1u,          /* run the instrumented kernel variant and fill PtStats */
    PT_FLAG_SIMPLE_KERNEL = 2u,  /* PT_MODE_PATH, PT_MODE_REFERENCE: one-pixel-per-lane kernel instead of the persistent megakernel (A/B checks) */
    PT_FLAG_BRUTE_FORCE = 4u,    /* modes 1, 2: ignore the BVH, test every triangle and every sphere of pt_set_spheres (config C1) */
    PT_FLAG_COMPACT = 8u         /* tile-major compact output (pt_compact_radiance / pt_set_compact_buffer) also when tile_count <= 1: the layout
                                    a one-member group gathers, so that one code path serves every group size */
};

/* Traversal counters of the last PT_FLAG_STATS render (algorithmic-bytes bookkeeping,
 * bytes = 32*nodes_examined + 36*tris_tested + 16*samples). */
typedef struct PtStats {
    uint64_t rays_closest, rays_shadow;
    uint64_t nodes_examined;     /* node records box-tested, each once per examination */
    uint64_t tris_tested;
    uint64_t stack_drops;        /* pushes dropped at the 64-entry cap (renderer.wgsl:337) */
    uint64_t max_stack;
    uint64_t samples;
} PtStats;

/* ---- library / context ------------------------------------------------------------ */

/* PathTracer.initialize() (PathTracer.js:97-173): picks the device, creates the stream and
 * the fixed-size buffers.  device_ordinal < 0 selects the current HIP device. */
PT_API int  pt_create(int device_ordinal, PtContext** out);
PT_API void pt_destroy(PtContext* ctx);
PT_API const char* pt_last_error(const PtContext* ctx);
PT_API const char* pt_version(void);
/* Issue all work of this context on a caller-owned hipStream_t (e.g. torch's current
 * stream) instead of the context's own stream.  NULL restores the own stream. */
PT_API int  pt_set_stream(PtContext* ctx, void* hip_stream);
PT_API int  pt_synchronize(PtContext* ctx);
/* The hipStream_t the context currently issues on (its own non-blocking stream unless pt_set_stream
 * replaced it) -- e.g. to wrap it as torch.cuda.ExternalStream so collectives order after the render. */
PT_API int  pt_get_stream(PtContext* ctx, void** hip_stream);

/* ---- host-side scene build (no GPU touched; the reference runs these in JS) ------------ */

/* computeBVH2Sizing / computeBVH4Sizing (PathTracer.js:227-238) */
PT_API int pt_compute_bvh2_sizing(uint32_t num_tris, uint32_t* num_nodes2, uint64_t* bytes);
PT_API int pt_compute_bvh4_sizing(uint32_t num_nodes4, uint64_t* bytes);
/* buildMortonAndSort (PathTracer.js:427-481): outputs hold num_tris words each */
PT_API int pt_morton_sort(const float* tris, uint32_t num_tris, uint32_t* morton_sorted, uint32_t* tri_index_sorted);
/* collapseLBVH2ToBVH4 (PathTracer.js:506-667): out holds up to 1 + 8*(2N-1) words */
PT_API int pt_collapse_lbvh2_to_bvh4(const uint32_t* bvh2, uint32_t num_tris, uint32_t* out, uint64_t out_words, uint32_t* num_nodes4);
/* BVH2 -> BVH4_wide promotion (tests/test.cpp:106-196): out holds 1 + 8*bvh2[0] words */
PT_API int pt_bvh2_to_bvh4_wide(const uint32_t* bvh2, uint64_t bvh2_words, uint32_t* out, uint64_t out_words);
/* data/BVH2.bin, data/BVH4_wide.bin: raw little-endian u32 dumps (src/server/api.js:27-31,
 * tests/test.cpp:16-33).  pt_file_read_u32 returns the word count through *words; call with
 * dst = NULL to query the size. */
PT_API int pt_file_write_u32(const char* path, const uint32_t* src, uint64_t words);
PT_API int pt_file_read_u32(const char* path, uint32_t* dst, uint64_t dst_words, uint64_t* words);

/* Deterministic procedural stand-in scenes (the reference's dragon.glb / Sponza are absent,
 * SURVEY.md 0.3).  kind 0 = "dragon-class" closed bumpy knot, kind 1 = "sponza-class"
 * interior.  Writes exactly num_tris triangles (9 f32 each), normalised to [-1,1]^3. */
PT_API int pt_scene_procedural(uint32_t kind, uint32_t seed, uint32_t num_tris, float* tris_out);

/* ---- device scene state ------------------------------------------------------------ */

/* device.queue.writeBuffer(triangles) (PathTracer.js:679); N <= 932,067 in the reference
 * (32 MiB buffer, :140-143) -- no such cap here. */
PT_API int pt_set_triangles(PtContext* ctx, const float* tris, uint32_t num_tris);
/* buildBVH (PathTracer.js:671-749): Morton codes + sort (:411-481), LBVH2 kernels (BVHBuilder.wgsl), greedy collapse to
 * BVH4 (:506-667) -- every step on the device, same BVH2 / BVH4 buffers bit for bit as the reference's CPU + WebGPU split
 * (the host entry points below mirror the JS steps one by one).  The bounds of the INTERNAL BVH2 nodes, which only a BVH2
 * read-back looks at, are filled in by the first pt_read_bvh2. */
PT_API int pt_build_bvh(PtContext* ctx);
/* LBVH2 kernels only, from caller-supplied sorted codes (the two dispatches at
 * PathTracer.js:709-728); result stays on the device for pt_read_bvh2. */
PT_API int pt_build_lbvh2(PtContext* ctx, const uint32_t* morton_sorted, const uint32_t* tri_index_sorted);
/* readBVH2 (PathTracer.js:485-502): copies min(bytes, buffer size) bytes */
PT_API int pt_read_bvh2(PtContext* ctx, uint32_t* dst, uint64_t bytes);
/* writeBuffer(BVH) (PathTracer.js:739-740): install a BVH4 buffer in the reference layout
 * (collapse output or BVH4_wide).  Validated: sizes, child indices, no node reachable twice. */
PT_API int pt_set_bvh4(PtContext* ctx, const uint32_t* bvh4, uint64_t words);
/* load a BVH2 buffer (data/BVH2.bin) and collapse it like buildBVH does after readback */
PT_API int pt_set_bvh2(PtContext* ctx, const uint32_t* bvh2, uint64_t words);
PT_API int pt_read_bvh4(PtContext* ctx, uint32_t* dst, uint64_t bytes);
/* Build-defined extension for BASELINE config C1 (Cornell box: triangles + analytic spheres, no BVH):
 * n spheres as (x, y, z, r) f32 quadruples, used only by PT_FLAG_BRUTE_FORCE renders. */
PT_API int pt_set_spheres(PtContext* ctx, const float* xyzr, uint32_t num_spheres);
PT_API int pt_scene_info(PtContext* ctx, uint32_t* num_tris, uint32_t* num_nodes2, uint32_t* num_nodes4);

/* ---- the hot path ----------------------------------------------------------------- */

/* PathTracer.render() compute pass (PathTracer.js:756-802 + renderer.wgsl:355-413).
 * Asynchronous on the context's stream; results are read with pt_read_radiance. */
PT_API int pt_render(PtContext* ctx, const PtRenderParams* params);
/* Batched submission (PT_MODE_PATH): queue `frames_per_launch` (1..256, default 1) consecutive pt_render calls of
 * the same shape and trace them with ONE persistent launch -- small frames (e.g. a 1/8 tile share of a
 * multi-GPU run) then fill the chip like a whole frame does.  A partial batch is launched by whatever
 * needs its result: pt_synchronize, any read-back, pt_compact_radiance / pt_deinterleave, scene changes.
 * Each frame still resolves into the output target that was current when it was submitted. */
PT_API int pt_set_batch(PtContext* ctx, uint32_t frames_per_launch);
/* Launch a partially filled batch now (asynchronous, no host synchronisation). */
PT_API int pt_flush(PtContext* ctx);
/* Device time of the last pt_render's kernel(s), by hipEvents on the stream it ran on.
 * Synchronises the stream. */
PT_API int pt_last_render_ms(PtContext* ctx, float* ms);
/* Per-launch kernel timing without host synchronisation inside a timed loop: after
 * pt_timing_begin(ctx, capacity) every pt_render records its own hipEvent pair (on the stream
 * the kernel is launched on) into a ring; pt_timing_collect synchronises once and returns the
 * elapsed milliseconds of the recorded launches (oldest first) and their count. */
PT_API int pt_timing_begin(PtContext* ctx, uint32_t capacity);
PT_API int pt_timing_collect(PtContext* ctx, float* ms, uint32_t capacity, uint32_t* count);
/* The same, plus when each recorded launch started relative to the first one (launches of consecutive batches overlap on the
 * context's side streams: the union of the [start, start + duration] intervals is the time the GPU was busy tracing). */
PT_API int pt_timing_collect_spans(PtContext* ctx, float* start_ms, float* dur_ms, uint32_t capacity, uint32_t* count);
PT_API int pt_get_stats(PtContext* ctx, PtStats* out);
/* Full-frame f32 RGBA W*H*4 (tile_count <= 1).  Synchronises. */
PT_API int pt_read_radiance(PtContext* ctx, float* dst, uint64_t dst_floats);
/* rgba8unorm equivalent of the reference's outputTex (PathTracer.js:163-172) */
PT_API int pt_read_rgba8(PtContext* ctx, uint8_t* dst, uint64_t dst_bytes);
/* tonemapper.wgsl:24-41 applied to the last frame: Reinhard, gamma 1/2.2, vertical flip;
 * from_rgba8 != 0 first quantises to rgba8unorm like the reference's texture. */
PT_API int pt_read_tonemapped(PtContext* ctx, int from_rgba8, uint8_t* dst, uint64_t dst_bytes);

/* ---- checkpoint / resume of a progressive accumulation (PtRenderParams.accumulate) ------------------------------------------
 * The reference has no accumulation; its only persistence is the raw BVH2 dump (src/server/api.js:27-31).  A
 * progressive render of BASELINE configuration C5 (64 spp as 16 accumulated frames) is long enough to want the same: the running
 * per-pixel state is ONE f32 RGBA buffer -- sums of the sample radiances in x, y, z, the sample count in w -- dumped and restored raw.
 * Whole frame: W*H*4 floats, row-major like the radiance; a tile share (tile_count > 1 or PT_FLAG_COMPACT): tiles*64*4 floats,
 * tile-major like the compact buffer (pt_tile_ids gives the order).  A restored context continues bit for bit: the next pt_render with
 * accumulate = 1 and the same shape adds to the restored sums, with frame indices continuing where the dumped run stopped. */
typedef struct PtAccumInfo {
    uint32_t width, height;
    uint32_t tile_rank, tile_count;   /* the share the sums cover (0 or 1 = every tile) */
    uint32_t compact;                 /* 1: tile-major share layout, 0: whole frame, row-major */
    uint32_t samples;                 /* samples per pixel accumulated so far (the w channel of every pixel) */
    uint64_t floats;                  /* size of the dump; 0 = no running accumulation */
} PtAccumInfo;
PT_API int pt_accum_info(PtContext* ctx, PtAccumInfo* out);
/* Launches what is queued and waits for it; dst holds at least info.floats floats. */
PT_API int pt_read_accum(PtContext* ctx, float* dst, uint64_t dst_floats);
/* Install running sums (the host array is copied during the call).  The scene must be set as for rendering; changing the scene
 * afterwards restarts the accumulation, like it does for one that was rendered. */
PT_API int pt_set_accum(PtContext* ctx, const PtAccumInfo* info, const float* src);

/* ---- pixel-tile sharding across GPUs (one context per GPU / rank) ------------------ */

/* Number of 8x8 tiles / pixels-slots this rank owns for a W x H frame split tile_count ways. */
PT_API int pt_tile_layout(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count,
                   uint32_t* num_tiles, uint64_t* compact_floats);
/* The tile ids (ty * ceil(W/8) + tx) this rank owns, in the order of its compact buffer: slot s of the buffer holds tile ids[s]
 * (64 pixels, row-major inside the 8x8 tile).  ids may be NULL to ask for the count only; capacity in entries.  This is the list
 * pt_render uploads for the share -- callers that assemble or check compact buffers themselves take the order from here. */
PT_API int pt_tile_ids(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count,
                uint32_t* ids, uint32_t capacity, uint32_t* num_tiles);
/* Device pointer + size of this rank's compact radiance buffer (tile-major, 64 px per tile,
 * f32 RGBA) after a render with tile_count > 1: the send buffer of the RCCL gather. */
PT_API int pt_compact_radiance(PtContext* ctx, void** device_ptr, uint64_t* floats);
/* Render tile-sharded frames straight into a caller-owned device buffer (e.g. a torch tensor
 * that is then handed to the RCCL gather) instead of the context's own compact buffer.
 * `floats` is the buffer's capacity.  The buffer that is current when a frame is submitted is that frame's
 * target, also for frames still queued by pt_set_batch when the target changes.  Lifetime rule: a buffer
 * handed in here must stay allocated until a pt_synchronize (or any read-back) issued after the last frame
 * that was submitted while it was current has returned -- or until pt_buffer_busy says it is free.  device_ptr = NULL
 * restores the internal buffer, launches whatever is still queued for caller-owned targets AND waits for it: when the call
 * returns no frame targets a caller-owned buffer any more. */
PT_API int pt_set_compact_buffer(PtContext* ctx, void* device_ptr, uint64_t floats);
/* The same for whole frames (tile_count <= 1): render into a caller-owned row-major f32 RGBA device buffer of at least
 * width*height*4 floats instead of the context's own frame buffer, e.g. one buffer per frame of a batched launch so that
 * every frame stays available (frames of one launch that share a target leave only the last one's result, exactly as if they
 * had been rendered one after the other).  The read-backs read the target of the last frame.  Same lifetime rule as
 * pt_set_compact_buffer; NULL restores the internal buffer, launches what is still queued and waits for it. */
PT_API int pt_set_output_buffer(PtContext* ctx, void* device_ptr, uint64_t floats);
/* *busy = 1 while a frame that is queued (pt_set_batch) or in flight (launched, not yet delivered) still writes into bytes that
 * overlap [device_ptr, device_ptr + bytes) -- whichever kernel renders it, however many launches were submitted after it: the check
 * to make before freeing or re-using a buffer that was handed to pt_set_compact_buffer / pt_set_output_buffer.  Does not wait. */
PT_API int pt_buffer_busy(PtContext* ctx, const void* device_ptr, uint64_t bytes, int* busy);
/* Rank 0: scatter `tile_count` gathered compact buffers (device memory, concatenated in rank
 * order, each padded to `stride_floats`) into the context's full-frame radiance buffer. */
PT_API int pt_deinterleave(PtContext* ctx, const void* gathered_device, uint64_t stride_floats,
                    uint32_t width, uint32_t height, uint32_t tile_count);
/* The same for a gathered BATCH of frames in one launch: rank r's share of frame j sits at gathered + r * rank_stride_floats +
 * j * frame_stride_floats (what one gather of `num_frames` consecutive compact buffers per rank leaves on the root); frame j is
 * scattered to frames_out_device + j * out_stride_floats (row-major f32 RGBA, caller-owned).  frames_out_device = NULL: the
 * context's own frame buffer, which holds one frame -- only the last frame of the batch is scattered (the earlier ones would
 * be replaced by it).  The read-backs then read the last frame. */
PT_API int pt_deinterleave_batch(PtContext* ctx, const void* gathered_device, uint64_t rank_stride_floats, uint64_t frame_stride_floats,
                          uint32_t num_frames, uint32_t width, uint32_t height, uint32_t tile_count,
                          void* frames_out_device, uint64_t out_stride_floats);

/* Packed tile shares: what a sharded frame needs to ship.  Outside the rectangle of tiles in which a camera ray can reach the scene's
 * root box at all, every pixel is the camera-miss value (renderer.wgsl:410) whatever is traced there -- two thirds of the tiles of the
 * dragon-class frame -- and alpha is 1 everywhere.  So a rank packs only its tiles INSIDE the rectangle, 12 bytes per pixel (its tiles
 * in row-major order, 64 x 3 floats each), the packed buffers are gathered, and rank 0 rebuilds the row-major frames: a quarter of the
 * bytes of the compact buffers for that frame.  Non-accumulating frames only (a running sum outside the rectangle depends on history).
 *
 * pt_traced_tile_rect: rect = {tx0, ty0, tx1, ty1}, half-open, in 8x8 tiles -- the rectangle this context's launches trace for a frame
 * with these parameters (the whole image when nothing can be left out); for several frames use the union. */
PT_API int pt_traced_tile_rect(PtContext* ctx, const PtRenderParams* params, uint32_t rect[4]);
/* Largest packed share over the ranks: tiles, and floats per frame (tiles * 192) -- the per-frame count of the gather. */
PT_API int pt_packed_layout(uint32_t width, uint32_t height, uint32_t tile_count, const uint32_t rect[4], uint32_t* max_tiles, uint64_t* floats_per_frame);
/* The tile ids (ty * ceil(W/8) + tx) of a rank's packed share, in the order of the packed buffer: slot s holds tile ids[s] as 64 pixels x
 * (r, g, b).  ids may be NULL to ask for the count only.  (What pt_tile_ids is for the compact buffer.) */
PT_API int pt_packed_tile_ids(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count, const uint32_t rect[4],
                              uint32_t* ids, uint32_t capacity, uint32_t* num_tiles);
/* Pack `num_frames` compact buffers of this rank (frame j at compact_device + j * frame_stride_floats, tile-major f32 RGBA as pt_render
 * leaves them) into packed_device + j * packed_frame_stride_floats.  Asynchronous on the context's stream, behind the frames' resolve. */
PT_API int pt_pack_shares(PtContext* ctx, const void* compact_device, uint64_t frame_stride_floats, uint32_t num_frames, uint32_t width, uint32_t height,
                          uint32_t tile_rank, uint32_t tile_count, const uint32_t rect[4], void* packed_device, uint64_t packed_frame_stride_floats);
/* Rank 0: rank r's packed share of frame j sits at gathered_device + r * rank_stride_floats + j * frame_stride_floats; frame j is rebuilt
 * at frames_out_device + j * out_stride_floats (row-major f32 RGBA; NULL: the context's own frame buffer, last frame only, as
 * pt_deinterleave_batch).  `spp` = samples per pixel of the frames (1 in the reference modes): the value outside the rectangle is the
 * mean of spp camera-miss samples, formed exactly as the resolve pass forms it. */
PT_API int pt_unpack_batch(PtContext* ctx, const void* gathered_device, uint64_t rank_stride_floats, uint64_t frame_stride_floats, uint32_t num_frames,
                           uint32_t width, uint32_t height, uint32_t tile_count, const uint32_t rect[4], uint32_t spp,
                           void* frames_out_device, uint64_t out_stride_floats);

/* ---- one image from all GPUs of the node: a group of contexts inside ONE process ------------------------------
 *
 * PathTracer.render() yields one image per call (src/main.js:54-76); a group keeps that call shape over N GPUs.  Member r renders
 * the 8x8 tiles with (tx + ty) % N == r into a compact buffer (pt_render with tile_rank / tile_count), RCCL gathers the compact
 * buffers on rank 0 over xGMI (ncclCommInitAll + one ncclGather per batch of frames, every sender on its own link to the root), rank 0
 * de-interleaves them into the row-major frame (pt_deinterleave).  The scene is replicated (each member builds it on its own GPU);
 * images are bit-identical for every N.  Accumulating frames (PtRenderParams.accumulate) keep their running sums on the members and
 * are gathered only when an image is asked for.  pt_group_last_error(NULL): last failure without a group. */
typedef struct PtGroup PtGroup;
enum {
    PT_GROUP_TRANSPORT_RCCL = 0,   /* ncclGather over xGMI; one distinct GPU per member */
    PT_GROUP_TRANSPORT_COPY = 1    /* diagnostics: hipMemcpyPeerAsync instead of the collective; members may share a GPU (what RCCL refuses),
                                      so the N > 1 logic can be exercised on a one-GPU machine */
};
/* device_ordinals = NULL: devices 0 .. num_devices-1; num_devices = 0: every visible device */
PT_API int  pt_group_create(const int* device_ordinals, uint32_t num_devices, uint32_t transport, PtGroup** out);
PT_API void pt_group_destroy(PtGroup* group);
PT_API const char* pt_group_last_error(const PtGroup* group);
PT_API int  pt_group_size(const PtGroup* group, uint32_t* num_members);
/* The member context of a rank (borrowed: valid until pt_group_destroy), e.g. rank 0 for pt_read_bvh2 / pt_scene_info. */
PT_API int  pt_group_context(PtGroup* group, uint32_t rank, PtContext** ctx);
/* scene, replicated on every member: pt_set_triangles / pt_build_bvh / pt_set_bvh2 / pt_set_bvh4 */
PT_API int  pt_group_set_triangles(PtGroup* group, const float* tris, uint32_t num_tris);
PT_API int  pt_group_build_bvh(PtGroup* group);
PT_API int  pt_group_set_bvh2(PtGroup* group, const uint32_t* bvh2, uint64_t words);
PT_API int  pt_group_set_bvh4(PtGroup* group, const uint32_t* bvh4, uint64_t words);
/* pt_set_batch for every member; the gather then moves one batch per collective.  A group delivers ONE image per batch: the batch's LAST frame
 * (pt_group_read_* return it; the frames before it are traced -- an accumulating sequence sums them -- but not rebuilt on rank 0).  The frames of a
 * batch may differ in camera: what travels is cut to the union of their traced tile rectangles (none at all when no frame can see the scene). */
PT_API int  pt_group_set_batch(PtGroup* group, uint32_t frames_per_launch);
/* One frame over all members (tile_rank / tile_count of `params` are ignored).  Asynchronous. */
PT_API int  pt_group_render(PtGroup* group, const PtRenderParams* params);
/* Launch, gather and de-interleave what is queued (a partial batch, an accumulating sequence).  Asynchronous. */
PT_API int  pt_group_flush(PtGroup* group);
PT_API int  pt_group_synchronize(PtGroup* group);
/* pt_read_radiance / pt_read_rgba8 / pt_read_tonemapped of the last gathered frame (rank 0).  Synchronise. */
PT_API int  pt_group_read_radiance(PtGroup* group, float* dst, uint64_t dst_floats);
PT_API int  pt_group_read_rgba8(PtGroup* group, uint8_t* dst, uint64_t dst_bytes);
PT_API int  pt_group_read_tonemapped(PtGroup* group, int from_rgba8, uint8_t* dst, uint64_t dst_bytes);

/* ---- diagnostics: NOT part of the drop-in surface -------------------------------------------------------------------------
 * Exported for this repository's own tests and tools (tests/test_gpu_parity.py, tools/ab/sweep.sh, tools/wave_timeline.py); a binding
 * for the reference has no use for them and they may change between builds of the library. */
/* Override one launch heuristic of this context ("GRIDDIV", "ROWS", "CHUNK", "XCD", "SHADE", "FILL", "SLOTS", "CULL",
 * "STATSBATCH", "QUAD", "FORK", "BOUNDED", "TIMELINE"); value 0xFFFFFFFF restores the measured default.  Launches the open batch first.  The same knobs are read
 * from PT_TUNE_<NAME> once, when a context is created.  "TIMELINE" = 1: ordinary (non-STATS) megakernel launches run the TIMELINE variant of the kernel --
 * the production kernel plus wave-uniform bookkeeping in scalar registers, same registers / occupancy, no scratch -- and leave the record pt_debug_wave_times reads. */
PT_API int pt_debug_set_tune(PtContext* ctx, const char* name, uint32_t value);
/* The launch heuristics as a pure function of the launch's shape (no GPU, no context; the measured defaults): out[12] = workgroups, rows / columns of
 * the queue's batch transposition, logical items, items per claim, items per XCD range (0: one queue), shade / fill thresholds, paths at which a
 * wavefront goes on with one ray per quad, shadow-ray forking, frame slots in rotation, frame slots set up.  (tests/test_launch_plan.py) */
PT_API int pt_debug_launch_plan(uint32_t num_cus, uint32_t frames, uint32_t tile_count, uint32_t launches_in_flight, uint32_t traced_batches, uint32_t batch_size, uint32_t out[12]);
/* Raw counter block (24 words) of the last PT_FLAG_STATS launch: PtStats order in [0..6], then the instrumented megakernel's own
 * diagnostics (stack pushes by depth, longest path / ray in traversal steps, re-seated wavefronts, ...). */
PT_API int pt_debug_counters(PtContext* ctx, unsigned long long* dst24);
/* Per-wavefront timeline of the last PT_FLAG_STATS or TIMELINE megakernel launch, 24 words per wavefront: [0] begin, [1] queue found dry, [2] end (100 MHz ticks),
 * [3] traversal steps, [4] shade passes, [5] refill passes, [6] steps / [8] traversing lanes / [15] shade passes when the queue ran dry, [7] traversing lanes summed
 * over the steps, [9] of them at a leaf, [16..18] tick / steps / lanes when the wavefront re-seated its paths one per quad (0: never), [20] the variant that wrote it
 * (1 COUNTERS, 2 TIMELINE); COUNTERS only: [10..13] cycle shares (shade, refill, step, step before queue-dry), [14] longest path that ended after queue-dry, [19].
 * *n_waves = wavefronts written (<= max_waves).  Read it after pt_synchronize. */
PT_API int pt_debug_wave_times(PtContext* ctx, unsigned long long* dst, uint32_t max_waves, uint32_t* n_waves);

#ifdef __cplusplus
}
#endif
#endif /* MI355PT_H */
