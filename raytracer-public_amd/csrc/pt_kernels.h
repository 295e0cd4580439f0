// pt_kernels.h -- kernel argument block and launchers shared by pt_kernels.hip and pt_api.cpp
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_KMODE_PACKET    0
#define PT_KMODE_REFERENCE 1
#define PT_KMODE_PATH      2

// megakernel tuning knobs (overridable with -D at build time)
#ifndef PT_SHORT_STACK
#define PT_SHORT_STACK 12          // LDS stack entries per lane (8 B each); deeper entries spill to global scratch
#endif
#ifndef PT_MEGA_WAVES_PER_SIMD
#define PT_MEGA_WAVES_PER_SIMD 6   // resident 256-thread blocks per CU = waves per SIMD
#endif
#ifndef PT_FRAME_SLOTS
#define PT_FRAME_SLOTS 3            // whole frames whose trace phases may be in flight at once (side streams); tile-sharded frames use 8
#endif
#ifndef PT_MEGA_BLOCK
#define PT_MEGA_BLOCK 64           // threads per workgroup of the persistent kernel: single-wave groups free their CU slot as soon as the wave drains
#endif
#ifndef PT_SHADE_THRESHOLD
#define PT_SHADE_THRESHOLD 16       // lanes waiting for a shade pass before one runs (or nothing traverses); 16 measured best in long launches, 8..16 equal for single frames (tools/ab/sweep.sh SHADE)
#endif
#ifndef PT_QUAD
#define PT_QUAD 1                  // 1: a wavefront with nothing left to start and at most PT_QUAD_LIVE paths goes on with one ray per quad of lanes (pt_megakernel.hip); 0: never; 2: quads from the first ray on (A/B builds)
#endif
#ifndef PT_FORK_LANES
#define PT_FORK_LANES 1            // 1: shadow rays are also handed to idle LANES once a wavefront has nothing left to start (fork_shadow >= 2); 0: only to idle quads in quad mode
#endif
#ifndef PT_FORK_SHADOW
#define PT_FORK_SHADOW 2          // quad mode: a path hands its shadow ray to an idle quad of its wavefront and goes on with the next bounce at once (pt_megakernel.hip)
#endif
#ifndef PT_QUAD_LIVE
#define PT_QUAD_LIVE 16            // paths a wavefront may hold when it re-seats them (16 quads per wavefront)
#endif
#ifndef PT_FILL_THRESHOLD
#define PT_FILL_THRESHOLD 4        // hand out ready camera rays when this many lanes of a wavefront are without a path (a fetch from the ray buffer is cheap: 4 beats 8 by 2 %)
#endif

namespace ptk {

constexpr uint32_t kWaveTimeWords = 24;  // STATS diagnostics: 64-bit words per wavefront in RenderArgs::wave_times
#define PT_MAX_BATCH 256    // frames per persistent launch (their per-frame parameters live in a small device array)
// Per-frame part of the UBO for a batched launch (several consecutive frames traced by one persistent launch).
struct FrameParams {
    float cam[3]; float focal;
    float quat[4];
    float aspect; uint32_t frame; uint32_t seed;
    uint32_t accum_mode;        // bits 0..7: 0 no accumulation buffer, 1 restart the running sum, 2 add to it;
                                // bit 8: a later frame of the same launch has the same output target (this frame's result is not kept)
};
constexpr int kFrameChunk = 32; // frames per upload kernel (their parameters travel as that kernel's argument)
struct FrameChunk { FrameParams f[kFrameChunk]; float4* o[kFrameChunk]; };

// Passed by value as the kernel argument (lives in SGPRs / the kernarg segment).
struct RenderArgs {
    // device scene, MI355X layouts
    const uint4*  nodes;        // WideNode[]: 64 B per internal BVH4 node, read as 4 x dwordx4
    const float4* tris;         // TriRecord[]: 64 B per triangle (three axis-major pieces + the normal)
    const uint4*  scene;        // the arena both arrays live in: triangle record t at byte 64 t, wide node i at byte node_off + 64 i;
                                // child references are positions in it in 16-byte units (packed references, pt_host.h)
    uint32_t      node_off;
    uint32_t      rcp_short;    // 1: the operands of every reciprocal and square root of this launch are bounded (pt_api.cpp::arith_is_bounded): launch_trace picks the BOUNDED kernel variant (short forms, pt_device.h)
    uint32_t      tri_gate;     // the leaf reference of triangle numTris (0x80000000 | 4 * numTris) when the UBO's numTris is smaller than the uploaded triangle count (leaves from it on are entered, not tested), else 0xFFFFFFFF
    // device scene, reference layouts (literal packet kernel, LBVH build, readback)
    const uint32_t* bvh4_ref;   // u32[1 + 8*M]   renderer.wgsl:91-111
    const float*    tris9;      // f32[9*N]       renderer.wgsl:82-89
    // outputs
    float4*   out;              // radiance: row-major W*H, or compact tile-major (64 px per owned tile)
    float4*   accum;            // running per-pixel sums (xyz) + sample count (w); nullptr in reference modes
    uint32_t* tri_ids;          // optional: closest-hit triangle of the primary ray (reference modes)
    const float4* spheres;      // config C1 brute-force scenes: (x, y, z, r) per sphere
    uint32_t num_spheres, brute;
    const uint32_t* tiles;      // owned tile ids, nullptr = every tile in row-major order
    unsigned long long* stats;  // 7 counters (PtStats order)
    uint32_t num_tiles, tiles_x;
    // RendererUBO (renderer.wgsl:14-19)
    uint32_t width, height; float focal, aspect;
    float cam[3]; uint32_t num_tris;
    float quat[4];
    uint32_t frame;
    // wide-BVH root
    uint32_t root_ref; uint32_t root_box[3]; uint32_t root_degenerate;
    // extension
    uint32_t spp, max_bounces, seed, accumulate, compact;
    // persistent megakernel
    float4*   samples;          // per-sample radiance, item = (slot*spp + s)*64 + lane_in_tile; primed per batch by the trace (camera-ray generation), read by resolve_kernel
    uint32_t* queue;            // global item cursor
    uint2*    spill;            // deep stack entries: [entry][grid lane]
    uint4*    raybuf;           // per wavefront of the grid: 64 camera-ray records of 3 x uint4 (o, d, inv, key, sample index), generated 64 at a time
    unsigned long long* wave_times;   // STATS diagnostics: kWaveTimeWords words per wave (begin, queue-empty, end ticks @100 MHz, loop counts, ...)
    // Queue enumeration vs sample storage.  The queue hands out (frame, traced tile, sample) batches of 64 pixel-samples;
    // `trace_slots` lists the owned-tile slots that are traced at all (nullptr = every owned tile): tiles whose every camera ray
    // provably misses the root box are left out (pt_api.cpp: screen rectangle of the root box) and keep the primed miss value.
    // Samples are stored by (frame, owned-tile slot, sample), whatever subset is traced.
    const uint32_t* trace_slots; uint32_t num_trace_tiles;
    uint32_t  trace_rect[4];                  // tiles [tx0, tx1) x [ty0, ty1) that are traced when trace_slots is set: the rest keeps the primed value (resolve_kernel writes it without reading)
    uint32_t  trace_bpf, trace_bpf_magic;     // traced batches per frame (num_trace_tiles * spp) and floor(2^32 / that)
    uint32_t  spp_magic, tiles_x_magic;       // floor(2^32 / spp), floor(2^32 / tiles_x): division by multiply + one correction
    uint32_t  num_sample_batches;             // (frame, owned tile, sample) batches in the sample buffer = num_frames * batches_per_frame
    uint32_t  total_items, chunk_items;   // logical items (64*64*perm_cols) and items per queue claim
    uint32_t  xcd_span;                   // 0: one queue; else items per XCD range (8 cursors at queue[8..15])
    uint32_t  num_batches, perm_cols;     // real (frame, traced tile, sample) batches of the queue; columns of the batch transpose
    uint32_t  perm_rows, perm_rows_magic; // its rows and floor(2^32 / rows) for the division
    uint32_t  shade_threshold, fill_threshold;
    uint32_t  fork_shadow;      // quad mode: shadow rays of paths that go on are traced by idle quads, next to the path's next ray (0: by the path itself, in turn)
    uint32_t  quad_live;        // re-seat the paths one per quad once the wavefront has nothing left to start and holds at most this many (0: never)
    // batched launch: frames[i] / outs[i] for i < num_frames; items of frame i are batches [i*batches_per_frame, ...)
    const FrameParams* frames; float4* const* outs;     // device arrays of the frame slot, filled by launch_frame_params
    uint32_t  num_frames, batches_per_frame;
    uint32_t  prime;            // 1: launch_trace must zero the control block itself
    uint32_t  ref_mode;         // 1: PT_MODE_REFERENCE on the megakernel -- one primary ray through each pixel centre, shade() of renderer.wgsl:348-353 (spp = 1, no bounces)
};

hipError_t launch_render(const RenderArgs& args, int kmode, bool stats, hipStream_t stream);
// k0/k1 (optional): events recorded immediately around the trace_paths_kernel launches
hipError_t launch_trace(const RenderArgs& args, bool stats, uint32_t grid_blocks, hipStream_t stream, hipEvent_t k0, hipEvent_t k1);
hipError_t launch_prime(uint32_t* queue, hipStream_t stream);   // first use of a frame slot: its control block zeroed
// per-frame parameters and output targets of a launch into the slot's device arrays (asynchronous: the data travels as kernel arguments)
hipError_t launch_frame_params(const FrameParams* frames, float4* const* outs, uint32_t n, FrameParams* d_frames, float4** d_outs, hipStream_t stream);
hipError_t launch_resolve(const RenderArgs& args, hipStream_t stream);
uint32_t megakernel_grid(int num_cus);
uint32_t megakernel_block();
// refit = false leaves the internal BVH2 nodes without bounds (the BVH4 collapse does not read them); launch_lbvh2_refit adds them
hipError_t launch_lbvh2(uint32_t* bvh2, const float* tris9, const uint32_t* morton, const uint32_t* tri_index,
                        uint32_t* parent, uint32_t* flags, uint32_t num_tris, bool refit, hipStream_t stream);
hipError_t launch_lbvh2_refit(uint32_t* bvh2, const uint32_t* parent, uint32_t* flags, uint32_t num_tris, hipStream_t stream);
// ---- device-side scene build (pt_build.hip) -------------------------------------------------
constexpr int kBuildCounters = 256;      // one append counter per BVH4 level (an LBVH2 over 30-bit codes + index bits is < 64 deep)
struct BuildBuffers {
    unsigned long long* bounds;          // [6] centroid bounds as order-preserving u64 keys
    uint32_t* counters;                  // [kBuildCounters]
    uint32_t *code_tmp, *index_tmp;      // [n] Morton codes / triangle ids before the sort
    uint32_t *morton, *tri_index;        // [n] after the sort: the inputs of launch_lbvh2
    void* temp; size_t temp_bytes;       // rocPRIM scratch (build_temp_bytes)
    uint32_t* node2; uint4* child_pos; uint32_t *subtree, *ids, *bnd;   // [2n-1] per BVH4 node in breadth-first order (bnd: 3 words each)
    uint32_t* host_word;                 // pinned host word for the per-level counts
};
size_t build_temp_bytes(uint32_t num_tris);
// edge_max (optional, zeroed by the caller): receives the largest |edge component| of the triangles as f32 bits
hipError_t launch_tri_records(const float* tris9, uint32_t num_tris, float4* records, uint32_t* edge_max, hipStream_t stream);
hipError_t launch_morton_sort(const BuildBuffers& B, const float* tris9, uint32_t num_tris, hipStream_t stream);
// synchronises the stream once per BVH4 level (the level sizes size the next launch); *num_nodes4 = M on return
hipError_t collapse_on_device(const BuildBuffers& B, const uint32_t* bvh2, uint32_t num_tris, uint32_t* bvh4, uint32_t* num_nodes4, hipStream_t stream);
// B.subtree[i] = 1 for internal node id i, B.ids = its exclusive prefix sum (the wide-node index)
hipError_t launch_internal_scan(const BuildBuffers& B, const uint32_t* bvh4, uint32_t num_nodes4, hipStream_t stream);
hipError_t launch_wide_nodes(const BuildBuffers& B, const uint32_t* bvh4, uint32_t num_nodes4, uint4* wide, uint32_t num_tris, uint32_t node_base16, hipStream_t stream);
// gathered: rank r's share of frame j at gathered + r * rank_stride_px + j * frame_stride_px; frame j goes to full + j * full_stride_px
hipError_t launch_deinterleave(const float4* gathered, uint64_t rank_stride_px, uint64_t frame_stride_px, uint32_t frames, float4* full, uint64_t full_stride_px,
                               uint32_t width, uint32_t height, uint32_t count, hipStream_t stream);
// packed tile shares (pt_kernels.hip): rank's tiles inside the tile rectangle rect = {tx0, ty0, tx1, ty1}, 64 x 3 floats each
hipError_t launch_pack_shares(const float4* compact, uint64_t frame_stride_px, uint32_t frames, float* packed, uint64_t packed_stride_floats, uint32_t width,
                              uint32_t rank, uint32_t count, const uint32_t rect[4], hipStream_t stream);
hipError_t launch_unpack_frames(const float* gathered, uint64_t rank_stride_floats, uint64_t frame_stride_floats, uint32_t frames, float4* full, uint64_t full_stride_px,
                                uint32_t width, uint32_t height, uint32_t count, const uint32_t rect[4], uint32_t spp, hipStream_t stream);
hipError_t launch_rgba8(const float4* src, uint32_t* dst, uint32_t n, hipStream_t stream);
hipError_t launch_tonemap(const float4* src, uint32_t* dst, uint32_t width, uint32_t height, int from_rgba8, hipStream_t stream);

} // namespace ptk
