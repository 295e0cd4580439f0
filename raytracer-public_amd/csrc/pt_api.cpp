// pt_api.cpp -- the C ABI of libmi355pt (include/mi355pt.h): context, device buffers, scene
// upload / build, render dispatch, readback.  Mirrors the host side of the reference's
// PathTracer class (src/libs/PathTracer.js); each entry point cites what it replaces in the
// header.  No CPU fallback for device work: without a usable HIP device pt_create fails.
#include "mi355pt.h"
#include "pt_host.h"
#include "pt_kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_global_error;

template <typename T>
struct DevBuf {
    T* ptr = nullptr; size_t cap = 0;   // capacity in elements
    hipError_t ensure(size_t n) {
        if (n <= cap && ptr) return hipSuccess;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
        if (n == 0) n = 1;
        hipError_t e = hipMalloc((void**)&ptr, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

} // namespace

// Development overrides of the megakernel's launch heuristics.  kAuto = use the measured default for the launch at hand.
// Read from PT_TUNE_* ONCE, when the context is created (a shipped library does not consult the environment per launch);
// tests and tools/ab/sweep.sh change them through pt_debug_set_tune.
struct PtTune {
    static constexpr uint32_t kAuto = 0xFFFFFFFFu;
    uint32_t grid_div = kAuto, rows = kAuto, chunk = kAuto, xcd = kAuto, shade = kAuto, fill = kAuto,
             slots = kAuto, cull = kAuto, stats_batch = kAuto, quad = kAuto, fork = kAuto, bounded = kAuto, timeline = kAuto;
    uint32_t* find(const char* name) {
        static const struct { const char* n; uint32_t PtTune::* m; } tab[] = {
            {"GRIDDIV", &PtTune::grid_div}, {"ROWS", &PtTune::rows}, {"CHUNK", &PtTune::chunk}, {"XCD", &PtTune::xcd}, {"SHADE", &PtTune::shade},
            {"FILL", &PtTune::fill}, {"SLOTS", &PtTune::slots}, {"CULL", &PtTune::cull}, {"STATSBATCH", &PtTune::stats_batch}, {"QUAD", &PtTune::quad}, {"FORK", &PtTune::fork}, {"BOUNDED", &PtTune::bounded}, {"TIMELINE", &PtTune::timeline}};
        for (const auto& t : tab) if (std::strcmp(name, t.n) == 0) return &(this->*(t.m));
        return nullptr;
    }
    void from_environment() {
        static const char* names[] = {"GRIDDIV", "ROWS", "CHUNK", "XCD", "SHADE", "FILL", "SLOTS", "CULL", "STATSBATCH", "QUAD", "FORK", "BOUNDED", "TIMELINE"};
        for (const char* n : names) {
            const std::string key = std::string("PT_TUNE_") + n;
            const char* v = std::getenv(key.c_str());
            if (v && *v) *find(n) = uint32_t(std::strtoul(v, nullptr, 10));
        }
    }
    static uint32_t pick(uint32_t knob, uint32_t dflt) { return knob == kAuto ? dflt : knob; }
};

struct PtContext {
    int device = 0;
    PtTune tune;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool timed = false;
    std::string err;

    // scene (host mirrors kept for rebuilds / readback of small metadata only)
    uint32_t num_tris = 0, num_nodes2 = 0, num_nodes4 = 0;
    bool edges_small = false;        // every |edge component| of the uploaded triangles is below 2^20 (pt_set_triangles; see arith_is_bounded)
    DevBuf<uint32_t> d_edge_max;
    bool have_tris = false, have_bvh = false, have_bvh2 = false;
    bool bvh2_refit_pending = false;  // pt_build_bvh leaves the internal BVH2 bounds to the first pt_read_bvh2 (nothing on the render path reads them)
    pt::WideBvh wide_meta;           // root info; nodes vector emptied after upload

    DevBuf<float> d_tris9;           // reference layout
    // One arena for the two record arrays the traversal gathers from: [triangle records, 64 B each, padded to a multiple of
    // 64 B | wide nodes, 64 B each]: one allocation, one base address, 32-bit byte offsets reach both kinds.
    DevBuf<uint4> d_scene; uint64_t node_off = 0, node_cap = 0; uint32_t scene_tris = 0;
    float4* trirec() const { return (float4*)d_scene.ptr; }
    uint4* wide() const { return d_scene.ptr + node_off / 16; }
    DevBuf<uint32_t> d_bvh2, d_bvh4; // reference layouts
    DevBuf<uint32_t> d_morton, d_triidx, d_parent, d_flags;
    // device-side scene build (pt_build.hip): scratch kept for rebuilds
    DevBuf<unsigned long long> d_bounds; DevBuf<uint32_t> d_counters, d_code_tmp, d_index_tmp, d_node2, d_subtree, d_ids, d_bnd;
    DevBuf<uint4> d_child_pos; DevBuf<unsigned char> d_build_temp; uint32_t* h_word = nullptr;
    DevBuf<float4> d_spheres; uint32_t num_spheres = 0;

    // frame
    DevBuf<float4> d_out, d_accum, d_compact, d_compact_accum;
    DevBuf<uint32_t> d_tiles, d_u32tmp;
    // Frame slots: the trace phase of consecutive pt_render calls runs on alternating side streams
    // (own sample / control / scratch buffers each), so the sparse tail of one frame overlaps the
    // dense start of the next; the resolve passes stay in call order on the main stream.
    struct FrameSlot {
        hipStream_t side = nullptr; hipEvent_t resolved = nullptr, done = nullptr; bool used = false;
        uint64_t resolved_seq = 0;                                      // launch sequence number of the latest record of `resolved` (pt_buffer_busy)
        DevBuf<uint32_t> queue; DevBuf<float4> samples; DevBuf<uint2> spill; DevBuf<uint4> rays;
        // owned-tile slots that the launch in this slot traces (the others are culled: every camera ray misses the root box)
        DevBuf<uint32_t> trace_slots; uint32_t* h_trace = nullptr; size_t h_trace_cap = 0; hipEvent_t trace_copied = nullptr;
        uint32_t cull_key[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint32_t num_trace_tiles = 0; bool cull_valid = false;
        DevBuf<ptk::FrameParams> frame_params; DevBuf<float4*> frame_outs;      // per-frame parameters / targets of the launch in this slot
        const void* primed_ptr = nullptr; size_t primed_samples = 0;   // what the resident prefill covers
    };
    static constexpr int kMaxSlots = 16;
    FrameSlot slots[kMaxSlots]; int num_slots = 0; uint32_t next_slot = 0;
    // frames queued for one batched launch (pt_set_batch): launched when full or when anything needs their result
    uint32_t batch_size = 1; uint32_t pending = 0; ptk::RenderArgs pendingA; bool pending_ring = false; uint32_t pending_rank = 0, pending_count = 1;
    std::vector<ptk::FrameParams> pending_frames; std::vector<float4*> pending_outs;
    DevBuf<unsigned long long> d_wave_times; uint32_t wave_times_n = 0;
    int num_cus = 0;
    DevBuf<unsigned long long> d_stats;
    std::vector<uint32_t> tiles_host; uint32_t tiles_w = 0, tiles_h = 0, tiles_rank = 0, tiles_count = 0;
    uint32_t out_w = 0, out_h = 0;   // dimensions of the last full-frame result in d_out
    uint32_t accum_w = 0, accum_h = 0, accum_count = 0, accum_rank = 0;
    uint64_t compact_floats = 0;
    uint32_t share_w = 0, share_h = 0, share_count = 0, share_max_tiles = 0;      // cached: largest tile share of a (W, H, count) split (pt_deinterleave*)
    float4* ext_compact = nullptr; uint64_t ext_compact_floats = 0;
    float4* ext_out = nullptr; uint64_t ext_out_floats = 0;      // caller-owned whole-frame target (pt_set_output_buffer)
    const float4* last_full = nullptr;                          // where the last whole-frame result lives (d_out or a caller's buffer)
    bool last_stats = false;
    uint64_t stats_culled = 0;          // pixel-samples of the last instrumented launch that were culled (counted as one root-box miss each)
    std::vector<hipEvent_t> ring;    // start/stop pairs recorded by pt_render while timing is on
    uint32_t ring_used = 0;
    // pt_buffer_busy: the byte ranges that launches not yet known to be delivered write into.  Every launch appends its targets with its
    // sequence number.  The context's stream is in order and every launch's last write is followed there by an event that exists anyway
    // (the frame slot's `resolved`, re-recorded by the slot's next launch; `misc_fence` behind the kernels that do not use a slot), each
    // remembering the sequence number of its LATEST record: once such an event has completed, every launch up to that number has been
    // delivered -- whatever frame slot it ran in and however many launches were submitted after it.  (No event of its own per launch:
    // one more marker on the stream cost the reference's 0.16 ms frame 18 % with one render() per launch.)
    struct InFlight { const char* lo; const char* hi; uint64_t seq; };
    std::vector<InFlight> inflight;
    uint64_t launch_seq = 0, delivered_seq = 0;
    hipEvent_t misc_fence = nullptr; uint64_t misc_fence_seq = 0;
};

namespace {

int fail(PtContext* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_global_error = msg;
    return code;
}
int fail_hip(PtContext* ctx, hipError_t e, const char* what) {
    return fail(ctx, PT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define PT_HIP(ctx, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail_hip((ctx), e__, #call); } while (0)

int bind(PtContext* ctx) {
    if (!ctx) return fail(nullptr, PT_ERR_INVALID_ARG, "null context");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    return PT_OK;
}

// which pixels a running accumulation covers: tile rank (bits 0..11), tile count (12..23), compact tile-major layout (bit 31)
uint32_t accum_share_key(uint32_t rank, uint32_t count, bool compact) { return (rank & 0xfffu) | ((count & 0xfffu) << 12) | (compact ? 0x80000000u : 0u); }

// ---- pt_buffer_busy bookkeeping (PtContext::inflight) ----
void prune_inflight(PtContext* ctx) {
    if (ctx->inflight.empty()) return;
    uint64_t delivered = ctx->delivered_seq;
    auto done = [&](hipEvent_t ev) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); return false; }         // an answer, not an error: do not leave it behind for the launch checks
        return true;                                                                  // complete (or failed: nothing will be written any more)
    };
    for (auto& sl : ctx->slots) if (sl.side && sl.used && sl.resolved_seq > delivered && done(sl.resolved)) delivered = sl.resolved_seq;
    if (ctx->misc_fence && ctx->misc_fence_seq > delivered && done(ctx->misc_fence)) delivered = ctx->misc_fence_seq;
    if (delivered != ctx->delivered_seq) {
        ctx->delivered_seq = delivered;
        ctx->inflight.erase(std::remove_if(ctx->inflight.begin(), ctx->inflight.end(), [&](const PtContext::InFlight& f) { return f.seq <= delivered; }), ctx->inflight.end());
    }
}
void everything_delivered(PtContext* ctx) {         // after a host wait on the context's stream
    ctx->delivered_seq = ctx->launch_seq; ctx->inflight.clear();
}
// the launch just submitted writes `bytes` bytes at each of targets[0..n): remembered until an event recorded behind it on the context's
// stream has completed; returns the launch's sequence number (the caller stores it next to the event it records)
uint64_t track_targets(PtContext* ctx, float4* const* targets, uint32_t n, size_t bytes) {
    if (ctx->inflight.size() >= 64) prune_inflight(ctx);                 // bounded without a query per launch
    const uint64_t seq = ++ctx->launch_seq;
    for (uint32_t i = 0; i < n; ++i) {
        if (!targets[i]) continue;
        bool dup = false;
        for (uint32_t g = 0; g < i && !dup; ++g) dup = targets[g] == targets[i];
        if (!dup) ctx->inflight.push_back({(const char*)targets[i], (const char*)targets[i] + bytes, seq});
    }
    return seq;
}

// Triangle records (64 B each), then one all-zero record (what a leaf with an out-of-range triangle index points at: never hit).
uint64_t tri_region_bytes(uint32_t num_tris) { return (uint64_t(num_tris) + 1u) * 64u; }

// Room for `num_tris` triangle records and `nodes` wide nodes; triangle records that are already there survive a regrowth.
int ensure_scene(PtContext* ctx, uint32_t num_tris, uint64_t nodes) {
    const uint64_t tri_bytes = tri_region_bytes(num_tris);
    const uint64_t need = tri_bytes + (nodes + 1u) * 64u;
    if (need >= 0xFFFFFFFFull) return fail(ctx, PT_ERR_INVALID_ARG, "scene too large: triangle records and BVH nodes are addressed by 32-bit byte offsets (4 GiB)");
    if (ctx->d_scene.ptr && ctx->node_off == tri_bytes && ctx->node_cap >= nodes) {
        if (ctx->scene_tris != num_tris)      // another triangle count in the same region: the never-hit record moves
            PT_HIP(ctx, hipMemset((char*)ctx->d_scene.ptr + uint64_t(num_tris) * 64u, 0, tri_bytes - uint64_t(num_tris) * 64u));
        ctx->scene_tris = num_tris; return PT_OK;
    }
    const uint64_t cap_nodes = nodes + nodes / 8u + 16u;
    uint4* fresh = nullptr;
    PT_HIP(ctx, hipMalloc((void**)&fresh, tri_bytes + (cap_nodes + 1u) * 64u));
    if (ctx->d_scene.ptr && ctx->scene_tris == num_tris && num_tris)          // the records of the current triangles move along
        PT_HIP(ctx, hipMemcpy(fresh, ctx->d_scene.ptr, uint64_t(num_tris) * 64u, hipMemcpyDeviceToDevice));
    if (ctx->d_scene.ptr) (void)hipFree(ctx->d_scene.ptr);
    PT_HIP(ctx, hipMemset((char*)fresh + uint64_t(num_tris) * 64u, 0, tri_bytes - uint64_t(num_tris) * 64u));      // the never-hit record
    ctx->d_scene.ptr = fresh; ctx->d_scene.cap = size_t((tri_bytes + (cap_nodes + 1u) * 64u) / 16u);
    ctx->node_off = tri_bytes; ctx->node_cap = cap_nodes; ctx->scene_tris = num_tris;
    return PT_OK;
}

int upload_wide(PtContext* ctx, const uint32_t* bvh4, uint64_t words) {
    pt::WideBvh w; std::string err;
    const uint32_t tris_now = ctx->have_tris ? ctx->num_tris : 0u;
    if (!pt::build_wide_bvh(bvh4, words, tris_now, uint32_t(tri_region_bytes(tris_now) / 16u), w, err)) return fail(ctx, PT_ERR_BAD_BVH, err);
    PT_HIP(ctx, ctx->d_bvh4.ensure(words));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_bvh4.ptr, bvh4, words * 4, hipMemcpyHostToDevice, ctx->stream));
    if (int rc = ensure_scene(ctx, tris_now, w.nodes.size())) return rc;
    if (!w.nodes.empty())
        PT_HIP(ctx, hipMemcpyAsync(ctx->wide(), w.nodes.data(), w.nodes.size() * sizeof(pt::WideNode), hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // host staging vectors die at return
    ctx->num_nodes4 = w.num_nodes4;
    w.nodes.clear(); w.nodes.shrink_to_fit();
    ctx->wide_meta = w;
    ctx->have_bvh = true;
    return PT_OK;
}

// Tiles whose every camera ray provably misses the root box need no tracing: their samples keep the primed miss value.  The
// rectangle is the bounding box of the eight projected corners of the root's (f16, exactly representable) bounds, in tile
// units, widened by a margin of two pixels -- orders of magnitude more than the rounding of the ray set-up (renderer.wgsl:387-395)
// can move a ray.  No culling when a corner is beside or behind the eye, or when the quaternion is not of unit length to within
// 1e-5 (the reference's rotateVectorByQuat is a rotation only then).  Returns false for "trace everything".
struct TileRect { uint32_t tx0, ty0, tx1, ty1; };     // half-open, in tiles
bool root_box_rect(const pt::WideBvh& w, const ptk::FrameParams& f, uint32_t width, uint32_t height, TileRect& out) {
    const double qx = f.quat[0], qy = f.quat[1], qz = f.quat[2], qw = f.quat[3];
    const double qn = qx * qx + qy * qy + qz * qz + qw * qw;
    // |q|^2 within 1e-5 of 1 (an f32-normalised quaternion passes): for a longer or shorter q the reference's rotateVectorByQuat
    // scales and shears the direction by about |q|^2 - 1, i.e. up to that times half the image width in pixels at the screen edge --
    // 1e-5 x 16384 px stays far inside the two-pixel margin below; anything sloppier traces every tile
    if (!(qn > 1.0 - 1e-5 && qn < 1.0 + 1e-5) || !(f.focal > 1e-3f) || !(f.aspect > 1e-3f)) return false;
    const double mn[3] = {pt::half_to_float(w.root_box[0] & 0xffffu), pt::half_to_float(w.root_box[0] >> 16), pt::half_to_float(w.root_box[1] & 0xffffu)};
    const double mx[3] = {pt::half_to_float(w.root_box[1] >> 16), pt::half_to_float(w.root_box[2] & 0xffffu), pt::half_to_float(w.root_box[2] >> 16)};
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int c = 0; c < 8; ++c) {
        const double p[3] = {((c & 1) ? mx[0] : mn[0]) - f.cam[0], ((c & 2) ? mx[1] : mn[1]) - f.cam[1], ((c & 4) ? mx[2] : mn[2]) - f.cam[2]};
        if (!(p[0] == p[0] && p[1] == p[1] && p[2] == p[2])) return false;
        // v = conj(q) * p * q: the inverse of the camera-to-world rotation
        const double ux = -qx, uy = -qy, uz = -qz;
        const double cx = uy * p[2] - uz * p[1], cy = uz * p[0] - ux * p[2], cz = ux * p[1] - uy * p[0];
        const double dx = uy * cz - uz * cy, dy = uz * cx - ux * cz, dz = ux * cy - uy * cx;
        const double vx = p[0] + 2.0 * (qw * cx + dx), vy = p[1] + 2.0 * (qw * cy + dy), vz = p[2] + 2.0 * (qw * cz + dz);
        if (!(vz < -1e-4)) return false;                                 // beside / behind the eye (the camera looks down -z)
        const double sx = vx / -vz * f.focal / f.aspect, sy = vy / -vz * f.focal;          // p.x, p.y of renderer.wgsl:388
        const double fx = (sx + 1.0) * 0.5 * width, fy = (sy + 1.0) * 0.5 * height;
        x0 = fx < x0 ? fx : x0; x1 = fx > x1 ? fx : x1; y0 = fy < y0 ? fy : y0; y1 = fy > y1 ? fy : y1;
    }
    const double margin = 2.0;
    const uint32_t tiles_x = (width + pt::kTile - 1) / pt::kTile, tiles_y = (height + pt::kTile - 1) / pt::kTile;
    auto lo = [&](double v, uint32_t n) { v = (v - margin) / pt::kTile; return v <= 0.0 ? 0u : (v >= n ? n : uint32_t(v)); };
    auto hi = [&](double v, uint32_t n) { v = (v + margin) / pt::kTile + 1.0; return v <= 0.0 ? 0u : (v >= n ? n : uint32_t(v)); };
    out.tx0 = lo(x0, tiles_x); out.tx1 = hi(x1, tiles_x); out.ty0 = lo(y0, tiles_y); out.ty1 = hi(y1, tiles_y);
    return true;
}

// May the launch use the short forms of the correctly rounded reciprocal and square root (pt_device.h::rcp_normal / sqrt_normal: bit-identical to the
// IEEE operations for operands of magnitude 2^-64 ... 2^64, and for sqrt(0))?  Their operands, site by site:
//   * 1 / det of the triangle test, det = e1 . (d x e2), so |det| <= |e1| |e2| |d|.  Every edge component is below 2^20 (edges_small: |e1| |e2| < 3 * 2^40).
//     Shadow and bounce directions are unit vectors; a camera direction is a unit vector v taken through rotateVectorByQuat (renderer.wgsl:66-72),
//     v + 2 (s u x v + u x (u x v)) with q = (u, s): |2 s u x v| <= |q|^2 and |2 u x (u x v)| <= 2 |q|^2, so |d| <= 1 + 3 |q|^2 -- NOT |q|^2.  With every
//     frame's |q|^2 below 2^18 that is |d| < 2^19.6 and |det| < 2^61.2: 3.8 binades under the 2^65 the short forms are proven for (tools/probes/rcp_exact.hip).
//     A det below the range fails the test's own |det| < 1e-7 and its quotient is never used;
//   * the three reciprocals of a ray's direction: taken only for |component| > 1e-8 (safeInvDir), bounded above as just said;
//   * normalize() of the camera-space direction (p.x * aspect, p.y, -focal) with |p| <= 1: its squared length lies in [focal^2, aspect^2 + 1 + focal^2],
//     so focal and aspect have to be of ordinary magnitude (1e-6 ... 1e6 is asked here);
//   * the sampling (cosine_dir, Russian roulette): sqrt of u1 and 1 - u1 in [0, 1] (multiples of 2^-24: zero or >= 2^-24), 1 / (1 + |n.z|), 1 / max(T) with
//     max(T) >= 0.3 -- always inside.
// A NaN fails every compare below.
bool arith_is_bounded(const PtContext* ctx, const std::vector<ptk::FrameParams>& frames, uint32_t nf) {
    if (!ctx->edges_small) return false;
    for (uint32_t i = 0; i < nf; ++i) {
        const float* q = frames[i].quat;
        const double n2 = double(q[0]) * q[0] + double(q[1]) * q[1] + double(q[2]) * q[2] + double(q[3]) * q[3];
        if (!(n2 < 262144.0)) return false;                              // 2^18, see above
        if (!(frames[i].focal > 1e-6f && frames[i].focal < 1e6f && frames[i].aspect < 1e6f && frames[i].aspect > -1e6f)) return false;
    }
    return true;
}

// ---- a launch of the queued frames, in three steps: PLAN (pure arithmetic: the launch heuristics), PREPARE (frame slot, buffers, the list of
// traced tiles), LAUNCH (trace on the slot's side stream, resolve on the context's stream).

// What the heuristics need to know about a launch ...
struct PlanInput {
    uint32_t num_cus;            // compute units of the device
    uint32_t frames;             // frames of the launch
    uint32_t tile_count;         // tile shares the frame is cut into (1: whole frames)
    bool     sharded;            // compact (tile-major) output
    bool     stats;              // instrumented launch: slot 0, never overlapped
    uint32_t launches_in_flight; // earlier launches of this context that have not finished (only asked for single whole frames)
    uint32_t traced_batches;     // (frame, traced tile, sample) batches of 64 pixel-samples in the launch
    uint32_t batch_size;         // pt_set_batch of the context: what a FULL launch of the current setting carries
};
// ... and what they decide (measured defaults, each overridable through PtTune; sources: DESIGN.md section 6.1, profiles/HISTORY.md, tools/ab/sweep.sh)
struct LaunchPlan {
    uint32_t grid;               // workgroups of the persistent kernel
    uint32_t perm_rows, perm_cols, total_items, chunk_items, xcd_span;       // the queue's walk through the batches
    uint32_t shade_threshold, fill_threshold, quad_live, fork_shadow;
    int      slots, setup_slots; // frame slots launches of this size rotate through / slots to have allocated (a full batch's count as well)
    const char* error;           // nullptr, or why the launch cannot be made
};

int slots_for(const PtTune& tune, uint32_t frames, uint32_t tile_count, bool sharded) {
    // How many launches to keep in flight depends on the work in one launch (in whole frames): a long launch only needs its tail covered by the
    // next one (and each extra overlapped launch stretches every launch's own duration), small sharded launches need several in flight to fill
    // the chip.  Measured: tools/ab/sweep.sh SLOTS, tools/ab/pipe_sweep.sh.
    const uint32_t w8 = frames * 8u / (tile_count ? tile_count : 1u);      // eighths of a whole frame
    const int n = int(PtTune::pick(tune.slots, w8 >= 256u ? 3u : (w8 >= 64u ? 4u : (w8 >= 16u ? PT_FRAME_SLOTS : (w8 >= 8u ? (sharded ? 4u : 6u) : (sharded ? 8u : PT_FRAME_SLOTS))))));
    return n < 1 ? 1 : (n > PtContext::kMaxSlots ? PtContext::kMaxSlots : n);
}

LaunchPlan plan_launch(const PtTune& tune, const PlanInput& in) {
    LaunchPlan P; std::memset(&P, 0, sizeof(P));
    const uint32_t nf = in.frames, count = in.tile_count ? in.tile_count : 1u;
    const int cus = in.num_cus > 0 ? int(in.num_cus) : 256;
    P.grid = ptk::megakernel_grid(cus);
    {   // A tile-sharded frame is a fraction of the work: fewer, denser wavefronts (measured: 1/8 frame best at grid/4); a batch of nf such frames is
        // nf times the work again.  A single whole frame: the grid shrinks with the launches already in flight (one render() per frame without host
        // waits, the reference's call shape: six frame slots, a quarter of the grid each -- 1.21 -> 1.01 ms per frame; a lone frame keeps the whole
        // grid: 2.9 ms, 3.8 ms on a quarter of it).  tools/ab/pipe_sweep.sh
        uint32_t div = PtTune::pick(tune.grid_div, count >= 8u ? 4u : (count >= 2u ? 2u : std::min(4u, 1u + in.launches_in_flight)));
        if (nf > 1u) div = div > nf ? div / nf : 1u;
        if (div > 1u) P.grid = (P.grid + div - 1u) / div;
    }
    const uint32_t grid_lanes = P.grid * ptk::megakernel_block();
    // Rows of the batch transposition (the order in which the queue walks the (frame, tile, sample) batches).  Rows that are a multiple or a divisor
    // of the frame count keep the frames of a launch aligned: all rows are at the same image position at the same time, so the frames share the BVH
    // nodes they pull through L2.  Long launches take one row per frame (few places in flight = locality); short launches cut every frame into 128
    // segments (fine interleave of object and background tiles = balance when each wavefront only sees a few chunks).  tools/ab/sweep.sh ROWS
    const uint32_t work8 = nf * 8u / count;
    // (one row per frame from 3 frames of work on: launches of 3 .. 7 frames of work 1 .. 7 % faster than with 128 segments per frame, profiles/r05_m2_midsize_rows.txt)
    P.perm_rows = PtTune::pick(tune.rows, work8 >= 24u ? nf : (work8 >= 8u ? 128u * nf : 64u * nf));
    if (P.perm_rows < 1u) P.perm_rows = 1u;
    if (P.perm_rows > 4096u) P.perm_rows = 4096u;
    P.perm_cols = (in.traced_batches + P.perm_rows - 1u) / P.perm_rows;
    if (uint64_t(P.perm_cols) * P.perm_rows * 64ull > 0xFFFFFFFFull) { P.error = "pt_render: batch too large (more than 2^32 items per launch)"; return P; }
    P.total_items = P.perm_cols * P.perm_rows * 64u;
    // two batches per claim: what a wavefront still holds when the queue runs dry is what the launch ends on (512: 20-frame launches 16 % slower, lone
    // frames 25 %; 64: one atomic per batch costs 5 % in long launches); whole batches of 64: the kernel generates camera rays a batch at a time
    P.chunk_items = ((PtTune::pick(tune.chunk, 128u) + 63u) / 64u) * 64u;
    if (P.chunk_items < 64u) P.chunk_items = 64u;
    // every wavefront adds chunk_items to a 32-bit cursor once more after it has found the queue dry (once per XCD range with the XCD-aware queue): the
    // cursor must not wrap, or items would be handed out twice and the launch would never end
    if (uint64_t(P.total_items) + uint64_t(grid_lanes / 64u + 1u) * P.chunk_items > 0xFFFFFFFFull) {
        P.error = "pt_render: launch too large for the 32-bit work-queue cursor (more than 2^32 - grid * chunk items; lower spp, the resolution or the batch)"; return P;
    }
    {   // XCD-aware queue for long launches: 8 ranges of the logical item order, one cursor per XCD (chunk aligned; 0 = one queue).  Measured (tools/ab/sweep.sh
        // XCD): 32-frame launches +2..3.5 %, HBM fetch traffic halved (L2 hit rate 85 -> 91 %); no gain at 8 frames of work, a loss for a single frame
        const uint32_t per = (P.total_items + 7u) / 8u;
        P.xcd_span = PtTune::pick(tune.xcd, work8 >= 64u ? 1u : 0u) ? ((per + P.chunk_items - 1u) / P.chunk_items) * P.chunk_items : 0u;
    }
    P.shade_threshold = PtTune::pick(tune.shade, PT_SHADE_THRESHOLD); P.fill_threshold = PtTune::pick(tune.fill, PT_FILL_THRESHOLD);
    P.quad_live = std::min(16u, PtTune::pick(tune.quad, PT_QUAD_LIVE));     // 16 quads per wavefront
    P.fork_shadow = PtTune::pick(tune.fork, PT_FORK_SHADOW);
    P.slots = in.stats ? 1 : slots_for(tune, nf, count, in.sharded);
    // slots are set up for what a FULL batch of the current setting rotates through as well: a partial launch (a warm-up, a flush before a read-back) must
    // not leave the first full one to allocate and prime a slot
    P.setup_slots = in.stats ? 1 : std::max(P.slots, slots_for(tune, std::max(nf, in.batch_size), count, in.sharded));
    return P;
}

// Which owned tiles are traced at all: the union over the launch's frames of the root box's screen rectangle (root_box_rect).  Returns false for
// "every owned tile"; otherwise `traced` lists the owned-tile slots inside `rect`.
bool traced_tiles(const PtContext* ctx, const ptk::RenderArgs& A, uint32_t nf, bool sharded, TileRect& rect, std::vector<uint32_t>& traced) {
    traced.clear(); rect = {0, 0, 0, 0};
    if (!(ctx->have_bvh && ctx->wide_meta.root_ref != pt::kInvalid && !ctx->wide_meta.root_degenerate && A.num_tris != 0u && ctx->tune.cull != 0u)) return false;
    for (uint32_t i = 0; i < nf; ++i) {
        TileRect r;
        if (!root_box_rect(ctx->wide_meta, ctx->pending_frames[i], A.width, A.height, r)) return false;
        if (i == 0) rect = r;
        else { rect.tx0 = std::min(rect.tx0, r.tx0); rect.ty0 = std::min(rect.ty0, r.ty0); rect.tx1 = std::max(rect.tx1, r.tx1); rect.ty1 = std::max(rect.ty1, r.ty1); }
    }
    const uint32_t tiles_y = (A.height + pt::kTile - 1) / pt::kTile;
    if (rect.tx0 == 0u && rect.ty0 == 0u && rect.tx1 >= A.tiles_x && rect.ty1 >= tiles_y) return false;     // nothing to leave out
    auto inside = [&](uint32_t tile) { const uint32_t tx = tile % A.tiles_x, ty = tile / A.tiles_x; return tx >= rect.tx0 && tx < rect.tx1 && ty >= rect.ty0 && ty < rect.ty1; };
    if (sharded) { for (uint32_t sl = 0; sl < A.num_tiles; ++sl) if (inside(ctx->tiles_host[sl])) traced.push_back(sl); }
    else { for (uint32_t ty = rect.ty0; ty < rect.ty1; ++ty) for (uint32_t tx = rect.tx0; tx < rect.tx1; ++tx) traced.push_back(ty * A.tiles_x + tx); }
    return true;
}

// Streams, events and buffers of the frame slots a launch of this plan may use: every slot is sized for a full batch of the current setting (largest
// grid) and its control block zeroed the first time it is needed; from then on the resolve passes rewind it, whatever prefix a later launch uses.
int prepare_slots(PtContext* ctx, const LaunchPlan& P, const ptk::RenderArgs& A, uint32_t nf, bool stats) {
    const size_t n_samples = size_t(A.num_sample_batches) * 64u;
    const size_t cap_samples = std::max(n_samples, size_t(A.batches_per_frame) * 64u * size_t(ctx->batch_size));
    if (cap_samples > 0xFFFFFFFFull) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: batch too large (more than 2^32 samples per launch)");
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    const uint32_t full_lanes = ptk::megakernel_grid(cus) * ptk::megakernel_block();
    for (int si = 0; si < (stats ? 1 : P.setup_slots); ++si) {
        PtContext::FrameSlot& s = ctx->slots[si];
        if (!s.side) {
            PT_HIP(ctx, hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking));
            PT_HIP(ctx, hipEventCreateWithFlags(&s.resolved, hipEventDisableTiming));
            PT_HIP(ctx, hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        }
        PT_HIP(ctx, s.queue.ensure(16));
        PT_HIP(ctx, s.samples.ensure(cap_samples));
        PT_HIP(ctx, s.spill.ensure(size_t(full_lanes) * size_t(64 - PT_SHORT_STACK)));
        PT_HIP(ctx, s.rays.ensure(size_t(full_lanes) * 3u));          // 64 ray records of 3 x uint4 per wavefront
        PT_HIP(ctx, s.frame_params.ensure(std::max<size_t>(ctx->batch_size, nf))); PT_HIP(ctx, s.frame_outs.ensure(std::max<size_t>(ctx->batch_size, nf)));
        if (s.primed_ptr != (const void*)s.samples.ptr || s.primed_samples < cap_samples) {
            if (s.used) PT_HIP(ctx, hipStreamWaitEvent(s.side, s.resolved, 0));
            PT_HIP(ctx, ptk::launch_prime(s.queue.ptr, s.side));
            s.primed_ptr = s.samples.ptr; s.primed_samples = cap_samples;
        }
    }
    return PT_OK;
}

// The list of traced tiles of a launch into the slot's device array, through a pinned staging buffer of the slot; rebuilt only when the rectangle
// (or the tile share) changes.  The copy itself is queued by the caller's launch step (on the slot's stream, behind what it still waits for).
int stage_traced_tiles(PtContext* ctx, PtContext::FrameSlot& sl, const uint32_t key[8], const std::vector<uint32_t>& traced) {
    if (sl.cull_valid && std::memcmp(key, sl.cull_key, 8 * sizeof(uint32_t)) == 0) return PT_OK;
    if (!sl.trace_copied) PT_HIP(ctx, hipEventCreateWithFlags(&sl.trace_copied, hipEventDisableTiming));
    else PT_HIP(ctx, hipEventSynchronize(sl.trace_copied));                   // the previous copy out of the staging buffer is done
    if (sl.h_trace_cap < traced.size()) {
        if (sl.h_trace) (void)hipHostFree(sl.h_trace);
        sl.h_trace = nullptr; sl.h_trace_cap = 0;
        PT_HIP(ctx, hipHostMalloc((void**)&sl.h_trace, std::max<size_t>(traced.size(), 1024) * sizeof(uint32_t), hipHostMallocDefault));
        sl.h_trace_cap = std::max<size_t>(traced.size(), 1024);
    }
    PT_HIP(ctx, sl.trace_slots.ensure(std::max<size_t>(traced.size(), 1)));
    if (!traced.empty()) std::memcpy(sl.h_trace, traced.data(), traced.size() * sizeof(uint32_t));
    sl.cull_valid = false;
    return PT_OK;
}

// Launch the queued frames as one persistent launch (trace on a side stream, resolve on the main stream).
int flush_pending_stats(PtContext* ctx, bool stats, bool sharded, uint32_t count) {
    if (!ctx->pending) return PT_OK;
    ptk::RenderArgs A = ctx->pendingA;
    const uint32_t nf = ctx->pending;
    ctx->pending = 0;
    const bool ring = ctx->ring_used + 2 <= ctx->ring.size();
    hipEvent_t e0 = ring ? ctx->ring[ctx->ring_used] : ctx->ev_start, e1 = ring ? ctx->ring[ctx->ring_used + 1] : ctx->ev_stop;
    A.num_frames = nf;
    if (uint64_t(A.num_tiles) * A.spp * nf * 64ull > 0xFFFFFFFFull)     // item and sample indices are 32-bit
        return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: more than 2^32 pixel-samples in one launch (lower spp, the resolution or the batch)");
    A.batches_per_frame = A.num_tiles * A.spp;
    A.num_sample_batches = A.batches_per_frame * nf;

    // ---- plan ---------------------------------------------------------------------------------------------------------------
    std::vector<uint32_t> traced;                 // owned-tile slots inside the rectangle (cull == false: all of them)
    TileRect rect;
    const bool cull = traced_tiles(ctx, A, nf, sharded, rect, traced);
    A.num_trace_tiles = cull ? uint32_t(traced.size()) : A.num_tiles;
    ctx->stats_culled = 0;
    if (stats && cull) {                          // the oracle traces these rays too: one closest ray, one root record, one sample each
        uint64_t px_all = 0, px_traced = 0;
        auto tile_px = [&](uint32_t tile) { const uint32_t tx = tile % A.tiles_x, ty = tile / A.tiles_x; return uint64_t(std::min(8u, A.width - tx * 8u)) * std::min(8u, A.height - ty * 8u); };
        for (uint32_t sl = 0; sl < A.num_tiles; ++sl) px_all += tile_px(sharded ? ctx->tiles_host[sl] : sl);
        for (uint32_t sl : traced) px_traced += tile_px(sharded ? ctx->tiles_host[sl] : sl);
        ctx->stats_culled = (px_all - px_traced) * A.spp * nf;
    }
    A.trace_bpf = A.num_trace_tiles * A.spp;
    A.num_batches = A.trace_bpf * nf;
    auto magic = [](uint32_t d) { return d > 1u ? uint32_t(0x100000000ull / d) : 0xFFFFFFFFu; };      // d == 1: mulhi gives n - 1 (n > 0), corrected in the kernel
    A.trace_bpf_magic = magic(A.trace_bpf ? A.trace_bpf : 1u); A.spp_magic = magic(A.spp); A.tiles_x_magic = magic(A.tiles_x);
    if (A.num_batches == 0u) { A.trace_bpf = 1u; A.trace_bpf_magic = magic(1u); }      // every owned tile is culled (or there is none): nothing to trace, the resolve pass delivers the primed miss values
    PlanInput in; std::memset(&in, 0, sizeof(in));
    in.num_cus = uint32_t(ctx->num_cus > 0 ? ctx->num_cus : 256); in.frames = nf; in.tile_count = count; in.sharded = sharded; in.stats = stats;
    in.traced_batches = A.num_batches; in.batch_size = ctx->batch_size;
    if (count < 2u && nf == 1u && !stats) {       // a single whole frame: how many earlier launches are still in flight?
        for (const auto& o : ctx->slots) if (o.side && o.used && hipEventQuery(o.done) == hipErrorNotReady) ++in.launches_in_flight;
        (void)hipGetLastError();                  // hipErrorNotReady is an answer, not an error: do not leave it behind for the launch checks
    }
    const LaunchPlan P = plan_launch(ctx->tune, in);
    if (P.error) return fail(ctx, PT_ERR_INVALID_ARG, P.error);
    A.perm_rows = P.perm_rows; A.perm_rows_magic = magic(P.perm_rows); A.perm_cols = P.perm_cols; A.total_items = P.total_items;
    A.chunk_items = P.chunk_items; A.xcd_span = P.xcd_span;
    A.shade_threshold = P.shade_threshold; A.fill_threshold = P.fill_threshold; A.quad_live = P.quad_live; A.fork_shadow = P.fork_shadow;
    A.rcp_short = (PtTune::pick(ctx->tune.bounded, 1u) != 0u && arith_is_bounded(ctx, ctx->pending_frames, nf)) ? 1u : 0u;      // knob BOUNDED = 0: always the general forms (tests)

    // ---- prepare: frame slots (instrumented launches always use slot 0 and are not overlapped), traced-tile list ------------------------
    ctx->num_slots = P.slots;
    if (int rc = prepare_slots(ctx, P, A, nf, stats)) return rc;
    PtContext::FrameSlot& sl = ctx->slots[stats ? 0 : (ctx->next_slot++ % uint32_t(P.slots))];
    A.prime = stats ? 1u : 0u;                   // instrumented launches start from a freshly primed prefix
    A.samples = sl.samples.ptr; A.queue = sl.queue.ptr; A.spill = sl.spill.ptr; A.raybuf = sl.rays.ptr;
    A.trace_slots = nullptr;
    const uint32_t cull_key[8] = {rect.tx0, rect.ty0, rect.tx1, rect.ty1, A.width, A.height, ctx->pending_rank | (ctx->pending_count << 16), A.num_trace_tiles};
    if (cull) { if (int rc = stage_traced_tiles(ctx, sl, cull_key, traced)) return rc; }
    // knob TIMELINE = 1 (diagnostics): an ordinary launch -- its plan, its slot, its overlap with its neighbours all as in production -- runs the TIMELINE variant
    // of the kernel and leaves the per-wavefront record of pt_debug_wave_times (the last such launch's: read it after a pt_synchronize)
    const bool timeline = !stats && PtTune::pick(ctx->tune.timeline, 0u) != 0u;
    if (stats || timeline) {
        const uint32_t stat_waves = P.grid * ptk::megakernel_block() / 64u;
        PT_HIP(ctx, ctx->d_wave_times.ensure(size_t(stat_waves) * ptk::kWaveTimeWords));
        A.wave_times = ctx->d_wave_times.ptr; ctx->wave_times_n = stat_waves;
    }

    // ---- launch ---------------------------------------------------------------------------------------------------------------
    // Dependencies: scene uploads are host-synchronous, so the trace only has to wait for the resolve that last read this slot's sample buffer (NOT
    // for the previous frame's resolve -- that is what lets consecutive frames overlap); the resolve on the main stream waits for the trace.
    if (!ring) PT_HIP(ctx, hipEventRecord(e0, ctx->stream));
    if (sl.used) PT_HIP(ctx, hipStreamWaitEvent(sl.side, sl.resolved, 0));
    if (cull) {
        if (!sl.cull_valid) {
            if (!traced.empty()) PT_HIP(ctx, hipMemcpyAsync(sl.trace_slots.ptr, sl.h_trace, traced.size() * sizeof(uint32_t), hipMemcpyHostToDevice, sl.side));
            PT_HIP(ctx, hipEventRecord(sl.trace_copied, sl.side));
            std::memcpy(sl.cull_key, cull_key, sizeof cull_key); sl.cull_valid = true; sl.num_trace_tiles = A.num_trace_tiles;
        }
        A.trace_slots = sl.trace_slots.ptr;
        A.trace_rect[0] = rect.tx0; A.trace_rect[1] = rect.tx1; A.trace_rect[2] = rect.ty0; A.trace_rect[3] = rect.ty1;
    }
    if (stats) {
        // the counter blocks are zeroed on the stream the instrumented kernel runs on, behind everything that stream still has to wait for; the previous
        // reader (pt_get_stats / pt_debug_*) copied them synchronously
        PT_HIP(ctx, hipMemsetAsync(ctx->d_stats.ptr, 0, 24 * sizeof(unsigned long long), sl.side));
        PT_HIP(ctx, hipMemsetAsync(ctx->d_wave_times.ptr, 0, size_t(ctx->wave_times_n) * ptk::kWaveTimeWords * 8u, sl.side));
    }
    if (timeline) PT_HIP(ctx, hipMemsetAsync(ctx->d_wave_times.ptr, 0, size_t(ctx->wave_times_n) * ptk::kWaveTimeWords * 8u, sl.side));
    {   // per-frame parameters and targets into the slot's device arrays; a frame whose target a later frame of this launch
        // overwrites is marked (its result would not survive one-launch-per-frame rendering either)
        std::vector<ptk::FrameParams>& F = ctx->pending_frames;
        for (uint32_t i = 0; i < nf; ++i) {
            bool superseded = false;
            if ((F[i].accum_mode & 0xffu) == 0u)
                for (uint32_t g = i + 1u; g < nf && !superseded; ++g) superseded = ctx->pending_outs[g] == ctx->pending_outs[i];
            F[i].accum_mode = (F[i].accum_mode & 0xffu) | (superseded ? 0x100u : 0u);
        }
        PT_HIP(ctx, ptk::launch_frame_params(F.data(), ctx->pending_outs.data(), nf, sl.frame_params.ptr, sl.frame_outs.ptr, sl.side));
        A.frames = sl.frame_params.ptr; A.outs = sl.frame_outs.ptr;
    }
    // timing ring: events tightly around the trace kernels on the stream they run on
    PT_HIP(ctx, ptk::launch_trace(A, stats, P.grid, sl.side, ring ? e0 : nullptr, ring ? e1 : nullptr));
    PT_HIP(ctx, hipEventRecord(sl.done, sl.side));
    PT_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.done, 0));
    PT_HIP(ctx, ptk::launch_resolve(A, ctx->stream));
    PT_HIP(ctx, hipEventRecord(sl.resolved, ctx->stream)); sl.used = true;
    sl.resolved_seq = track_targets(ctx, ctx->pending_outs.data(), nf, (A.compact ? size_t(A.num_tiles) * 64u : size_t(A.width) * A.height) * sizeof(float4));
    if (!ring) PT_HIP(ctx, hipEventRecord(e1, ctx->stream));
    if (ring) ctx->ring_used += 2;
    ctx->timed = !ring;
    return PT_OK;
}

int flush_pending(PtContext* ctx) {
    if (!ctx->pending) return PT_OK;
    const uint32_t count = ctx->tiles_count ? ctx->tiles_count : 1u;
    const bool sharded = ctx->pendingA.compact != 0u;
    return flush_pending_stats(ctx, ctx->pendingA.stats != nullptr, sharded, sharded ? count : 1u);
}

} // namespace

extern "C" {

#define PT_STR2(x) #x
#define PT_STR(x) PT_STR2(x)
// names the build options that change what the megakernel does
const char* pt_version(void) { return "mi355pt 0.1 (gfx950; workgroup " PT_STR(PT_MEGA_BLOCK) ", quad " PT_STR(PT_QUAD) ")"; }

const char* pt_last_error(const PtContext* ctx) { return ctx ? ctx->err.c_str() : g_global_error.c_str(); }

int pt_create(int device_ordinal, PtContext** out) {
    if (!out) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_create: null out pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(nullptr, PT_ERR_NO_DEVICE, std::string("pt_create: no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0") + "); libmi355pt has no CPU path");
    int dev = device_ordinal;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= count) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_create: device ordinal out of range");
    e = hipSetDevice(dev);
    if (e != hipSuccess) return fail_hip(nullptr, e, "hipSetDevice");
    PtContext* ctx = new PtContext();
    ctx->device = dev;
    ctx->tune.from_environment();
    {   // the context's stream carries the short resolve passes: highest priority, so their blocks are placed ahead of
        // the persistent trace launches (normal-priority side streams) whenever CU slots free up
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_start);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_stop);
    if (e == hipSuccess) e = ctx->d_stats.ensure(24);
    if (e == hipSuccess) { hipDeviceProp_t prop; e = hipGetDeviceProperties(&prop, dev); if (e == hipSuccess) ctx->num_cus = prop.multiProcessorCount; }
    if (e != hipSuccess) { int rc = fail_hip(nullptr, e, "pt_create"); pt_destroy(ctx); return rc; }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return PT_OK;
}

void pt_destroy(PtContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)flush_pending(ctx);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    ctx->d_tris9.release(); ctx->d_scene.release(); ctx->d_bvh2.release(); ctx->d_bvh4.release();
    ctx->d_spheres.release(); ctx->d_morton.release(); ctx->d_triidx.release(); ctx->d_parent.release(); ctx->d_flags.release();
    ctx->d_out.release(); ctx->d_accum.release(); ctx->d_compact.release(); ctx->d_compact_accum.release();
    ctx->d_tiles.release(); ctx->d_u32tmp.release(); ctx->d_stats.release();
    ctx->d_wave_times.release();
    ctx->d_bounds.release(); ctx->d_counters.release(); ctx->d_code_tmp.release(); ctx->d_index_tmp.release(); ctx->d_node2.release();
    ctx->d_subtree.release(); ctx->d_ids.release(); ctx->d_bnd.release(); ctx->d_child_pos.release(); ctx->d_build_temp.release();
    if (ctx->h_word) (void)hipHostFree(ctx->h_word);
    for (auto& sl : ctx->slots) {
        sl.queue.release(); sl.samples.release(); sl.spill.release(); sl.rays.release(); sl.trace_slots.release();
        if (sl.h_trace) (void)hipHostFree(sl.h_trace);
        if (sl.trace_copied) (void)hipEventDestroy(sl.trace_copied); sl.frame_params.release(); sl.frame_outs.release();
        if (sl.resolved) (void)hipEventDestroy(sl.resolved);
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.side) { (void)hipStreamSynchronize(sl.side); (void)hipStreamDestroy(sl.side); }
    }
    for (hipEvent_t e : ctx->ring) (void)hipEventDestroy(e);
    if (ctx->misc_fence) (void)hipEventDestroy(ctx->misc_fence);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int pt_set_stream(PtContext* ctx, void* hip_stream) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    everything_delivered(ctx);
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    ctx->timed = false;
    return PT_OK;
}

int pt_get_stream(PtContext* ctx, void** hip_stream) {
    if (!ctx || !hip_stream) return fail(ctx, PT_ERR_INVALID_ARG, "pt_get_stream: null argument");
    *hip_stream = (void*)ctx->stream;
    return PT_OK;
}

int pt_synchronize(PtContext* ctx) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    everything_delivered(ctx);
    return PT_OK;
}

// ---- host-side scene build ------------------------------------------------------------

int pt_compute_bvh2_sizing(uint32_t num_tris, uint32_t* num_nodes2, uint64_t* bytes) {
    const uint32_t nn = num_tris ? 2 * num_tris - 1 : 0;
    if (num_nodes2) *num_nodes2 = nn;
    if (bytes) *bytes = num_tris ? 4ull * (1ull + 6ull * nn) : 4ull;
    return PT_OK;
}
int pt_compute_bvh4_sizing(uint32_t num_nodes4, uint64_t* bytes) {
    if (bytes) *bytes = num_nodes4 ? 4ull * (1ull + 8ull * num_nodes4) : 4ull;
    return PT_OK;
}
int pt_morton_sort(const float* tris, uint32_t num_tris, uint32_t* morton_sorted, uint32_t* tri_index_sorted) {
    if (num_tris && (!tris || !morton_sorted || !tri_index_sorted)) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_morton_sort: null pointer");
    pt::morton_codes_sorted(tris, num_tris, morton_sorted, tri_index_sorted);
    return PT_OK;
}
int pt_collapse_lbvh2_to_bvh4(const uint32_t* bvh2, uint32_t num_tris, uint32_t* out, uint64_t out_words, uint32_t* num_nodes4) {
    if (!out || (num_tris && !bvh2)) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_collapse_lbvh2_to_bvh4: null pointer");
    std::vector<uint32_t> v; std::string err;
    if (!pt::collapse_to_bvh4(bvh2, num_tris, v, err)) return fail(nullptr, PT_ERR_BAD_BVH, err);
    if (v.size() > out_words) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_collapse_lbvh2_to_bvh4: output buffer too small");
    std::memcpy(out, v.data(), v.size() * 4);
    if (num_nodes4) *num_nodes4 = v[0];
    return PT_OK;
}
int pt_bvh2_to_bvh4_wide(const uint32_t* bvh2, uint64_t bvh2_words, uint32_t* out, uint64_t out_words) {
    if (!bvh2 || !out) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_bvh2_to_bvh4_wide: null pointer");
    std::vector<uint32_t> v; std::string err;
    if (!pt::promote_to_bvh4_wide(bvh2, bvh2_words, v, err)) return fail(nullptr, PT_ERR_BAD_BVH, err);
    if (v.size() > out_words) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_bvh2_to_bvh4_wide: output buffer too small");
    std::memcpy(out, v.data(), v.size() * 4);
    return PT_OK;
}
int pt_file_write_u32(const char* path, const uint32_t* src, uint64_t words) {
    if (!path || (!src && words)) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_file_write_u32: null pointer");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(nullptr, PT_ERR_IO, std::string("cannot open for writing: ") + path);
    const size_t n = std::fwrite(src, 4, words, f);
    const int rc = std::fclose(f);
    if (n != words || rc != 0) return fail(nullptr, PT_ERR_IO, std::string("short write: ") + path);
    return PT_OK;
}
int pt_file_read_u32(const char* path, uint32_t* dst, uint64_t dst_words, uint64_t* words) {
    if (!path) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_file_read_u32: null path");
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(nullptr, PT_ERR_IO, std::string("cannot open: ") + path);
    std::fseek(f, 0, SEEK_END);
    const long size = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (size <= 0 || (size & 3)) { std::fclose(f); return fail(nullptr, PT_ERR_IO, std::string("not a u32 dump (size): ") + path); }   // tests/test.cpp:20
    const uint64_t n = uint64_t(size) / 4;
    if (words) *words = n;
    if (!dst) { std::fclose(f); return PT_OK; }
    if (dst_words < n) { std::fclose(f); return fail(nullptr, PT_ERR_INVALID_ARG, "pt_file_read_u32: destination too small"); }
    const size_t got = std::fread(dst, 4, n, f);
    std::fclose(f);
    if (got != n) return fail(nullptr, PT_ERR_IO, std::string("short read: ") + path);
    return PT_OK;
}
int pt_scene_procedural(uint32_t kind, uint32_t seed, uint32_t num_tris, float* tris_out) {
    std::string err;
    if (!pt::procedural_scene(kind, seed, num_tris, tris_out, err)) return fail(nullptr, PT_ERR_INVALID_ARG, err);
    return PT_OK;
}

// ---- device scene state ----------------------------------------------------------------

int pt_set_triangles(PtContext* ctx, const float* tris, uint32_t num_tris) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (num_tris && !tris) return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_triangles: null triangles");
    if (num_tris >= 0x7fffffffu) return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_triangles: too many triangles for the 31-bit leaf index");
    PT_HIP(ctx, ctx->d_tris9.ensure(size_t(num_tris) * 9));
    ctx->scene_tris = ~0u;                          // new triangles: nothing in the arena is worth keeping
    if (int rc = ensure_scene(ctx, num_tris, uint64_t(num_tris) + 16u)) return rc;
    uint32_t edge_bits = 0u;
    if (num_tris) {
        PT_HIP(ctx, ctx->d_edge_max.ensure(1));
        PT_HIP(ctx, hipMemsetAsync(ctx->d_edge_max.ptr, 0, sizeof(uint32_t), ctx->stream));
        PT_HIP(ctx, hipMemcpyAsync(ctx->d_tris9.ptr, tris, size_t(num_tris) * 36, hipMemcpyHostToDevice, ctx->stream));
        PT_HIP(ctx, ptk::launch_tri_records(ctx->d_tris9.ptr, num_tris, ctx->trirec(), ctx->d_edge_max.ptr, ctx->stream));   // 64 B records, DESIGN.md section 5
        PT_HIP(ctx, hipMemcpyAsync(&edge_bits, ctx->d_edge_max.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    }
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->edges_small = edge_bits < 0x49800000u;     // 2^20 as f32 bits (NaN and infinity lie above)
    ctx->num_tris = num_tris;
    ctx->have_tris = true;
    ctx->have_bvh = false; ctx->have_bvh2 = false; ctx->bvh2_refit_pending = false;
    ctx->accum_count = 0;
    return PT_OK;
}

int pt_build_lbvh2(PtContext* ctx, const uint32_t* morton_sorted, const uint32_t* tri_index_sorted) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!ctx->have_tris) return fail(ctx, PT_ERR_NO_SCENE, "pt_build_lbvh2: no triangles uploaded");
    const uint32_t n = ctx->num_tris;
    uint32_t nn2 = 0; uint64_t bytes = 0;
    pt_compute_bvh2_sizing(n, &nn2, &bytes);
    PT_HIP(ctx, ctx->d_bvh2.ensure(bytes / 4));
    ctx->num_nodes2 = nn2;
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_bvh2.ptr, &nn2, 4, hipMemcpyHostToDevice, ctx->stream));   // BVH2[0] = numNodes2, PathTracer.js:699
    if (n == 0) { PT_HIP(ctx, hipStreamSynchronize(ctx->stream)); ctx->have_bvh2 = true; return PT_OK; }
    if (!morton_sorted || !tri_index_sorted) return fail(ctx, PT_ERR_INVALID_ARG, "pt_build_lbvh2: null sorted arrays");
    for (uint32_t i = 0; i < n; ++i)
        if (tri_index_sorted[i] >= n) return fail(ctx, PT_ERR_INVALID_ARG, "pt_build_lbvh2: triangle index out of range");
    for (uint32_t i = 1; i < n; ++i)
        if (morton_sorted[i] < morton_sorted[i - 1]) return fail(ctx, PT_ERR_INVALID_ARG, "pt_build_lbvh2: Morton codes are not sorted");
    PT_HIP(ctx, ctx->d_morton.ensure(n)); PT_HIP(ctx, ctx->d_triidx.ensure(n));
    PT_HIP(ctx, ctx->d_parent.ensure(nn2)); PT_HIP(ctx, ctx->d_flags.ensure(n > 1 ? n - 1 : 1));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_morton.ptr, morton_sorted, size_t(n) * 4, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_triidx.ptr, tri_index_sorted, size_t(n) * 4, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, ptk::launch_lbvh2(ctx->d_bvh2.ptr, ctx->d_tris9.ptr, ctx->d_morton.ptr, ctx->d_triidx.ptr, ctx->d_parent.ptr, ctx->d_flags.ptr, n, true, ctx->stream));
    ctx->bvh2_refit_pending = false;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));   // device.queue.onSubmittedWorkDone(), PathTracer.js:727
    ctx->have_bvh2 = true;
    return PT_OK;
}

int pt_read_bvh2(PtContext* ctx, uint32_t* dst, uint64_t bytes) {
    if (int rc = bind(ctx)) return rc;
    if (!ctx->have_bvh2) return fail(ctx, PT_ERR_NO_SCENE, "pt_read_bvh2: no BVH2 on the device");
    if (!dst) return fail(ctx, PT_ERR_INVALID_ARG, "pt_read_bvh2: null destination");
    uint64_t have = 0; pt_compute_bvh2_sizing(ctx->num_tris, nullptr, &have);
    const uint64_t n = bytes < have ? bytes : have;
    if (ctx->bvh2_refit_pending) {                  // BVHBuilder.wgsl:242-275, deferred by pt_build_bvh
        PT_HIP(ctx, ptk::launch_lbvh2_refit(ctx->d_bvh2.ptr, ctx->d_parent.ptr, ctx->d_flags.ptr, ctx->num_tris, ctx->stream));
        ctx->bvh2_refit_pending = false;
    }
    PT_HIP(ctx, hipMemcpyAsync(dst, ctx->d_bvh2.ptr, n, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_set_bvh4(PtContext* ctx, const uint32_t* bvh4, uint64_t words) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!bvh4 || words < 1) return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_bvh4: empty buffer");
    ctx->accum_count = 0;
    return upload_wide(ctx, bvh4, words);
}

int pt_set_bvh2(PtContext* ctx, const uint32_t* bvh2, uint64_t words) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!ctx->have_tris) return fail(ctx, PT_ERR_NO_SCENE, "pt_set_bvh2: upload triangles first");
    uint32_t nn2 = 0; uint64_t bytes = 0; pt_compute_bvh2_sizing(ctx->num_tris, &nn2, &bytes);
    if (!bvh2 || words * 4 < bytes || bvh2[0] != nn2) return fail(ctx, PT_ERR_BAD_BVH, "pt_set_bvh2: buffer does not match 2N-1 nodes of the uploaded triangles");
    PT_HIP(ctx, ctx->d_bvh2.ensure(bytes / 4));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_bvh2.ptr, bvh2, bytes, hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->num_nodes2 = nn2; ctx->have_bvh2 = true; ctx->bvh2_refit_pending = false;
    std::vector<uint32_t> b4; std::string err;
    if (!pt::collapse_to_bvh4(bvh2, ctx->num_tris, b4, err)) return fail(ctx, PT_ERR_BAD_BVH, err);
    ctx->accum_count = 0;
    return upload_wide(ctx, b4.data(), b4.size());
}

int pt_build_bvh(PtContext* ctx) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!ctx->have_tris) return fail(ctx, PT_ERR_NO_SCENE, "pt_build_bvh: no triangles uploaded");
    const uint32_t n = ctx->num_tris;
    if (n == 0) {                                   // PathTracer.js:701-707: empty BVH4
        const uint32_t zero = 0;
        if (int rc = pt_build_lbvh2(ctx, nullptr, nullptr)) return rc;
        return upload_wide(ctx, &zero, 1);
    }
    // Every step of PathTracer.buildBVH (:671-749) on the device, nothing crosses PCIe but a few counters:
    // Morton codes + stable radix sort (:411-481), LBVH2 kernels (BVHBuilder.wgsl), collapse to BVH4 (:506-667), device layouts.
    uint32_t nn2 = 0; uint64_t bytes2 = 0;
    pt_compute_bvh2_sizing(n, &nn2, &bytes2);
    PT_HIP(ctx, ctx->d_bvh2.ensure(bytes2 / 4));
    PT_HIP(ctx, ctx->d_morton.ensure(n)); PT_HIP(ctx, ctx->d_triidx.ensure(n));
    PT_HIP(ctx, ctx->d_parent.ensure(nn2)); PT_HIP(ctx, ctx->d_flags.ensure(n > 1 ? n - 1 : 1));
    PT_HIP(ctx, ctx->d_bounds.ensure(6)); PT_HIP(ctx, ctx->d_counters.ensure(ptk::kBuildCounters));
    PT_HIP(ctx, ctx->d_code_tmp.ensure(n)); PT_HIP(ctx, ctx->d_index_tmp.ensure(n));
    PT_HIP(ctx, ctx->d_node2.ensure(nn2)); PT_HIP(ctx, ctx->d_child_pos.ensure(nn2)); PT_HIP(ctx, ctx->d_subtree.ensure(nn2));
    PT_HIP(ctx, ctx->d_ids.ensure(nn2)); PT_HIP(ctx, ctx->d_bnd.ensure(size_t(nn2) * 3));
    const size_t temp_bytes = ptk::build_temp_bytes(n);
    PT_HIP(ctx, ctx->d_build_temp.ensure(temp_bytes));
    if (!ctx->h_word) PT_HIP(ctx, hipHostMalloc((void**)&ctx->h_word, 64, hipHostMallocDefault));
    PT_HIP(ctx, ctx->d_bvh4.ensure(1 + size_t(nn2) * 8));      // M <= 2N-1 nodes
    ptk::BuildBuffers B;
    B.bounds = ctx->d_bounds.ptr; B.counters = ctx->d_counters.ptr; B.code_tmp = ctx->d_code_tmp.ptr; B.index_tmp = ctx->d_index_tmp.ptr;
    B.morton = ctx->d_morton.ptr; B.tri_index = ctx->d_triidx.ptr; B.temp = ctx->d_build_temp.ptr; B.temp_bytes = temp_bytes;
    B.node2 = ctx->d_node2.ptr; B.child_pos = ctx->d_child_pos.ptr; B.subtree = ctx->d_subtree.ptr; B.ids = ctx->d_ids.ptr; B.bnd = ctx->d_bnd.ptr;
    B.host_word = ctx->h_word;
    ctx->have_bvh = false; ctx->have_bvh2 = false;
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_bvh2.ptr, &nn2, 4, hipMemcpyHostToDevice, ctx->stream));   // BVH2[0] = numNodes2, PathTracer.js:699
    PT_HIP(ctx, ptk::launch_morton_sort(B, ctx->d_tris9.ptr, n, ctx->stream));
    PT_HIP(ctx, ptk::launch_lbvh2(ctx->d_bvh2.ptr, ctx->d_tris9.ptr, ctx->d_morton.ptr, ctx->d_triidx.ptr, ctx->d_parent.ptr, ctx->d_flags.ptr, n, false, ctx->stream));
    ctx->bvh2_refit_pending = true;                 // internal BVH2 bounds: on demand (pt_read_bvh2)
    ctx->num_nodes2 = nn2;
    uint32_t m = 0;
    {
        hipError_t e = ptk::collapse_on_device(B, ctx->d_bvh2.ptr, n, ctx->d_bvh4.ptr, &m, ctx->stream);
        if (e == hipErrorInvalidValue) return fail(ctx, PT_ERR_BAD_BVH, "pt_build_bvh: the LBVH2 is not a tree of 2N-1 nodes");
        PT_HIP(ctx, e);
    }
    ctx->have_bvh2 = true;
    PT_HIP(ctx, ptk::launch_internal_scan(B, ctx->d_bvh4.ptr, m, ctx->stream));
    uint32_t tail[2] = {0, 0}, root[8];
    PT_HIP(ctx, hipMemcpyAsync(&tail[0], ctx->d_ids.ptr + (m - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipMemcpyAsync(&tail[1], ctx->d_subtree.ptr + (m - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipMemcpyAsync(root, ctx->d_bvh4.ptr + 1, 32, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t internal = tail[0] + tail[1];
    if (int rc = ensure_scene(ctx, n, internal)) return rc;
    PT_HIP(ctx, ptk::launch_wide_nodes(B, ctx->d_bvh4.ptr, m, ctx->wide(), n, uint32_t(ctx->node_off / 16u), ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pt::WideBvh meta;
    meta.num_nodes4 = m;
    meta.root_box[0] = root[0]; meta.root_box[1] = root[1]; meta.root_box[2] = root[2];
    meta.root_degenerate = pt::half_to_float(root[0] & 0xffffu) > pt::half_to_float(root[1] >> 16) || pt::half_to_float(root[0] >> 16) > pt::half_to_float(root[2] & 0xffffu) ||
                           pt::half_to_float(root[1] & 0xffffu) > pt::half_to_float(root[2] >> 16);
    meta.root_ref = (root[7] & pt::kLeafFlag) ? pt::packed_leaf_ref(root[7] & 0x7fffffffu, n) : uint32_t(ctx->node_off / 16u);
    ctx->wide_meta = meta;
    ctx->num_nodes4 = m;
    ctx->have_bvh = true;
    ctx->accum_count = 0;
    return PT_OK;
}

int pt_read_bvh4(PtContext* ctx, uint32_t* dst, uint64_t bytes) {
    if (int rc = bind(ctx)) return rc;
    if (!ctx->have_bvh) return fail(ctx, PT_ERR_NO_SCENE, "pt_read_bvh4: no BVH on the device");
    if (!dst) return fail(ctx, PT_ERR_INVALID_ARG, "pt_read_bvh4: null destination");
    const uint64_t have = 4ull * (1ull + 8ull * ctx->num_nodes4);
    const uint64_t n = bytes < have ? bytes : have;
    PT_HIP(ctx, hipMemcpyAsync(dst, ctx->d_bvh4.ptr, n, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_set_spheres(PtContext* ctx, const float* xyzr, uint32_t num_spheres) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (num_spheres && !xyzr) return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_spheres: null spheres");
    PT_HIP(ctx, ctx->d_spheres.ensure(num_spheres));
    if (num_spheres) PT_HIP(ctx, hipMemcpy(ctx->d_spheres.ptr, xyzr, size_t(num_spheres) * 16, hipMemcpyHostToDevice));
    ctx->num_spheres = num_spheres; ctx->accum_count = 0;
    return PT_OK;
}

int pt_scene_info(PtContext* ctx, uint32_t* num_tris, uint32_t* num_nodes2, uint32_t* num_nodes4) {
    if (!ctx) return fail(nullptr, PT_ERR_INVALID_ARG, "null context");
    if (num_tris) *num_tris = ctx->num_tris;
    if (num_nodes2) *num_nodes2 = ctx->have_bvh2 ? ctx->num_nodes2 : 0;
    if (num_nodes4) *num_nodes4 = ctx->have_bvh ? ctx->num_nodes4 : 0;
    return PT_OK;
}

// ---- the hot path -----------------------------------------------------------------------

int pt_tile_layout(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count, uint32_t* num_tiles, uint64_t* compact_floats) {
    if (tile_count == 0) tile_count = 1;
    if (tile_rank >= tile_count) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_tile_layout: rank >= count");
    const uint32_t n = pt::tile_count_of(width, height, tile_rank, tile_count);      // closed form: no list is built
    if (num_tiles) *num_tiles = n;
    if (compact_floats) *compact_floats = uint64_t(n) * 64ull * 4ull;
    return PT_OK;
}

int pt_tile_ids(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count, uint32_t* ids, uint32_t capacity, uint32_t* num_tiles) {
    if (tile_count == 0) tile_count = 1;
    if (tile_rank >= tile_count) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_tile_ids: rank >= count");
    std::vector<uint32_t> t; pt::tile_list(width, height, tile_rank, tile_count, t);      // the list pt_render uploads for this share
    if (num_tiles) *num_tiles = uint32_t(t.size());
    if (ids) {
        if (capacity < t.size()) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_tile_ids: destination too small");
        if (!t.empty()) std::memcpy(ids, t.data(), t.size() * sizeof(uint32_t));
    }
    return PT_OK;
}

int pt_render(PtContext* ctx, const PtRenderParams* p) {
    if (int rc = bind(ctx)) return rc;
    if (!p) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: null params");
    const bool brute = (p->flags & PT_FLAG_BRUTE_FORCE) != 0;
    if (!ctx->have_tris || (!ctx->have_bvh && !brute)) return fail(ctx, PT_ERR_NO_SCENE, "pt_render: scene not set (triangles + BVH)");   // PathTracer.js:757
    if (brute && p->mode == PT_MODE_REFERENCE_PACKET) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: brute-force scenes render in modes 1 and 2");
    if (p->width == 0 || p->height == 0 || p->width > 32768 || p->height > 32768) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: bad resolution");
    if (p->num_tris > ctx->num_tris) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: num_tris exceeds the uploaded triangle count");
    if (p->mode > PT_MODE_PATH) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: unknown mode");
    if (p->mode == PT_MODE_PATH && p->spp == 0) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: spp must be >= 1");
    const uint32_t count = p->tile_count ? p->tile_count : 1;
    if (p->tile_rank >= count) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: tile_rank >= tile_count");
    if (count > 0xfffu) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: tile_count > 4095 (the share a running accumulation covers is remembered as 12 + 12 bits; pt_set_accum has the same limit)");
    const bool sharded = count > 1 || (p->flags & PT_FLAG_COMPACT) != 0;
    const bool stats = (p->flags & PT_FLAG_STATS) != 0;

    // A frame joins the open batch only if it has the same shape (resolution, spp, bounces, triangle count, tile share,
    // accumulation) and runs on the megakernel; anything else launches the open batch FIRST -- before any output buffer
    // is resized or the tile list is rewritten under the queued frames.
    if (ctx->pending) {
        const ptk::RenderArgs& Q = ctx->pendingA;
        const bool ref = p->mode == PT_MODE_REFERENCE;
        const bool stats_joins = stats && PtTune::pick(ctx->tune.stats_batch, 0u) != 0u && Q.stats != nullptr;      // diagnostics: an instrumented multi-frame launch (tools/wave_timeline.py)
        const bool mega = (p->mode == PT_MODE_PATH || ref) && !(p->flags & PT_FLAG_SIMPLE_KERNEL) && !brute && (!stats || stats_joins);
        const bool same = mega && Q.width == p->width && Q.height == p->height && Q.ref_mode == (ref ? 1u : 0u) && (ref || (Q.spp == p->spp && Q.max_bounces == p->max_bounces)) &&
                          Q.num_tris == p->num_tris && ctx->pending_rank == p->tile_rank && ctx->pending_count == count && (Q.accum != nullptr) == (!ref && p->accumulate != 0);
        if (!same) { if (int rc = flush_pending(ctx)) return rc; }
    }

    ptk::RenderArgs A; std::memset(&A, 0, sizeof(A));
    A.nodes = ctx->wide(); A.tris = ctx->trirec(); A.scene = ctx->d_scene.ptr; A.node_off = uint32_t(ctx->node_off); A.bvh4_ref = ctx->d_bvh4.ptr; A.tris9 = ctx->d_tris9.ptr;
    A.width = p->width; A.height = p->height; A.focal = p->focal; A.aspect = p->aspect;
    std::memcpy(A.cam, p->cam_pos, 12); std::memcpy(A.quat, p->cam_quat, 16);
    A.num_tris = p->num_tris; A.frame = p->frame;
    A.tri_gate = p->num_tris < ctx->num_tris ? (0x80000000u | (4u * p->num_tris)) : 0xFFFFFFFFu;
    A.root_ref = ctx->wide_meta.root_ref; std::memcpy(A.root_box, ctx->wide_meta.root_box, 12);
    A.root_degenerate = ctx->wide_meta.root_degenerate ? 1u : 0u;
    A.spheres = ctx->d_spheres.ptr; A.num_spheres = brute ? ctx->num_spheres : 0u; A.brute = brute ? 1u : 0u;
    A.spp = p->spp; A.max_bounces = p->max_bounces; A.seed = p->seed; A.accumulate = 0; A.compact = sharded ? 1u : 0u;
    A.tiles_x = (p->width + pt::kTile - 1) / pt::kTile;
    const uint32_t tiles_y = (p->height + pt::kTile - 1) / pt::kTile;

    const size_t npx = size_t(p->width) * p->height;
    if (sharded) {
        if (ctx->tiles_w != p->width || ctx->tiles_h != p->height || ctx->tiles_rank != p->tile_rank || ctx->tiles_count != count) {
            pt::tile_list(p->width, p->height, p->tile_rank, count, ctx->tiles_host);
            PT_HIP(ctx, ctx->d_tiles.ensure(ctx->tiles_host.size()));
            if (!ctx->tiles_host.empty())
                PT_HIP(ctx, hipMemcpyAsync(ctx->d_tiles.ptr, ctx->tiles_host.data(), ctx->tiles_host.size() * 4, hipMemcpyHostToDevice, ctx->stream));
            PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->tiles_w = p->width; ctx->tiles_h = p->height; ctx->tiles_rank = p->tile_rank; ctx->tiles_count = count;
        }
        A.tiles = ctx->d_tiles.ptr; A.num_tiles = uint32_t(ctx->tiles_host.size());
        ctx->compact_floats = uint64_t(A.num_tiles) * 64ull * 4ull;
        if (ctx->ext_compact) {
            if (ctx->ext_compact_floats < ctx->compact_floats) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: caller-owned compact buffer too small");
            A.out = ctx->ext_compact;
        } else {
            PT_HIP(ctx, ctx->d_compact.ensure(size_t(A.num_tiles) * 64));
            A.out = ctx->d_compact.ptr;
        }
    } else {
        A.tiles = nullptr; A.num_tiles = A.tiles_x * tiles_y;
        if (ctx->ext_out) {
            if (ctx->ext_out_floats < uint64_t(npx) * 4ull) return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: caller-owned output buffer too small");
            A.out = ctx->ext_out;
        } else {
            PT_HIP(ctx, ctx->d_out.ensure(npx));
            A.out = ctx->d_out.ptr;
        }
        ctx->last_full = A.out;
        ctx->out_w = p->width; ctx->out_h = p->height;
    }
    if (p->mode == PT_MODE_PATH && p->accumulate) {
        DevBuf<float4>& acc = sharded ? ctx->d_compact_accum : ctx->d_accum;
        const size_t need = sharded ? size_t(A.num_tiles) * 64 : npx;
        const uint32_t share_key = accum_share_key(p->tile_rank, count, sharded);
        const bool cont = ctx->accum_count > 0 && ctx->accum_w == p->width && ctx->accum_h == p->height && ctx->accum_rank == share_key && acc.cap >= need;
        PT_HIP(ctx, acc.ensure(need));
        A.accum = acc.ptr; A.accumulate = cont ? 1u : 0u;
        ctx->accum_w = p->width; ctx->accum_h = p->height; ctx->accum_rank = share_key;
        ctx->accum_count = cont ? ctx->accum_count + p->spp : p->spp;
    } else {
        ctx->accum_count = 0;
    }
    // The persistent megakernel traces PT_MODE_PATH and PT_MODE_REFERENCE (one primary ray through each pixel centre + shade():
    // its camera rays, traversal and sample / resolve machinery with spp = 1 and no bounces); PT_FLAG_SIMPLE_KERNEL selects the
    // one-pixel-per-lane kernel for either.
    const bool mega_ref = p->mode == PT_MODE_REFERENCE && !(p->flags & PT_FLAG_SIMPLE_KERNEL) && !brute;
    const bool mega_path = (p->mode == PT_MODE_PATH && !(p->flags & PT_FLAG_SIMPLE_KERNEL) && !brute) || mega_ref;
    if (mega_ref) { A.ref_mode = 1u; A.spp = 1u; A.max_bounces = 0u; }
    if (stats) {
        if (!mega_path) {      // the megakernel zeroes the block itself, on the stream its trace runs on (flush_pending_stats)
            if (int rc = flush_pending(ctx)) return rc;
            ctx->stats_culled = 0;
            PT_HIP(ctx, hipMemsetAsync(ctx->d_stats.ptr, 0, 24 * sizeof(unsigned long long), ctx->stream));
        }
        A.stats = ctx->d_stats.ptr;
    }
    ctx->last_stats = stats;
    const bool ring = ctx->ring_used + 2 <= ctx->ring.size();
    hipEvent_t e0 = ring ? ctx->ring[ctx->ring_used] : ctx->ev_start, e1 = ring ? ctx->ring[ctx->ring_used + 1] : ctx->ev_stop;
    const int kmode = p->mode == PT_MODE_REFERENCE_PACKET ? PT_KMODE_PACKET : (p->mode == PT_MODE_REFERENCE ? PT_KMODE_REFERENCE : PT_KMODE_PATH);
    if (mega_path) {
        // ---- persistent megakernel: frames are queued and launched in batches of ctx->batch_size
        if (uint64_t(A.num_tiles) * A.spp * 64ull * ctx->batch_size > 0xFFFFFFFFull)      // item and sample indices are 32-bit
            return fail(ctx, PT_ERR_INVALID_ARG, "pt_render: more than 2^32 pixel-samples per launch (lower spp, the resolution or pt_set_batch)");
        ptk::FrameParams fp; std::memset(&fp, 0, sizeof(fp));
        std::memcpy(fp.cam, p->cam_pos, 12); std::memcpy(fp.quat, p->cam_quat, 16);
        fp.focal = p->focal; fp.aspect = p->aspect; fp.frame = p->frame; fp.seed = p->seed;
        fp.accum_mode = A.accum ? (A.accumulate ? 2u : 1u) : 0u;
        if (ctx->pending) {
            const ptk::RenderArgs& Q = ctx->pendingA;   // a frame joins the open batch only if it has the same shape and targets
            const bool same = Q.width == A.width && Q.height == A.height && Q.spp == A.spp && Q.max_bounces == A.max_bounces && Q.num_tris == A.num_tris &&
                              Q.tiles == A.tiles && Q.num_tiles == A.num_tiles && Q.compact == A.compact && Q.accum == A.accum && Q.ref_mode == A.ref_mode && Q.stats == A.stats &&
                              (!stats || PtTune::pick(ctx->tune.stats_batch, 0u) != 0u);
            if (!same || ctx->pending >= PT_MAX_BATCH) { if (int rc = flush_pending(ctx)) return rc; }
        }
        if (!ctx->pending) { ctx->pendingA = A; ctx->pending_ring = false; ctx->pending_rank = p->tile_rank; ctx->pending_count = count; }
        if (ctx->pending == 0) { ctx->pending_frames.clear(); ctx->pending_outs.clear(); }
        ctx->pending_frames.push_back(fp); ctx->pending_outs.push_back(A.out);
        ++ctx->pending;
        ctx->timed = false;
        if ((stats && PtTune::pick(ctx->tune.stats_batch, 0u) == 0u) || ctx->pending >= ctx->batch_size) return flush_pending_stats(ctx, stats, sharded, count);
        return PT_OK;
    } else {
        if (int rc = flush_pending(ctx)) return rc;
        PT_HIP(ctx, hipEventRecord(e0, ctx->stream));
        PT_HIP(ctx, ptk::launch_render(A, kmode, stats, ctx->stream));
        PT_HIP(ctx, hipEventRecord(e1, ctx->stream));
        {   // these kernels use no frame slot: their own fence for pt_buffer_busy
            float4* t = A.out;
            if (!ctx->misc_fence) PT_HIP(ctx, hipEventCreateWithFlags(&ctx->misc_fence, hipEventDisableTiming));
            PT_HIP(ctx, hipEventRecord(ctx->misc_fence, ctx->stream));
            ctx->misc_fence_seq = track_targets(ctx, &t, 1, (A.compact ? size_t(A.num_tiles) * 64u : npx) * sizeof(float4));
        }
    }
    if (ring) ctx->ring_used += 2;
    ctx->timed = !ring;
    return PT_OK;
}

int pt_set_batch(PtContext* ctx, uint32_t frames_per_launch) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (frames_per_launch < 1u || frames_per_launch > PT_MAX_BATCH) return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_batch: 1..256 frames per launch");
    ctx->batch_size = frames_per_launch;
    return PT_OK;
}

int pt_flush(PtContext* ctx) {
    if (int rc = bind(ctx)) return rc;
    return flush_pending(ctx);
}

int pt_timing_begin(PtContext* ctx, uint32_t capacity) {
    if (int rc = bind(ctx)) return rc;
    while (ctx->ring.size() < size_t(capacity) * 2) {
        hipEvent_t e = nullptr;
        PT_HIP(ctx, hipEventCreate(&e));
        ctx->ring.push_back(e);
    }
    while (ctx->ring.size() > size_t(capacity) * 2) { (void)hipEventDestroy(ctx->ring.back()); ctx->ring.pop_back(); }
    ctx->ring_used = 0;
    return PT_OK;
}

int pt_timing_collect_spans(PtContext* ctx, float* start_ms, float* dur_ms, uint32_t capacity, uint32_t* count) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t n = ctx->ring_used / 2;
    uint32_t got = 0;
    for (uint32_t i = 0; i < n && i < capacity; ++i, ++got) {
        if (dur_ms) PT_HIP(ctx, hipEventElapsedTime(&dur_ms[i], ctx->ring[2 * i], ctx->ring[2 * i + 1]));
        if (start_ms) { start_ms[i] = 0.0f; if (i) PT_HIP(ctx, hipEventElapsedTime(&start_ms[i], ctx->ring[0], ctx->ring[2 * i])); }
    }
    if (count) *count = got;
    for (hipEvent_t e : ctx->ring) (void)hipEventDestroy(e);
    ctx->ring.clear(); ctx->ring_used = 0;
    return PT_OK;
}

int pt_timing_collect(PtContext* ctx, float* ms, uint32_t capacity, uint32_t* count) {
    return pt_timing_collect_spans(ctx, nullptr, ms, capacity, count);
}

int pt_last_render_ms(PtContext* ctx, float* ms) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!ms) return fail(ctx, PT_ERR_INVALID_ARG, "pt_last_render_ms: null output");
    if (!ctx->timed) return fail(ctx, PT_ERR_NO_SCENE, "pt_last_render_ms: nothing rendered yet");
    PT_HIP(ctx, hipEventSynchronize(ctx->ev_stop));
    PT_HIP(ctx, hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
    return PT_OK;
}

int pt_get_stats(PtContext* ctx, PtStats* out) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!out) return fail(ctx, PT_ERR_INVALID_ARG, "pt_get_stats: null output");
    if (!ctx->last_stats) return fail(ctx, PT_ERR_NO_SCENE, "pt_get_stats: last render did not run with PT_FLAG_STATS");
    unsigned long long h[8];
    PT_HIP(ctx, hipMemcpyAsync(h, ctx->d_stats.ptr, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->rays_closest = h[0]; out->rays_shadow = h[1]; out->nodes_examined = h[2]; out->tris_tested = h[3];
    out->stack_drops = h[4]; out->max_stack = h[5]; out->samples = h[6];
    if (ctx->stats_culled) {                      // culled camera rays: generated, one root record examined, missed (renderer.wgsl:240-262)
        out->rays_closest += ctx->stats_culled; out->nodes_examined += ctx->stats_culled; out->samples += ctx->stats_culled;
        if (out->max_stack < 1u) out->max_stack = 1u;
    }
    return PT_OK;
}

/* diagnostics (include/mi355pt.h, last section): override one launch heuristic of this context
 * value 0xFFFFFFFF restores the measured default).  Launches the open batch first. */
int pt_debug_set_tune(PtContext* ctx, const char* name, uint32_t value) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    uint32_t* knob = name ? ctx->tune.find(name) : nullptr;
    if (!knob) return fail(ctx, PT_ERR_INVALID_ARG, "pt_debug_set_tune: unknown knob");
    *knob = value;
    return PT_OK;
}

/* diagnostics: the launch heuristics as a pure function (no GPU, no context: the measured defaults) -- what a launch of `frames` frames of a 1/tile_count
 * share with `traced_batches` batches of 64 pixel-samples would be given */
int pt_debug_launch_plan(uint32_t num_cus, uint32_t frames, uint32_t tile_count, uint32_t launches_in_flight, uint32_t traced_batches, uint32_t batch_size, uint32_t out[12]) {
    if (!out || frames == 0u) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_debug_launch_plan: bad arguments");
    PlanInput in; std::memset(&in, 0, sizeof(in));
    in.num_cus = num_cus; in.frames = frames; in.tile_count = tile_count ? tile_count : 1u; in.sharded = in.tile_count > 1u; in.stats = false;
    in.launches_in_flight = launches_in_flight; in.traced_batches = traced_batches; in.batch_size = batch_size ? batch_size : 1u;
    const LaunchPlan P = plan_launch(PtTune(), in);
    if (P.error) return fail(nullptr, PT_ERR_INVALID_ARG, P.error);
    const uint32_t v[12] = {P.grid, P.perm_rows, P.perm_cols, P.total_items, P.chunk_items, P.xcd_span, P.shade_threshold, P.fill_threshold, P.quad_live, P.fork_shadow,
                            uint32_t(P.slots), uint32_t(P.setup_slots)};
    std::memcpy(out, v, sizeof v);
    return PT_OK;
}

/* diagnostics: raw counter block of the last STATS launch */
int pt_debug_counters(PtContext* ctx, unsigned long long* dst24) {
    if (int rc = bind(ctx)) return rc;
    PT_HIP(ctx, hipMemcpy(dst24, ctx->d_stats.ptr, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PT_OK;
}

/* diagnostics: per-wave timeline of the last STATS megakernel launch */
int pt_debug_wave_times(PtContext* ctx, unsigned long long* dst, uint32_t max_waves, uint32_t* n_waves) {
    if (int rc = bind(ctx)) return rc;
    const uint32_t n = ctx->wave_times_n < max_waves ? ctx->wave_times_n : max_waves;
    if (n) PT_HIP(ctx, hipMemcpy(dst, ctx->d_wave_times.ptr, size_t(n) * ptk::kWaveTimeWords * 8u, hipMemcpyDeviceToHost));
    if (n_waves) *n_waves = n;
    return PT_OK;
}

int pt_read_radiance(PtContext* ctx, float* dst, uint64_t dst_floats) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!dst) return fail(ctx, PT_ERR_INVALID_ARG, "pt_read_radiance: null destination");
    const uint64_t need = uint64_t(ctx->out_w) * ctx->out_h * 4;
    if (need == 0) return fail(ctx, PT_ERR_NO_SCENE, "pt_read_radiance: no full-frame result (render with tile_count <= 1 or call pt_deinterleave)");
    if (dst_floats < need) return fail(ctx, PT_ERR_INVALID_ARG, "pt_read_radiance: destination too small");
    PT_HIP(ctx, hipMemcpyAsync(dst, ctx->last_full, need * 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_read_rgba8(PtContext* ctx, uint8_t* dst, uint64_t dst_bytes) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    const uint64_t npx = uint64_t(ctx->out_w) * ctx->out_h;
    if (npx == 0) return fail(ctx, PT_ERR_NO_SCENE, "pt_read_rgba8: no full-frame result");
    if (!dst || dst_bytes < npx * 4) return fail(ctx, PT_ERR_INVALID_ARG, "pt_read_rgba8: destination too small");
    PT_HIP(ctx, ctx->d_u32tmp.ensure(npx));
    PT_HIP(ctx, ptk::launch_rgba8(ctx->last_full, ctx->d_u32tmp.ptr, uint32_t(npx), ctx->stream));
    PT_HIP(ctx, hipMemcpyAsync(dst, ctx->d_u32tmp.ptr, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_read_tonemapped(PtContext* ctx, int from_rgba8, uint8_t* dst, uint64_t dst_bytes) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    const uint64_t npx = uint64_t(ctx->out_w) * ctx->out_h;
    if (npx == 0) return fail(ctx, PT_ERR_NO_SCENE, "pt_read_tonemapped: no full-frame result");
    if (!dst || dst_bytes < npx * 4) return fail(ctx, PT_ERR_INVALID_ARG, "pt_read_tonemapped: destination too small");
    PT_HIP(ctx, ctx->d_u32tmp.ensure(npx));
    PT_HIP(ctx, ptk::launch_tonemap(ctx->last_full, ctx->d_u32tmp.ptr, ctx->out_w, ctx->out_h, from_rgba8, ctx->stream));
    PT_HIP(ctx, hipMemcpyAsync(dst, ctx->d_u32tmp.ptr, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

// ---- checkpoint / resume of a progressive accumulation -------------------------------------------------------------------------
int pt_accum_info(PtContext* ctx, PtAccumInfo* out) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!out) return fail(ctx, PT_ERR_INVALID_ARG, "pt_accum_info: null output");
    std::memset(out, 0, sizeof(*out));
    if (ctx->accum_count == 0) return PT_OK;                       // no running accumulation: floats = 0
    out->width = ctx->accum_w; out->height = ctx->accum_h;
    out->tile_rank = ctx->accum_rank & 0xfffu; out->tile_count = (ctx->accum_rank >> 12) & 0xfffu; out->compact = ctx->accum_rank >> 31;
    out->samples = ctx->accum_count;
    out->floats = out->compact ? uint64_t(pt::tile_count_of(out->width, out->height, out->tile_rank, out->tile_count)) * 256ull : uint64_t(out->width) * out->height * 4ull;
    return PT_OK;
}

int pt_read_accum(PtContext* ctx, float* dst, uint64_t dst_floats) {
    PtAccumInfo info;
    if (int rc = pt_accum_info(ctx, &info)) return rc;
    if (info.samples == 0) return fail(ctx, PT_ERR_NO_SCENE, "pt_read_accum: no running accumulation (render with accumulate = 1 first)");
    if (info.floats == 0) return PT_OK;             // a share that owns no tile (more shares than tiles): a running accumulation of nothing
    if (!dst || dst_floats < info.floats) return fail(ctx, PT_ERR_INVALID_ARG, "pt_read_accum: destination too small (pt_accum_info gives the size)");
    const DevBuf<float4>& acc = info.compact ? ctx->d_compact_accum : ctx->d_accum;
    PT_HIP(ctx, hipMemcpyAsync(dst, acc.ptr, info.floats * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // A compact dump is tile-major (tile slot * 64 + ly * 8 + lx); where the frame is not a multiple of 8 the edge tiles carry pixels outside the image
    // that resolve_kernel never writes: zeroed here, so that two dumps of the same state are the same bytes.
    if (info.compact && ((info.width | info.height) & 7u)) {
        std::vector<uint32_t> tiles; pt::tile_list(info.width, info.height, info.tile_rank, info.tile_count ? info.tile_count : 1u, tiles);
        const uint32_t tiles_x = (info.width + 7u) / 8u;
        for (size_t slot = 0; slot < tiles.size(); ++slot) {
            const uint32_t x0 = (tiles[slot] % tiles_x) * 8u, y0 = (tiles[slot] / tiles_x) * 8u;
            if (x0 + 8u <= info.width && y0 + 8u <= info.height) continue;
            for (uint32_t p = 0; p < 64u; ++p)
                if (x0 + (p & 7u) >= info.width || y0 + (p >> 3) >= info.height) std::memset(dst + (slot * 64u + p) * 4u, 0, 4 * sizeof(float));
        }
    }
    return PT_OK;
}

int pt_set_accum(PtContext* ctx, const PtAccumInfo* info, const float* src) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!info || !src) return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_accum: null argument");
    const uint32_t count = info->tile_count ? info->tile_count : 1u;
    if (info->width == 0 || info->height == 0 || info->width > 32768 || info->height > 32768 || info->tile_rank >= count || count > 0xfffu || info->samples == 0)
        return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_accum: bad shape (resolution, tile share or a sample count of 0)");
    const bool compact = info->compact != 0 || count > 1u;
    const uint64_t need = compact ? uint64_t(pt::tile_count_of(info->width, info->height, info->tile_rank, count)) * 256ull : uint64_t(info->width) * info->height * 4ull;
    if (info->floats != need) return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_accum: `floats` does not match the shape (whole frame: W*H*4; tile share: tiles*64*4)");
    DevBuf<float4>& acc = compact ? ctx->d_compact_accum : ctx->d_accum;
    if (need == 0) {                                                // a share that owns no tile (more shares than tiles): nothing to restore, nothing to read
        ctx->accum_w = info->width; ctx->accum_h = info->height; ctx->accum_rank = accum_share_key(info->tile_rank, count, compact); ctx->accum_count = info->samples;
        return PT_OK;
    }
    // every pixel's w is the number of samples summed into it (resolve_kernel divides by it) and it is the same for all: a dump whose `samples`
    // disagrees with its own data (edited by hand, truncated) would give means and checkpoint metadata that contradict each other.  Checked on pixels that
    // are inside the image whatever the resolution: the first pixel of the first and of the last tile of a compact dump (a tile's origin is always inside;
    // its last pixel is not when width or height is not a multiple of 8), the first and last pixel of a whole frame.
    const uint64_t last_w = compact ? need - 256u + 3u : need - 1u;
    if (src[3] != float(info->samples) || src[last_w] != float(info->samples))
        return fail(ctx, PT_ERR_INVALID_ARG, "pt_set_accum: `samples` does not match the sample count stored in the data (w of the first / last pixel)");
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));                 // nothing in flight still reads or writes the running sums
    everything_delivered(ctx);
    PT_HIP(ctx, acc.ensure(size_t(need / 4u)));
    PT_HIP(ctx, hipMemcpyAsync(acc.ptr, src, need * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));                 // the host array is not retained
    ctx->accum_w = info->width; ctx->accum_h = info->height; ctx->accum_rank = accum_share_key(info->tile_rank, count, compact); ctx->accum_count = info->samples;
    return PT_OK;
}

int pt_set_compact_buffer(PtContext* ctx, void* device_ptr, uint64_t floats) {
    if (int rc = bind(ctx)) return rc;
    if (!device_ptr) {       // "no longer yours to keep": everything queued or in flight for the caller's buffers has been delivered when this returns
        if (int rc = flush_pending(ctx)) return rc;
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->ext_compact = (float4*)device_ptr;
    ctx->ext_compact_floats = device_ptr ? floats : 0;
    return PT_OK;
}

int pt_set_output_buffer(PtContext* ctx, void* device_ptr, uint64_t floats) {
    if (int rc = bind(ctx)) return rc;
    if (!device_ptr) {
        if (int rc = flush_pending(ctx)) return rc;
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->ext_out = (float4*)device_ptr;
    ctx->ext_out_floats = device_ptr ? floats : 0;
    return PT_OK;
}

int pt_compact_radiance(PtContext* ctx, void** device_ptr, uint64_t* floats) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (ctx->compact_floats == 0 || !(ctx->ext_compact || ctx->d_compact.ptr)) return fail(ctx, PT_ERR_NO_SCENE, "pt_compact_radiance: no tile-sharded render yet");
    if (device_ptr) *device_ptr = ctx->ext_compact ? (void*)ctx->ext_compact : (void*)ctx->d_compact.ptr;
    if (floats) *floats = ctx->compact_floats;
    return PT_OK;
}

// largest share of a W x H frame split `count` ways, in tiles: cached per context (the group / bench call this once per batch)
static uint32_t max_share_tiles(PtContext* ctx, uint32_t width, uint32_t height, uint32_t count) {
    if (ctx->share_w != width || ctx->share_h != height || ctx->share_count != count) {
        uint32_t m = 0;
        for (uint32_t r = 0; r < count; ++r) m = std::max(m, pt::tile_count_of(width, height, r, count));
        ctx->share_w = width; ctx->share_h = height; ctx->share_count = count; ctx->share_max_tiles = m;
    }
    return ctx->share_max_tiles;
}

int pt_deinterleave_batch(PtContext* ctx, const void* gathered_device, uint64_t rank_stride_floats, uint64_t frame_stride_floats, uint32_t num_frames,
                          uint32_t width, uint32_t height, uint32_t tile_count, void* frames_out_device, uint64_t out_stride_floats) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!gathered_device || tile_count == 0 || num_frames == 0 || (rank_stride_floats & 3) || (frame_stride_floats & 3) || (out_stride_floats & 3) || width == 0 || height == 0)
        return fail(ctx, PT_ERR_INVALID_ARG, "pt_deinterleave_batch: bad arguments");
    const uint64_t share = uint64_t(max_share_tiles(ctx, width, height, tile_count)) * 256ull;
    if (share > frame_stride_floats && num_frames > 1u) return fail(ctx, PT_ERR_INVALID_ARG, "pt_deinterleave_batch: frame stride smaller than a rank's compact buffer");
    if (share + uint64_t(num_frames - 1u) * frame_stride_floats > rank_stride_floats) return fail(ctx, PT_ERR_INVALID_ARG, "pt_deinterleave_batch: rank stride smaller than the frames of one rank");
    const size_t npx = size_t(width) * height;
    float4* out = (float4*)frames_out_device; uint64_t out_stride_px = out_stride_floats / 4u;
    if (out) {
        if (num_frames > 1u && out_stride_px < npx) return fail(ctx, PT_ERR_INVALID_ARG, "pt_deinterleave_batch: output stride smaller than a frame");
    } else {
        // the context's own frame buffer holds ONE frame: earlier frames of the batch are delivered and replaced (frames of one launch that
        // share a target leave the last one's result, as with pt_render); only the last one is scattered
        PT_HIP(ctx, ctx->d_out.ensure(npx));
        out = ctx->d_out.ptr; out_stride_px = 0;
        gathered_device = (const char*)gathered_device + size_t(num_frames - 1u) * frame_stride_floats * sizeof(float);
        num_frames = 1u;
    }
    PT_HIP(ctx, ptk::launch_deinterleave((const float4*)gathered_device, rank_stride_floats / 4u, frame_stride_floats / 4u, num_frames, out, out_stride_px,
                                         width, height, tile_count, ctx->stream));
    ctx->out_w = width; ctx->out_h = height; ctx->last_full = out + size_t(num_frames - 1u) * out_stride_px;
    return PT_OK;
}

// ---- packed tile shares: only the tiles a camera ray can reach the scene in travel, at 12 bytes per pixel ---------------------------
int pt_traced_tile_rect(PtContext* ctx, const PtRenderParams* p, uint32_t rect[4]) {
    if (!ctx || !p || !rect) return fail(ctx, PT_ERR_INVALID_ARG, "pt_traced_tile_rect: null argument");
    if (p->width == 0 || p->height == 0 || p->width > 32768 || p->height > 32768) return fail(ctx, PT_ERR_INVALID_ARG, "pt_traced_tile_rect: bad resolution");
    const uint32_t tiles_x = (p->width + pt::kTile - 1) / pt::kTile, tiles_y = (p->height + pt::kTile - 1) / pt::kTile;
    rect[0] = 0; rect[1] = 0; rect[2] = tiles_x; rect[3] = tiles_y;                       // nothing can be left out: the whole image
    const bool brute = (p->flags & PT_FLAG_BRUTE_FORCE) != 0;
    if (brute || !ctx->have_bvh || ctx->wide_meta.root_ref == pt::kInvalid || ctx->wide_meta.root_degenerate || p->num_tris == 0u || ctx->tune.cull == 0u) return PT_OK;
    ptk::FrameParams f; std::memset(&f, 0, sizeof(f));
    std::memcpy(f.cam, p->cam_pos, 12); std::memcpy(f.quat, p->cam_quat, 16); f.focal = p->focal; f.aspect = p->aspect;
    TileRect r;
    if (root_box_rect(ctx->wide_meta, f, p->width, p->height, r)) { rect[0] = r.tx0; rect[1] = r.ty0; rect[2] = std::max(r.tx0, r.tx1); rect[3] = std::max(r.ty0, r.ty1); }
    return PT_OK;
}

static bool rect_ok(const uint32_t rect[4], uint32_t width, uint32_t height) {
    const uint32_t tiles_x = (width + pt::kTile - 1) / pt::kTile, tiles_y = (height + pt::kTile - 1) / pt::kTile;
    return rect && rect[0] <= rect[2] && rect[1] <= rect[3] && rect[2] <= tiles_x && rect[3] <= tiles_y;
}

int pt_packed_layout(uint32_t width, uint32_t height, uint32_t tile_count, const uint32_t rect[4], uint32_t* max_tiles, uint64_t* floats_per_frame) {
    if (tile_count == 0) tile_count = 1;
    if (width == 0 || height == 0 || !rect_ok(rect, width, height)) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_packed_layout: bad rectangle");
    uint32_t m = 0;
    for (uint32_t r = 0; r < tile_count; ++r) m = std::max(m, pt::rect_tile_count_of(r, tile_count, rect));
    if (max_tiles) *max_tiles = m;
    if (floats_per_frame) *floats_per_frame = uint64_t(m) * 192ull;
    return PT_OK;
}

int pt_packed_tile_ids(uint32_t width, uint32_t height, uint32_t tile_rank, uint32_t tile_count, const uint32_t rect[4], uint32_t* ids, uint32_t capacity, uint32_t* num_tiles) {
    if (tile_count == 0) tile_count = 1;
    if (tile_rank >= tile_count || width == 0 || height == 0 || !rect_ok(rect, width, height)) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_packed_tile_ids: bad arguments");
    const uint32_t tiles_x = (width + pt::kTile - 1) / pt::kTile;
    uint32_t n = 0;
    for (uint32_t ty = rect[1]; ty < rect[3]; ++ty)
        for (uint32_t tx = rect[0]; tx < rect[2]; ++tx)
            if ((tx + ty) % tile_count == tile_rank) { if (ids) { if (n >= capacity) return fail(nullptr, PT_ERR_INVALID_ARG, "pt_packed_tile_ids: destination too small"); ids[n] = ty * tiles_x + tx; } ++n; }
    if (num_tiles) *num_tiles = n;
    return PT_OK;
}

int pt_pack_shares(PtContext* ctx, const void* compact_device, uint64_t frame_stride_floats, uint32_t num_frames, uint32_t width, uint32_t height,
                   uint32_t tile_rank, uint32_t tile_count, const uint32_t rect[4], void* packed_device, uint64_t packed_frame_stride_floats) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (tile_count == 0) tile_count = 1;
    if (!compact_device || !packed_device || num_frames == 0 || (frame_stride_floats & 3) || width == 0 || height == 0 || tile_rank >= tile_count || !rect_ok(rect, width, height))
        return fail(ctx, PT_ERR_INVALID_ARG, "pt_pack_shares: bad arguments");
    const uint64_t own = uint64_t(pt::tile_count_of(width, height, tile_rank, tile_count)) * 256ull, packed = uint64_t(pt::rect_tile_count_of(tile_rank, tile_count, rect)) * 192ull;
    if (num_frames > 1u && (frame_stride_floats < own || packed_frame_stride_floats < packed)) return fail(ctx, PT_ERR_INVALID_ARG, "pt_pack_shares: a frame stride is smaller than the share it holds");
    PT_HIP(ctx, ptk::launch_pack_shares((const float4*)compact_device, frame_stride_floats / 4u, num_frames, (float*)packed_device, packed_frame_stride_floats, width, tile_rank, tile_count, rect, ctx->stream));
    return PT_OK;
}

int pt_unpack_batch(PtContext* ctx, const void* gathered_device, uint64_t rank_stride_floats, uint64_t frame_stride_floats, uint32_t num_frames, uint32_t width, uint32_t height,
                    uint32_t tile_count, const uint32_t rect[4], uint32_t spp, void* frames_out_device, uint64_t out_stride_floats) {
    if (int rc = bind(ctx)) return rc;
    if (int rc = flush_pending(ctx)) return rc;
    if (!gathered_device || tile_count == 0 || num_frames == 0 || (out_stride_floats & 3) || width == 0 || height == 0 || spp == 0 || !rect_ok(rect, width, height))
        return fail(ctx, PT_ERR_INVALID_ARG, "pt_unpack_batch: bad arguments");
    uint32_t max_tiles = 0; uint64_t share = 0;
    if (int rc = pt_packed_layout(width, height, tile_count, rect, &max_tiles, &share)) return rc;
    if (share > frame_stride_floats && num_frames > 1u) return fail(ctx, PT_ERR_INVALID_ARG, "pt_unpack_batch: frame stride smaller than a rank's packed share");
    if (share + uint64_t(num_frames - 1u) * frame_stride_floats > rank_stride_floats && tile_count > 1u) return fail(ctx, PT_ERR_INVALID_ARG, "pt_unpack_batch: rank stride smaller than the frames of one rank");
    const size_t npx = size_t(width) * height;
    float4* out = (float4*)frames_out_device; uint64_t out_stride_px = out_stride_floats / 4u;
    if (out) {
        if (num_frames > 1u && out_stride_px < npx) return fail(ctx, PT_ERR_INVALID_ARG, "pt_unpack_batch: output stride smaller than a frame");
    } else {                        // the context's own frame buffer holds one frame: the last of the batch (as pt_deinterleave_batch)
        PT_HIP(ctx, ctx->d_out.ensure(npx));
        out = ctx->d_out.ptr; out_stride_px = 0;
        gathered_device = (const float*)gathered_device + size_t(num_frames - 1u) * frame_stride_floats;
        num_frames = 1u;
    }
    PT_HIP(ctx, ptk::launch_unpack_frames((const float*)gathered_device, rank_stride_floats, frame_stride_floats, num_frames, out, out_stride_px, width, height, tile_count, rect, spp, ctx->stream));
    ctx->out_w = width; ctx->out_h = height; ctx->last_full = out + size_t(num_frames - 1u) * out_stride_px;
    return PT_OK;
}

int pt_deinterleave(PtContext* ctx, const void* gathered_device, uint64_t stride_floats, uint32_t width, uint32_t height, uint32_t tile_count) {
    return pt_deinterleave_batch(ctx, gathered_device, stride_floats, 0, 1, width, height, tile_count, nullptr, 0);
}

int pt_buffer_busy(PtContext* ctx, const void* device_ptr, uint64_t bytes, int* busy) {
    if (int rc = bind(ctx)) return rc;
    if (!busy) return fail(ctx, PT_ERR_INVALID_ARG, "pt_buffer_busy: null output");
    *busy = 0;
    if (!device_ptr || bytes == 0) return PT_OK;
    const char* lo = (const char*)device_ptr; const char* hi = lo + bytes;
    auto overlaps = [&](const char* t_lo, const char* t_hi) { return t_lo < hi && lo < t_hi; };
    if (ctx->pending) {                                                                   // queued, not launched yet
        const ptk::RenderArgs& Q = ctx->pendingA;
        const size_t fb = (Q.compact ? size_t(Q.num_tiles) * 64u : size_t(Q.width) * Q.height) * sizeof(float4);
        for (uint32_t i = 0; i < ctx->pending && !*busy; ++i) if (overlaps((const char*)ctx->pending_outs[i], (const char*)ctx->pending_outs[i] + fb)) *busy = 1;
    }
    if (!*busy) {
        prune_inflight(ctx);                                                              // launched: busy until the launch's fence has completed
        for (const auto& f : ctx->inflight) if (overlaps(f.lo, f.hi)) { *busy = 1; break; }
    }
    return PT_OK;
}

} // extern "C"
