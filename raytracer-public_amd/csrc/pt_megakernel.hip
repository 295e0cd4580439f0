// pt_megakernel.hip -- the production form of the hot path for PT_MODE_PATH on gfx950:
// a persistent-wavefront traversal + shade megakernel.
//
//   * work item = one pixel-sample; items are claimed in chunks from a global queue
//     (one atomic per chunk and wave), so the grid is sized to the chip, not to the frame;
//   * every lane is a small state machine  IDLE -> TRAVERSE -> DONE -> (TRAVERSE | IDLE):
//     lanes whose path ended are refilled with new samples (ray regeneration), lanes whose
//     ray ended wait until enough of the wavefront is waiting (__ballot / popcount) and are
//     shaded together -- this is the wavefront-level compaction that keeps the 64 lanes busy
//     across bounces although only ~13 % of the camera rays hit anything;
//   * the traversal stack is a per-wavefront LDS short stack (kShort entries per lane, 8 B
//     each, bank-conflict-free by construction) that spills its rare deep entries to a global
//     scratch area, keeping the reference's 64-entry semantics;
//   * per-sample radiance goes to a sample buffer; resolve_kernel sums the samples of a pixel
//     in sample order (deterministic, bit-identical to the oracle) and applies accumulation.
//
// Traversal order, tie-breaking and arithmetic are exactly those of render_rays_kernel /
// DESIGN.md sections 3-6 (the oracle checks both kernels bit-for-bit).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_kernels.h"
#include "pt_device.h"

namespace ptk {

// A wavefront that has found the queue dry only finishes the paths it holds: from then on it is on the critical path of its launch
// (the longest chain of dependent steps ends it) while the wavefronts of the next launch next to it are throughput work.
#ifndef PT_TAIL_PRIORITY
#define PT_TAIL_PRIORITY 3
#endif
#define PT_TAIL_PRIO do { if (PT_TAIL_PRIORITY) __builtin_amdgcn_s_setprio(PT_TAIL_PRIORITY); } while (0)
constexpr int kShort = PT_SHORT_STACK;    // LDS entries per lane
constexpr uint32_t kPhaseIdle = 0, kPhaseTrav = 1, kPhaseDone = 2, kPhaseParked = 3;   // parked: waiting at a closest-ray boundary to be donated

struct Lane {
    // ray
    F3 o, d, inv;
    RaySel sel;           // near / far picks of the slab test, from the signs of inv (pt_device.h::slab_sel)
    float best_t; uint32_t best_tri;
    uint32_t cur; int sp;
    // path
    F3 T, rad;
    F3 d_next, contrib;
    uint32_t key, item;
    uint32_t bounce;      // bits 0..15 bounce index, bit 16 = current ray is a shadow ray, bit 17 = path continues after the shadow ray
};

constexpr uint32_t kShadowBit = 1u << 16, kContBit = 1u << 17;

// x + (this lane's bit of a wavefront mask): one v_addc_co_u32 with the mask as the carry input
// (s_nop 1: two wait states between a vector compare that wrote the mask and its use as a carry input, whatever was scheduled between)
__device__ __forceinline__ void add_lane_bits3(int x, unsigned long long m0, unsigned long long m1, unsigned long long m2, int& x1, int& x2, int& x3) {
    asm("s_nop 1\n\tv_addc_co_u32_e64 %0, vcc, 0, %3, %4\n\tv_addc_co_u32_e64 %1, vcc, 0, %0, %5\n\tv_addc_co_u32_e64 %2, vcc, 0, %1, %6"
        : "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(x), "s"(m0), "s"(m1), "s"(m2) : "vcc");
}
// v_min_f32 / v_min3_f32 on values that are never NaN (no canonicalisation of the inputs)
__device__ __forceinline__ float min_f32(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float min3_f32(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// CONT = false: pass 0, items are pixel-samples generated from the permuted batch queue.
// CONT = true : continuation pass, items are path records flushed by the previous pass.
template <bool STATS, bool CONT>
__global__ __launch_bounds__(PT_MEGA_BLOCK, PT_MEGA_WAVES_PER_SIMD) void trace_paths_kernel(const RenderArgs A) {
    __shared__ unsigned long long lds_stack[PT_MEGA_BLOCK / 64][kShort][64];

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    unsigned long long* const stk = &lds_stack[wave][0][lane];          // entry i at stk[i * 64]: (tmin bits << 32) | ref
    const uint32_t stk_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned long long*)stk;    // the same as an LDS byte address
    unsigned long long* const spill = (unsigned long long*)A.spill + ((size_t)blockIdx.x * PT_MEGA_BLOCK + threadIdx.x);   // entry j at spill[j * spill_stride]
    const size_t spill_stride = (size_t)gridDim.x * PT_MEGA_BLOCK;

    const F3 base = f3(0.9f, 0.7f, 0.3f);
    const F3 L = light_dir();
    const F3 invL = safe_inv(L);
    const bool scene_empty = (A.root_ref == kInvalidRef) || (A.num_tris == 0u);

    unsigned long long t_begin = 0, t_qempty = 0;
    uint32_t n_iter = 0, n_shade = 0, n_fill = 0, n_iter_q = 0; unsigned long long cy_shade = 0, cy_fill = 0, cy_step = 0, cy_step_q = 0, cy_mark = 0; unsigned long long lanes_sum = 0, lanes_sum_q = 0, leaf_lanes = 0; uint32_t spill_ops = 0, push_ops = 0, push8_ops = 0, push12_ops = 0;
    if (STATS) t_begin = wall_clock64();
    // CONT passes consume what the previous pass left in its path pool: records [head, min(tail, capacity))
    uint32_t in_base = 0, in_count = 0;
    if (CONT) {
        const uint32_t tail = min(A.in_ctrl[0], A.pool_capacity), head = A.in_ctrl[1];
        in_base = head; in_count = tail > head ? tail - head : 0u;
    }
    const uint32_t total_items = CONT ? in_count : A.total_items;
    const bool pool_on = A.flush_threshold != 0u;
    // per-frame parameters: a small device array (per-lane index: lanes of one refill may straddle two frames)
    const FrameParams* const frames = A.frames;
    const uint32_t chunk_items = CONT ? 64u : A.chunk_items;
    // the XCD this wavefront runs on (HW_REG_XCC_ID, bits 3:0) and how many other XCDs' ranges it has moved on to
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
    uint32_t xcd_hop = 0;
    uint32_t chunk_next = 0, chunk_end = 0;   // wave-uniform: this wave's private item range
    bool queue_empty = false;                 // wave-uniform

    Lane S;
    uint32_t phase = kPhaseIdle;
    S.sp = 0; S.cur = 0; S.best_t = kInfT; S.best_tri = kInvalidRef; S.bounce = 0; S.key = 0; S.item = 0;
    S.o = S.d = S.inv = S.T = S.rad = S.d_next = S.contrib = f3(0, 0, 0); S.sel = ray_selectors(S.inv);
    uint32_t c_nodes = 0, c_tris = 0, c_drops = 0, c_maxstack = 0, c_closest = 0, c_shadow = 0, c_samples = 0;

    // ---- helpers as lambdas (all inlined) -------------------------------------------------
    auto push = [&](uint32_t ref, float tmin) {
        if (S.sp < kStackMax) {
            const unsigned long long e = ((unsigned long long)__float_as_uint(tmin) << 32) | ref;
            if (__builtin_expect(S.sp < kShort, 1)) stk[S.sp * 64] = e; else spill[(size_t)(S.sp - kShort) * spill_stride] = e;
            if (STATS) { if (S.sp >= kShort) ++spill_ops; if (S.sp >= 8) ++push8_ops; if (S.sp >= 12) ++push12_ops; ++push_ops; }
            ++S.sp;
        } else if (STATS) ++c_drops;
    };
    // start a ray from S.o / S.d / S.inv: root box test; returns false when it misses outright
    auto begin_ray = [&]() -> bool {
        S.best_t = kInfT; S.best_tri = kInvalidRef; S.sp = 0;
        if (scene_empty) return false;
        if (STATS) { c_nodes += 1; if (c_maxstack < 1u) c_maxstack = 1u; }   // the root record is fetched before its degenerate check (renderer.wgsl:240-244)
        if (A.root_degenerate != 0u) return false;
        Ray r; r.o = S.o; r.d = S.d; r.inv = S.inv;
        float troot;
        if (!slab(r, A.root_box[0], A.root_box[1], A.root_box[2], kInfT, troot)) return false;
        S.cur = A.root_ref;
        S.sel = ray_selectors(S.inv);
        return true;
    };

    // the wavefront's ray buffer (camera rays generated 64 at a time, see the refill pass): 64 records of 3 x uint4
    uint4* const rb = CONT ? nullptr : A.raybuf + ((size_t)blockIdx.x * (PT_MEGA_BLOCK / 64) + wave) * (64 * 3);
    uint32_t rb_next = 0, rb_count = 0;       // wave-uniform
    // nothing left to start: the queue is dry and (pass 0) so is the ray buffer
    auto source_dry = [&]() -> bool { return queue_empty && (CONT || rb_next == rb_count); };
    // claim the next chunk of the queue (one atomic per wavefront and chunk); sets queue_empty when there is none
    auto claim_chunk = [&]() {
        if (!CONT && A.xcd_span != 0u) {
            // XCD-aware queue: the logical item space is cut into 8 contiguous ranges, one cursor each; a wavefront
            // works through the range of the XCD it runs on (its neighbours in the queue order are then traced by
            // wavefronts that share its L2) and moves on to the next XCD's range when its own has run dry
            for (;;) {
                const uint32_t qi = (xcc + xcd_hop) & 7u;
                const uint32_t q_begin = qi * A.xcd_span;
                const uint32_t q_end = min(q_begin + A.xcd_span, total_items);
                uint32_t start = 0;
                if (lane == 0) start = atomicAdd(A.queue + 8 + qi, chunk_items);
                start = __builtin_amdgcn_readfirstlane(start);
                if (q_begin < q_end && start < q_end - q_begin) { chunk_next = q_begin + start; chunk_end = min(chunk_next + chunk_items, q_end); break; }
                if (++xcd_hop == 8u) { queue_empty = true; PT_TAIL_PRIO; if (STATS) { t_qempty = wall_clock64(); n_iter_q = n_iter; lanes_sum_q = lanes_sum; } break; }
            }
        } else {
            uint32_t start = 0;
            if (lane == 0) start = atomicAdd(A.queue, chunk_items);
            start = __builtin_amdgcn_readfirstlane(start);
            if (start >= total_items) { queue_empty = true; PT_TAIL_PRIO; if (STATS) { t_qempty = wall_clock64(); n_iter_q = n_iter; lanes_sum_q = lanes_sum; } }
            else { chunk_next = start; chunk_end = min(start + chunk_items, total_items); }
        }
    };
    // Lanes whose ray has ended wait for a shade pass until shade_threshold of them are there: the pass costs a wavefront as much
    // as 1.5 traversal steps whatever the number of lanes in it.  Once the queue is dry a wavefront thins out, and a lane that
    // waited for a crowd would wait for the other lanes' whole rays -- the paths of a wavefront would run one after the other,
    // and it is the longest chain of dependent steps that ends a launch.  So then a quarter of the live lanes is enough.
    auto shade_due = [&](unsigned long long m_live) -> uint32_t {
        if (!source_dry()) return A.shade_threshold;
        const uint32_t q = (uint32_t)__popcll(m_live) >> 2;
        return q < 1u ? 1u : (q < A.shade_threshold ? q : A.shade_threshold);
    };
    for (;;) {
        // ------------------------------------------------------------------ shade DONE lanes
        {
            const unsigned long long m_done = __ballot(phase == kPhaseDone);
            const unsigned long long m_trav = __ballot(phase == kPhaseTrav);
            if (m_done != 0ull && ((uint32_t)__popcll(m_done) >= shade_due(m_trav | m_done) || m_trav == 0ull)) {
                if (STATS) { ++n_shade; cy_mark = __builtin_amdgcn_s_memtime(); }
                // sparse wavefront and nothing left to regenerate from: hand the surviving paths to the
                // next pass (at a closest-ray boundary) instead of finishing them at low lane utilisation
                const bool flush_now = pool_on && source_dry() && (uint32_t)__popcll(__ballot(phase != kPhaseIdle)) < A.flush_threshold;
                bool do_flush = false;
                if (phase == kPhaseDone) {
                    const bool hit = S.best_tri != kInvalidRef;
                    const uint32_t bounce = S.bounce & 0xffffu;
                    bool finish = false, launch = false, next_bounce = false;      // next_bounce: the path goes on with its sampled direction (one site for the three reciprocals)
                    if (S.bounce & kShadowBit) {
                        if (!hit) S.rad = S.rad + S.contrib;
                        if (S.bounce & kContBit) next_bounce = true;
                        else finish = true;
                    } else if (!hit) {
                        S.rad = S.rad + S.T * ((bounce == 0u) ? kBgPrimary : kSkyAmbient);
                        finish = true;
                    } else {
                        const F3 n = tri_normal_ref(A, S.best_tri);
                        const F3 hp = S.o + S.d * S.best_t;
                        const F3 nf = (dot3(n, S.d) < 0.0f) ? n : f3(-n.x, -n.y, -n.z);
                        const F3 so = hp + nf * kEpsOrigin;
                        const float ndl = dot3(nf, L);
                        S.contrib = (S.T * base) * ndl;
                        bool cont = bounce < A.max_bounces;
                        if (cont) {
                            F3 Tn = S.T * base;
                            if (bounce >= kRRStart) {
                                const float p = wmax(wmax(Tn.x, Tn.y), Tn.z);
                                if (rnd(S.key, bounce, 4) >= p) cont = false;
                                else Tn = Tn * (1.0f / p);
                            }
                            if (cont) { S.T = Tn; S.d_next = cosine_dir(nf, rnd(S.key, bounce, 2), rnd(S.key, bounce, 3)); }
                        }
                        S.o = so;
                        if (ndl > 0.0f) {
                            S.d = L; S.inv = invL;
                            S.bounce = bounce | kShadowBit | (cont ? kContBit : 0u);
                            launch = true;
                        } else if (cont) next_bounce = true;
                        else finish = true;
                    }
                    if (next_bounce) { S.d = S.d_next; S.inv = safe_inv(S.d); S.bounce = bounce + 1u; launch = true; }
                    if (launch && flush_now && !(S.bounce & kShadowBit)) { launch = false; do_flush = true; }
                    if (launch) {
                        if (STATS) { if (S.bounce & kShadowBit) ++c_shadow; else ++c_closest; }
                        phase = begin_ray() ? kPhaseTrav : kPhaseDone;   // a root miss is shaded on the next pass
                    }
                    if (finish) {
                        A.samples[S.item] = make_float4(S.rad.x, S.rad.y, S.rad.z, 1.0f);
                        phase = kPhaseIdle;
                    }
                }
                if (do_flush) phase = kPhaseParked;            // wait for the rest of the wavefront to reach a ray boundary
                if (STATS) cy_shade += __builtin_amdgcn_s_memtime() - cy_mark;
            }
        }
        // ------------------------------------------------------------------ donate parked paths (once per wavefront)
        {
            const unsigned long long m_park = __ballot(phase == kPhaseParked);
            if (m_park != 0ull && __ballot(phase == kPhaseTrav || phase == kPhaseDone) == 0ull) {
                // every live lane is parked: publish them in one event (records, release fence, flags) and leave
                const bool parked = phase == kPhaseParked;
                uint32_t base_idx = 0;
                if (lane == 0) base_idx = atomicAdd(&A.pool_ctrl[0], (uint32_t)__popcll(m_park));   // reserve
                base_idx = __builtin_amdgcn_readfirstlane(base_idx);
                const uint32_t idx = base_idx + (uint32_t)__popcll(m_park & ((1ull << lane) - 1ull));
                const bool stored = parked && idx < A.pool_capacity;
                if (stored) {
                    // write-through (sc1) stores: the record reaches memory without an L2 write-back fence
                    uint32_t* rec = (uint32_t*)(A.pool + (size_t)idx * 4);
                    const uint32_t w[15] = {__float_as_uint(S.o.x), __float_as_uint(S.o.y), __float_as_uint(S.o.z), __float_as_uint(S.d.x),
                                            __float_as_uint(S.d.y), __float_as_uint(S.d.z), __float_as_uint(S.T.x), __float_as_uint(S.T.y),
                                            __float_as_uint(S.T.z), __float_as_uint(S.rad.x), __float_as_uint(S.rad.y), __float_as_uint(S.rad.z),
                                            S.key, S.item, S.bounce};
#pragma unroll
                    for (int k = 0; k < 15; ++k) __hip_atomic_store(rec + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    phase = kPhaseIdle;
                } else if (parked) {                             // pool full: the path simply continues here
                    if (STATS) ++c_closest;
                    phase = begin_ray() ? kPhaseTrav : kPhaseDone;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every record store has left before its flag is set
                if (stored) __hip_atomic_store(&A.pool_flags[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ------------------------------------------------------------------ refill IDLE lanes
        {
            const unsigned long long m_idle = __ballot(phase == kPhaseIdle);
            const unsigned long long m_trav = __ballot(phase == kPhaseTrav);
            if (m_idle != 0ull && !source_dry() && ((uint32_t)__popcll(m_idle) >= A.fill_threshold || m_trav == 0ull)) {
                const uint32_t want = (uint32_t)__popcll(m_idle);
                const uint32_t rank = (uint32_t)__popcll(m_idle & ((1ull << lane) - 1ull));
                if (STATS) { ++n_fill; cy_mark = __builtin_amdgcn_s_memtime(); }
                if (CONT) {
                    if (chunk_next == chunk_end) claim_chunk();
                    if (!queue_empty) {
                        const uint32_t avail = chunk_end - chunk_next;
                        if (phase == kPhaseIdle && rank < avail) {
                            const float4* rec = A.in_pool + (size_t)(in_base + chunk_next + rank) * 4;
                            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
                            S.o = f3(r0.x, r0.y, r0.z); S.d = f3(r0.w, r1.x, r1.y); S.inv = safe_inv(S.d);
                            S.T = f3(r1.z, r1.w, r2.x); S.rad = f3(r2.y, r2.z, r2.w);
                            S.key = __float_as_uint(r3.x); S.item = __float_as_uint(r3.y); S.bounce = __float_as_uint(r3.z);
                            if (STATS) ++c_closest;
                            phase = begin_ray() ? kPhaseTrav : kPhaseDone;
                        }
                        chunk_next += min(want, avail);
                    }
                } else {
                    // Camera rays are GENERATED by the whole wavefront, one (frame, tile, sample) batch of 64 at a time -- item decode,
                    // RNG key, jittered ray, its rotation, the three correctly rounded reciprocals and the root-box test are ~300
                    // instructions whether 8 lanes need a ray or 64, so all 64 lanes compute (whatever their own ray is doing: only
                    // temporaries are touched), and the rays that enter the root box go, compacted, into the wavefront's ray buffer.
                    // Idle lanes then only FETCH a ready ray (three dwordx4).  A camera ray that misses the root box is never stored:
                    // its sample keeps the primed miss value (0 + 1 * 0.01, renderer.wgsl:410).
                    while (rb_next == rb_count && !queue_empty) {
                        if (chunk_next == chunk_end) claim_chunk();
                        if (queue_empty) break;
                        const uint32_t first = chunk_next; chunk_next += 64u;       // chunks are multiples of 64 items
                        const ItemInfo it = decode_item(A, first + lane);
                        bool enters = false; Ray r; uint32_t key = 0u;
                        r.o = r.d = r.inv = f3(0.0f, 0.0f, 0.0f);
                        if (it.valid) {
                            // every sample of the batch starts as the camera-ray miss value (0 + 1 * 0.01, renderer.wgsl:410); a path that
                            // hits something overwrites its own later (same wavefront, same address: in order).  Priming here, 64 lanes
                            // at a time, is what keeps the resolve pass read-only on the sample buffer.
                            A.samples[it.sample_index] = make_float4(0.0f + 1.0f * kBgPrimary, 0.0f + 1.0f * kBgPrimary, 0.0f + 1.0f * kBgPrimary, 1.0f);
                            const FrameParams fp = frames[it.fid];
                            key = sample_key(fp.seed, it.py * A.width + it.px, fp.frame * A.spp + it.s);
                            r = primary_ray_fp(A, fp, (float)it.px + rnd(key, 0, 0), (float)it.py + rnd(key, 0, 1));
                            if (STATS) { ++c_closest; ++c_samples; }
                            if (!scene_empty) {
                                if (STATS) { c_nodes += 1; if (c_maxstack < 1u) c_maxstack = 1u; }   // the root record is fetched before its degenerate check (renderer.wgsl:240-244)
                                float troot;
                                if (A.root_degenerate == 0u) enters = slab(r, A.root_box[0], A.root_box[1], A.root_box[2], kInfT, troot);
                            }
                        }
                        const unsigned long long m_in = __ballot(enters);
                        if (enters) {
                            uint4* rec = rb + (size_t)__popcll(m_in & ((1ull << lane) - 1ull)) * 3;
                            rec[0] = make_uint4(__float_as_uint(r.o.x), __float_as_uint(r.o.y), __float_as_uint(r.o.z), __float_as_uint(r.d.x));
                            rec[1] = make_uint4(__float_as_uint(r.d.y), __float_as_uint(r.d.z), __float_as_uint(r.inv.x), __float_as_uint(r.inv.y));
                            rec[2] = make_uint4(__float_as_uint(r.inv.z), key, it.sample_index, 0u);
                        }
                        rb_count = (uint32_t)__popcll(m_in); rb_next = 0u;
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the records are written before any lane of this wavefront fetches one
                    }
                    const uint32_t avail = rb_count - rb_next;
                    if (phase == kPhaseIdle && rank < avail) {
                        const uint4* rec = rb + (size_t)(rb_next + rank) * 3;
                        const uint4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                        S.o = f3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z));
                        S.d = f3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
                        S.inv = f3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x));
                        S.key = r2.y; S.item = r2.z; S.bounce = 0u;
                        S.T = f3(1.0f, 1.0f, 1.0f); S.rad = f3(0.0f, 0.0f, 0.0f);
                        S.best_t = kInfT; S.best_tri = kInvalidRef; S.sp = 0; S.cur = A.root_ref; S.sel = ray_selectors(S.inv);
                        phase = kPhaseTrav;
                    }
                    rb_next += min(want, avail);
                }
                if (STATS) cy_fill += __builtin_amdgcn_s_memtime() - cy_mark;
            }
        }
        // ------------------------------------------------------------------ exit / idle-spin
        const unsigned long long m_trav = __ballot(phase == kPhaseTrav);
        if (m_trav == 0ull) {
            const unsigned long long m_done = __ballot(phase == kPhaseDone);
            if (m_done == 0ull && source_dry()) break;    // donated paths are picked up by the continuation pass
            continue;
        }
        // ------------------------------------------------------------------ traversal steps, until one of the passes above is due again
        for (;;) {
        if (STATS) { cy_mark = __builtin_amdgcn_s_memtime(); ++n_iter; lanes_sum += __popcll(__ballot(phase == kPhaseTrav)); leaf_lanes += __popcll(__ballot(phase == kPhaseTrav && (S.cur & kLeaf))); }
        if (phase == kPhaseTrav) {
            bool need_pop = false;
            // Unified fetch: a lane is either at an internal node (64 B record) or at a leaf (48 B triangle record).  Both kinds
            // live in one arena and a reference is the record's position in it (packed references, pt_host.h), so the same four
            // dwordx4 loads at `cur << 4` fetch either kind BEFORE the node/leaf branch: a wavefront with mixed lanes pays one
            // memory latency per step, not two.  (A leaf lane over-reads 16 B into the next record; the arena is padded.)
            const bool at_leaf = (S.cur & kLeaf) != 0u;
            const uint4* rec = arena_record(A, S.cur);
            uint4 n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
            // keep the four loads here: without this the compiler sinks them into the two branches again
            asm volatile("" : "+v"(n0.x), "+v"(n1.x), "+v"(n2.x), "+v"(n3.x));
            if (at_leaf) {
                const float4 a = make_float4(__uint_as_float(n0.x), __uint_as_float(n0.y), __uint_as_float(n0.z), __uint_as_float(n0.w));
                const float4 b = make_float4(__uint_as_float(n1.x), __uint_as_float(n1.y), __uint_as_float(n1.z), __uint_as_float(n1.w));
                const float4 c = make_float4(__uint_as_float(n2.x), __uint_as_float(n2.y), __uint_as_float(n2.z), __uint_as_float(n2.w));
                // a leaf whose triangle index is out of range points at the all-zero record behind the last triangle: the test
                // below fails on |det| < eps, as if it had not been made (renderer.wgsl:262); only the counter has to know
                if (STATS) { if ((S.cur & 0x7fffffffu) < 3u * A.num_tris) ++c_tris; }      // A.num_tris = the UBO's numTris <= uploaded
                const F3 v0 = f3(a.x, a.y, a.z), e1 = f3(a.w, b.x, b.y), e2 = f3(b.z, b.w, c.x);
                // branch-free Moller-Trumbore (renderer.wgsl:185-205): same operations and comparisons,
                // rejections combined at the end, so the 48 B record is fetched in one go
                const F3 pv = cross3(S.d, e2);
                const float det = dot3(e1, pv);
                const bool ok_det = !(fabsf(det) < kTriEps);
                const float inv_det = 1.0f / det;
                const F3 sv = S.o - v0;
                const float u = inv_det * dot3(sv, pv);
                const bool ok_u = !((u < 0.0f) | (u > 1.0f));
                const F3 q = cross3(sv, e1);
                const float v = inv_det * dot3(S.d, q);
                const bool ok_v = !((v < 0.0f) | ((u + v) > 1.0f));
                const float t = inv_det * dot3(e2, q);
                bool ok = ok_det & ok_u & ok_v & (t > kTriEps) & (t < S.best_t);
                // the UBO's numTris may be smaller than the uploaded triangle count: `ti < numTris` (renderer.wgsl:267) then hides the
                // triangles past it (a wavefront-uniform branch: the usual frame has numTris == uploaded and never takes it)
                if (A.tri_gate3 != 0xFFFFFFFFu) ok &= (S.cur & 0x7fffffffu) < A.tri_gate3;
                if (ok) {
                    S.best_t = t; S.best_tri = S.cur;
                    if (S.bounce & kShadowBit) phase = kPhaseDone;   // any-hit: first accepted hit ends the ray
                }
                need_pop = true;
            } else {
                float t0, t1, t2, t3;
                // all four slab tests are evaluated unconditionally: one 64 B fetch, no per-child branches or dependent
                // waits.  Empty slots and degenerate children (kDegenerateRef: counted below, never entered) hold the inverted
                // box, which slab_sel rejects by itself
                const unsigned long long H0 = slab_sel(S.o, S.inv, S.sel, n0.x, n0.y, n0.z, S.best_t, t0);
                const unsigned long long H1 = slab_sel(S.o, S.inv, S.sel, n0.w, n1.x, n1.y, S.best_t, t1);
                const unsigned long long H2 = slab_sel(S.o, S.inv, S.sel, n1.z, n1.w, n2.x, S.best_t, t2);
                const unsigned long long H3 = slab_sel(S.o, S.inv, S.sel, n2.y, n2.z, n2.w, S.best_t, t3);
                if (STATS) c_nodes += (n3.x != kInvalidRef) + (n3.y != kInvalidRef) + (n3.z != kInvalidRef) + (n3.w != kInvalidRef);
                // Branch-free form of renderer.wgsl:314-342 for one lane.  Hit children keep slot order;
                // the nearest (first minimum of tmin) is entered next and trades places with the first
                // hit, so the stacked entry for slot s is the first hit's when s is the nearest slot, the
                // slot's own child otherwise -- and a slot is stacked iff it is hit and is not the first hit.
                // All predicates are wavefront masks: their logic runs on the scalar unit, the vector unit only selects.
                const float kBig = 3.0e38f;
                const float e0 = lane_of(H0) ? t0 : kBig, e1 = lane_of(H1) ? t1 : kBig, e2 = lane_of(H2) ? t2 : kBig, e3 = lane_of(H3) ? t3 : kBig;
                const float tn = min_f32(min3_f32(e0, e1, e2), e3);
                // nearest slot = first slot (in slot order) whose tmin equals the minimum (with any hit, tn < kBig, so a slot that
                // equals it is a hit slot)
                const unsigned long long Q0 = __builtin_amdgcn_ballot_w64(e0 == tn), Q1 = __builtin_amdgcn_ballot_w64(e1 == tn), Q2 = __builtin_amdgcn_ballot_w64(e2 == tn);
                const unsigned long long M1 = Q1 & ~Q0, M2 = Q2 & ~(Q0 | Q1), M3 = ~(Q0 | Q1 | Q2);
                uint32_t rn = n3.x;                                   // three independent selects (a nested ?: becomes branches)
                rn = lane_of(M1) ? n3.y : rn; rn = lane_of(M2) ? n3.z : rn; rn = lane_of(M3) ? n3.w : rn;
                if (lane_of(H0 | H1 | H2 | H3)) {
                    // slot s (1..3) is stacked iff it is hit and an earlier slot is hit too (so it is not the first hit); the first hit
                    // among the slots before s is then what the nearest slot hands over
                    const unsigned long long P3 = H3 & (H0 | H1 | H2), P2 = H2 & (H0 | H1), P1 = H1 & H0;
                    const float tf01 = lane_of(H0) ? t0 : t1; const uint32_t rf01 = lane_of(H0) ? n3.x : n3.y;                     // first hit of slots 0..1 (when there is one)
                    const float tf012 = lane_of(H0 | H1) ? tf01 : t2; const uint32_t rf012 = lane_of(H0 | H1) ? rf01 : n3.z;     // first hit of slots 0..2
                    const unsigned long long w3 = ((unsigned long long)__float_as_uint(lane_of(M3) ? tf012 : t3) << 32) | (lane_of(M3) ? rf012 : n3.w);
                    const unsigned long long w2 = ((unsigned long long)__float_as_uint(lane_of(M2) ? tf01 : t2) << 32) | (lane_of(M2) ? rf01 : n3.z);
                    const unsigned long long w1 = ((unsigned long long)__float_as_uint(lane_of(M1) ? t0 : t1) << 32) | (lane_of(M1) ? n3.x : n3.y);
                    if (__builtin_expect(S.sp + 3 <= kShort, 1)) {
                        // fast path: three unconditional LDS stores far -> near; an entry that is not
                        // stacked is simply overwritten by the next one (the slot index does not advance)
                        int sp1, sp2, sp;
                        add_lane_bits3(S.sp, P3, P2, P1, sp1, sp2, sp);
                        stk[S.sp * 64] = w3; stk[sp1 * 64] = w2; stk[sp2 * 64] = w1;
                        if (STATS) { const int k = sp - S.sp; push_ops += k; for (int j = S.sp; j < sp; ++j) { if (j >= 8) ++push8_ops; if (j >= 12) ++push12_ops; } }
                        S.sp = sp;
                        if (STATS) { const uint32_t depth = (uint32_t)S.sp + 1u; if (depth > c_maxstack) c_maxstack = depth; }   // entries incl. the nearest child
                        S.cur = rn;
                    } else {
                        if (lane_of(P3)) push((uint32_t)w3, __uint_as_float((uint32_t)(w3 >> 32)));
                        if (lane_of(P2)) push((uint32_t)w2, __uint_as_float((uint32_t)(w2 >> 32)));
                        if (lane_of(P1)) push((uint32_t)w1, __uint_as_float((uint32_t)(w1 >> 32)));
                        if (STATS) { const uint32_t depth = (uint32_t)S.sp + (S.sp < kStackMax ? 1u : 0u); if (depth > c_maxstack) c_maxstack = depth; }   // entries incl. the nearest child, if its push fitted
                        if (S.sp < kStackMax) S.cur = rn;
                        else { need_pop = true; if (STATS) ++c_drops; }
                    }
                } else need_pop = true;
            }
            if (need_pop && phase == kPhaseTrav) {
                bool found = false;
                // deep entries (index >= kShort) live in the spill area: rare
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(S.sp > kShort) != 0ull, 0)) {     // wavefront-uniform: usually nobody is that deep
                    while (S.sp > kShort) {
                        --S.sp;
                        const unsigned long long e = *(volatile unsigned long long*)&spill[(size_t)(S.sp - kShort) * spill_stride];
                        if (__uint_as_float((uint32_t)(e >> 32)) < S.best_t) { S.cur = (uint32_t)e; found = true; break; }
                    }
                }
                if (!found && S.sp > 0) {
                    // Entries whose box the ray no longer reaches (tmin >= best) are skipped -- after a hit that is most of what a lane has
                    // stacked, and the wavefront loops as long as its slowest lane.  The loop is hand-written: per entry one address
                    // step, the read, two compares and three scalar instructions (lanes drop out of EXEC as they find their entry or run
                    // out of entries); the compiler's form of the same loop costs 5 vector + 10 scalar instructions per entry.
                    const uint32_t base = stk_lds, top = base + (uint32_t)S.sp * 512u;     // byte addresses: entry 0, one past the top entry
                    uint32_t addr = top, lo, hi; unsigned long long saved_exec;
                    asm volatile("s_mov_b64 %[save], exec\n"
                                 "1:\n\t"
                                 "v_add_u32 %[addr], 0xfffffe00, %[addr]\n\t"
                                 "ds_read_b32 %[hi], %[addr] offset:4\n\t"
                                 "ds_read_b32 %[lo], %[addr]\n\t"
                                 "s_waitcnt lgkmcnt(0)\n\t"
                                 "v_cmp_ngt_f32 vcc, %[best], %[hi]\n\t"         // not this one: go on
                                 "s_and_b64 exec, exec, vcc\n\t"
                                 "v_cmp_gt_u32 vcc, %[addr], %[base]\n\t"        // entries left below it
                                 "s_and_b64 exec, exec, vcc\n\t"
                                 "s_cbranch_execnz 1b\n\t"
                                 "s_mov_b64 exec, %[save]"
                                 : [addr] "+v"(addr), [lo] "=&v"(lo), [hi] "=&v"(hi), [save] "=&s"(saved_exec)
                                 : [best] "v"(S.best_t), [base] "v"(base)
                                 : "vcc", "memory");
                    found = __uint_as_float(hi) < S.best_t;          // the entry the lane stopped at: its own, or entry 0 when it ran out
                    S.sp = (int)((addr - base) >> 9);
                    if (found) S.cur = lo;
                }
                // a camera ray that found nothing leaves its sample at the primed miss value (0 + 1 * 0.01, renderer.wgsl:410): no shade pass, the lane is free
                if (!found) phase = (S.bounce == 0u && S.best_tri == kInvalidRef) ? kPhaseIdle : kPhaseDone;
            }
        }
        if (STATS) { const unsigned long long dt = __builtin_amdgcn_s_memtime() - cy_mark; cy_step += dt; if (!queue_empty) cy_step_q += dt; }
        // the conditions of the shade / donate / refill passes, exactly as they are tested at the top of the outer loop
        const unsigned long long now_trav = __ballot(phase == kPhaseTrav);
        if (now_trav == 0ull) break;
        if ((uint32_t)__popcll(__ballot(phase == kPhaseDone)) >= shade_due(now_trav | __ballot(phase == kPhaseDone))) break;
        if (!source_dry() && (uint32_t)__popcll(__ballot(phase == kPhaseIdle)) >= A.fill_threshold) break;
        }
    }
    if (STATS) {
        if (!CONT && A.wave_times && lane == 0) {
            unsigned long long* w = A.wave_times + ((size_t)blockIdx.x * (PT_MEGA_BLOCK / 64) + wave) * 16u;
            w[0] = t_begin; w[1] = t_qempty; w[2] = wall_clock64(); w[3] = n_iter; w[4] = n_shade; w[5] = n_fill;
            w[6] = n_iter_q; w[7] = lanes_sum; w[8] = lanes_sum_q; w[9] = leaf_lanes;
            w[10] = cy_shade; w[11] = cy_fill; w[12] = cy_step; w[13] = cy_step_q;
        }
        atomicAdd(&A.stats[8], (unsigned long long)push_ops); atomicAdd(&A.stats[9], (unsigned long long)push8_ops);
        atomicAdd(&A.stats[10], (unsigned long long)push12_ops); atomicAdd(&A.stats[11], (unsigned long long)spill_ops);
        atomicAdd(&A.stats[0], (unsigned long long)c_closest);
        atomicAdd(&A.stats[1], (unsigned long long)c_shadow);
        atomicAdd(&A.stats[2], (unsigned long long)c_nodes);
        atomicAdd(&A.stats[3], (unsigned long long)c_tris);
        atomicAdd(&A.stats[4], (unsigned long long)c_drops);
        atomicMax(&A.stats[5], (unsigned long long)c_maxstack);
        atomicAdd(&A.stats[6], (unsigned long long)c_samples);
    }
}

// Sum the samples of each owned pixel in sample order; carry the running sum when accumulating.  blockIdx.y = frame of
// the batch when the frames are independent (ACCUM false); an accumulating sequence walks its frames in submission order
// (the running sum is order dependent).  Samples are read four at a time, so a wavefront keeps several loads in flight: the
// pass runs next to a persistent launch that leaves it about one wave slot per SIMD.  It only READS the sample buffer: the trace
// primes every batch it generates.
template <bool ACCUM>
__global__ __launch_bounds__(256) void resolve_kernel(const RenderArgs A) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slot = idx >> 6, p = idx & 63u;
    if (idx == 0u && blockIdx.y == 0u) { for (int k = 0; k < 16; ++k) A.queue[k] = 0u; }   // the slot's control block rewound for its next launch
    if (slot >= A.num_tiles) return;
    const uint32_t tile = A.tiles ? A.tiles[slot] : slot;
    const uint32_t tx = tile % A.tiles_x, ty = tile / A.tiles_x;
    const uint32_t px = tx * 8u + (p & 7u), py = ty * 8u + (p >> 3);
    if (px >= A.width || py >= A.height) return;
    const size_t out_index = A.compact ? (size_t)idx : ((size_t)py * A.width + px);
    const float bg = 0.0f + 1.0f * kBgPrimary;
    const FrameParams* const frames = A.frames;
    float4* const* const outs = A.outs;
    const uint32_t f_begin = ACCUM ? 0u : blockIdx.y, f_end = ACCUM ? A.num_frames : blockIdx.y + 1u;
    // a tile outside the traced rectangle (every camera ray provably misses the root box, pt_api.cpp) was never touched by the
    // trace: its samples would be the primed value, which is summed from a register -- nothing is read
    // (the same additions in the same order, so the same bits).  For the dragon-class frame that is two thirds of the pass's traffic.
    const bool untraced = A.trace_slots != nullptr && !(tx >= A.trace_rect[0] && tx < A.trace_rect[1] && ty >= A.trace_rect[2] && ty < A.trace_rect[3]);
    for (uint32_t fid = f_begin; fid < f_end; ++fid) {
        F3 sum = f3(0.0f, 0.0f, 0.0f);
        float4* const sp = A.samples + (((size_t)fid * A.batches_per_frame + (size_t)slot * A.spp) * 64u + p);   // sample s at sp[s * 64]
        if (untraced) {
            if (!ACCUM && (frames[fid].accum_mode & 0x100u)) continue;
            for (uint32_t s = 0; s < A.spp; ++s) sum = sum + f3(bg, bg, bg);
        } else {
        if (!ACCUM) {
            // independent frames resolve in parallel; where several share one output target the last submitted one is the
            // result (what resolving them in order would leave), the others only hand their sample slots back
            if (frames[fid].accum_mode & 0x100u) continue;
        }
        for (uint32_t s0 = 0; s0 < A.spp; s0 += 4u) {
            const uint32_t n = min(4u, A.spp - s0);
            float4 v[4];
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) v[j] = (j < n) ? sp[(size_t)(s0 + j) * 64u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) if (j < n) sum = sum + f3(v[j].x, v[j].y, v[j].z);
        }
        }
        float count = (float)A.spp;
        if (ACCUM) {
            const uint32_t am = frames[fid].accum_mode & 0xffu;
            const float4 acc = (am == 2u) ? A.accum[out_index] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            sum = f3(acc.x + sum.x, acc.y + sum.y, acc.z + sum.z);
            count = acc.w + count;
            A.accum[out_index] = make_float4(sum.x, sum.y, sum.z, count);
        }
        const float inv = 1.0f / count;
        outs[fid][out_index] = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, 1.0f);
    }
}

// Trace phase of one frame (pass 0 + continuation passes) on `stream`; k0/k1 (optional)
// are recorded immediately around the trace_paths_kernel launches.
hipError_t launch_trace(const RenderArgs& A0, bool stats, uint32_t grid_blocks, hipStream_t stream, hipEvent_t k0, hipEvent_t k1) {
    RenderArgs A = A0;
    hipError_t e = hipSuccess;
    if (A.total_items == 0u) return hipSuccess;
    if (A0.prime) {
        // first use of this slot's control block: zero it; afterwards resolve_kernel rewinds it for the slot's next launch.  (The
        // sample buffer needs no preparation: the trace primes every batch it generates, and nothing else is ever read.)
        e = hipMemsetAsync(A.queue, 0, 16 * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    // control block (u32): [0..3] item cursors of the passes; [4],[5] tail/head of pool A; [6],[7] tail/head of pool B
    uint32_t* const ctrl = A0.queue;
    float4* const pool_a = A0.pool; float4* const pool_b = A0.pool + (size_t)A0.pool_capacity * 4;
    uint32_t* const flags_a = A0.pool_flags; uint32_t* const flags_b = A0.pool_flags + A0.pool_capacity;
    const uint32_t passes = (A.max_bounces == 0u || A0.flush_threshold == 0u) ? 0u : (A0.cont_passes < 3u ? A0.cont_passes : 3u);   // nothing is donated without a flush threshold
    if (passes > 0u) {
        e = hipMemsetAsync(A0.pool_flags, 0, (size_t)A0.pool_capacity * 2u * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
    }
    // pass 0: pixel-samples; with a flush threshold, sparse wavefronts donate their paths into pool A
    A.queue = ctrl; A.pool = pool_a; A.pool_flags = flags_a; A.pool_ctrl = ctrl + 4;
    A.flush_threshold = passes > 0u ? A0.flush_threshold : 0u;
    if (k0) { e = hipEventRecord(k0, stream); if (e != hipSuccess) return e; }
    if (stats) hipLaunchKernelGGL((trace_paths_kernel<true, false>), dim3(grid_blocks), dim3(PT_MEGA_BLOCK), 0, stream, A);
    else       hipLaunchKernelGGL((trace_paths_kernel<false, false>), dim3(grid_blocks), dim3(PT_MEGA_BLOCK), 0, stream, A);
    e = hipGetLastError(); if (e != hipSuccess) return e;
    // continuation passes pick up what is left in the previous pool; the last one never donates
    for (uint32_t pass = 1; pass <= passes; ++pass) {
        const bool odd = (pass & 1u) != 0u;
        A.queue = ctrl + pass;
        A.in_pool = odd ? pool_a : pool_b; A.in_ctrl = ctrl + (odd ? 4 : 6);
        A.pool = odd ? pool_b : pool_a; A.pool_flags = odd ? flags_b : flags_a; A.pool_ctrl = ctrl + (odd ? 6 : 4);
        A.flush_threshold = (pass < passes) ? A0.flush_threshold : 0u;
        if (pass == 2u && passes > 2u) {   // pool A is reused as the output of pass 2
            e = hipMemsetAsync(flags_a, 0, (size_t)A0.pool_capacity * sizeof(uint32_t), stream); if (e != hipSuccess) return e;
            e = hipMemsetAsync(ctrl + 4, 0, 2 * sizeof(uint32_t), stream); if (e != hipSuccess) return e;
        }
        if (stats) hipLaunchKernelGGL((trace_paths_kernel<true, true>), dim3(grid_blocks), dim3(PT_MEGA_BLOCK), 0, stream, A);
        else       hipLaunchKernelGGL((trace_paths_kernel<false, true>), dim3(grid_blocks), dim3(PT_MEGA_BLOCK), 0, stream, A);
        e = hipGetLastError(); if (e != hipSuccess) return e;
    }
    if (k1) { e = hipEventRecord(k1, stream); if (e != hipSuccess) return e; }   // k0..k1 = all trace_paths passes
    return hipSuccess;
}

hipError_t launch_resolve(const RenderArgs& A, hipStream_t stream) {
    if (A.num_tiles == 0u || A.num_frames == 0u) return hipSuccess;       // total_items may be 0 (every tile culled): the primed miss values are still delivered
    const uint32_t n = A.num_tiles * 64u;
    if (A.accum) hipLaunchKernelGGL(resolve_kernel<true>, dim3((n + 255u) / 256u, 1), dim3(256), 0, stream, A);
    else         hipLaunchKernelGGL(resolve_kernel<false>, dim3((n + 255u) / 256u, A.num_frames), dim3(256), 0, stream, A);
    return hipGetLastError();
}

__global__ void write_frame_params_kernel(const FrameChunk c, FrameParams* d_frames, float4** d_outs, uint32_t offset, uint32_t n) {
    const uint32_t i = threadIdx.x;
    if (i < n) { d_frames[offset + i] = c.f[i]; d_outs[offset + i] = c.o[i]; }
}

hipError_t launch_frame_params(const FrameParams* frames, float4* const* outs, uint32_t n, FrameParams* d_frames, float4** d_outs, hipStream_t stream) {
    for (uint32_t off = 0; off < n; off += (uint32_t)kFrameChunk) {
        FrameChunk c;
        const uint32_t m = n - off < (uint32_t)kFrameChunk ? n - off : (uint32_t)kFrameChunk;
        for (uint32_t i = 0; i < m; ++i) { c.f[i] = frames[off + i]; c.o[i] = outs[off + i]; }
        for (uint32_t i = m; i < (uint32_t)kFrameChunk; ++i) { c.f[i] = frames[off]; c.o[i] = nullptr; }
        hipLaunchKernelGGL(write_frame_params_kernel, dim3(1), dim3(kFrameChunk), 0, stream, c, d_frames, d_outs, off, m);
        const hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_prime(uint32_t* queue, float4* samples, uint32_t n_samples, hipStream_t stream) {
    (void)samples; (void)n_samples;       // the trace primes the samples of every batch it generates
    return hipMemsetAsync(queue, 0, 16 * sizeof(uint32_t), stream);
}

uint32_t megakernel_grid(int num_cus) { return (uint32_t)num_cus * PT_MEGA_WAVES_PER_SIMD * (256 / PT_MEGA_BLOCK); }
uint32_t megakernel_block() { return PT_MEGA_BLOCK; }


} // namespace ptk
