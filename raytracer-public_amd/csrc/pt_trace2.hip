// pt_trace2.hip -- the hot path for PT_MODE_PATH on gfx950, second generation: a persistent-wavefront megakernel whose
// lanes carry TWO rays each.
//
// Why two.  trace_paths_kernel (pt_megakernel.hip, one ray per lane) is bound by vector-instruction issue (profiles/README.md:
// VALU issue is the saturated unit; tools/probes/gather64.hip: the 64-byte gathers hide completely behind ~250 VALU instructions
// per iteration, whatever the number of wavefronts per SIMD), and more than half of the issued lane-slots do nothing: in every
// iteration a wavefront runs the node step (~135 instructions) for the ~86 % of its traversing lanes that stand at an internal
// node AND the triangle step (~100 instructions) for the ~14 % that stand at a leaf, and lanes whose ray has ended idle until
// enough of them wait for a shade pass.  Here every lane owns two independent ray slots and an iteration does ONE kind of work
// for the whole wavefront:
//     node step     every lane that has a slot standing at an internal node advances that slot (one 64 B record, four slab tests)
//     leaf step     run only when enough slots wait at a leaf: every lane that has such a slot tests its triangle
//     service pass  run only when enough slots wait with an ended ray (shade, next-event estimation, next bounce) or empty
//                   (new pixel-sample from the queue); both end in the same root-box test and write-back
// A slot that waits for its kind of step costs nothing, because the lane advances its other slot meanwhile; the steps run
// dense.  The visit order of every ray, its arithmetic and therefore images and counters are exactly those of the one-ray
// kernels (DESIGN.md sections 3-6; renderer.wgsl:210-346 with a one-lane packet): only WHEN a step happens changes.
//
// Per-slot traversal state lives in registers (13 VGPRs: origin, direction, inverse direction, best t, best triangle, current
// reference, stack depth), selected per lane with v_cndmask; per-path state that only the service pass touches (throughput,
// radiance, next direction, pending NEE contribution, RNG key, sample index, bounce) lives in a wavefront-private 64 B record
// per slot and lane in global memory (L2 resident, read and written with whole-wavefront coalesced dwordx4 accesses).
// The traversal stacks are two LDS short stacks per lane (conflict-free: entry i of lane l at (i*64 + l)*8), deep entries
// spill to a global area; the reference's 64-entry cap and silent drop (renderer.wgsl:337) are kept.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_kernels.h"
#include "pt_device.h"

namespace ptk {

namespace {

constexpr int kS2 = PT2_SHORT_STACK;                 // LDS stack entries per slot and lane
constexpr uint32_t kPhIdle = 0, kPhNode = 1, kPhLeaf = 2, kPhHit = 3, kPhRet = 4;
constexpr uint32_t kShadow = 1u << 16, kCont = 1u << 17;     // bits of the path's bounce word: current ray is a shadow ray / path goes on after it

struct Slot { F3 o, d, inv; float best_t; uint32_t best_tri, cur; int sp; };

__device__ __forceinline__ float sel(bool c, float a, float b) { return c ? a : b; }
__device__ __forceinline__ uint32_t sel(bool c, uint32_t a, uint32_t b) { return c ? a : b; }
__device__ __forceinline__ int sel(bool c, int a, int b) { return c ? a : b; }
__device__ __forceinline__ F3 sel(bool c, F3 a, F3 b) { return f3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
__device__ __forceinline__ uint32_t popc(unsigned long long m) { return (uint32_t)__popcll(m); }

}  // namespace

template <bool STATS>
__global__ __launch_bounds__(64, PT2_WAVES_PER_SIMD) void trace2_kernel(const RenderArgs A) {
    __shared__ unsigned long long lds_stack[2][kS2][64];
    const uint32_t lane = threadIdx.x;
    unsigned long long* const stk0 = &lds_stack[0][0][lane];        // entry i of slot s at stk0[(s * kS2 + i) * 64]: (tmin bits << 32) | ref
    unsigned long long* const spill0 = (unsigned long long*)A.spill + ((size_t)blockIdx.x * 64u + lane);   // entry j of slot s at spill0[(s * (64 - kS2) + j) * spill_stride]
    const size_t spill_stride = (size_t)gridDim.x * 64u;
    uint4* const path0 = A.path_state + ((size_t)blockIdx.x * 8u) * 64u + lane;                           // quarter q of slot s at path0[(s * 4 + q) * 64]

    const F3 base = f3(0.9f, 0.7f, 0.3f);
    const F3 L = light_dir();
    const F3 invL = safe_inv(L);
    const bool scene_empty = (A.root_ref == kInvalidRef) || (A.num_tris == 0u);
    const uint32_t root_phase = (A.root_ref & kLeaf) ? kPhLeaf : kPhNode;
    const FrameParams* const frames = A.frames;
    const uint32_t total_items = A.total_items, chunk_items = A.chunk_items;
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;     // HW_REG_XCC_ID
    uint32_t xcd_hop = 0, chunk_next = 0, chunk_end = 0;
    bool queue_empty = false;

    Slot s0, s1;
    s0.o = s0.d = s0.inv = f3(0, 0, 0); s0.best_t = kInfT; s0.best_tri = kInvalidRef; s0.cur = 0; s0.sp = 0;
    s1 = s0;
    uint32_t ph0 = kPhIdle, ph1 = kPhIdle;   // phase of each slot
    uint32_t flags = 0;                      // bit s: slot s carries a shadow (any-hit) ray; bit 2 + s: slot s carries a camera ray
    // wave-uniform census of the 128 slots
    uint32_t n_node = 0, n_leaf = 0, n_hit = 0, n_ret = 0, n_idle = 128;

    unsigned long long t_begin = 0, t_qempty = 0;
    uint32_t it_node = 0, it_leaf = 0, it_hit = 0, it_ret = 0, it_cam = 0, it_node_q = 0;
    unsigned long long lanes_node = 0, lanes_leaf = 0, lanes_hit = 0, lanes_ret = 0, lanes_cam = 0, lanes_node_q = 0;
    uint32_t c_nodes = 0, c_tris = 0, c_drops = 0, c_maxstack = 0, c_closest = 0, c_shadow = 0, c_samples = 0;
    uint32_t push_ops = 0, push8_ops = 0, push12_ops = 0, spill_ops = 0;
    if (STATS) t_begin = wall_clock64();

    // The common tail of the three service passes: root-box test of the ray (o, d, inv) that slot `sl1` of this lane starts, and
    // the write-back of the slot.  Returns the slot's new phase.
    auto start_ray = [&](bool sl1, F3 o, F3 d, F3 inv, bool is_shadow, bool is_camera) -> uint32_t {
        if (STATS) { if (is_shadow) ++c_shadow; else ++c_closest; }
        bool enters = false;
        if (!scene_empty) {
            if (STATS) { c_nodes += 1; if (c_maxstack < 1u) c_maxstack = 1u; }       // the root record is fetched before its degenerate check (renderer.wgsl:240-244)
            if (A.root_degenerate == 0u) {
                Ray r; r.o = o; r.d = d; r.inv = inv;
                float troot;
                enters = slab(r, A.root_box[0], A.root_box[1], A.root_box[2], kInfT, troot);
            }
        }
        const bool w0 = !sl1, w1 = sl1;
        s0.o = sel(w0, o, s0.o); s0.d = sel(w0, d, s0.d); s0.inv = sel(w0, inv, s0.inv);
        s1.o = sel(w1, o, s1.o); s1.d = sel(w1, d, s1.d); s1.inv = sel(w1, inv, s1.inv);
        s0.best_t = sel(w0, kInfT, s0.best_t); s1.best_t = sel(w1, kInfT, s1.best_t);
        s0.best_tri = sel(w0, kInvalidRef, s0.best_tri); s1.best_tri = sel(w1, kInvalidRef, s1.best_tri);
        s0.cur = sel(w0, A.root_ref, s0.cur); s1.cur = sel(w1, A.root_ref, s1.cur);
        s0.sp = sel(w0, 0, s0.sp); s1.sp = sel(w1, 0, s1.sp);
        const uint32_t sbit = sl1 ? 2u : 1u, cbit = sl1 ? 8u : 4u;
        flags = (flags & ~(sbit | cbit)) | (is_shadow ? sbit : 0u) | (is_camera ? cbit : 0u);
        // A camera ray that misses keeps the prefilled sample value (0 + 1 * 0.01, renderer.wgsl:410): its slot is simply free again.
        // Any other closest ray that misses, and every shadow ray, comes back through the return pass.
        return enters ? root_phase : (is_camera ? kPhIdle : kPhRet);
    };

    for (;;) {
        const uint32_t traversable = n_node + n_leaf;
        // a class of waiting slots is served when it has reached its threshold, or a quarter of what is being traversed (sparse
        // wavefronts: nobody waits for a crowd that will not come), or when nothing else is left to do
        const uint32_t quarter = max(1u, traversable >> 2);
        // =========================================================================================== hit pass
        // closest rays that ended on a triangle: next-event estimation, Russian roulette, BSDF sample; starts the shadow ray or
        // the next bounce
        if (n_hit != 0u && n_hit >= min(A.shade_threshold, quarter)) {
            const bool sl1 = ph0 != kPhHit;
            const bool mine = (ph0 == kPhHit) | (ph1 == kPhHit);
            uint32_t nph = kPhIdle;
            if (STATS) { ++it_hit; lanes_hit += popc(__ballot(mine)); }
            if (mine) {
                uint4* const pst = path0 + (size_t)(sl1 ? 4u : 0u) * 64u;
                const uint4 q0 = pst[0], q1 = pst[64], q2 = pst[128];
                F3 T = f3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
                const F3 rad = f3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
                const uint32_t key = q1.z, item = q1.w, b = q2.x & 0xffffu;
                const F3 o = sel(sl1, s1.o, s0.o), d = sel(sl1, s1.d, s0.d);
                const float best_t = sel(sl1, s1.best_t, s0.best_t);
                const uint32_t best_tri = sel(sl1, s1.best_tri, s0.best_tri);
                const F3 n = tri_normal(A, best_tri);
                const F3 hp = o + d * best_t;
                const F3 nf = (dot3(n, d) < 0.0f) ? n : f3(-n.x, -n.y, -n.z);
                const F3 so = hp + nf * kEpsOrigin;
                const float ndl = dot3(nf, L);
                const F3 contrib = (T * base) * ndl;
                F3 d_next = f3(0, 0, 0);
                bool cont = b < A.max_bounces;
                if (cont) {
                    F3 Tn = T * base;
                    if (b >= kRRStart) {
                        const float p = wmax(wmax(Tn.x, Tn.y), Tn.z);
                        if (rnd(key, b, 4) >= p) cont = false;
                        else Tn = Tn * (1.0f / p);
                    }
                    if (cont) { T = Tn; d_next = cosine_dir(nf, rnd(key, b, 2), rnd(key, b, 3)); }
                }
                if (ndl > 0.0f) {
                    pst[0] = make_uint4(__float_as_uint(T.x), __float_as_uint(T.y), __float_as_uint(T.z), q0.w);
                    pst[128] = make_uint4(b | kShadow | (cont ? kCont : 0u), __float_as_uint(contrib.x), __float_as_uint(contrib.y), __float_as_uint(contrib.z));
                    pst[192] = make_uint4(__float_as_uint(d_next.x), __float_as_uint(d_next.y), __float_as_uint(d_next.z), 0u);
                    nph = start_ray(sl1, so, L, invL, true, false);
                } else if (cont) {
                    pst[0] = make_uint4(__float_as_uint(T.x), __float_as_uint(T.y), __float_as_uint(T.z), q0.w);
                    pst[128] = make_uint4(b + 1u, 0u, 0u, 0u);
                    nph = start_ray(sl1, so, d_next, safe_inv(d_next), false, false);
                } else {
                    A.samples[item] = make_float4(rad.x, rad.y, rad.z, 1.0f);
                    nph = kPhIdle;
                }
                if (sl1) ph1 = nph; else ph0 = nph;
            }
            const uint32_t k = popc(__ballot(mine));
            const uint32_t k_node = popc(__ballot(mine & (nph == kPhNode))), k_leaf = popc(__ballot(mine & (nph == kPhLeaf))), k_ret = popc(__ballot(mine & (nph == kPhRet)));
            n_hit -= k; n_node += k_node; n_leaf += k_leaf; n_ret += k_ret; n_idle += k - k_node - k_leaf - k_ret;
            continue;
        }
        // =========================================================================================== return pass
        // shadow rays that came back (add the pending contribution if nothing was hit, then go on with the sampled direction or
        // end the path) and paths that left the scene (sky radiance, end)
        if (n_ret != 0u && n_ret >= min(A.shade_threshold, quarter)) {
            const bool sl1 = ph0 != kPhRet;
            const bool mine = (ph0 == kPhRet) | (ph1 == kPhRet);
            uint32_t nph = kPhIdle;
            if (STATS) { ++it_ret; lanes_ret += popc(__ballot(mine)); }
            if (mine) {
                uint4* const pst = path0 + (size_t)(sl1 ? 4u : 0u) * 64u;
                const uint4 q0 = pst[0], q1 = pst[64], q2 = pst[128], q3 = pst[192];
                const F3 T = f3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
                F3 rad = f3(__uint_as_float(q0.w), __uint_as_float(q1.x), __uint_as_float(q1.y));
                const uint32_t item = q1.w, bounce = q2.x, b = bounce & 0xffffu;
                const bool hit = sel(sl1, s1.best_tri, s0.best_tri) != kInvalidRef;
                bool go_on = false;
                if (bounce & kShadow) {
                    if (!hit) rad = rad + f3(__uint_as_float(q2.y), __uint_as_float(q2.z), __uint_as_float(q2.w));
                    go_on = (bounce & kCont) != 0u;
                } else {
                    rad = rad + T * ((b == 0u) ? kBgPrimary : kSkyAmbient);       // b == 0 only for a camera ray restarted by ... (never: camera misses do not come here)
                }
                if (go_on) {
                    const F3 dn = f3(__uint_as_float(q3.x), __uint_as_float(q3.y), __uint_as_float(q3.z));
                    pst[0] = make_uint4(q0.x, q0.y, q0.z, __float_as_uint(rad.x));
                    pst[64] = make_uint4(__float_as_uint(rad.y), __float_as_uint(rad.z), q1.z, q1.w);
                    pst[128] = make_uint4(b + 1u, 0u, 0u, 0u);
                    nph = start_ray(sl1, sel(sl1, s1.o, s0.o), dn, safe_inv(dn), false, false);
                } else {
                    A.samples[item] = make_float4(rad.x, rad.y, rad.z, 1.0f);
                    nph = kPhIdle;
                }
                if (sl1) ph1 = nph; else ph0 = nph;
            }
            const uint32_t k = popc(__ballot(mine));
            const uint32_t k_node = popc(__ballot(mine & (nph == kPhNode))), k_leaf = popc(__ballot(mine & (nph == kPhLeaf))), k_ret = popc(__ballot(mine & (nph == kPhRet)));
            n_ret -= k; n_node += k_node; n_leaf += k_leaf; n_ret += k_ret; n_idle += k - k_node - k_leaf - k_ret;
            continue;
        }
        // =========================================================================================== camera pass
        // empty slots take new pixel-samples from the queue (one atomic per wavefront and chunk)
        if (!queue_empty && n_idle != 0u && (n_idle >= A.fill_threshold || traversable + n_hit + n_ret == 0u)) {
            const bool sl1 = ph0 != kPhIdle;
            const bool mine = (ph0 == kPhIdle) | (ph1 == kPhIdle);
            const unsigned long long m_fill = __ballot(mine);
            bool got_item = false; uint32_t logical = 0;
            if (chunk_next == chunk_end) {
                if (A.xcd_span != 0u) {
                    // XCD-aware queue: 8 ranges of the logical item order, one cursor each; a wavefront works through the range of
                    // the XCD it runs on and moves on to the next XCD's when its own is dry
                    for (;;) {
                        const uint32_t qi = (xcc + xcd_hop) & 7u;
                        const uint32_t q_begin = qi * A.xcd_span, q_end = min(q_begin + A.xcd_span, total_items);
                        uint32_t start = 0;
                        if (lane == 0) start = atomicAdd(A.queue + 8 + qi, chunk_items);
                        start = __builtin_amdgcn_readfirstlane(start);
                        if (q_begin < q_end && start < q_end - q_begin) { chunk_next = q_begin + start; chunk_end = min(chunk_next + chunk_items, q_end); break; }
                        if (++xcd_hop == 8u) { queue_empty = true; break; }
                    }
                } else {
                    uint32_t start = 0;
                    if (lane == 0) start = atomicAdd(A.queue, chunk_items);
                    start = __builtin_amdgcn_readfirstlane(start);
                    if (start >= total_items) queue_empty = true;
                    else { chunk_next = start; chunk_end = min(start + chunk_items, total_items); }
                }
                if (STATS && queue_empty) { t_qempty = wall_clock64(); it_node_q = it_node; lanes_node_q = lanes_node; }
            }
            if (queue_empty) continue;
            {
                const uint32_t rank = popc(m_fill & ((1ull << lane) - 1ull)), avail = chunk_end - chunk_next;
                got_item = mine & (rank < avail);
                logical = chunk_next + rank;
                chunk_next += min(popc(m_fill), avail);
            }
            if (STATS) { ++it_cam; lanes_cam += popc(__ballot(got_item)); }
            uint32_t nph = kPhIdle;
            if (got_item) {
                const ItemInfo it = decode_item(A, logical);
                if (it.valid) {
                    const uint32_t fid = it.fid, px = it.px, py = it.py, s = it.s, item = it.sample_index;
                    const FrameParams fp = frames[fid];
                    const uint32_t key = sample_key(fp.seed, py * A.width + px, fp.frame * A.spp + s);
                    const Ray r = primary_ray_fp(A, fp, (float)px + rnd(key, 0, 0), (float)py + rnd(key, 0, 1));
                    if (STATS) ++c_samples;
                    nph = start_ray(sl1, r.o, r.d, r.inv, false, true);
                    if (nph != kPhIdle) {
                        uint4* const pst = path0 + (size_t)(sl1 ? 4u : 0u) * 64u;
                        pst[0] = make_uint4(__float_as_uint(1.0f), __float_as_uint(1.0f), __float_as_uint(1.0f), 0u);
                        pst[64] = make_uint4(0u, 0u, key, item);
                        pst[128] = make_uint4(0u, 0u, 0u, 0u);
                    }
                }
                if (sl1) ph1 = nph; else ph0 = nph;
            }
            const uint32_t k_node = popc(__ballot(got_item & (nph == kPhNode))), k_leaf = popc(__ballot(got_item & (nph == kPhLeaf)));
            n_idle -= k_node + k_leaf; n_node += k_node; n_leaf += k_leaf;
            continue;
        }
        if (traversable == 0u) {
            if (n_hit + n_ret == 0u) break;        // queue dry (or nothing could be fetched), every slot empty
            continue;                              // unreachable by construction: a waiting class with nothing traversing passes its threshold
        }
        // =========================================================================================== one traversal step
        // the wavefront does ONE kind of step: triangle tests when enough slots wait at a leaf, node steps otherwise; a lane
        // advances the first of its slots that is due for that kind
        const bool do_leaf = (n_leaf != 0u) & ((n_leaf >= min(A.leaf_threshold, max(1u, n_node >> 2))) | (n_node == 0u));
        const uint32_t want = do_leaf ? kPhLeaf : kPhNode;
        const bool a0 = ph0 == want, a1 = ph1 == want;
        const bool active = a0 | a1;
        const bool sl1 = !a0;
        uint32_t nph = want;
        if (STATS) { const uint32_t k = popc(__ballot(active)); if (do_leaf) { ++it_leaf; lanes_leaf += k; } else { ++it_node; lanes_node += k; } }
        if (active) {
            uint32_t cur = sel(sl1, s1.cur, s0.cur);
            int sp = sel(sl1, s1.sp, s0.sp);
            float best_t = sel(sl1, s1.best_t, s0.best_t);
            const F3 o = sel(sl1, s1.o, s0.o);
            unsigned long long* const stk = stk0 + (size_t)(sl1 ? kS2 * 64 : 0);
            unsigned long long* const spill = spill0 + (size_t)(sl1 ? (64 - kS2) : 0) * spill_stride;
            bool need_pop = false, ended = false, has_hit = false;
            if (do_leaf) {
                // ---- triangle test (renderer.wgsl:171-208), branch-free: same operations and comparisons, rejections combined at the end
                const F3 d = sel(sl1, s1.d, s0.d);
                const uint32_t ti = cur & 0x7fffffffu;
                if (ti < A.num_tris) {
                    const float4* tp = A.tris + (size_t)ti * 3;
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    if (STATS) ++c_tris;
                    const F3 v0 = f3(a.x, a.y, a.z), e1 = f3(a.w, b.x, b.y), e2 = f3(b.z, b.w, c.x);
                    const F3 pv = cross3(d, e2);
                    const float det = dot3(e1, pv);
                    const bool ok_det = !(fabsf(det) < kTriEps);
                    const float inv_det = 1.0f / det;
                    const F3 sv = o - v0;
                    const float u = inv_det * dot3(sv, pv);
                    const bool ok_u = !((u < 0.0f) | (u > 1.0f));
                    const F3 q = cross3(sv, e1);
                    const float v = inv_det * dot3(d, q);
                    const bool ok_v = !((v < 0.0f) | ((u + v) > 1.0f));
                    const float t = inv_det * dot3(e2, q);
                    if (ok_det & ok_u & ok_v & (t > kTriEps) & (t < best_t)) {
                        best_t = t;
                        if (sl1) { s1.best_t = t; s1.best_tri = ti; } else { s0.best_t = t; s0.best_tri = ti; }
                        if (flags & (sl1 ? 2u : 1u)) { ended = true; has_hit = true; }        // any-hit: the first accepted hit ends a shadow ray
                    }
                }
                need_pop = !ended;
            } else {
                // ---- node step: one 64 B record, four slab tests, branch-free child ordering (renderer.wgsl:283-342 for one lane)
                const F3 inv = sel(sl1, s1.inv, s0.inv);
                Ray r; r.o = o; r.d = f3(0, 0, 0); r.inv = inv;
                const uint4* rec = A.nodes + (size_t)cur * 4;
                const uint4 n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
                float t0, t1, t2, t3;
                const bool c0 = slab(r, n0.x, n0.y, n0.z, best_t, t0);
                const bool c1 = slab(r, n0.w, n1.x, n1.y, best_t, t1);
                const bool c2 = slab(r, n1.z, n1.w, n2.x, best_t, t2);
                const bool c3 = slab(r, n2.y, n2.z, n2.w, best_t, t3);
                // enterable slots: neither empty (kInvalidRef) nor a degenerate child (kDegenerateRef: counted, never entered)
                const bool h0 = c0 & (n3.x < kDegenerateRef), h1 = c1 & (n3.y < kDegenerateRef);
                const bool h2 = c2 & (n3.z < kDegenerateRef), h3 = c3 & (n3.w < kDegenerateRef);
                if (STATS) c_nodes += (n3.x != kInvalidRef) + (n3.y != kInvalidRef) + (n3.z != kInvalidRef) + (n3.w != kInvalidRef);
                // Hit children keep slot order; the nearest (first minimum of tmin) is entered next and trades places with the
                // first hit, so the stacked entry for slot s is the first hit's when s is the nearest slot, the slot's own child
                // otherwise -- and a slot is stacked iff it is hit and is not the first hit.
                const float kBig = 3.0e38f;
                const float e0 = h0 ? t0 : kBig, e1 = h1 ? t1 : kBig, e2 = h2 ? t2 : kBig, e3 = h3 ? t3 : kBig;
                const float tn = wmin(wmin(e0, e1), wmin(e2, e3));
                const bool any = h0 | h1 | h2 | h3;
                const bool m0 = h0 & (e0 == tn), m1 = h1 & (e1 == tn) & !m0, m2 = h2 & (e2 == tn) & !(m0 | m1);
                const bool m3 = !(m0 | m1 | m2);
                const uint32_t rn = m0 ? n3.x : m1 ? n3.y : m2 ? n3.z : n3.w;
                const uint32_t rf = h0 ? n3.x : h1 ? n3.y : h2 ? n3.z : n3.w;      // first hit
                const float tf = h0 ? t0 : h1 ? t1 : h2 ? t2 : t3;
                if (!any) need_pop = true;
                else {
                    const bool p3 = h3 & (h0 | h1 | h2), p2 = h2 & (h0 | h1), p1 = h1 & h0;
                    const unsigned long long en = ((unsigned long long)__float_as_uint(tf) << 32) | rf;
                    const unsigned long long w3 = m3 ? en : (((unsigned long long)__float_as_uint(t3) << 32) | n3.w);
                    const unsigned long long w2 = m2 ? en : (((unsigned long long)__float_as_uint(t2) << 32) | n3.z);
                    const unsigned long long w1 = m1 ? en : (((unsigned long long)__float_as_uint(t1) << 32) | n3.y);
                    if (__builtin_expect(sp + 3 <= kS2, 1)) {
                        // fast path: three unconditional LDS stores far -> near; an entry that is not stacked is overwritten by the next
                        const int sp_in = sp;
                        stk[sp * 64] = w3; sp += p3 ? 1 : 0;
                        stk[sp * 64] = w2; sp += p2 ? 1 : 0;
                        stk[sp * 64] = w1; sp += p1 ? 1 : 0;
                        if (STATS) { push_ops += sp - sp_in; for (int j = sp_in; j < sp; ++j) { if (j >= 8) ++push8_ops; if (j >= 12) ++push12_ops; } }
                    } else {
                        auto push = [&](unsigned long long e) {
                            if (sp < kStackMax) {
                                if (sp < kS2) stk[sp * 64] = e; else spill[(size_t)(sp - kS2) * spill_stride] = e;
                                if (STATS) { ++push_ops; if (sp >= 8) ++push8_ops; if (sp >= 12) ++push12_ops; if (sp >= kS2) ++spill_ops; }
                                ++sp;
                            } else if (STATS) ++c_drops;
                        };
                        if (p3) push(w3);
                        if (p2) push(w2);
                        if (p1) push(w1);
                    }
                    if (STATS) { const uint32_t depth = (uint32_t)sp + (sp < kStackMax ? 1u : 0u); if (depth > c_maxstack) c_maxstack = depth; }
                    if (sp < kStackMax) cur = rn;                              // the push of the nearest child would have fitted
                    else { need_pop = true; if (STATS) ++c_drops; }            // ... it is the first to be dropped (renderer.wgsl:337)
                }
            }
            if (need_pop) {
                bool found = false;
                while (sp > 0) {
                    --sp;
                    unsigned long long e = stk[(sp < kS2 ? sp : kS2 - 1) * 64];
                    if (__builtin_expect(sp >= kS2, 0)) e = *(volatile unsigned long long*)&spill[(size_t)(sp - kS2) * spill_stride];
                    if (__uint_as_float((uint32_t)(e >> 32)) < best_t) { cur = (uint32_t)e; found = true; break; }
                }
                if (!found) { ended = true; has_hit = sel(sl1, s1.best_tri, s0.best_tri) != kInvalidRef; }
            }
            if (ended) {
                // where an ended ray goes: shadow rays and closest rays that missed come back through the return pass, closest hits
                // through the hit pass; a camera ray that missed leaves its sample at the prefilled miss value and frees the slot
                const bool is_shadow = (flags & (sl1 ? 2u : 1u)) != 0u, is_camera = (flags & (sl1 ? 8u : 4u)) != 0u;
                nph = is_shadow ? kPhRet : (has_hit ? kPhHit : (is_camera ? kPhIdle : kPhRet));
            } else {
                nph = (cur & kLeaf) ? kPhLeaf : kPhNode;
            }
            if (sl1) { s1.cur = cur; s1.sp = sp; ph1 = nph; } else { s0.cur = cur; s0.sp = sp; ph0 = nph; }
        }
        // census: the advanced slots leave `want` and arrive where nph says
        {
            const uint32_t k_act = popc(__ballot(active));
            const uint32_t k_node = popc(__ballot(active & (nph == kPhNode))), k_leaf = popc(__ballot(active & (nph == kPhLeaf)));
            const uint32_t k_hit = popc(__ballot(active & (nph == kPhHit))), k_ret = popc(__ballot(active & (nph == kPhRet)));
            if (do_leaf) n_leaf -= k_act; else n_node -= k_act;
            n_node += k_node; n_leaf += k_leaf; n_hit += k_hit; n_ret += k_ret; n_idle += k_act - k_node - k_leaf - k_hit - k_ret;
        }
    }
    if (STATS) {
        if (A.wave_times && lane == 0) {
            unsigned long long* w = A.wave_times + (size_t)blockIdx.x * 16u;
            w[0] = t_begin; w[1] = t_qempty; w[2] = wall_clock64(); w[3] = it_node; w[4] = it_leaf; w[5] = it_hit;
            w[6] = it_ret; w[7] = it_cam; w[8] = it_node_q; w[9] = lanes_node; w[10] = lanes_leaf; w[11] = lanes_hit; w[12] = lanes_ret; w[13] = lanes_cam; w[14] = lanes_node_q;
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

hipError_t launch_trace2(const RenderArgs& A, bool stats, uint32_t grid_blocks, hipStream_t stream, hipEvent_t k0, hipEvent_t k1) {
    hipError_t e = hipSuccess;
    if (A.total_items == 0u) return hipSuccess;
    if (A.prime) {
        e = launch_prime(A.queue, A.samples, A.num_sample_batches * 64u, stream);
        if (e != hipSuccess) return e;
    }
    if (k0) { e = hipEventRecord(k0, stream); if (e != hipSuccess) return e; }
    if (stats) hipLaunchKernelGGL((trace2_kernel<true>), dim3(grid_blocks), dim3(64), 0, stream, A);
    else       hipLaunchKernelGGL((trace2_kernel<false>), dim3(grid_blocks), dim3(64), 0, stream, A);
    e = hipGetLastError(); if (e != hipSuccess) return e;
    if (k1) { e = hipEventRecord(k1, stream); if (e != hipSuccess) return e; }
    return hipSuccess;
}

uint32_t trace2_grid(int num_cus) { return (uint32_t)num_cus * 4u * PT2_WAVES_PER_SIMD; }
uint32_t trace2_spill_entries() { return 2u * (64u - (uint32_t)PT2_SHORT_STACK); }

}  // namespace ptk
