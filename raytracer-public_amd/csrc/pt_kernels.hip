// pt_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4):
//   * render_rays_kernel      the per-pixel-sample hot path of renderer.wgsl (ray generation,
//                             BVH4 traversal, Moller-Trumbore, shade) + the build-defined
//                             path-tracing extension, one ray per lane
//   * render_packet_kernel    the literal 2x2-packet form of renderer.wgsl:355-413, four lanes of a quad = the four rays of a packet
//   * lbvh2_internal_kernel / lbvh2_leaves_kernel   BVHBuilder.wgsl:152-306
//   * deinterleave_kernel, tonemap/quantise kernels
//
// Arithmetic contract: compiled with -ffp-contract=off; the only fused
// operations are the explicit __builtin_fmaf sites; '/' and sqrtf are correctly rounded
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); f32 denormals are kept.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_kernels.h"
#include "pt_device.h"

namespace ptk {
static_assert(kBgPrimary == 0.01f, "renderer.wgsl:410: packed tile shares (pt_group, bench.py) rebuild pixels outside the traced rectangle from this constant in every render mode");

struct Counters { uint32_t nodes, tris, drops, maxstack; };

// One-ray BVH4 traversal over the wide layout.  Visit order, tie-breaking and the 64-entry
// stack cap are those of traverseBVH4Packet (renderer.wgsl:210-346) run with a single active
// lane: hit children keep slot order, the nearest one (first minimum) trades places with the
// first hit and is entered next; a stacked child is re-validated at pop by tmin < best.
template <bool ANYHIT, bool STATS>
__device__ __forceinline__ bool traverse(const RenderArgs& A, const Ray& r, float& best_t, uint32_t& best_tri,
                                         uint2* __restrict__ stk, Counters& cnt) {
    best_t = kInfT; best_tri = kInvalidRef;
    if (A.root_ref == kInvalidRef || A.num_tris == 0u) return false;
    if (STATS) { cnt.nodes += 1; if (cnt.maxstack < 1u) cnt.maxstack = 1u; }   // the root record is fetched before its degenerate check (renderer.wgsl:240-244)
    if (A.root_degenerate) return false;
    float troot;
    if (!slab(r, A.root_box[0], A.root_box[1], A.root_box[2], best_t, troot)) return false;
    uint32_t cur = A.root_ref;
    int sp = 0;
    for (;;) {
        bool need_pop = false;
        if (cur & kLeaf) {
            const uint32_t ti4 = cur & 0x7fffffffu;               // packed reference: 4 * triangle index (16-byte units of 64 B records)
            if (ti4 < 4u * A.num_tris) {                          // renderer.wgsl:262 (an out-of-range leaf points at record num_tris)
                const uint32_t ti = ti4 >> 2;
                const float4* tp = (const float4*)arena_record(A, cur);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                if (STATS) cnt.tris += 1;
                const F3 v0 = f3(a.x, b.x, c.x), e1 = f3(a.y, b.y, c.y), e2 = f3(a.z, b.z, c.z);     // axis-major record (pt_host.h::TriRecord)
                const F3 p = cross3(r.d, e2);                    // renderer.wgsl:185-205
                const float det = dot3(e1, p);
                if (!(fabsf(det) < kTriEps)) {
                    const float inv_det = 1.0f / det;
                    const F3 s = r.o - v0;
                    const float u = inv_det * dot3(s, p);
                    if (!(u < 0.0f || u > 1.0f)) {
                        const F3 q = cross3(s, e1);
                        const float v = inv_det * dot3(r.d, q);
                        if (!(v < 0.0f || (u + v) > 1.0f)) {
                            const float t = inv_det * dot3(e2, q);
                            if (t > kTriEps && t < best_t) {
                                best_t = t; best_tri = ti;
                                if (ANYHIT) return true;
                            }
                        }
                    }
                }
            }
            need_pop = true;
        } else {
            const uint4* np = arena_record(A, cur);
            const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
            float t0, t1, t2, t3;
            // child-major record (pt_host.h::WideNode): piece k = child k's box words + its reference
            const uint32_t r0 = n0.w, r1 = n1.w, r2 = n2.w, r3 = n3.w;
            const bool h0 = (r0 < kDegenerateRef) && slab(r, n0.x, n0.y, n0.z, best_t, t0);
            const bool h1 = (r1 < kDegenerateRef) && slab(r, n1.x, n1.y, n1.z, best_t, t1);
            const bool h2 = (r2 < kDegenerateRef) && slab(r, n2.x, n2.y, n2.z, best_t, t2);
            const bool h3 = (r3 < kDegenerateRef) && slab(r, n3.x, n3.y, n3.z, best_t, t3);
            if (STATS) cnt.nodes += (r0 != kInvalidRef) + (r1 != kInvalidRef) + (r2 != kInvalidRef) + (r3 != kInvalidRef);
            // nearest = first minimum in slot order (renderer.wgsl:315-318); first = first hit
            int nslot = -1, fslot = -1; float tn = kInfT, tf = 0.0f; uint32_t rn = kInvalidRef, rf = kInvalidRef;
            if (h0) { nslot = 0; tn = t0; rn = r0; fslot = 0; tf = t0; rf = r0; }
            if (h1) { if (nslot < 0 || t1 < tn) { nslot = 1; tn = t1; rn = r1; } if (fslot < 0) { fslot = 1; tf = t1; rf = r1; } }
            if (h2) { if (nslot < 0 || t2 < tn) { nslot = 2; tn = t2; rn = r2; } if (fslot < 0) { fslot = 2; tf = t2; rf = r2; } }
            if (h3) { if (nslot < 0 || t3 < tn) { nslot = 3; tn = t3; rn = r3; } if (fslot < 0) { fslot = 3; tf = t3; rf = r3; } }
            if (nslot < 0) {
                need_pop = true;
            } else {
                // pushes far -> near (renderer.wgsl:336-342); the slot the nearest child left holds the first hit
#define PT_PUSH(REF, TMIN) do { if (sp < kStackMax) { stk[sp] = make_uint2((REF), __float_as_uint(TMIN)); ++sp; } else if (STATS) { cnt.drops += 1; } } while (0)
                if (h3) { if (nslot == 3) { if (fslot != 3) PT_PUSH(rf, tf); } else if (fslot != 3) PT_PUSH(r3, t3); }
                if (h2) { if (nslot == 2) { if (fslot != 2) PT_PUSH(rf, tf); } else if (fslot != 2) PT_PUSH(r2, t2); }
                if (h1) { if (nslot == 1) { if (fslot != 1) PT_PUSH(rf, tf); } else if (fslot != 1) PT_PUSH(r1, t1); }
#undef PT_PUSH
                if (STATS) { const uint32_t depth = (uint32_t)sp + (sp < kStackMax ? 1u : 0u); if (depth > cnt.maxstack) cnt.maxstack = depth; }   // entries incl. the nearest child, if its push fitted
                if (sp < kStackMax) cur = rn;          // the push of the nearest child would have fitted
                else { need_pop = true; if (STATS) cnt.drops += 1; }
            }
        }
        if (need_pop) {
            bool found = false;
            while (sp > 0) {
                --sp;
                const uint2 e = stk[sp];
                if (__uint_as_float(e.y) < best_t) { cur = e.x; found = true; break; }
            }
            if (!found) break;
        }
    }
    return best_tri != kInvalidRef;
}

constexpr uint32_t kSphereFlag = 0x40000000u;

// BUILD-DEFINED config C1: no BVH -- every triangle record in index order with the
// reference's Moller-Trumbore, then every analytic sphere (x,y,z,r) in index order; strict t < best.
template <bool ANYHIT, bool STATS>
__device__ __forceinline__ bool brute_trace(const RenderArgs& A, const Ray& r, float& best_t, uint32_t& best_prim, Counters& cnt) {
    best_t = kInfT; best_prim = kInvalidRef;
    for (uint32_t ti = 0; ti < A.num_tris; ++ti) {
        const float4* tp = A.tris + (size_t)ti * 4;
        const float4 a = tp[0], b = tp[1], c = tp[2];
        if (STATS) cnt.tris += 1;
        const F3 v0 = f3(a.x, b.x, c.x), e1 = f3(a.y, b.y, c.y), e2 = f3(a.z, b.z, c.z);     // axis-major record (pt_host.h::TriRecord)
        const F3 p = cross3(r.d, e2);
        const float det = dot3(e1, p);
        if (fabsf(det) < kTriEps) continue;
        const float inv_det = 1.0f / det;
        const F3 s = r.o - v0;
        const float u = inv_det * dot3(s, p);
        if (u < 0.0f || u > 1.0f) continue;
        const F3 q = cross3(s, e1);
        const float v = inv_det * dot3(r.d, q);
        if (v < 0.0f || (u + v) > 1.0f) continue;
        const float t = inv_det * dot3(e2, q);
        if (t > kTriEps && t < best_t) { best_t = t; best_prim = ti; if (ANYHIT) return true; }
    }
    for (uint32_t si = 0; si < A.num_spheres; ++si) {
        const float4 sp = A.spheres[si];
        const F3 oc = r.o - f3(sp.x, sp.y, sp.z);
        const float a = dot3(r.d, r.d), hb = dot3(oc, r.d), cc = dot3(oc, oc) - sp.w * sp.w;
        const float disc = hb * hb - a * cc;
        if (disc < 0.0f) continue;
        const float sq = sqrtf(disc);
        const float t0 = (-hb - sq) / a, t1 = (-hb + sq) / a;
        const float t = (t0 > kTriEps) ? t0 : t1;
        if (t > kTriEps && t < best_t) { best_t = t; best_prim = kSphereFlag | si; if (ANYHIT) return true; }
    }
    return best_prim != kInvalidRef;
}

template <bool ANYHIT, bool STATS, bool BRUTE>
__device__ __forceinline__ bool trace_any(const RenderArgs& A, const Ray& r, float& t, uint32_t& prim, uint2* stk, Counters& cnt) {
    if (BRUTE) return brute_trace<ANYHIT, STATS>(A, r, t, prim, cnt);
    return traverse<ANYHIT, STATS>(A, r, t, prim, stk, cnt);
}

template <bool BRUTE>
__device__ __forceinline__ F3 hit_normal(const RenderArgs& A, uint32_t prim, const Ray& r, float t) {
    if (BRUTE && (prim & kSphereFlag)) {
        const float4 sp = A.spheres[prim & ~kSphereFlag];
        const F3 p = r.o + r.d * t;
        return normalize3(p - f3(sp.x, sp.y, sp.z));
    }
    return tri_normal(A, prim);
}

// One work item = one pixel.  A wavefront owns one 8x8 tile (lane = y*8+x inside the tile).
template <int MODE, bool STATS, bool BRUTE>
__global__ __launch_bounds__(256) void render_rays_kernel(const RenderArgs A) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t slot = item >> 6, lane = item & 63u;
    if (slot >= A.num_tiles) return;
    const uint32_t tile = A.tiles ? A.tiles[slot] : slot;
    const uint32_t tx = tile % A.tiles_x, ty = tile / A.tiles_x;
    const uint32_t px = tx * 8u + (lane & 7u), py = ty * 8u + (lane >> 3);
    if (px >= A.width || py >= A.height) return;
    const size_t out_index = A.compact ? (size_t)item : ((size_t)py * A.width + px);

    uint2 stk[kStackMax];
    Counters cnt; cnt.nodes = cnt.tris = cnt.drops = cnt.maxstack = 0;
    uint32_t n_closest = 0, n_shadow = 0;
    const F3 base = f3(0.9f, 0.7f, 0.3f);
    const F3 L = light_dir();

    if (MODE == PT_KMODE_REFERENCE) {
        const Ray r = primary_ray(A, (float)px + 0.5f, (float)py + 0.5f);
        float t; uint32_t tri;
        n_closest = 1;
        const bool hit = trace_any<false, STATS, BRUTE>(A, r, t, tri, stk, cnt);
        F3 col = f3(kBgPrimary, kBgPrimary, kBgPrimary);      // renderer.wgsl:410 -- the ONE camera-miss value of every mode (unpack_frames_kernel fills untraced tiles with it)
        if (hit) {                                           // shade(), renderer.wgsl:348-353
            const float ndotl = wmax(dot3(hit_normal<BRUTE>(A, tri, r, t), L), 0.0f);
            col = base * (0.15f + ndotl);
        }
        A.out[out_index] = make_float4(col.x, col.y, col.z, 1.0f);
        if (A.tri_ids) A.tri_ids[out_index] = tri;
    } else {
        const uint32_t pixel = py * A.width + px;
        F3 sum = f3(0.0f, 0.0f, 0.0f);
        for (uint32_t s = 0; s < A.spp; ++s) {
            const uint32_t key = sample_key(A.seed, pixel, A.frame * A.spp + s);
            Ray r = primary_ray(A, (float)px + rnd(key, 0, 0), (float)py + rnd(key, 0, 1));
            F3 rad = f3(0.0f, 0.0f, 0.0f), T = f3(1.0f, 1.0f, 1.0f);
            for (uint32_t bounce = 0;; ++bounce) {
                float t; uint32_t tri;
                ++n_closest;
                const bool hit = trace_any<false, STATS, BRUTE>(A, r, t, tri, stk, cnt);
                if (!hit) { rad = rad + T * ((bounce == 0u) ? kBgPrimary : kSkyAmbient); break; }
                const F3 n = hit_normal<BRUTE>(A, tri, r, t);
                const F3 hp = r.o + r.d * t;
                const F3 nf = (dot3(n, r.d) < 0.0f) ? n : f3(-n.x, -n.y, -n.z);
                const F3 so = hp + nf * kEpsOrigin;
                const float ndl = dot3(nf, L);
                if (ndl > 0.0f) {
                    Ray sr; sr.o = so; sr.d = L; sr.inv = safe_inv(L);
                    float st; uint32_t stri;
                    ++n_shadow;
                    if (!trace_any<true, STATS, BRUTE>(A, sr, st, stri, stk, cnt)) rad = rad + (T * base) * ndl;
                }
                if (bounce >= A.max_bounces) break;
                T = T * base;
                if (bounce >= kRRStart) {
                    const float p = wmax(wmax(T.x, T.y), T.z);
                    if (rnd(key, bounce, 4) >= p) break;
                    T = T * (1.0f / p);
                }
                r.d = cosine_dir(nf, rnd(key, bounce, 2), rnd(key, bounce, 3));
                r.o = so; r.inv = safe_inv(r.d);
            }
            sum = sum + rad;
        }
        float count = (float)A.spp;
        if (A.accum) {
            float4 acc = A.accumulate ? A.accum[out_index] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            sum = f3(acc.x + sum.x, acc.y + sum.y, acc.z + sum.z);
            count = acc.w + count;
            A.accum[out_index] = make_float4(sum.x, sum.y, sum.z, count);
        }
        const float inv = 1.0f / count;
        A.out[out_index] = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, 1.0f);
    }
    if (STATS) {
        atomicAdd(&A.stats[0], (unsigned long long)n_closest);
        atomicAdd(&A.stats[1], (unsigned long long)n_shadow);
        atomicAdd(&A.stats[2], (unsigned long long)cnt.nodes);
        atomicAdd(&A.stats[3], (unsigned long long)cnt.tris);
        atomicAdd(&A.stats[4], (unsigned long long)cnt.drops);
        atomicMax(&A.stats[5], (unsigned long long)cnt.maxstack);
        atomicAdd(&A.stats[6], (unsigned long long)(MODE == PT_KMODE_REFERENCE ? 1u : A.spp));
    }
}

// ------------------------------------------------------------------------------------
// The reference's own traversal, literally: traverseBVH4Packet (renderer.wgsl:210-346) with its 2x2 ray packets, the shared
// 64-entry stack with a 4-lane mask per entry, the double node fetch, the nearest-child swap and the silent drop -- laid out for
// the wavefront machine instead of one thread per packet: the FOUR LANES OF A QUAD are the four rays of a packet (16 packets per
// wavefront).  What the WGSL keeps per thread in private arrays becomes
//   * the packet's stack: 64 entries of (node, lane mask) in LDS, one column per packet, bank-conflict free for any mix of depths
//     (entry i of packet q at (i * 64 + q) * 8; the quad's four lanes read the same address);
//   * a lane mask: the quad's nibble of a wavefront ballot;
//   * "min over the packet's hit lanes": two v_min_f32 with DPP quad permutes;
//   * a node fetch: one 32 B record read by the four lanes at the same address (one L1 request for the quad), the four child
//     records of an internal node requested together.
// Everything that steers the traversal (popped entry, masks, child distances, order, pushes) is identical in the four lanes of a
// quad by construction, so a quad never diverges; packets of a wavefront do, like rays in the one-ray kernels.
// Reference layouts throughout (BVH u32[1 + 8 M], triangles f32[9 N]): this is the fidelity mode, PT_MODE_REFERENCE_PACKET.
// ------------------------------------------------------------------------------------
struct PNode { F3 mn, mx; uint32_t c[4]; uint32_t tri; bool leaf; };
__device__ __forceinline__ PNode load_ref_node(const uint32_t* __restrict__ bvh, uint32_t i) {   // renderer.wgsl:91-111
    const uint32_t* p = bvh + 1 + (size_t)i * 8;
    PNode n;
    const uint32_t a = p[0], b = p[1], c = p[2];
    n.mn = f3(half_lo(a), half_hi(a), half_lo(b));
    n.mx = f3(half_hi(b), half_lo(c), half_hi(c));
    n.c[0] = p[3]; n.c[1] = p[4]; n.c[2] = p[5]; n.c[3] = p[6];
    n.leaf = (p[7] & kLeaf) != 0u; n.tri = p[7] & 0x7fffffffu;
    return n;
}
__device__ __forceinline__ bool degenerate_box(const PNode& n) { return n.mn.x > n.mx.x || n.mn.y > n.mx.y || n.mn.z > n.mx.z; }
// this lane's part of intersectAABBPacketMask (renderer.wgsl:147-159): hit flag and tmin of its own ray
__device__ __forceinline__ bool lane_aabb(const Ray& r, F3 mn, F3 mx, float best, float& tmin_out) {
    const F3 t1 = (mn - r.o) * r.inv, t2 = (mx - r.o) * r.inv;
    const float tmin = wmax(wmax(wmin(t1.x, t2.x), wmin(t1.y, t2.y)), wmin(t1.z, t2.z));
    const float tmax = wmin(wmin(wmax(t1.x, t2.x), wmax(t1.y, t2.y)), wmax(t1.z, t2.z));
    tmin_out = tmin;
    return (tmax >= wmax(tmin, 0.0f)) && (tmin < best);
}
// the quad's nibble of a wavefront mask (bit k = lane k of this lane's quad)
__device__ __forceinline__ uint32_t quad_nibble(unsigned long long m, uint32_t lane) { return (uint32_t)(m >> (lane & 60u)) & 15u; }
// minimum over the four lanes of a quad (every lane gets it): quad_perm [1,0,3,2] then [2,3,0,1]
__device__ __forceinline__ float quad_min(float v) {
    float o = __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0xB1, 0xf, 0xf, true));
    v = wmin(v, o);
    o = __uint_as_float((uint32_t)__builtin_amdgcn_mov_dpp((int)__float_as_uint(v), 0x4E, 0xf, 0xf, true));
    return wmin(v, o);
}

// STATS: the oracle's packet-mode counters, counted once per packet (by the quad's first lane): node records examined (the root + every
// valid child slot), getBVHNode4 calls as the reference makes them (one per pop + one per valid child: the double fetch), leaves whose
// triangle was tested, dropped pushes, deepest stack.
template <bool STATS>
__global__ __launch_bounds__(256) void render_packet_kernel(const RenderArgs A) {
    __shared__ uint2 pstack[kStackMax][64];                  // [entry][packet of the block]: (node index, lane mask)
    const uint32_t lane = threadIdx.x & 63u, sub = threadIdx.x & 3u;
    const uint32_t pk = threadIdx.x >> 2;                    // packet of the block
    uint32_t px, py; size_t out_index; bool owned = true;
    if (A.compact != 0u) {
        // a tile share (pixel-tile sharding): an 8 x 8 tile is 4 x 4 whole packets (packets start at even pixels,
        // renderer.wgsl:359), so every packet of the image belongs to exactly one rank; a block takes four owned tiles, the output is
        // the compact tile-major buffer
        const uint32_t slot = blockIdx.x * 4u + (pk >> 4), pin = pk & 15u;
        owned = slot < A.num_tiles;
        const uint32_t tile = owned ? (A.tiles ? A.tiles[slot] : slot) : 0u;
        const uint32_t tx = tile % A.tiles_x, ty = tile / A.tiles_x;
        px = tx * 8u + (pin & 3u) * 2u + (sub & 1u); py = ty * 8u + (pin >> 2) * 2u + (sub >> 1);
        out_index = (size_t)slot * 64u + (py & 7u) * 8u + (px & 7u);
    } else {
        const uint32_t gx = blockIdx.x * 8u + (pk & 7u), gy = blockIdx.y * 8u + (pk >> 3);       // 8 x 8 packets = 16 x 16 pixels
        px = gx * 2u + (sub & 1u); py = gy * 2u + (sub >> 1);
        out_index = (size_t)py * A.width + px;
    }
    const bool in_image = owned && px < A.width && py < A.height;     // renderer.wgsl:376-385: lanes outside the image stay inactive
    Ray r;
    if (in_image) r = primary_ray(A, (float)px + 0.5f, (float)py + 0.5f);
    else { r.o = f3(0, 0, 0); r.d = f3(0, 0, -1.0f); r.inv = f3(kInfT, kInfT, kInfT); }
    float best = kInfT; uint32_t btri = kInvalidRef; F3 bn = f3(0, 0, 0);
    const uint32_t lanes0 = quad_nibble(__ballot(in_image), lane);          // the packet's initial lane mask
    const uint32_t num_nodes = A.bvh4_ref[0];
    int sp = -1;
    uint32_t c_nodes = 0, c_fetch = 0, c_tris = 0, c_drops = 0, c_maxstack = 0;
    if (num_nodes != 0u && A.num_tris != 0u && lanes0 != 0u) {     // renderer.wgsl:224-233
        sp = 0; if (sub == 0u) pstack[0][pk] = make_uint2(0u, lanes0);
        if (STATS) { c_nodes = 1; c_maxstack = 1; }
    }
    // (one wavefront holds whole quads, so a packet's LDS accesses are ordered by its own wavefront: no barrier is needed)
    while (__ballot(sp >= 0) != 0ull) {
        const bool live = sp >= 0;
        uint32_t ni = 0u, lm = 0u;
        if (live) { const uint2 e = pstack[sp][pk]; ni = e.x; lm = e.y; --sp; }
        PNode node;
        if (live) node = load_ref_node(A.bvh4_ref, ni);
        if (STATS && live) ++c_fetch;
        bool go = live && !degenerate_box(node);             // renderer.wgsl:244-246
        float tmin = kInfT;
        const bool hit = go && ((lm >> sub) & 1u) != 0u && lane_aabb(r, node.mn, node.mx, best, tmin);
        const uint32_t hm = quad_nibble(__ballot(hit), lane);    // re-test at pop with the current best t (renderer.wgsl:248-252)
        go = go && hm != 0u;
        if (go && node.leaf) {
            if (node.tri < A.num_tris) {                     // renderer.wgsl:262-283
                if (STATS) ++c_tris;
                const float* tp = A.tris9 + (size_t)node.tri * 9;
                const F3 v0 = f3(tp[0], tp[1], tp[2]), v1 = f3(tp[3], tp[4], tp[5]), v2 = f3(tp[6], tp[7], tp[8]);
                const F3 e1 = v1 - v0, e2 = v2 - v0;
                const F3 tn = normalize3(cross3(e1, e2));
                if ((hm >> sub) & 1u) {                      // intersectTrianglePacket, this lane (renderer.wgsl:185-205)
                    const F3 p = cross3(r.d, e2);
                    const float det = dot3(e1, p);
                    if (!(fabsf(det) < kTriEps)) {
                        const float inv_det = 1.0f / det;
                        const F3 s = r.o - v0;
                        const float u = inv_det * dot3(s, p);
                        if (!(u < 0.0f || u > 1.0f)) {
                            const F3 q = cross3(s, e1);
                            const float v = inv_det * dot3(r.d, q);
                            if (!(v < 0.0f || (u + v) > 1.0f)) {
                                const float t = inv_det * dot3(e2, q);
                                if (t > kTriEps && t < best) { best = t; bn = tn; btri = node.tri; }
                            }
                        }
                    }
                }
            }
            go = false;
        }
        // internal node (renderer.wgsl:286-343): the four child records are requested together, tested by every lane of the mask
        uint32_t cidx[4]; float cdist[4]; uint32_t cmask[4]; int cc = 0;
        {
            PNode ch[4]; bool valid[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t ci = node.c[c];
                valid[c] = go && ci != kInvalidRef && ci < num_nodes;
                if (valid[c]) ch[c] = load_ref_node(A.bvh4_ref, ci);
                if (STATS && valid[c]) { ++c_fetch; ++c_nodes; }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool ok = valid[c] && !degenerate_box(ch[c]);
                float ct = kInfT;
                const bool chit = ok && ((hm >> sub) & 1u) != 0u && lane_aabb(r, ch[c].mn, ch[c].mx, best, ct);
                const uint32_t m = quad_nibble(__ballot(chit), lane);
                const float cm = quad_min(chit ? ct : kInfT);          // min over the packet's hit lanes (kInfT when none)
                if (ok && m != 0u) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (k == cc) { cidx[k] = node.c[c]; cdist[k] = cm; cmask[k] = m; }
                    ++cc;
                }
            }
        }
        if (go && cc > 0) {
            int bi = 0;
#pragma unroll
            for (int i = 1; i < 4; ++i) if (i < cc) bi = (cdist[i] < cdist[bi]) ? i : bi;
            if (bi != 0) {                                   // the nearest child trades places with slot 0 (renderer.wgsl:320-330)
#pragma unroll
                for (int k = 1; k < 4; ++k) if (k == bi) {
                    const uint32_t ti = cidx[0]; const float td = cdist[0]; const uint32_t tm = cmask[0];
                    cidx[0] = cidx[k]; cdist[0] = cdist[k]; cmask[0] = cmask[k];
                    cidx[k] = ti; cdist[k] = td; cmask[k] = tm;
                }
            }
#pragma unroll
            for (int i = 3; i >= 0; --i)                     // far -> near; a push beyond the 64 entries is dropped (renderer.wgsl:336-342)
                if (i < cc) {
                    if (sp + 1 < kStackMax) { ++sp; if (sub == 0u) pstack[sp][pk] = make_uint2(cidx[i], cmask[i]); }
                    else if (STATS) ++c_drops;
                }
            if (STATS && (uint32_t)(sp + 1) > c_maxstack) c_maxstack = (uint32_t)(sp + 1);
        }
    }
    if (in_image) {                                          // renderer.wgsl:401-412
        const F3 base = f3(0.9f, 0.7f, 0.3f);
        const F3 L = light_dir();
        F3 col = f3(kBgPrimary, kBgPrimary, kBgPrimary);      // renderer.wgsl:410 -- the ONE camera-miss value of every mode (unpack_frames_kernel fills untraced tiles with it)
        if (btri != kInvalidRef) col = base * (0.15f + wmax(dot3(bn, L), 0.0f));
        A.out[out_index] = make_float4(col.x, col.y, col.z, 1.0f);
        if (A.tri_ids) A.tri_ids[out_index] = btri;
    }
    if (STATS) {
        if (in_image) { atomicAdd(&A.stats[0], 1ull); atomicAdd(&A.stats[6], 1ull); }      // one closest ray / sample per pixel of the image
        if (sub == 0u) {                                                                   // per packet
            atomicAdd(&A.stats[2], (unsigned long long)c_nodes); atomicAdd(&A.stats[3], (unsigned long long)c_tris);
            atomicAdd(&A.stats[4], (unsigned long long)c_drops); atomicMax(&A.stats[5], (unsigned long long)c_maxstack);
            atomicAdd(&A.stats[7], (unsigned long long)c_fetch);                           // node_fetches_ref: the reference's double fetch
        }
    }
}

// ------------------------------------------------------------------------------------
// LBVH2 build (BVHBuilder.wgsl).  f32 -> f16 is round-to-nearest-even (v_cvt_f16_f32).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t f16_bits_rtne(float v) { return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)v); }
__device__ __forceinline__ uint32_t step_f16(float v, bool up) {        // BVHBuilder.wgsl:63-81, returns f16 bits
    const uint32_t bits = f16_bits_rtne(v);
    uint32_t ord = (bits & 0x8000u) ? ((~bits) & 0xFFFFu) : (bits ^ 0x8000u);
    ord = up ? ord + 1u : ord - 1u;
    return ((ord & 0x8000u) ? (ord ^ 0x8000u) : ((~ord) & 0xFFFFu)) & 0xFFFFu;
}
__device__ __forceinline__ void store_bounds2(uint32_t* bvh2, uint32_t node, F3 mn, F3 mx) {   // BVHBuilder.wgsl:83-102
    uint32_t* p = bvh2 + 1 + (size_t)node * 6;
    const uint32_t w0 = step_f16(mn.x, false) | (step_f16(mn.y, false) << 16);
    const uint32_t w1 = step_f16(mn.z, false) | (step_f16(mx.x, true) << 16);
    const uint32_t w2 = step_f16(mx.y, true) | (step_f16(mx.z, true) << 16);
    __hip_atomic_store(p + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int lbvh_delta(const uint32_t* __restrict__ morton, int i, int j, int n) {   // BVHBuilder.wgsl:134-149
    if (j < 0 || j >= n) return -1;
    const uint32_t x = morton[i] ^ morton[j];
    if (x == 0u) return 32 + __clz((int)((uint32_t)i ^ (uint32_t)j));
    return __clz((int)x);
}

__global__ __launch_bounds__(256) void lbvh2_internal_kernel(uint32_t* __restrict__ bvh2, const uint32_t* __restrict__ morton,
                                                              uint32_t* __restrict__ parent, uint32_t* __restrict__ flags, uint32_t num_tris) {
    const uint32_t iu = blockIdx.x * blockDim.x + threadIdx.x;     // BVHBuilder.wgsl:152-240
    if (num_tris <= 1u || iu >= num_tris - 1u) return;
    const int n = (int)num_tris, i = (int)iu;
    flags[iu] = 0u;
    const int d = (lbvh_delta(morton, i, i + 1, n) - lbvh_delta(morton, i, i - 1, n)) > 0 ? 1 : -1;
    const int dmin = lbvh_delta(morton, i, i - d, n);
    int lmax = 2;
    while (lbvh_delta(morton, i, i + lmax * d, n) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t > 0; t >>= 1)
        if (lbvh_delta(morton, i, i + (l + t) * d, n) > dmin) l += t;
    const int j = i + l * d;
    const int first = min(i, j), last = max(i, j);
    const int dnode = lbvh_delta(morton, first, last, n);
    int split = first, step = last - first;
    while (step > 1) {
        step = (step + 1) >> 1;
        const int ns = split + step;
        if (ns < last && lbvh_delta(morton, first, ns, n) > dnode) split = ns;
    }
    const uint32_t leaf_base = num_tris - 1u;
    const uint32_t lc = (split == first) ? leaf_base + (uint32_t)split : (uint32_t)split;
    const uint32_t rc = (split + 1 == last) ? leaf_base + (uint32_t)(split + 1) : (uint32_t)(split + 1);
    uint32_t* p = bvh2 + 1 + (size_t)iu * 6;
    p[3] = lc; p[4] = rc; p[5] = 0u;
    parent[lc] = iu; parent[rc] = iu;
    if (iu == 0u) parent[0] = kInvalidRef;
}

// REFIT = false: only the leaf records (all the BVH4 collapse needs: it re-unions bounds from the leaves up);
// REFIT = true : leaf records and the bottom-up walk that gives the internal BVH2 nodes their bounds;
// LEAVES = false with REFIT: the walk alone, over leaf records written by an earlier launch (pt_read_bvh2 after pt_build_bvh).
template <bool LEAVES, bool REFIT>
__global__ __launch_bounds__(256) void lbvh2_leaves_kernel(uint32_t* bvh2, const float* __restrict__ tris, const uint32_t* __restrict__ tri_index,
                                                            const uint32_t* __restrict__ parent, uint32_t* flags, uint32_t num_tris) {
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;   // BVHBuilder.wgsl:278-306
    if (leaf >= num_tris) return;
    const uint32_t internal = num_tris - 1u;
    const uint32_t node = internal + leaf;
    if (LEAVES) {
        const uint32_t ti = tri_index[leaf];
        const float* tp = tris + (size_t)ti * 9;
        const F3 v0 = f3(tp[0], tp[1], tp[2]), v1 = f3(tp[3], tp[4], tp[5]), v2 = f3(tp[6], tp[7], tp[8]);
        const F3 mn = f3(wmin(v0.x, wmin(v1.x, v2.x)), wmin(v0.y, wmin(v1.y, v2.y)), wmin(v0.z, wmin(v1.z, v2.z)));
        const F3 mx = f3(wmax(v0.x, wmax(v1.x, v2.x)), wmax(v0.y, wmax(v1.y, v2.y)), wmax(v0.z, wmax(v1.z, v2.z)));
        store_bounds2(bvh2, node, mn, mx);
        uint32_t* p = bvh2 + 1 + (size_t)node * 6;
        p[3] = 0u; p[4] = 0u; p[5] = kLeaf | (ti & 0x7fffffffu);
    }
    if (!REFIT || internal == 0u) return;
    // bottom-up refit (BVHBuilder.wgsl:242-275): the second thread to arrive at a node unions its children's bounds.  The
    // reference orders nothing between a child's bounds store and the arrival count.  Here every bounds word is written and
    // read with agent-scope accesses (write-through stores, L1-bypassing loads: the 8 per-XCD L2s are not coherent with each
    // other for plain accesses), a thread's stores have left the CU (s_waitcnt vmcnt(0)) before it counts itself in, and the
    // reads of the sibling's bounds depend on the returned count -- no cache write-back / invalidate per step (two
    // __threadfence() per step made this walk 4x as long as the whole rest of the build).
    uint32_t cur = node;
    for (;;) {
        const uint32_t par = parent[cur];
        if (par == kInvalidRef || par >= internal) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(&flags[par], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0u) break;
        const uint32_t* pp = bvh2 + 1 + (size_t)par * 6;
        const uint32_t l = pp[3], r = pp[4];
        const uint32_t* lp = bvh2 + 1 + (size_t)l * 6; const uint32_t* rp = bvh2 + 1 + (size_t)r * 6;
        const uint32_t l0 = __hip_atomic_load(lp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), l1 = __hip_atomic_load(lp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), l2 = __hip_atomic_load(lp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t r0 = __hip_atomic_load(rp + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), r1 = __hip_atomic_load(rp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), r2 = __hip_atomic_load(rp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const F3 umn = f3(wmin(half_lo(l0), half_lo(r0)), wmin(half_hi(l0), half_hi(r0)), wmin(half_lo(l1), half_lo(r1)));
        const F3 umx = f3(wmax(half_hi(l1), half_hi(r1)), wmax(half_lo(l2), half_lo(r2)), wmax(half_hi(l2), half_hi(r2)));
        store_bounds2(bvh2, par, umn, umx);
        cur = par;
    }
}

// ------------------------------------------------------------------------------------
// post passes
// ------------------------------------------------------------------------------------
// blockIdx.y = frame of the batch: every frame of a gathered batch is scattered by ONE launch
__global__ __launch_bounds__(256) void deinterleave_kernel(const float4* __restrict__ gathered, uint64_t stride_px, uint64_t frame_stride_px, float4* __restrict__ full,
                                                           uint64_t full_stride_px, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t tiles_y, uint32_t count) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tile = item >> 6, lane = item & 63u;
    if (tile >= tiles_x * tiles_y) return;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t px = tx * 8u + (lane & 7u), py = ty * 8u + (lane >> 3);
    if (px >= width || py >= height) return;
    const uint32_t rank = (tx + ty) % count;
    // slot of (tx,ty) in its rank's row-major list.  Row r holds the rank's tiles at
    // tx = first(r), first(r)+count, ... with first(r) = (rank - r) mod count; over `count`
    // consecutive rows first() takes every residue once, so such a block holds tiles_x tiles.
    uint32_t slot = (ty / count) * tiles_x;
    for (uint32_t r = ty - ty % count; r < ty; ++r) {
        const uint32_t f = (rank + count - (r % count)) % count;
        slot += (f < tiles_x) ? (tiles_x - f + count - 1) / count : 0u;
    }
    const uint32_t first = (rank + count - (ty % count)) % count;
    slot += (tx - first) / count;
    full[(size_t)blockIdx.y * full_stride_px + (size_t)py * width + px] = gathered[(size_t)rank * stride_px + (size_t)blockIdx.y * frame_stride_px + (size_t)slot * 64 + lane];
}

// ---- packed tile shares: what a sharded frame ships to rank 0 ---------------------------------------------------------------------
// A rank's compact buffer holds every owned tile at 16 bytes per pixel.  Two thirds of a dragon-class frame's tiles lie outside the
// rectangle of tiles in which a camera ray can reach the scene at all (pt_traced_tile_rect): their pixels are the camera-miss value, a
// constant; and alpha is 1 everywhere.  So only the owned tiles INSIDE the rectangle travel, as 12 bytes per pixel: rank r's share of a
// frame is its tiles inside the rectangle in row-major order, 64 x 3 floats each.  Tile (tx, ty) belongs to rank (tx + ty) % count.
struct TileRectArg { uint32_t tx0, ty0, tx1, ty1; };
// slot of tile (tx, ty) among its rank's tiles of the half-open tile range [x0, x1) x [y0, ..): over `count` consecutive rows a rank's first
// column takes every residue once, so such a block of rows holds (x1 - x0) of its tiles
__device__ __forceinline__ uint32_t share_slot(uint32_t tx, uint32_t ty, uint32_t rank, uint32_t count, uint32_t x0, uint32_t x1, uint32_t y0) {
    const uint32_t rows = ty - y0;
    uint32_t slot = (rows / count) * (x1 - x0);
    auto first_col = [&](uint32_t row) { return x0 + (rank + count - ((row + x0) % count)) % count; };       // the rank's first tile column >= x0 in that row
    for (uint32_t r = ty - rows % count; r < ty; ++r) {
        const uint32_t f = first_col(r);
        slot += f < x1 ? (x1 - f + count - 1u) / count : 0u;
    }
    return slot + (tx - first_col(ty)) / count;
}

// blockIdx.y = frame; one wavefront per tile of the rectangle, the tiles of other ranks are skipped
__global__ __launch_bounds__(256) void pack_shares_kernel(const float4* __restrict__ compact, uint64_t frame_stride_px, float* __restrict__ packed, uint64_t packed_stride_floats,
                                                          uint32_t tiles_x, uint32_t rank, uint32_t count, TileRectArg rc) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = item >> 6, lane = item & 63u;
    const uint32_t rw = rc.tx1 - rc.tx0;
    if (t >= rw * (rc.ty1 - rc.ty0)) return;
    const uint32_t tx = rc.tx0 + t % rw, ty = rc.ty0 + t / rw;
    if ((tx + ty) % count != rank) return;
    const uint32_t owned = share_slot(tx, ty, rank, count, 0u, tiles_x, 0u), inside = share_slot(tx, ty, rank, count, rc.tx0, rc.tx1, rc.ty0);
    const float4 v = compact[(size_t)blockIdx.y * frame_stride_px + (size_t)owned * 64u + lane];
    float* o = packed + (size_t)blockIdx.y * packed_stride_floats + ((size_t)inside * 64u + lane) * 3u;
    o[0] = v.x; o[1] = v.y; o[2] = v.z;
}

// rank 0: blockIdx.y = frame; every pixel of the row-major frame from the gathered packed shares, or -- outside the rectangle -- the
// camera-miss mean: `spp` additions of the miss value and the multiplication by 1 / spp, as resolve_kernel forms it (the same bits)
__global__ __launch_bounds__(256) void unpack_frames_kernel(const float* __restrict__ gathered, uint64_t rank_stride_floats, uint64_t frame_stride_floats, float4* __restrict__ full,
                                                            uint64_t full_stride_px, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t tiles_y, uint32_t count,
                                                            TileRectArg rc, uint32_t spp) {
    const uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t tile = item >> 6, lane = item & 63u;
    if (tile >= tiles_x * tiles_y) return;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const uint32_t px = tx * 8u + (lane & 7u), py = ty * 8u + (lane >> 3);
    if (px >= width || py >= height) return;
    float4 v;
    if (tx >= rc.tx0 && tx < rc.tx1 && ty >= rc.ty0 && ty < rc.ty1) {
        const uint32_t rank = (tx + ty) % count;
        const float* src = gathered + (size_t)rank * rank_stride_floats + (size_t)blockIdx.y * frame_stride_floats + ((size_t)share_slot(tx, ty, rank, count, rc.tx0, rc.tx1, rc.ty0) * 64u + lane) * 3u;
        v = make_float4(src[0], src[1], src[2], 1.0f);
    } else {
        const float bg = 0.0f + 1.0f * kBgPrimary;
        float sum = 0.0f;
        for (uint32_t s = 0; s < spp; ++s) sum = sum + bg;
        const float m = sum * (1.0f / (float)spp);
        v = make_float4(m, m, m, 1.0f);
    }
    full[(size_t)blockIdx.y * full_stride_px + (size_t)py * width + px] = v;
}

__global__ __launch_bounds__(256) void rgba8_kernel(const float4* __restrict__ src, uint32_t* __restrict__ dst, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = src[i];
    auto q = [](float v) { v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); return (uint32_t)floorf(v * 255.0f + 0.5f); };
    dst[i] = q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | (q(c.w) << 24);
}

// x^(1/2.2) for x in [0, 1] as ONE pinned f32 evaluation (WGSL leaves pow's precision to the implementation, and libm's powf differs from the
// device's by an LSB of the 8-bit result): x = m 2^e with m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh(s), s = (m - 1) / (m + 1), as an odd
// polynomial in s evaluated with fmaf; z = (e + ln m / ln 2) / 2.2; 2^z = 2^k 2^r with k = floor(z), r in [0, 1): a degree-7 polynomial with
// fmaf, the scale by exponent arithmetic.  Only +, -, *, /, fmaf, floorf and integer operations: bit-identical to oracle/pt_oracle.cpp::pow_1_2_2.
__device__ __forceinline__ float pow_1_2_2(float x) {
    if (!(x > 1.17549435e-38f)) return 0.0f;                       // zero, negatives, subnormals, NaN
    if (x >= 1.0f) return 1.0f;                                     // the Reinhard curve stays below 1
    uint32_t bits = __float_as_uint(x);
    int e = (int)(bits >> 23) - 127;
    float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);  // [1, 2)
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }                  // [sqrt(1/2), sqrt(2))
    const float s = (m - 1.0f) / (m + 1.0f), s2 = s * s;
    float p = __builtin_fmaf(s2, 0.11111111f, 0.14285715f);
    p = __builtin_fmaf(s2, p, 0.2f);
    p = __builtin_fmaf(s2, p, 0.33333334f);
    p = __builtin_fmaf(s2, p, 1.0f);
    const float ln_m = 2.0f * s * p;
    const float z = ((float)e + ln_m * 1.44269504f) * 0.45454547f;  // log2(x) / 2.2
    const float kf = floorf(z), r = z - kf;
    float q = __builtin_fmaf(r, 1.5252734e-5f, 1.5403530e-4f);      // 2^r = sum (r ln 2)^n / n!
    q = __builtin_fmaf(r, q, 1.3333558e-3f);
    q = __builtin_fmaf(r, q, 9.6181291e-3f);
    q = __builtin_fmaf(r, q, 5.5504109e-2f);
    q = __builtin_fmaf(r, q, 2.4022651e-1f);
    q = __builtin_fmaf(r, q, 6.9314718e-1f);
    q = __builtin_fmaf(r, q, 1.0f);
    const int k = (int)kf;
    if (k < -126) return 0.0f;
    return q * __uint_as_float((uint32_t)(k + 127) << 23);
}

// tonemapper.wgsl:24-41 (Reinhard, gamma 1/2.2) with the vertical flip of the full-screen pass
__global__ __launch_bounds__(256) void tonemap_kernel(const float4* __restrict__ src, uint32_t* __restrict__ dst, uint32_t width, uint32_t height, int from_rgba8) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= width * height) return;
    const uint32_t x = i % width, y = i / width;
    const float4 c = src[(size_t)(height - 1u - y) * width + x];
    auto tm = [from_rgba8](float v) {
        if (from_rgba8) { v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); v = floorf(v * 255.0f + 0.5f) / 255.0f; }
        const float m = v / (v + 1.0f);
        float g = pow_1_2_2(m);
        g = g < 0.0f ? 0.0f : (g > 1.0f ? 1.0f : g);
        return (uint32_t)floorf(g * 255.0f + 0.5f);
    };
    dst[i] = tm(c.x) | (tm(c.y) << 8) | (tm(c.z) << 16) | (255u << 24);
}

// ------------------------------------------------------------------------------------
// launchers (called from pt_api.cpp)
// ------------------------------------------------------------------------------------
hipError_t launch_render(const RenderArgs& A, int kmode, bool stats, hipStream_t stream) {
    if (kmode == PT_KMODE_PACKET) {
        // 256 lanes = 64 packets of 2 x 2 pixels: 16 x 16 pixels of a whole frame, four owned 8 x 8 tiles of a tile share
        const dim3 grid = A.compact ? dim3((A.num_tiles + 3u) / 4u, 1) : dim3((A.width + 15) / 16, (A.height + 15) / 16);
        if (grid.x == 0) return hipSuccess;
        if (stats) hipLaunchKernelGGL(render_packet_kernel<true>, grid, dim3(256), 0, stream, A);
        else       hipLaunchKernelGGL(render_packet_kernel<false>, grid, dim3(256), 0, stream, A);
        return hipGetLastError();
    }
    const uint32_t items = A.num_tiles * 64u;
    const dim3 grid((items + 255u) / 256u), block(256);
    if (grid.x == 0) return hipSuccess;
    const bool brute = A.brute != 0u;
#define PT_LAUNCH(M, S, B) hipLaunchKernelGGL((render_rays_kernel<M, S, B>), grid, block, 0, stream, A)
    if (kmode == PT_KMODE_REFERENCE) {
        if (brute) { if (stats) PT_LAUNCH(PT_KMODE_REFERENCE, true, true); else PT_LAUNCH(PT_KMODE_REFERENCE, false, true); }
        else       { if (stats) PT_LAUNCH(PT_KMODE_REFERENCE, true, false); else PT_LAUNCH(PT_KMODE_REFERENCE, false, false); }
    } else {
        if (brute) { if (stats) PT_LAUNCH(PT_KMODE_PATH, true, true); else PT_LAUNCH(PT_KMODE_PATH, false, true); }
        else       { if (stats) PT_LAUNCH(PT_KMODE_PATH, true, false); else PT_LAUNCH(PT_KMODE_PATH, false, false); }
    }
#undef PT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_lbvh2(uint32_t* bvh2, const float* tris9, const uint32_t* morton, const uint32_t* tri_index,
                        uint32_t* parent, uint32_t* flags, uint32_t num_tris, bool refit, hipStream_t stream) {
    if (num_tris == 0) return hipSuccess;
    if (num_tris > 1) {
        hipLaunchKernelGGL(lbvh2_internal_kernel, dim3((num_tris - 1 + 255) / 256), dim3(256), 0, stream, bvh2, morton, parent, flags, num_tris);
        hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    }
    if (refit) hipLaunchKernelGGL((lbvh2_leaves_kernel<true, true>), dim3((num_tris + 255) / 256), dim3(256), 0, stream, bvh2, tris9, tri_index, parent, flags, num_tris);
    else       hipLaunchKernelGGL((lbvh2_leaves_kernel<true, false>), dim3((num_tris + 255) / 256), dim3(256), 0, stream, bvh2, tris9, tri_index, parent, flags, num_tris);
    return hipGetLastError();
}

// the bottom-up walk alone (the arrival flags are still zero from lbvh2_internal_kernel)
hipError_t launch_lbvh2_refit(uint32_t* bvh2, const uint32_t* parent, uint32_t* flags, uint32_t num_tris, hipStream_t stream) {
    if (num_tris <= 1) return hipSuccess;
    hipLaunchKernelGGL((lbvh2_leaves_kernel<false, true>), dim3((num_tris + 255) / 256), dim3(256), 0, stream, bvh2, nullptr, nullptr, parent, flags, num_tris);
    return hipGetLastError();
}

hipError_t launch_deinterleave(const float4* gathered, uint64_t rank_stride_px, uint64_t frame_stride_px, uint32_t frames, float4* full, uint64_t full_stride_px,
                               uint32_t width, uint32_t height, uint32_t count, hipStream_t stream) {
    const uint32_t tx = (width + 7) / 8, ty = (height + 7) / 8;
    const uint32_t items = tx * ty * 64u;
    if (frames == 0u) return hipSuccess;
    hipLaunchKernelGGL(deinterleave_kernel, dim3((items + 255) / 256, frames), dim3(256), 0, stream, gathered, rank_stride_px, frame_stride_px, full, full_stride_px, width, height, tx, ty, count);
    return hipGetLastError();
}

hipError_t launch_pack_shares(const float4* compact, uint64_t frame_stride_px, uint32_t frames, float* packed, uint64_t packed_stride_floats, uint32_t width,
                              uint32_t rank, uint32_t count, const uint32_t rect[4], hipStream_t stream) {
    const TileRectArg rc = {rect[0], rect[1], rect[2], rect[3]};
    const uint32_t items = (rc.tx1 - rc.tx0) * (rc.ty1 - rc.ty0) * 64u;
    if (frames == 0u || items == 0u) return hipSuccess;
    hipLaunchKernelGGL(pack_shares_kernel, dim3((items + 255) / 256, frames), dim3(256), 0, stream, compact, frame_stride_px, packed, packed_stride_floats, (width + 7) / 8, rank, count, rc);
    return hipGetLastError();
}

hipError_t launch_unpack_frames(const float* gathered, uint64_t rank_stride_floats, uint64_t frame_stride_floats, uint32_t frames, float4* full, uint64_t full_stride_px,
                                uint32_t width, uint32_t height, uint32_t count, const uint32_t rect[4], uint32_t spp, hipStream_t stream) {
    const uint32_t tx = (width + 7) / 8, ty = (height + 7) / 8;
    const TileRectArg rc = {rect[0], rect[1], rect[2], rect[3]};
    if (frames == 0u) return hipSuccess;
    hipLaunchKernelGGL(unpack_frames_kernel, dim3((tx * ty * 64u + 255) / 256, frames), dim3(256), 0, stream, gathered, rank_stride_floats, frame_stride_floats, full, full_stride_px,
                       width, height, tx, ty, count, rc, spp);
    return hipGetLastError();
}

hipError_t launch_rgba8(const float4* src, uint32_t* dst, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(rgba8_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_tonemap(const float4* src, uint32_t* dst, uint32_t width, uint32_t height, int from_rgba8, hipStream_t stream) {
    hipLaunchKernelGGL(tonemap_kernel, dim3((width * height + 255) / 256), dim3(256), 0, stream, src, dst, width, height, from_rgba8);
    return hipGetLastError();
}

} // namespace ptk
