// pt_device.h -- device-side arithmetic shared by the HIP kernels (pt_kernels.hip,
// pt_megakernel.hip): explicit-order f32 vector helpers, the reference's ray setup and slab
// test, and the build-defined sampling functions of DESIGN.md section 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_kernels.h"

namespace ptk {

// ------------------------------------------------------------------------------------
// small vector helpers (explicit operation order, never contracted)
// ------------------------------------------------------------------------------------
struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot3(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ F3 cross3(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// ---- short forms of the correctly rounded reciprocal and square root --------------------------------------------------------
// 1.0f / d for 2^-64 <= |d| < 2^64 in THREE instructions instead of the compiler's eleven: v_rcp_f32 and ONE Newton step with the exact residual,
// r + r * fma(-d, r, 1).  The compiler's IEEE sequence (v_div_scale x2, v_rcp, four fma, v_mul, v_div_fmas, v_div_fixup) scales operands near the ends of
// the exponent range and patches zero / infinity / NaN -- nothing an operand of ordinary size needs -- and then refines twice more than gfx950's v_rcp_f32
// (1 ulp) requires: the residual of a reciprocal cannot come closer to a rounding boundary than one step resolves.  That is not taken on trust:
// tools/probes/rcp_exact.hip compares this form with `1.0f / x` over EVERY f32 bit pattern of the range on the GPU (2,164,260,864 patterns, 0 differences,
// profiles/r05_d3_rcp3_exact.txt) -- so it IS the IEEE quotient there, bit for bit, on this hardware (the library is built for gfx950 only).
// sqrtf(x) for x == 0 or 2^-64 <= x < 2^64 in 9 instructions instead of 16: the compiler's sequence without its scaling of small operands and its
// zero / infinity fix-up -- v_sqrt, then the neighbours one ulp down and up are tried against the exact residuals (same probe, 0 differences).
// Outside those ranges the general forms are used: the megakernel is compiled in two variants (BOUNDED), and the host picks the short one only where it
// can bound every operand (pt_api.cpp::arith_is_bounded).
__device__ __forceinline__ float rcp_normal(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float sqrt_normal(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = (r_dn <= 0.0f) ? s_dn : s;
    return (r_up > 0.0f) ? s_up : s;
}
template <bool B> __device__ __forceinline__ float rcp_f(float d) { if constexpr (B) return rcp_normal(d); else return 1.0f / d; }
template <bool B> __device__ __forceinline__ float sqrt_f(float x) { if constexpr (B) return sqrt_normal(x); else return sqrtf(x); }

template <bool B = false> __device__ __forceinline__ F3 normalize3(F3 v) { const float inv = rcp_f<B>(sqrt_f<B>(dot3(v, v))); return v * inv; }
// WGSL min/max on non-NaN data (sign of zero never reaches a comparison result)
// -> v_min_f32 / v_max_f32 / v_min3_f32 / v_max3_f32
// v_min_f32 / v_max_f32: -0 orders below +0 (the ISA's pseudo-code names the two zero cases).  In the slab test the sign of a zero never reaches a
// comparison; in the LBVH2 builder it does (incrementF16 steps -0 and +0 differently), and the oracle's builder pins the same order (min_oz / max_oz).
__device__ __forceinline__ float wmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float wmax(float a, float b) { return __builtin_fmaxf(a, b); }

__device__ __forceinline__ float half_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float half_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }

// float(low / high half of w) - o in one instruction
__device__ __forceinline__ float half_lo_minus(uint32_t w, float o) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(o)); return r; }
__device__ __forceinline__ float half_hi_minus(uint32_t w, float o) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(o)); return r; }

constexpr float kInfT = 1e30f;          // renderer.wgsl:64
constexpr float kTriEps = 1e-7f;        // renderer.wgsl:178
constexpr uint32_t kLeaf = 0x80000000u;
constexpr uint32_t kInvalidRef = 0xFFFFFFFFu;
constexpr uint32_t kDegenerateRef = 0xFFFFFFFEu;   // child slot whose record the reference fetches and rejects (renderer.wgsl:289-291): counted, never entered
constexpr int kStackMax = 64;           // renderer.wgsl:8

struct Ray { F3 o, d, inv; };

template <bool B = false> __device__ __forceinline__ F3 safe_inv(F3 d) {      // renderer.wgsl:74-80
    return f3(fabsf(d.x) > 1e-8f ? rcp_f<B>(d.x) : kInfT,
              fabsf(d.y) > 1e-8f ? rcp_f<B>(d.y) : kInfT,
              fabsf(d.z) > 1e-8f ? rcp_f<B>(d.z) : kInfT);
}

__device__ __forceinline__ F3 rotate_quat(F3 v, const float* q) {   // renderer.wgsl:66-72
    const F3 u = f3(q[0], q[1], q[2]); const float s = q[3];
    const F3 uv = cross3(u, v), uuv = cross3(u, uv);
    return f3(__builtin_fmaf(2.0f, __builtin_fmaf(s, uv.x, uuv.x), v.x),
              __builtin_fmaf(2.0f, __builtin_fmaf(s, uv.y, uuv.y), v.y),
              __builtin_fmaf(2.0f, __builtin_fmaf(s, uv.z, uuv.z), v.z));
}

__device__ __forceinline__ Ray primary_ray(const RenderArgs& A, float fx, float fy) {   // renderer.wgsl:387-395
    const float uvx = fx / (float)A.width, uvy = fy / (float)A.height;
    const float px = __builtin_fmaf(uvx, 2.0f, -1.0f), py = __builtin_fmaf(uvy, 2.0f, -1.0f);
    Ray r;
    r.d = rotate_quat(normalize3(f3(px * A.aspect, py, -A.focal)), A.quat);
    r.o = f3(A.cam[0], A.cam[1], A.cam[2]);
    r.inv = safe_inv(r.d);
    return r;
}

template <bool B = false> __device__ __forceinline__ Ray primary_ray_fp(const RenderArgs& A, const FrameParams& fp, float fx, float fy) {   // renderer.wgsl:387-395
    const float uvx = fx / (float)A.width, uvy = fy / (float)A.height;
    const float px = __builtin_fmaf(uvx, 2.0f, -1.0f), py = __builtin_fmaf(uvy, 2.0f, -1.0f);
    Ray r;
    r.d = rotate_quat(normalize3<B>(f3(px * fp.aspect, py, -fp.focal)), fp.quat);
    r.o = f3(fp.cam[0], fp.cam[1], fp.cam[2]);
    r.inv = safe_inv<B>(r.d);
    return r;
}

// slab test of one packed f16 box (renderer.wgsl:147-159); returns hit, writes tmin
__device__ __forceinline__ bool slab(const Ray& r, uint32_t w0, uint32_t w1, uint32_t w2, float best, float& tmin_out) {
    // (bound - origin) straight from the packed halves: v_fma_mix_f32 widens an f16 source on the fly, and fma(h, 1.0, -o) is the
    // f32 subtraction (one rounding of the exact difference) -- bit-identical to v_cvt_f32_f16 + v_sub_f32 for every half value
    // (tools/probes/fma_mix.hip), one instruction instead of two, 24 fewer per node
    const F3 t1 = f3(half_lo_minus(w0, r.o.x), half_hi_minus(w0, r.o.y), half_lo_minus(w1, r.o.z)) * r.inv;
    const F3 t2 = f3(half_hi_minus(w1, r.o.x), half_lo_minus(w2, r.o.y), half_hi_minus(w2, r.o.z)) * r.inv;
    const float tmin = wmax(wmax(wmin(t1.x, t2.x), wmin(t1.y, t2.y)), wmin(t1.z, t2.z));
    const float tmax = wmin(wmin(wmax(t1.x, t2.x), wmax(t1.y, t2.y)), wmax(t1.z, t2.z));
    tmin_out = tmin;
    return (tmax >= wmax(tmin, 0.0f)) & (tmin < best);
}

// The same test with the near / far bound of every axis picked by the sign of the ray's inverse direction instead of by
// min / max: for inv >= 0, fl(mn - o) <= fl(mx - o) (rounding is monotone and mn <= mx) and the products keep that order, so
// min(t1, t2) IS t1 and max(t1, t2) IS t2, bit for bit (for inv < 0 the other way round; equal values are the same number,
// and a zero's sign never reaches a comparison result).  One v_perm_b32 per axis builds {near, far} as a pair of halves from
// the packed box words -- 3 instructions in place of the 6 min / max.  The selectors are per ray (ray_selectors).
// Empty and degenerate child slots hold the inverted box (+inf, -inf) (kEmptyBox*): near = +inf, far = -inf for either sign,
// so the test fails by itself and the child references need no check.  (The min / max form above would accept such a box.)
struct RaySel { uint32_t x, y, z; };
__device__ __forceinline__ RaySel ray_selectors(F3 inv) {
    RaySel s;
    s.x = inv.x < 0.0f ? 0x01000706u : 0x07060100u;     // v_perm_b32(w1, w0): mn.x = bytes 0-1, mx.x = bytes 6-7
    s.y = inv.y < 0.0f ? 0x03020504u : 0x05040302u;     // v_perm_b32(w2, w0): mn.y = bytes 2-3, mx.y = bytes 4-5
    s.z = inv.z < 0.0f ? 0x01000706u : 0x07060100u;     // v_perm_b32(w2, w1): mn.z = bytes 0-1, mx.z = bytes 6-7
    return s;
}
constexpr uint32_t kEmptyBox0 = 0x7C007C00u, kEmptyBox1 = 0xFC007C00u, kEmptyBox2 = 0xFC00FC00u;   // mn = +inf, mx = -inf (f16)
// returns the hit mask of the wavefront's active lanes (the two comparisons are ballots of their own, combined on the scalar unit)
// (Round 4 tried the 18 instructions up to max(tmin, 0) as ONE asm statement in a fixed order: on gfx950 the compiler puts an `s_nop 0` behind
// every one-instruction asm statement -- it must assume a partial register write, the dst_sel forwarding hazard --, 24 idle issue slots per
// node here.  The single statement removed them and measured 0.6 % SLOWER in dense launches, same-session (profiles/r04_q5_slab_asm_ab.txt):
// the step is not bound by issue slots, and eleven early-clobber temporaries per box cost more than the no-ops.  Not kept.)
__device__ __forceinline__ unsigned long long slab_sel(const F3& o, const F3& inv, const RaySel& sel, uint32_t w0, uint32_t w1, uint32_t w2, float best, float& tmin_out) {
    const uint32_t bx = __builtin_amdgcn_perm(w1, w0, sel.x), by = __builtin_amdgcn_perm(w2, w0, sel.y), bz = __builtin_amdgcn_perm(w2, w1, sel.z);
    const float nx = half_lo_minus(bx, o.x) * inv.x, ny = half_lo_minus(by, o.y) * inv.y, nz = half_lo_minus(bz, o.z) * inv.z;
    const float fx = half_hi_minus(bx, o.x) * inv.x, fy = half_hi_minus(by, o.y) * inv.y, fz = half_hi_minus(bz, o.z) * inv.z;
    const float tmin = wmax(wmax(nx, ny), nz);
    const float tmax = wmin(wmin(fx, fy), fz);
    tmin_out = tmin;
    return __builtin_amdgcn_ballot_w64(tmax >= wmax(tmin, 0.0f)) & __builtin_amdgcn_ballot_w64(tmin < best);
}
// this lane's bit of a wavefront mask as a predicate (no instruction: the mask is used as the lane mask it is)
__device__ __forceinline__ bool lane_of(unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

// ---- build-defined sampling; integer hash + fixed fmaf polynomials ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ uint32_t sample_key(uint32_t seed, uint32_t pixel, uint32_t sidx) {
    uint32_t h = mix32(seed + 0x9E3779B9u);
    h = mix32(h ^ pixel);
    return mix32(h ^ sidx);
}
__device__ __forceinline__ float rnd(uint32_t key, uint32_t bounce, uint32_t dim) {
    const uint32_t h = mix32(key ^ (bounce * 8u + dim + 1u) * 0x9E3779B1u);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ void sincos_2pi(float u, float& c, float& s) {
    const float q = u * 4.0f;
    const float kf = floorf(q + 0.5f);
    const float y = (q - kf) * 1.57079632679489662f;
    const float y2 = y * y;
    float sp = __builtin_fmaf(y2, 2.7557319e-6f, -1.9841270e-4f);
    sp = __builtin_fmaf(y2, sp, 8.3333333e-3f);
    sp = __builtin_fmaf(y2, sp, -1.6666667e-1f);
    sp = __builtin_fmaf(y2, sp, 1.0f);
    const float sy = y * sp;
    float cp = __builtin_fmaf(y2, -2.7557319e-7f, 2.4801587e-5f);
    cp = __builtin_fmaf(y2, cp, -1.3888889e-3f);
    cp = __builtin_fmaf(y2, cp, 4.1666667e-2f);
    cp = __builtin_fmaf(y2, cp, -0.5f);
    const float cy = __builtin_fmaf(y2, cp, 1.0f);
    const int k = (int)kf & 3;
    c = (k == 0) ? cy : (k == 1) ? -sy : (k == 2) ? -cy : sy;
    s = (k == 0) ? sy : (k == 1) ? cy : (k == 2) ? -sy : -cy;
}
// cosine-weighted direction around n in two halves: the sample in the local frame (independent of n: the megakernel's shade pass computes it while the hit
// triangle's normal is still on its way from memory) and its transfer into the branch-free orthonormal basis of n (Duff et al.)
template <bool B = false> __device__ __forceinline__ F3 cosine_local(float u1, float u2) {
    float c, s; sincos_2pi(u2, c, s);
    const float r = sqrt_f<B>(u1);
    return f3(r * c, r * s, sqrt_f<B>(1.0f - u1));
}
template <bool B = false> __device__ __forceinline__ F3 cosine_world(F3 n, F3 l) {
    const float sign = copysignf(1.0f, n.z);
    const float a = B ? -rcp_normal(sign + n.z) : -1.0f / (sign + n.z);        // (the quotient's sign is exact: -(1 / x) and (-1) / x are the same bits)
    const float b = n.x * n.y * a;
    const F3 t = f3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    const F3 bt = f3(b, sign + n.y * n.y * a, -n.y);
    return (t * l.x + bt * l.y) + n * l.z;
}
template <bool B = false> __device__ __forceinline__ F3 cosine_dir(F3 n, float u1, float u2) { return cosine_world<B>(n, cosine_local<B>(u1, u2)); }

// n / d and n % d for a divisor whose magic = floor(2^32 / d) comes from the host: estimate by multiply-high (never too large, at
// most one too small), one correction
__device__ __forceinline__ void divmod_magic(uint32_t n, uint32_t d, uint32_t magic, uint32_t& q, uint32_t& r) {
    q = __umulhi(n, magic); r = n - q * d;
    if (r >= d) { r -= d; ++q; }
}

// One logical queue item -> the pixel-sample it stands for.  64 consecutive logical items are one (frame, traced tile, sample) batch;
// consecutive logical batches are perm_cols batches apart in (frame, tile, sample) order (a perm_rows-row transposition, pt_api.cpp).
struct ItemInfo { uint32_t sample_index, fid, px, py, s; bool valid; };
__device__ __forceinline__ ItemInfo decode_item(const RenderArgs& A, uint32_t logical) {
    ItemInfo it;
    const uint32_t lb = logical >> 6, p = logical & 63u;
    uint32_t pcol, prow; divmod_magic(lb, A.perm_rows, A.perm_rows_magic, pcol, prow);
    const uint32_t q = prow * A.perm_cols + pcol;                       // batch in (frame, traced tile, sample) order
    const bool in_range = q < A.num_batches;
    uint32_t fid, qq; divmod_magic(in_range ? q : 0u, A.trace_bpf, A.trace_bpf_magic, fid, qq);
    uint32_t tslot, s; divmod_magic(qq, A.spp, A.spp_magic, tslot, s);
    const uint32_t slot = A.trace_slots ? A.trace_slots[tslot] : tslot;  // owned-tile slot
    const uint32_t tile = A.tiles ? A.tiles[slot] : slot;
    uint32_t ty, tx; divmod_magic(tile, A.tiles_x, A.tiles_x_magic, ty, tx);
    it.px = tx * 8u + (p & 7u); it.py = ty * 8u + (p >> 3);
    it.fid = fid; it.s = s;
    it.sample_index = ((fid * A.batches_per_frame + slot * A.spp + s) << 6) + p;
    it.valid = in_range & (it.px < A.width) & (it.py < A.height);
    return it;
}

constexpr float kEpsOrigin = 1e-4f;
constexpr float kBgPrimary = 0.01f;    // renderer.wgsl:410
constexpr float kSkyAmbient = 0.15f;   // renderer.wgsl:352
constexpr uint32_t kRRStart = 2;

__device__ __forceinline__ F3 light_dir() { return normalize3(f3(1.0f, 1.5f, 1.0f)); }   // renderer.wgsl:349
// triangle records (pt_host.h::TriRecord): 64 bytes, pieces 0..2 = (v0[a], e1[a], e2[a], 0) per axis a, piece 3 = the normal
__device__ __forceinline__ F3 tri_normal(const RenderArgs& A, uint32_t ti) {
    const float4 n = A.tris[(size_t)ti * 4 + 3];
    return f3(n.x, n.y, n.z);
}
// Packed references (pt_host.h): a record's position in the scene arena in 16-byte units, leaf flag in bit 31 -- `ref << 4` is
// the byte offset of the record whichever kind it is (the shift drops the flag).
__device__ __forceinline__ const uint4* arena_record(const RenderArgs& A, uint32_t ref) {
    return (const uint4*)((const char*)A.scene + (ref << 4));
}
__device__ __forceinline__ F3 tri_normal_ref(const RenderArgs& A, uint32_t leaf_ref) {
    const uint4 n = arena_record(A, leaf_ref)[3];
    return f3(__uint_as_float(n.x), __uint_as_float(n.y), __uint_as_float(n.z));
}


} // namespace ptk
