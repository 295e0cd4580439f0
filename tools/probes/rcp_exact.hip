// rcp_exact.hip -- are the short reciprocal and square root of pt_device.h (rcp_normal, sqrt_normal) bit-identical to the compiler's IEEE
// `1.0f / x` and `sqrtf(x)`?
// The compiler lowers 1.0f / x to v_div_scale x2, v_rcp, fma x4, v_mul, v_div_fmas, v_div_fixup (11 instructions); for a denominator
// whose exponent keeps v_div_scale from scaling (2^-95 <= |x| <= 2^125 with the numerator 1.0) the two v_div_scale return their inputs, v_div_fmas
// is a plain fma and v_div_fixup returns the quotient: what is left is v_rcp + 6 fma.  Likewise sqrtf is v_sqrt + a try of the two neighbouring
// values against exact residuals, wrapped into a scaling of small operands and a zero / infinity fix-up that operands of ordinary size do not need.
// This probe runs the short and the compiler's forms over EVERY f32 bit pattern with 2^-64 <= |x| < 2^65 (both signs for the reciprocal; the positive
// ones and zero for the square root) -- the range the short forms are used in (pt_api.cpp::arith_is_bounded) -- and counts differing results.
//   build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/rcp_exact.hip -o /tmp/rcp_exact && /tmp/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
// the compiler's sequence without its scaling and fix-up (round 5's first short form: 7 instructions)
__device__ __forceinline__ float rcp_short(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    const float e1 = __builtin_fmaf(-d, r, 1.0f);
    const float q1 = __builtin_fmaf(e1, r, r);
    const float e2 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(e2, r, q1);
}
// a shorter candidate: two Newton steps (5 instructions, 5 deep) -- does it round correctly everywhere?
__device__ __forceinline__ float rcp_5(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float rcp_3(float d) { const float r = __builtin_amdgcn_rcpf(d); return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r); }
// candidate: reciprocal square root + one coupled Newton step (5 instructions, 4 deep); x == 0 must give 0
__device__ __forceinline__ float sqrt_5(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ float sqrt_short(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = (r_dn <= 0.0f) ? s_dn : s;
    return (r_up > 0.0f) ? s_up : s;
}
// one thread per (sign, exponent 63 .. 191, upper 13 mantissa bits); it walks the low 10 mantissa bits
__global__ void check(unsigned long long* bad, uint32_t* first_bad, unsigned long long* bad_sqrt, unsigned long long* bad5, unsigned long long* bad3, unsigned long long* bads5) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sign = i >> 21, exp = 63u + ((i >> 13) & 0xffu), mh = i & 0x1fffu;
    if (exp > 191u) return;
    unsigned long long n = 0, m = 0, n5 = 0, n3 = 0, m5 = 0;
    for (uint32_t lo = 0; lo < 1024u; ++lo) {
        const uint32_t bits = (sign << 31) | (exp << 23) | (mh << 10) | lo;
        const float x = __uint_as_float(bits);
        const float a = 1.0f / x, b = rcp_short(x);
        if (__float_as_uint(a) != __float_as_uint(b)) { ++n; atomicMin(first_bad, bits & 0x7fffffffu); }
        if (__float_as_uint(a) != __float_as_uint(rcp_5(x))) ++n5;
        if (__float_as_uint(a) != __float_as_uint(rcp_3(x))) ++n3;
        if (sign == 0u && __float_as_uint(sqrtf(x)) != __float_as_uint(sqrt_short(x))) ++m;
        if (sign == 0u && __float_as_uint(sqrtf(x)) != __float_as_uint(sqrt_5(x))) ++m5;
    }
    if (i == 0u && __float_as_uint(sqrtf(0.0f)) != __float_as_uint(sqrt_short(0.0f))) ++m;
    if (n) atomicAdd(bad, n);
    if (m) atomicAdd(bad_sqrt, m);
    if (n5) atomicAdd(bad5, n5);
    if (n3) atomicAdd(bad3, n3);
    if (m5) atomicAdd(bads5, m5);
}
int main() {
    unsigned long long *d_bad, *d_bad_sqrt, *d_bad5; unsigned long long bad5 = 0; hipMalloc(&d_bad5, 8); hipMemcpy(d_bad5, &bad5, 8, hipMemcpyHostToDevice);
    unsigned long long* d_bads5; unsigned long long bads5 = 0; hipMalloc(&d_bads5, 8); hipMemcpy(d_bads5, &bads5, 8, hipMemcpyHostToDevice);
    unsigned long long* d_bad3; unsigned long long bad3 = 0; hipMalloc(&d_bad3, 8); hipMemcpy(d_bad3, &bad3, 8, hipMemcpyHostToDevice);
    uint32_t* d_first; unsigned long long bad = 0, bad_sqrt = 0; uint32_t first = 0xffffffffu;
    hipMalloc(&d_bad_sqrt, 8); hipMemcpy(d_bad_sqrt, &bad_sqrt, 8, hipMemcpyHostToDevice);
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 4); hipMemcpy(d_bad, &bad, 8, hipMemcpyHostToDevice); hipMemcpy(d_first, &first, 4, hipMemcpyHostToDevice);
    check<<<(1u << 22) / 256, 256>>>(d_bad, d_first, d_bad_sqrt, d_bad5, d_bad3, d_bads5);
    hipDeviceSynchronize();
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&bad_sqrt, d_bad_sqrt, 8, hipMemcpyDeviceToHost);
    printf("positive f32 patterns with 2^-64 <= x < 2^65 and zero checked: %llu; results differing from sqrtf(x): %llu\n", 129ull * (1ull << 23) + 1ull, bad_sqrt);
    printf("f32 patterns with 2^-64 <= |x| < 2^65 checked: %llu; results differing from 1.0f / x: %llu (smallest |x| bits 0x%08x)\n", 2ull * 129ull * (1ull << 23), bad, first);
    hipMemcpy(&bad5, d_bad5, 8, hipMemcpyDeviceToHost);
    printf("(candidate: v_rcp + two Newton steps, 5 instructions: %llu results differ from 1.0f / x -- %s)\n", bad5, bad5 ? "not usable" : "usable");
    hipMemcpy(&bad3, d_bad3, 8, hipMemcpyDeviceToHost);
    printf("(pt_device.h::rcp_normal = v_rcp + ONE Newton step, 3 instructions: %llu results differ from 1.0f / x -- %s)\n", bad3, bad3 ? "not usable" : "usable");
    hipMemcpy(&bads5, d_bads5, 8, hipMemcpyDeviceToHost);
    printf("(candidate: v_rsq + one coupled Newton step for the square root, 5 instructions: %llu results differ (x = 0 not counted here) -- %s)\n", bads5, bads5 ? "not usable" : "usable");
    return bad != 0 || bad_sqrt != 0 || bad3 != 0;
}
