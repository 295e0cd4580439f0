// rcp_exact.hip -- is the short reciprocal of pt_device.h::rcp_ieee bit-identical to the compiler's IEEE `1.0f / x`?
// The compiler lowers 1.0f / x to v_div_scale x2, v_rcp, fma x4, v_mul, v_div_fmas, v_div_fixup (11 instructions); for a denominator
// whose exponent keeps v_div_scale from scaling (2^-95 <= |x| <= 2^125 with the numerator 1.0) the two v_div_scale return their inputs, v_div_fmas
// is a plain fma and v_div_fixup returns the quotient: what is left is v_rcp + 6 fma.  This probe runs BOTH over EVERY f32 bit pattern with
// 2^-64 <= |x| < 2^65 (the range rcp_ieee is used in; outside it the caller takes the division) and counts differing results.
//   build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probes/rcp_exact.hip -o /tmp/rcp_exact && /tmp/rcp_exact
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float rcp_short(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    const float e1 = __builtin_fmaf(-d, r, 1.0f);
    const float q1 = __builtin_fmaf(e1, r, r);
    const float e2 = __builtin_fmaf(-d, q1, 1.0f);
    return __builtin_fmaf(e2, r, q1);
}
// one thread per (sign, exponent 63 .. 191, upper 13 mantissa bits); it walks the low 10 mantissa bits
__global__ void check(unsigned long long* bad, uint32_t* first_bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sign = i >> 21, exp = 63u + ((i >> 13) & 0xffu), mh = i & 0x1fffu;
    if (exp > 191u) return;
    unsigned long long n = 0;
    for (uint32_t lo = 0; lo < 1024u; ++lo) {
        const uint32_t bits = (sign << 31) | (exp << 23) | (mh << 10) | lo;
        const float x = __uint_as_float(bits);
        const float a = 1.0f / x, b = rcp_short(x);
        if (__float_as_uint(a) != __float_as_uint(b)) { ++n; atomicMin(first_bad, bits & 0x7fffffffu); }
    }
    if (n) atomicAdd(bad, n);
}
int main() {
    unsigned long long* d_bad; uint32_t* d_first; unsigned long long bad = 0; uint32_t first = 0xffffffffu;
    hipMalloc(&d_bad, 8); hipMalloc(&d_first, 4); hipMemcpy(d_bad, &bad, 8, hipMemcpyHostToDevice); hipMemcpy(d_first, &first, 4, hipMemcpyHostToDevice);
    check<<<(1u << 22) / 256, 256>>>(d_bad, d_first);
    hipDeviceSynchronize();
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("f32 patterns with 2^-64 <= |x| < 2^65 checked: %llu; results differing from 1.0f / x: %llu (smallest |x| bits 0x%08x)\n", 2ull * 129ull * (1ull << 23), bad, first);
    return bad != 0;
}
