// Probe: v_fma_mix_f32 with an f16 source computes (float)half - o in ONE instruction (fma(h, 1.0, -o) rounds once, like the
// subtraction); check it bit for bit against v_cvt_f32_f16 + v_sub_f32 for every half value and a set of origins.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ float sub_lo(uint32_t w, float o) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(o)); return r; }
__device__ __forceinline__ float sub_hi(uint32_t w, float o) { float r; asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(w), "v"(o)); return r; }
__device__ __forceinline__ float half_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float half_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }

__global__ void check(const float* origins, int n_o, unsigned* bad, unsigned* first) {
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;      // all 65536 half patterns
    if (h >= 65536u) return;
    const uint32_t w = h | ((h ^ 0x5a5au) << 16);
    for (int i = 0; i < n_o; ++i) {
        const float o = origins[i];
        const float a = half_lo(w) - o, b = sub_lo(w, o);
        const float c = half_hi(w) - o, d = sub_hi(w, o);
        const bool nan_ok1 = (a != a) && (b != b), nan_ok2 = (c != c) && (d != d);
        if ((__float_as_uint(a) != __float_as_uint(b) && !nan_ok1) || (__float_as_uint(c) != __float_as_uint(d) && !nan_ok2)) {
            if (atomicAdd(bad, 1u) == 0u) { first[0] = w; first[1] = __float_as_uint(o); first[2] = __float_as_uint(a); first[3] = __float_as_uint(b); first[4] = __float_as_uint(c); first[5] = __float_as_uint(d); }
        }
    }
}

int main() {
    float ho[] = {0.0f, -0.0f, 1.0f, -1.0f, 2.5f, 0.1f, -0.3333333f, 1e-8f, -1e-8f, 6.1e-5f, 5.96e-8f, 65504.0f, -65504.0f, 1e30f, -1e30f, 3.4e38f, 1e-40f, -1e-40f, 1.17549435e-38f, 0.999999f, 123.456f};
    const int n = sizeof(ho) / 4;
    float* d_o; unsigned *bad, *first;
    CK(hipMalloc((void**)&d_o, sizeof(ho))); CK(hipMemcpy(d_o, ho, sizeof(ho), hipMemcpyHostToDevice));
    CK(hipMalloc((void**)&bad, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMalloc((void**)&first, 32)); CK(hipMemset(first, 0, 32));
    hipLaunchKernelGGL(check, dim3(256), dim3(256), 0, 0, d_o, n, bad, first);
    CK(hipDeviceSynchronize());
    unsigned hb = 0, hf[8]; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost));
    printf("mismatches: %u of %d\n", hb, 65536 * n * 2);
    if (hb) printf("first: w=%08x o=%08x  lo: sub=%08x mix=%08x  hi: sub=%08x mix=%08x\n", hf[0], hf[1], hf[2], hf[3], hf[4], hf[5]);
    return 0;
}
