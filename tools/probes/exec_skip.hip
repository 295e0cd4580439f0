// Probe: does a wave64 VALU instruction cost less when whole 16-lane quarters of EXEC are off?
// (if so, compacting the few active lanes of a divergent phase into one quarter would pay)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void spin(float* out, unsigned long long mask, int iters) {
    const unsigned lane = threadIdx.x & 63u;
    float a = (float)lane * 0.001f + 1.0f, b = 1.0001f, c = 0.5f, d = 0.25f;
    if ((mask >> lane) & 1ull) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f; }
        }
    }
    out[blockIdx.x * 64 + lane] = a + b + c + d;
}

int main() {
    float* out; CK(hipMalloc((void**)&out, 4096 * 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* name; unsigned long long mask; } cases[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xFFFFFFFFull}, {"lanes 0-15", 0xFFFFull}, {"lanes 0-7", 0xFFull}, {"lane 0", 1ull},
        {"16 lanes, one in four", 0x1111111111111111ull}, {"4 lanes, one per quarter", 0x0001000100010001ull}, {"lanes 48-63", 0xFFFF000000000000ull},
        {"lanes 0-15 + 32-47", 0x0000FFFF0000FFFFull},
    };
    const int grid = 256 * 4 * 6;      // 6 single-wave workgroups per SIMD, like the megakernel
    for (auto& c : cases) {
        hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, 0, out, c.mask, 200);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(spin, dim3(grid), dim3(64), 0, 0, out, c.mask, 4000);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.3f ms\n", c.name, ms);
    }
    return 0;
}
