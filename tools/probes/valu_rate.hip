// Probe: the vector unit's issue rate per SIMD for wave64 instructions, against the number of resident wavefronts -- the peak bench.py's `valu_issue` fraction
// is priced against comes from the guide (a wave64 instruction occupies a SIMD-32 for 2 cycles: 0.5 per SIMD and cycle, 1,228.8 G/s chip-wide at 2.4 GHz);
// this measures what the chip sustains for plain f32 FMAs / multiplies / min-max, with eight independent chains per wavefront (no instruction waits for its
// predecessor) and with one dependent chain.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define V8(a, b, c, d, e, f, g, h) \
    "v_fma_f32 " a ", " a ", %8, %9\n\tv_mul_f32 " b ", " b ", %8\n\tv_max_f32 " c ", " c ", %9\n\tv_fma_f32 " d ", " d ", %8, %9\n\t" \
    "v_mul_f32 " e ", " e ", %8\n\tv_min_f32 " f ", " f ", %9\n\tv_fma_f32 " g ", " g ", %8, %9\n\tv_sub_f32 " h ", " h ", %9\n\t"

template <int CHAINS>
__global__ __launch_bounds__(64) void valu_spin(float* out, int iters, float k0, float k1) {
    float a = threadIdx.x * 1e-3f, b = 1.0f, c = 2.0f, d = 3.0f, e = 1.5f, f = 4.0f, g = 0.5f, h = 9.0f;
    for (int i = 0; i < iters; ++i) {
        if (CHAINS == 8)
            asm volatile(V8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") V8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") V8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") V8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7")
                         V8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") V8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") V8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") V8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k0), "v"(k1));
        else
            asm volatile(V8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") V8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") V8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") V8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0")
                         V8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") V8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") V8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") V8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(k0), "v"(k1));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    float* out; CK(hipMalloc((void**)&out, (size_t)cus * 32 * 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("device: %d CUs, %d SIMDs, clock %d MHz (hipDeviceProp)\n", cus, simds, prop.clockRate / 1000);
    printf("%-7s %-8s %10s %14s %26s\n", "chains", "W/SIMD", "ms", "G VALU / s", "per SIMD and cycle @2.4GHz");
    const int iters = 20000;
    for (int chains : {8, 1}) {
        for (int w : {1, 2, 3, 4, 6, 8}) {
            const int grid = simds * w;          // W single-wave workgroups per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (chains == 8) hipLaunchKernelGGL(valu_spin<8>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
                else             hipLaunchKernelGGL(valu_spin<1>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            const double insts = (double)grid * iters * 64.0;
            printf("%-7d %-8d %10.3f %14.1f %26.3f\n", chains, w, ms, insts / (ms * 1e-3) / 1e9, insts / (ms * 1e-3) / simds / 2.4e9);
        }
    }
    return 0;
}
