// Probe: the scalar unit's issue rate per CU, against the number of resident wavefronts (the guide gives "1 scalar unit per CU" and no rate).
// bench.py's roofline prices the megakernel's SQ_INSTS_SALU against this figure (fractions.salu_issue, peak_source "probe").
//
// Every wavefront runs a long run of scalar ALU instructions (s_add_u32 / s_xor_b32 / s_lshl_b32 / s_and_b32 on CHAINS independent register chains: 8 = no
// instruction waits for its predecessor's result, 1 = every instruction does), W single-wave workgroups per CU (W / 4 per SIMD), nothing else.
// Reported: chip-wide G scalar instructions / s, per CU and cycle at the 2.4 GHz of the guide, and per CU and s_memtime tick-cycle as measured.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/salu_rate.hip -o /tmp/salu_rate && /tmp/salu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define S8(a, b, c, d, e, f, g, h) \
    "s_add_u32 " a ", " a ", %8\n\ts_xor_b32 " b ", " b ", %8\n\ts_lshl_b32 " c ", " c ", 1\n\ts_and_b32 " d ", " d ", %8\n\t" \
    "s_add_u32 " e ", " e ", %8\n\ts_xor_b32 " f ", " f ", %8\n\ts_lshl_b32 " g ", " g ", 1\n\ts_and_b32 " h ", " h ", %8\n\t"

template <int CHAINS>
__global__ __launch_bounds__(64) void salu_spin(unsigned* out, unsigned long long* ticks, int iters, unsigned k) {
    unsigned a = blockIdx.x, b = 2u, c = 3u, d = 0xffffu, e = 5u, f = 6u, g = 7u, h = 0xff00ffu;
    a = __builtin_amdgcn_readfirstlane(a);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (CHAINS == 8)      // 64 instructions per pass, eight independent chains
            asm volatile(S8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") S8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") S8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") S8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7")
                         S8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") S8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") S8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7") S8("%0", "%1", "%2", "%3", "%4", "%5", "%6", "%7")
                         : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : "s"(k) : "scc");
        else                  // the same 64 instructions, all on ONE register: a dependent chain
            asm volatile(S8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") S8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") S8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") S8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0")
                         S8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") S8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") S8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0") S8("%0", "%0", "%0", "%0", "%0", "%0", "%0", "%0")
                         : "+s"(a), "+s"(b), "+s"(c), "+s"(d), "+s"(e), "+s"(f), "+s"(g), "+s"(h) : "s"(k) : "scc");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h; ticks[blockIdx.x] = t1 - t0; }
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int max_blocks = cus * 32;
    unsigned* out; unsigned long long* ticks;
    CK(hipMalloc((void**)&out, max_blocks * 4)); CK(hipMalloc((void**)&ticks, max_blocks * 8));
    unsigned long long* h_ticks = (unsigned long long*)malloc(max_blocks * 8);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("device: %s, %d CUs, clock %d MHz (hipDeviceProp)\n", prop.name, cus, prop.clockRate / 1000);
    printf("%-7s %-6s %10s %14s %22s %26s\n", "chains", "W/CU", "ms", "G SALU / s", "per CU and cycle @2.4GHz", "per CU and s_memtime tick");
    const int iters = 20000;
    for (int chains : {8, 1}) {
        for (int w : {1, 2, 4, 6, 8, 12, 16, 20, 24, 32}) {
            const int grid = cus * w;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0, 0));
                if (chains == 8) hipLaunchKernelGGL(salu_spin<8>, dim3(grid), dim3(64), 0, 0, out, ticks, iters, 1u);
                else             hipLaunchKernelGGL(salu_spin<1>, dim3(grid), dim3(64), 0, 0, out, ticks, iters, 1u);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            }
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(h_ticks, ticks, grid * 8, hipMemcpyDeviceToHost));
            double tick_sum = 0; for (int i = 0; i < grid; ++i) tick_sum += (double)h_ticks[i];
            const double insts = (double)grid * iters * 64.0;
            const double per_wave_tick = (double)iters * 64.0 / (tick_sum / grid);           // one wavefront's scalar instructions per tick of its own clock counter
            printf("%-7d %-6d %10.3f %14.1f %22.3f %26.3f\n", chains, w, ms, insts / (ms * 1e-3) / 1e9, insts / (ms * 1e-3) / cus / 2.4e9, per_wave_tick * w);
        }
    }
    return 0;
}
