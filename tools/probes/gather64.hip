// gather64.hip -- how many dependent 64-byte record gathers per second does the chip sustain, and does the way a
// wavefront asks for them matter?  (The traversal step of trace_paths_kernel is one such gather per lane and iteration.)
//
//   V0  every lane reads its own record with 4 x global_load_dwordx4 (what the kernel does): each instruction touches
//       64 different cache lines
//   V1  every lane reads only the first 16 bytes (1 x dwordx4): the request count without the bytes
//   V2  quad-cooperative: lane l reads piece (l & 3) of the record of lane (l >> 2) + 16 k in instruction k, so each
//       instruction touches 16 lines and uses all 64 bytes of each; the pieces go back to their owners through LDS
//   V3  like V0 with 2 x dwordx4 (32-byte records)
// FMA = dependent v_fma_f32 per iteration next to the gather (the traversal step has ~250 VALU instructions).
// build: hipcc --offload-arch=gfx950 -O3 -o gather64 gather64.hip ;  run: ./gather64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int V, int FMA, int WPS = 6>
__global__ __launch_bounds__(64, WPS) void gather_kernel(const uint4* __restrict__ table, uint32_t mask, uint32_t hot_mask, uint32_t hot_shift, int iters, uint32_t* out) {
    __shared__ uint4 xch[64 * 4];
    __shared__ __attribute__((aligned(16))) char land[4 * 1040];
    const uint32_t lane = threadIdx.x;
    uint32_t idx = mix(blockIdx.x * 64u + lane) & mask;
    float acc = (float)lane;
    uint32_t sum = 0;
    for (int it = 0; it < iters; ++it) {
        uint4 a, b, c, d;
        if (V == 0) {
            const uint4* r = table + (size_t)idx * 4;
            a = r[0]; b = r[1]; c = r[2]; d = r[3];
        } else if (V == 1) {
            const uint4* r = table + (size_t)idx * 4;
            a = r[0]; b = a; c = a; d = a;
        } else if (V == 3) {
            const uint4* r = table + (size_t)idx * 4;
            a = r[0]; b = r[1]; c = a; d = b;
        } else if (V == 6) {
            // like V0, but the record's cache line is brought in by the first 16-byte load alone: the other three issue only after it has
            // returned, so they are plain L1 hits instead of accesses to a line whose fill is still in flight
            const uint4* r = table + (size_t)idx * 4;
            a = r[0];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w) :: "memory");
            b = r[1]; c = r[2]; d = r[3];
        } else if (V == 7) {
            // like V6 with a one-dword touch
            const uint4* r = table + (size_t)idx * 4;
            uint32_t touch = ((const uint32_t*)r)[0];
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(touch) :: "memory");
            a = r[0]; b = r[1]; c = r[2]; d = r[3]; a.x ^= touch & 0u;
        } else if (V == 4) {
            // LDS-DMA, owners inside the quad: in instruction k the four lanes of a quad fetch the four 16-byte pieces of the record
            // wanted by the quad's lane k, straight into LDS (landing zone k, 16 bytes per lane); the owner reads its 64 bytes back
#define DMA_K(K) { const uint32_t oidx = (uint32_t)__builtin_amdgcn_mov_dpp((int)idx, (K) * 0x55, 0xf, 0xf, true);   /* quad_perm:[K,K,K,K] */ \
                 const uint4* src = table + (size_t)oidx * 4 + (lane & 3u); \
                 __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, \
                                                  (__attribute__((address_space(3))) void*)((char*)land + (K) * 1040), 16, 0, 0); }
            DMA_K(0) DMA_K(1) DMA_K(2) DMA_K(3)
#undef DMA_K
            __builtin_amdgcn_s_waitcnt(0x0f70);                                                      // vmcnt(0)
            const uint4* mine = (const uint4*)((const char*)land + (lane & 3u) * 1040 + (lane >> 2) * 64);
            a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
        } else {
            // instruction k: lane l fetches piece (l & 3) of the record wanted by lane (l >> 2) + 16 k
            uint4 p[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t owner = (lane >> 2) + 16u * k;
                const uint32_t oidx = __shfl(idx, owner);
                p[k] = table[(size_t)oidx * 4 + (lane & 3u)];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) xch[((lane >> 2) + 16u * k) * 4 + (lane & 3u)] = p[k];   // owner-major: record of owner o at xch[4 o .. 4 o + 3]
            __builtin_amdgcn_s_waitcnt(0xc07f);                                                      // lgkmcnt(0): single-wave workgroup, no barrier needed
            a = xch[lane * 4 + 0]; b = xch[lane * 4 + 1]; c = xch[lane * 4 + 2]; d = xch[lane * 4 + 3];
        }
        const uint32_t h = a.x ^ b.y ^ c.z ^ d.w;
        sum += h;
#pragma unroll
        for (int f = 0; f < FMA; ++f) acc = __builtin_fmaf(acc, 1.0000001f, __uint_as_float((h & 0x007fffffu) | 0x3f000000u));
        // next record: a hot set (the top of a tree) with probability 1 - 2^-hot_shift ... modelled by masks
        const uint32_t r = mix(h + (uint32_t)it);
        idx = ((r >> 24) < hot_shift) ? (r & mask) : (r & hot_mask);
    }
    out[blockIdx.x * 64u + lane] = sum ^ __float_as_uint(acc);
}

template <int V, int FMA, int WPS = 6>
static int run(const char* name, const uint4* table, uint32_t mask, uint32_t hot_mask, uint32_t cold_of_256, uint32_t* out, int grid, int iters) {
    grid = grid / 6 * WPS;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather_kernel<V, FMA, WPS>), dim3(grid), dim3(64), 0, 0, table, mask, hot_mask, cold_of_256, iters / 4, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((gather_kernel<V, FMA, WPS>), dim3(grid), dim3(64), 0, 0, table, mask, hot_mask, cold_of_256, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)grid * 64.0 * iters;
    printf("%-34s %8.3f ms  %7.1f G records/s  %6.2f TB/s (64 B each)  %.0f cycles/iteration/SIMD-slot\n", name, ms, recs / ms / 1e6, recs * 64 / ms / 1e9,
           ms * 1e-3 * 2.4e9 / iters / (double)WPS);
    return 0;
}

int main(int argc, char** argv) {
    const int grid = 256 * 4 * 6, iters = 2000;
    for (int pass = 0; pass < 2; ++pass) {
        // pass 0: 25 MB table, 7/8 of the reads in a 1 MB hot set (roughly the hit rates of the real traversal); pass 1: uniform over 25 MB
        const uint32_t records = 1u << 19;                       // 512 Ki records x 64 B = 32 MB (power of two for the mask)
        const uint32_t mask = records - 1u, hot_mask = pass == 0 ? (1u << 14) - 1u : mask;
        const uint32_t cold_of_256 = pass == 0 ? 32u : 256u;
        uint4* table; uint32_t* out;
        CK(hipMalloc((void**)&table, (size_t)records * 64)); CK(hipMalloc((void**)&out, (size_t)grid * 64 * 4));
        std::vector<uint32_t> h((size_t)records * 16);
        uint32_t s = 12345u; for (auto& w : h) { s = s * 1664525u + 1013904223u; w = s; }
        CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("---- %s ----\n", pass == 0 ? "32 MB table, 7/8 of the gathers in a 1 MB hot set" : "32 MB table, uniform");
        if (run<0, 0>("V0 4 x dwordx4 per lane", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<3, 0>("V3 2 x dwordx4 per lane (32 B)", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<1, 0>("V1 1 x dwordx4 per lane (16 B)", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<2, 0>("V2 quad-cooperative + LDS", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<6, 0>("V6 first 16 B, wait, then 48 B", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<7, 0>("V7 touch dword, wait, then 64 B", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<6, 100>("V6 + 100 dependent FMA", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<4, 100>("V4 + 100 dependent FMA", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<4, 0>("V4 LDS-DMA quad-cooperative", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<4, 250>("V4 + 250 dependent FMA", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<0, 250, 8>("V0 + 250 FMA, 8 waves/SIMD", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<0, 250, 4>("V0 + 250 FMA, 4 waves/SIMD", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<4, 250, 4>("V4 + 250 FMA, 4 waves/SIMD", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<4, 250, 8>("V4 + 250 FMA, 8 waves/SIMD", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<0, 100>("V0 + 100 dependent FMA", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<2, 100>("V2 + 100 dependent FMA", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<0, 250>("V0 + 250 dependent FMA", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<2, 250>("V2 + 250 dependent FMA", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        if (run<1, 250>("V1 + 250 dependent FMA", table, mask, hot_mask, cold_of_256, out, grid, iters)) return 1;
        CK(hipFree(table)); CK(hipFree(out));
    }
    return 0;
}
