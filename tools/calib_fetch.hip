// calib_fetch.hip -- calibrates rocprofv3's FETCH_SIZE for THIS workload's access pattern: every lane reads
// one 64-byte record (4 x global_load_dwordx4) at a pseudo-random 64-byte-aligned offset of a buffer much
// larger than the 256 MiB Infinity Cache, so (almost) every record comes from HBM and the true byte count is
// known: records x 64 B.   hipcc --offload-arch=gfx950 -O3 tools/calib_fetch.hip -o calib_fetch
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- ./calib_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void gather64(const uint4* __restrict__ buf, uint32_t num_records, uint32_t per_lane, uint4* __restrict__ sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (uint32_t k = 0; k < per_lane; ++k) {
        const uint32_t r = mix32(gid * 0x9E3779B1u + k * 0x85EBCA77u + 12345u) % num_records;
        const uint4* p = buf + (size_t)r * 4;
        const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc.x ^= a.x ^ b.y ^ c.z ^ d.w; acc.y += a.y + b.z; acc.z ^= c.w; acc.w += d.x;
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[gid & 63] = acc;   // never true in practice; keeps the loads alive
}
int main() {
    const size_t bytes = size_t(3) << 30;                 // 3 GiB >> 256 MiB Infinity Cache
    const uint32_t num_records = uint32_t(bytes / 64);
    uint4 *buf = nullptr, *sink = nullptr;
    if (hipMalloc((void**)&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 64 * 16) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    const uint32_t blocks = 256 * 8, threads = 256, per_lane = 16;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(gather64, dim3(blocks), dim3(threads), 0, 0, buf, num_records, per_lane, sink);
    hipDeviceSynchronize();
    const double true_bytes = double(blocks) * threads * per_lane * 64.0;
    std::printf("records per launch %.0f, true bytes per launch %.0f\n", true_bytes / 64.0, true_bytes);
    hipFree(buf); hipFree(sink);
    return 0;
}
