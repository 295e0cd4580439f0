// pt_build.hip -- device-side scene build of libmi355pt.  pt_build_bvh runs every step of the
// reference's PathTracer.buildBVH (src/libs/PathTracer.js:671-749) on the GPU:
//
//   Morton codes + sort  PathTracer.js:411-481 (CPU JavaScript in the reference)  -> morton kernels + rocPRIM radix sort
//   LBVH2                BVHBuilder.wgsl:152-306 (WebGPU in the reference)        -> pt_kernels.hip (launch_lbvh2)
//   collapse to BVH4     PathTracer.js:506-667 (CPU JavaScript in the reference)  -> level-synchronous kernels below
//   device layouts       (none in the reference)                                  -> wide nodes + triangle records
//
// Results are bit-identical to the host implementations in pt_host.cpp (which stay behind the
// C ABI's host entry points pt_morton_sort / pt_collapse_lbvh2_to_bvh4) and to the oracle:
// the same IEEE double / float operations in the same order (-ffp-contract=off), the same
// stable sort on the 30-bit code, the same DFS pre-order numbering.
#include "pt_kernels.h"
#include "pt_device.h"

#include <hipcub/hipcub.hpp>

namespace ptk {

namespace {

// ------------------------------------------------------------------------------------
// Morton codes (PathTracer.js:411-456).  Doubles throughout, like the JS.
// ------------------------------------------------------------------------------------
// order-preserving map double -> u64 (for atomicMin / atomicMax); never fed a NaN
__device__ __forceinline__ unsigned long long dkey(double v) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dunkey(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}
__device__ __forceinline__ double centroid_axis(const float* p, int k) {
    return (((double)p[k] + (double)p[3 + k]) + (double)p[6 + k]) / 3;      // (a + b + c) / 3, PathTracer.js:419-421
}

__global__ void build_init_kernel(unsigned long long* bounds, uint32_t* counters, uint32_t n_counters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3u) { bounds[i] = dkey(1e30); bounds[3 + i] = dkey(-1e30); }     // PathTracer.js:413-414
    if (i < n_counters) counters[i] = 0u;
}

// lo = min over centroids (c < lo ? c : lo), hi = max: NaN never wins a comparison, values beyond +-1e30 never win either.
// Grid-stride over the triangles, wavefront shuffle + LDS reduction: six atomics per block.
__global__ __launch_bounds__(256) void centroid_bounds_kernel(const float* __restrict__ tris, uint32_t n, unsigned long long* bounds) {
    __shared__ unsigned long long part[4][6];
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const float* p = tris + (size_t)t * 9;
        for (int k = 0; k < 3; ++k) {
            const double c = centroid_axis(p, k);
            if (c == c) { const unsigned long long key = dkey(c); lo[k] = key < lo[k] ? key : lo[k]; hi[k] = key > hi[k] ? key : hi[k]; }
        }
    }
    for (int k = 0; k < 3; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long a = __shfl_xor(lo[k], off), b = __shfl_xor(hi[k], off);
            lo[k] = a < lo[k] ? a : lo[k];
            hi[k] = b > hi[k] ? b : hi[k];
        }
    }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0u) { for (int k = 0; k < 3; ++k) { part[wave][k] = lo[k]; part[wave][3 + k] = hi[k]; } }
    __syncthreads();
    if (threadIdx.x < 6u) {
        const uint32_t k = threadIdx.x;
        unsigned long long v = part[0][k];
        for (uint32_t w = 1; w < (blockDim.x >> 6); ++w) { const unsigned long long o = part[w][k]; v = (k < 3u) ? (o < v ? o : v) : (o > v ? o : v); }
        if (k < 3u) atomicMin(&bounds[k], v); else atomicMax(&bounds[k], v);
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t v) {      // PathTracer.js:440-447
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ uint32_t quantize1023(double x) {    // max(0, min(1023, (x * 1023) | 0))
    const double s = x * 1023;
    if (!(s == s)) return 0u;
    if (s >= 1023.0) return 1023u;
    if (s <= 0.0) return 0u;
    return (uint32_t)(int32_t)s;
}

__global__ __launch_bounds__(256) void morton_kernel(const float* __restrict__ tris, uint32_t n, const unsigned long long* __restrict__ bounds,
                                                      uint32_t* __restrict__ code, uint32_t* __restrict__ index) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float* p = tris + (size_t)t * 9;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const double lo = dunkey(bounds[k]), hi = dunkey(bounds[3 + k]);
        const double d = hi - lo, ext = d > 1e-20 ? d : 1e-20;               // PathTracer.js:431-433
        q[k] = quantize1023((centroid_axis(p, k) - lo) / ext);
    }
    code[t] = (spread10(q[0]) << 2) | (spread10(q[1]) << 1) | spread10(q[2]);
    index[t] = t;
}

// ------------------------------------------------------------------------------------
// Triangle records: v0, e1, e2, n = normalize(cross(e1, e2)) -- the f32 operations
// renderer.wgsl:179-180,269 performs per visit (same as pt::build_tri_records on the host)
// ------------------------------------------------------------------------------------
// edge_max (optional): the largest |component| of any edge vector as f32 bits -- positive floats order like their bits, and a NaN's bits lie
// above those of infinity, so "small enough" is one unsigned compare on the host (what the traversal's short reciprocal may assume, pt_api.cpp)
__global__ __launch_bounds__(256) void tri_records_kernel(const float* __restrict__ tris, uint32_t n, float4* __restrict__ rec, uint32_t* __restrict__ edge_max) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = t < n;                       // (no early return: the lanes past n of the last wavefront take part in the reduction below, with 0)
    const float* p = tris + (size_t)(live ? t : 0u) * 9;
    const F3 v0 = f3(p[0], p[1], p[2]);
    const F3 e1 = f3(p[3] - p[0], p[4] - p[1], p[5] - p[2]), e2 = f3(p[6] - p[0], p[7] - p[1], p[8] - p[2]);
    if (edge_max) {
        auto bits = [](float v) { return __float_as_uint(v) & 0x7fffffffu; };
        uint32_t m = live ? max(max(max(bits(e1.x), bits(e1.y)), max(bits(e1.z), bits(e2.x))), max(bits(e2.y), bits(e2.z))) : 0u;
        for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
        if ((threadIdx.x & 63u) == 0u) atomicMax(edge_max, m);
    }
    if (!live) return;
    const F3 c = cross3(e1, e2);
    const float inv = 1.0f / sqrtf((c.x * c.x + c.y * c.y) + c.z * c.z);
    float4* r = rec + (size_t)t * 4;
    r[0] = make_float4(v0.x, e1.x, e2.x, 0.0f);           // pt_host.h::TriRecord: piece a = component a of v0, e1, e2; piece 3 = the normal
    r[1] = make_float4(v0.y, e1.y, e2.y, 0.0f);
    r[2] = make_float4(v0.z, e1.z, e2.z, 0.0f);
    r[3] = make_float4(c.x * inv, c.y * inv, c.z * inv, 0.0f);
}

// ------------------------------------------------------------------------------------
// Collapse LBVH2 -> BVH4 (PathTracer.js:506-667).  The JS walks the tree depth-first and numbers
// BVH4 nodes in pre-order; here the BVH4 nodes are discovered breadth-first (one launch per BVH4
// level), their subtree sizes and bounds are computed bottom-up and the pre-order ids top-down:
//   id(child s) = id(parent) + 1 + sum of the subtree sizes of the children in slots < s.
// A BVH4 node is identified during the build by its position in the breadth-first order.
// ------------------------------------------------------------------------------------
constexpr uint32_t kNone = 0xFFFFFFFFu;

__device__ __forceinline__ bool leaf2(const uint32_t* __restrict__ bvh2, uint32_t node) { return (bvh2[1 + (size_t)node * 6 + 5] & kLeaf) != 0u; }

// one thread per BVH4 node of this level: greedy expansion of its BVH2 subtree top into <= 4 entries
// (repeatedly replace the first internal entry by its two children), children appended to the next level
__global__ __launch_bounds__(256) void collapse_expand_kernel(const uint32_t* __restrict__ bvh2, uint32_t nn2, uint32_t* __restrict__ node2, uint4* __restrict__ child_pos,
                                                               uint32_t level_begin, uint32_t level_count, uint32_t next_begin, uint32_t capacity, uint32_t* next_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t kid[4] = {kNone, kNone, kNone, kNone};
    uint32_t nk = 0;
    const bool live = i < level_count;
    if (live) {
        const uint32_t node = node2[level_begin + i];
        if (!leaf2(bvh2, node)) {
            const uint32_t* p = bvh2 + 1 + (size_t)node * 6;
            kid[0] = p[3]; kid[1] = p[4]; nk = 2;
            for (int round = 0; round < 2 && nk < 4u; ++round) {
                uint32_t pos = nk;
                for (uint32_t s = 0; s < nk; ++s)
                    if (kid[s] < nn2 && !leaf2(bvh2, kid[s])) { pos = s; break; }
                if (pos == nk) break;
                const uint32_t* kp = bvh2 + 1 + (size_t)kid[pos] * 6;
                const uint32_t a = kp[3], b = kp[4];
                for (uint32_t m = nk; m > pos + 1u; --m) kid[m] = kid[m - 1];
                kid[pos] = a; kid[pos + 1] = b;
                ++nk;
            }
        }
    }
    // wave-aggregated append: exclusive prefix of nk over the wavefront, one atomic per wavefront
    uint32_t incl = nk;
    const uint32_t lane = threadIdx.x & 63u;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += v;
    }
    const uint32_t total = __shfl(incl, 63);
    uint32_t base = 0;
    if (lane == 63u && total) base = atomicAdd(next_count, total);
    base = __shfl(base, 63);
    if (!live) return;
    const uint32_t first = next_begin + base + (incl - nk);
    uint4 cp = make_uint4(kNone, kNone, kNone, kNone);
    uint32_t* cpw = &cp.x;
    for (uint32_t s = 0; s < nk; ++s) {
        const uint32_t pos = first + s;
        if (pos < capacity && kid[s] < nn2) { node2[pos] = kid[s]; cpw[s] = pos; }   // a malformed BVH2 cannot write out of bounds
    }
    child_pos[level_begin + i] = cp;
}

// JS Math.min / Math.max on numbers (sign of zero ordered, first operand wins a tie otherwise): pt_host.cpp js_min_f / js_max_f
__device__ __forceinline__ float js_min_f(float a, float b) { if (a < b) return a; if (b < a) return b; return (__float_as_uint(a) >> 31) ? a : b; }
__device__ __forceinline__ float js_max_f(float a, float b) { if (a > b) return a; if (b > a) return b; return (__float_as_uint(a) >> 31) ? b : a; }
__device__ __forceinline__ float half_exact(uint32_t h) {        // PathTracer.js:16-40, integer form (exact for subnormals whatever the FP mode)
    const uint32_t sign = (h & 0x8000u) << 16, mag = h & 0x7fffu;
    if (mag >= 0x7c00u) return __uint_as_float(sign | 0x7f800000u | ((mag & 0x3ffu) << 13));
    if (mag >= 0x0400u) return __uint_as_float(sign | ((mag + (112u << 10)) << 13));
    const float v = (float)mag * 5.9604644775390625e-8f;
    return __uint_as_float(__float_as_uint(v) | sign);
}
__device__ __forceinline__ uint32_t half_trunc(float v) {        // PathTracer.js:42-51: truncate, flush below the normal range, saturate
    const uint32_t u = __float_as_uint(v);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t e = (int32_t)((u >> 23) & 0xffu) - 112;
    if (e <= 0) return sign;
    if (e >= 31) return sign | 0x7c00u;
    return sign | ((uint32_t)e << 10) | ((u >> 13) & 0x3ffu);
}

// bottom-up, one launch per level (deepest first): subtree size and bounds (PathTracer.js:640-661)
__global__ __launch_bounds__(256) void collapse_up_kernel(const uint32_t* __restrict__ bvh2, const uint32_t* __restrict__ node2, const uint4* __restrict__ child_pos,
                                                           uint32_t* __restrict__ subtree, uint32_t* __restrict__ bnd, uint32_t level_begin, uint32_t level_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= level_count) return;
    const uint32_t p = level_begin + i;
    const uint4 cp = child_pos[p];
    const uint32_t c[4] = {cp.x, cp.y, cp.z, cp.w};
    const uint32_t node = node2[p];
    uint32_t w0, w1, w2, size = 1u;
    if (leaf2(bvh2, node)) {
        const uint32_t* q = bvh2 + 1 + (size_t)node * 6;
        w0 = q[0]; w1 = q[1]; w2 = q[2];
    } else {
        const float inf = __uint_as_float(0x7f800000u);
        float mn[3] = {inf, inf, inf}, mx[3] = {-inf, -inf, -inf};
        for (int s = 0; s < 4; ++s) {
            if (c[s] == kNone) continue;
            size += subtree[c[s]];
            const uint32_t* b = bnd + (size_t)c[s] * 3;
            const uint32_t b0 = b[0], b1 = b[1], b2 = b[2];
            mn[0] = js_min_f(mn[0], half_exact(b0 & 0xffffu)); mn[1] = js_min_f(mn[1], half_exact(b0 >> 16)); mn[2] = js_min_f(mn[2], half_exact(b1 & 0xffffu));
            mx[0] = js_max_f(mx[0], half_exact(b1 >> 16)); mx[1] = js_max_f(mx[1], half_exact(b2 & 0xffffu)); mx[2] = js_max_f(mx[2], half_exact(b2 >> 16));
        }
        w0 = half_trunc(mn[0]) | (half_trunc(mn[1]) << 16);
        w1 = half_trunc(mn[2]) | (half_trunc(mx[0]) << 16);
        w2 = half_trunc(mx[1]) | (half_trunc(mx[2]) << 16);
    }
    subtree[p] = size;
    uint32_t* o = bnd + (size_t)p * 3;
    o[0] = w0; o[1] = w1; o[2] = w2;
}

// top-down, one launch per level: a node knows its id, numbers its children and writes its record (renderer.wgsl:91-111 layout)
__global__ __launch_bounds__(256) void collapse_down_kernel(const uint32_t* __restrict__ bvh2, const uint32_t* __restrict__ node2, const uint4* __restrict__ child_pos,
                                                             const uint32_t* __restrict__ subtree, const uint32_t* __restrict__ bnd, uint32_t* __restrict__ ids,
                                                             uint32_t* __restrict__ bvh4, uint32_t level_begin, uint32_t level_count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= level_count) return;
    const uint32_t p = level_begin + i;
    const uint32_t id = (p == 0u) ? 0u : ids[p];
    const uint4 cp = child_pos[p];
    const uint32_t c[4] = {cp.x, cp.y, cp.z, cp.w};
    uint32_t* rec = bvh4 + 1 + (size_t)id * 8;
    const uint32_t* b = bnd + (size_t)p * 3;
    rec[0] = b[0]; rec[1] = b[1]; rec[2] = b[2];
    uint32_t next = id + 1u;
    for (int s = 0; s < 4; ++s) {
        if (c[s] == kNone) { rec[3 + s] = kInvalidRef; continue; }
        ids[c[s]] = next; rec[3 + s] = next;
        next += subtree[c[s]];
    }
    const uint32_t node = node2[p];
    rec[7] = leaf2(bvh2, node) ? bvh2[1 + (size_t)node * 6 + 5] : 0u;
    if (p == 0u) bvh4[0] = subtree[0];
}

// ------------------------------------------------------------------------------------
// Wide nodes (pt::build_wide_bvh for a BVH this library built itself): internal nodes keep
// their pre-order among themselves = exclusive prefix sum of "is internal" over the node ids
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void internal_flags_kernel(const uint32_t* __restrict__ bvh4, uint32_t m, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) flags[i] = (bvh4[1 + (size_t)i * 8 + 7] & kLeaf) ? 0u : 1u;
}

__device__ __forceinline__ bool box_degenerate(uint32_t w0, uint32_t w1, uint32_t w2) {     // renderer.wgsl:244, 291: any(mn > mx)
    return half_exact(w0 & 0xffffu) > half_exact(w1 >> 16) || half_exact(w0 >> 16) > half_exact(w2 & 0xffffu) || half_exact(w1 & 0xffffu) > half_exact(w2 >> 16);
}

__global__ __launch_bounds__(256) void wide_nodes_kernel(const uint32_t* __restrict__ bvh4, uint32_t m, const uint32_t* __restrict__ wide_index, uint4* __restrict__ wide,
                                                          uint32_t num_tris, uint32_t node_base16) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint32_t* r = bvh4 + 1 + (size_t)i * 8;
    if (r[7] & kLeaf) return;
    uint32_t box[12], ref[4];
    for (int s = 0; s < 4; ++s) {
        box[3 * s] = kEmptyBox0; box[3 * s + 1] = kEmptyBox1; box[3 * s + 2] = kEmptyBox2; ref[s] = kInvalidRef;   // the inverted box no ray enters (pt_device.h::slab_sel)
        const uint32_t c = r[3 + s];
        if (c == kInvalidRef || c >= m) continue;
        const uint32_t* cr = bvh4 + 1 + (size_t)c * 8;
        const uint32_t w0 = cr[0], w1 = cr[1], w2 = cr[2];
        if (box_degenerate(w0, w1, w2)) { ref[s] = kDegenerateRef; continue; }   // fetched by the reference, entered by no ray
        box[3 * s] = w0; box[3 * s + 1] = w1; box[3 * s + 2] = w2;
        const uint32_t tri = cr[7] & 0x7fffffffu;
        ref[s] = (cr[7] & kLeaf) ? (kLeaf | (4u * (tri < num_tris ? tri : num_tris))) : node_base16 + 4u * wide_index[c];     // packed references (pt_host.h)
    }
    uint4* o = wide + (size_t)wide_index[i] * 4;
    o[0] = make_uint4(box[0], box[1], box[2], ref[0]);       // child-major (pt_host.h::WideNode): piece k = child k's box + reference
    o[1] = make_uint4(box[3], box[4], box[5], ref[1]);
    o[2] = make_uint4(box[6], box[7], box[8], ref[2]);
    o[3] = make_uint4(box[9], box[10], box[11], ref[3]);
}

inline uint32_t blocks(uint32_t n) { return (n + 255u) / 256u; }

} // namespace

size_t build_temp_bytes(uint32_t num_tris) {
    if (num_tris == 0) return 0;
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)num_tris, 0, 30);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)(2u * num_tris));
    return sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
}

hipError_t launch_tri_records(const float* tris9, uint32_t num_tris, float4* records, uint32_t* edge_max, hipStream_t stream) {
    if (num_tris == 0) return hipSuccess;
    hipLaunchKernelGGL(tri_records_kernel, dim3(blocks(num_tris)), dim3(256), 0, stream, tris9, num_tris, records, edge_max);
    return hipGetLastError();
}

hipError_t launch_morton_sort(const BuildBuffers& B, const float* tris9, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(build_init_kernel, dim3(1), dim3(256), 0, stream, B.bounds, B.counters, (uint32_t)kBuildCounters);
    hipLaunchKernelGGL(centroid_bounds_kernel, dim3(blocks(n) < 1024u ? blocks(n) : 1024u), dim3(256), 0, stream, tris9, n, B.bounds);
    hipLaunchKernelGGL(morton_kernel, dim3(blocks(n)), dim3(256), 0, stream, tris9, n, B.bounds, B.code_tmp, B.index_tmp);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    size_t bytes = B.temp_bytes;
    // stable LSD radix sort on the 30-bit code: triangles start in index order, so the result is the JS sort by (code, triangle)
    return hipcub::DeviceRadixSort::SortPairs(B.temp, bytes, (const uint32_t*)B.code_tmp, B.morton, (const uint32_t*)B.index_tmp, B.tri_index, (int)n, 0, 30, stream);
}

hipError_t collapse_on_device(const BuildBuffers& B, const uint32_t* bvh2, uint32_t num_tris, uint32_t* bvh4, uint32_t* num_nodes4, hipStream_t stream) {
    *num_nodes4 = 0;
    if (num_tris == 0) return hipSuccess;
    const uint32_t nn2 = 2u * num_tris - 1u;
    hipError_t e = hipMemsetAsync(B.node2, 0, sizeof(uint32_t), stream);      // level 0 = the BVH2 root (node 0)
    if (e != hipSuccess) return e;
    uint32_t level_off[kBuildCounters + 2];
    uint32_t levels = 0;
    level_off[0] = 0; level_off[1] = 1;
    for (;;) {
        const uint32_t begin = level_off[levels], count = level_off[levels + 1] - begin;
        hipLaunchKernelGGL(collapse_expand_kernel, dim3(blocks(count)), dim3(256), 0, stream, bvh2, nn2, B.node2, B.child_pos, begin, count, level_off[levels + 1], nn2, B.counters + levels);
        e = hipGetLastError(); if (e != hipSuccess) return e;
        e = hipMemcpyAsync(B.host_word, B.counters + levels, sizeof(uint32_t), hipMemcpyDeviceToHost, stream); if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream); if (e != hipSuccess) return e;
        const uint32_t next = *B.host_word;
        ++levels;
        if (next == 0u) break;
        if (levels >= (uint32_t)kBuildCounters || (uint64_t)level_off[levels] + next > nn2) return hipErrorInvalidValue;   // not a tree of 2N-1 nodes
        level_off[levels + 1] = level_off[levels] + next;
    }
    for (uint32_t l = levels; l-- > 0u;)
        hipLaunchKernelGGL(collapse_up_kernel, dim3(blocks(level_off[l + 1] - level_off[l])), dim3(256), 0, stream, bvh2, B.node2, B.child_pos, B.subtree, B.bnd, level_off[l], level_off[l + 1] - level_off[l]);
    for (uint32_t l = 0; l < levels; ++l)
        hipLaunchKernelGGL(collapse_down_kernel, dim3(blocks(level_off[l + 1] - level_off[l])), dim3(256), 0, stream, bvh2, B.node2, B.child_pos, B.subtree, B.bnd, B.ids, bvh4, level_off[l], level_off[l + 1] - level_off[l]);
    *num_nodes4 = level_off[levels];
    return hipGetLastError();
}

hipError_t launch_internal_scan(const BuildBuffers& B, const uint32_t* bvh4, uint32_t num_nodes4, hipStream_t stream) {
    if (num_nodes4 == 0) return hipSuccess;
    // B.subtree / B.ids are free again after the collapse: flags and their exclusive prefix sum
    hipLaunchKernelGGL(internal_flags_kernel, dim3(blocks(num_nodes4)), dim3(256), 0, stream, bvh4, num_nodes4, B.subtree);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
    size_t bytes = B.temp_bytes;
    return hipcub::DeviceScan::ExclusiveSum(B.temp, bytes, (const uint32_t*)B.subtree, B.ids, (int)num_nodes4, stream);
}

hipError_t launch_wide_nodes(const BuildBuffers& B, const uint32_t* bvh4, uint32_t num_nodes4, uint4* wide, uint32_t num_tris, uint32_t node_base16, hipStream_t stream) {
    if (num_nodes4 == 0) return hipSuccess;
    hipLaunchKernelGGL(wide_nodes_kernel, dim3(blocks(num_nodes4)), dim3(256), 0, stream, bvh4, num_nodes4, B.ids, wide, num_tris, node_base16);
    return hipGetLastError();
}

} // namespace ptk
