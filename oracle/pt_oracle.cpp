// pt_oracle.cpp -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
//
// A plain, strict-f32 CPU restatement of the reference's hot path and of the
// scene-build steps that produce its inputs.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load this library; the product
// (raytracer-public_amd/) never links, imports or calls it.
//
// Every function cites the reference file:line it follows (paths relative to the
// reference checkout).  What the reference does not contain (bounce loop, NEE,
// Russian roulette, RNG -- SURVEY.md section 0.2) follows the build-defined spec in
// DESIGN.md section 4 and is marked "BUILD-DEFINED (parity unpinned)".
//
// Pinning status:
//   * f16 codec, Morton+sort, BVH2->BVH4 collapse: pinned against vectors generated
//     by importing the reference's own src/libs/PathTracer.js under Node
//     (tests/golden/gen_golden_js.js -> tests/golden/pathtracer_js_golden.json).
//   * BVH4_wide promotion: pinned against oracle/_ref/bvh4_wide_ref, the reference's
//     tests/test.cpp compiled where it lies (oracle/Makefile target `ref`).
//   * renderer.wgsl / BVHBuilder.wgsl restatements: the WGSL cannot execute in this
//     environment and the reference holds no golden image or known-answer test for
//     them -> "parity unpinned" beyond the hand-derived known answers in
//     tests/test_oracle_render.py.
//
// Floating-point contract: IEEE-754 binary32, round-to-nearest
// -even, no contraction except the explicit fma sites, correctly rounded / and sqrt,
// normalize(v) = v * (1 / sqrt(dot(v,v))), dot = ((x*x' + y*y') + z*z').
// Build with: g++ -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).

#include <cmath>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>

namespace {

constexpr uint32_t NODE2_STRIDE = 6;          // BVHBuilder.wgsl:5, PathTracer.js:8
constexpr uint32_t NODE4_STRIDE = 8;          // renderer.wgsl:10, PathTracer.js:13
constexpr uint32_t LEAF_FLAG    = 0x80000000u; // renderer.wgsl:11
constexpr uint32_t INVALID      = 0xFFFFFFFFu; // renderer.wgsl:12
constexpr float    INF_T        = 1e30f;       // renderer.wgsl:64 (finite sentinel)
constexpr int      STACK_MAX    = 64;          // renderer.wgsl:8
constexpr int      PACKET_W = 2, PACKET_H = 2, PACKET_SIZE = 4; // renderer.wgsl:4-6

// ---------------------------------------------------------------------------
// f16 codec
// ---------------------------------------------------------------------------

inline uint32_t f32_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_f32(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// PathTracer.js:16-40 (f16ToF32) == WGSL unpack2x16float: exact widening, subnormals kept.
float f16_to_f32(uint32_t h) {
    uint32_t s = (h & 0x8000u) << 16;
    int e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x03ffu;
    if (e == 0) {
        if (m == 0) return bits_f32(s);
        e = 1;
        while ((m & 0x0400u) == 0) { m <<= 1; e--; }
        m &= 0x03ffu;
    } else if (e == 31) {
        return bits_f32(s | 0x7f800000u | (m << 13));
    }
    return bits_f32(s | (uint32_t(e + 112) << 23) | (m << 13));
}

// PathTracer.js:42-51 (f32ToF16): TRUNCATING, flushes results with biased exponent
// <= 0 to signed zero, saturates exponent >= 31 to infinity (NaN payload dropped).
uint32_t f32_to_f16_trunc(float v) {
    uint32_t u = f32_bits(v);
    uint32_t s = (u >> 16) & 0x8000u;
    int e = int((u >> 23) & 0xff) - 112;
    uint32_t m = (u >> 13) & 0x03ffu;
    if (e <= 0) return s;
    if (e >= 31) return s | 0x7c00u;
    return s | (uint32_t(e) << 10) | m;
}

// WGSL pack2x16float (BVHBuilder.wgsl:65,99-101): rounding is implementation-defined in
// WGSL; this build pins round-to-nearest-even with subnormals.
uint32_t f32_to_f16_rtne(float v) {
    uint32_t u = f32_bits(v);
    uint32_t s = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) {                       // inf / nan
        return s | 0x7c00u | ((a > 0x7f800000u) ? 0x0200u : 0u);
    }
    if (a >= 0x477ff000u) return s | 0x7c00u;     // >= 65520 rounds to inf
    if (a < 0x33000001u) return s;                // <= 2^-25 rounds to zero (tie -> even = 0)
    int e = int(a >> 23) - 127;                   // unbiased
    uint32_t m = (a & 0x007fffffu) | 0x00800000u; // 24-bit significand
    int shift;                                    // bits to drop
    uint32_t he;
    if (e < -14) { shift = 13 + (-14 - e); he = 0; }   // subnormal half
    else         { shift = 13;             he = uint32_t(e + 15); }
    uint32_t q = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    // q carries the implicit bit for normals (bit 10); adding he<<10 - (1<<10) handles carry.
    uint32_t h = (he == 0) ? q : (((he - 1) << 10) + q);
    return s | h;
}

// BVHBuilder.wgsl:63-81 (incrementF16), iterations = 1.
float increment_f16(float value, bool up) {
    uint32_t bits = f32_to_f16_rtne(value) & 0xFFFFu;
    bool sign = (bits & 0x8000u) != 0u;
    uint32_t ord = sign ? ((~bits) & 0xFFFFu) : (bits ^ 0x8000u);
    ord = up ? (ord + 1u) : (ord - 1u);
    bool ordSign = (ord & 0x8000u) != 0u;
    uint32_t bits2 = ordSign ? (ord ^ 0x8000u) : ((~ord) & 0xFFFFu);
    return f16_to_f32(bits2 & 0xFFFFu);
}

inline uint32_t pack2x16_rtne(float a, float b) {
    return (f32_to_f16_rtne(a) & 0xFFFFu) | ((f32_to_f16_rtne(b) & 0xFFFFu) << 16);
}

// ---------------------------------------------------------------------------
// Scene build: Morton + sort (PathTracer.js:411-481), all arithmetic in double like JS
// ---------------------------------------------------------------------------

inline uint32_t expand_bits10(uint32_t v) {       // PathTracer.js:411-418
    v &= 1023u;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8))  & 0x0300f00fu;
    v = (v | (v << 4))  & 0x030c30c3u;
    v = (v | (v << 2))  & 0x09249249u;
    return v;
}
inline uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) { // PathTracer.js:420-425
    return (expand_bits10(x) << 2) | (expand_bits10(y) << 1) | expand_bits10(z);
}
inline double js_min(double a, double b) { return (b < a) ? b : a; }
inline double js_max(double a, double b) { return (b > a) ? b : a; }
inline int32_t js_to_int32_trunc(double d) {      // `(x) | 0` for finite |x| < 2^31; NaN -> 0
    if (!(d == d)) return 0;
    return int32_t(d);
}

void morton_sort(const float* t, uint32_t n, uint32_t* mortonSorted, uint32_t* triIndexSorted) {
    if (n == 0) return;
    double minX = 1e30, minY = 1e30, minZ = 1e30, maxX = -1e30, maxY = -1e30, maxZ = -1e30;
    for (uint32_t i = 0; i < n; i++) {            // PathTracer.js:436-444
        const float* b = t + size_t(i) * 9;
        double cx = (double(b[0]) + double(b[3]) + double(b[6])) / 3;
        double cy = (double(b[1]) + double(b[4]) + double(b[7])) / 3;
        double cz = (double(b[2]) + double(b[5]) + double(b[8])) / 3;
        minX = js_min(minX, cx); minY = js_min(minY, cy); minZ = js_min(minZ, cz);
        maxX = js_max(maxX, cx); maxY = js_max(maxY, cy); maxZ = js_max(maxZ, cz);
    }
    double dx = js_max(1e-20, maxX - minX), dy = js_max(1e-20, maxY - minY), dz = js_max(1e-20, maxZ - minZ);
    std::vector<uint64_t> keys(n);
    for (uint32_t i = 0; i < n; i++) {            // PathTracer.js:452-468
        const float* b = t + size_t(i) * 9;
        double cx = (double(b[0]) + double(b[3]) + double(b[6])) / 3;
        double cy = (double(b[1]) + double(b[4]) + double(b[7])) / 3;
        double cz = (double(b[2]) + double(b[5]) + double(b[8])) / 3;
        double nx = (cx - minX) / dx, ny = (cy - minY) / dy, nz = (cz - minZ) / dz;
        int32_t qx = std::max(0, std::min(1023, js_to_int32_trunc(nx * 1023)));
        int32_t qy = std::max(0, std::min(1023, js_to_int32_trunc(ny * 1023)));
        int32_t qz = std::max(0, std::min(1023, js_to_int32_trunc(nz * 1023)));
        uint32_t code = morton3d(uint32_t(qx), uint32_t(qy), uint32_t(qz));
        keys[i] = (uint64_t(code) << 32) | i;     // comparator (code, tri): PathTracer.js:470
    }
    std::sort(keys.begin(), keys.end());
    for (uint32_t i = 0; i < n; i++) {
        mortonSorted[i] = uint32_t(keys[i] >> 32);
        triIndexSorted[i] = uint32_t(keys[i]);
    }
}

// ---------------------------------------------------------------------------
// Scene build: LBVH2 (BVHBuilder.wgsl), sequential restatement of the two kernels
// ---------------------------------------------------------------------------

inline int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }

struct Lbvh2Ctx {
    const uint32_t* morton; int n;
    int delta(int i, int j) const {               // BVHBuilder.wgsl:134-149
        if (j < 0 || j >= n) return -1;
        uint32_t a = morton[i], b = morton[j], x = a ^ b;
        if (x == 0u) return 32 + clz32(uint32_t(i) ^ uint32_t(j));
        return clz32(x);
    }
};

// WGSL leaves min / max of (-0, +0) to the implementation; in the builder the choice is visible, because incrementF16 (BVHBuilder.wgsl:63-82) steps
// -0 up to +0 but +0 up to the smallest subnormal.  Pinned: -0 orders below +0 (IEEE 754-2019 minimum / maximum) -- what the GPUs' min / max
// instructions do (gfx950 v_min_f32 / v_max_f32 included), and what JS Math.min / Math.max do in the collapse (:644-645).  Found by
// tests/test_gpu_ingest.py: a mesh centred by normalizeMesh has vertices at +-1e-9 on both sides of 0.
inline float min_oz(float a, float b) { if (a < b) return a; if (b < a) return b; return std::signbit(a) ? a : b; }
inline float max_oz(float a, float b) { if (a > b) return a; if (b > a) return b; return std::signbit(a) ? b : a; }
inline float fmin3v(float a, float b, float c) { return min_oz(min_oz(a, b), c); }
inline float fmax3v(float a, float b, float c) { return max_oz(max_oz(a, b), c); }

void write_bounds2(uint32_t* bvh2, uint32_t node, const float mn[3], const float mx[3]) {
    // BVHBuilder.wgsl:83-102: widen by one f16 ULP outward in every component, then pack.
    uint32_t base = 1u + node * NODE2_STRIDE;
    float mnL[3], mxL[3];
    for (int k = 0; k < 3; k++) { mnL[k] = increment_f16(mn[k], false); mxL[k] = increment_f16(mx[k], true); }
    bvh2[base + 0] = pack2x16_rtne(mnL[0], mnL[1]);
    bvh2[base + 1] = pack2x16_rtne(mnL[2], mxL[0]);
    bvh2[base + 2] = pack2x16_rtne(mxL[1], mxL[2]);
}

void read_bounds2(const uint32_t* bvh2, uint32_t node, float mn[3], float mx[3]) {
    uint32_t base = 1u + node * NODE2_STRIDE;     // BVHBuilder.wgsl:104-114
    uint32_t a = bvh2[base], b = bvh2[base + 1], c = bvh2[base + 2];
    mn[0] = f16_to_f32(a & 0xFFFF); mn[1] = f16_to_f32(a >> 16); mn[2] = f16_to_f32(b & 0xFFFF);
    mx[0] = f16_to_f32(b >> 16);    mx[1] = f16_to_f32(c & 0xFFFF); mx[2] = f16_to_f32(c >> 16);
}

void build_lbvh2(const float* tris, uint32_t numTris, const uint32_t* morton,
                 const uint32_t* triIdx, uint32_t* bvh2) {
    if (numTris == 0) { bvh2[0] = 0; return; }
    const uint32_t numNodes2 = 2 * numTris - 1;
    bvh2[0] = numNodes2;                          // PathTracer.js:699
    const uint32_t internalCount = numTris - 1;
    std::vector<uint32_t> parent(numNodes2, 0u), flags(std::max(1u, internalCount), 0u);
    Lbvh2Ctx c{morton, int(numTris)};
    const int n = int(numTris);
    // Pass 1: buildInternal, BVHBuilder.wgsl:152-240
    for (uint32_t iU = 0; iU < internalCount; iU++) {
        int i = int(iU);
        int dLeft = c.delta(i, i - 1), dRight = c.delta(i, i + 1);
        int d = ((dRight - dLeft) > 0) ? 1 : -1;
        int deltaMin = c.delta(i, i - d);
        int lmax = 2;
        while (c.delta(i, i + lmax * d) > deltaMin) lmax <<= 1;
        int l = 0;
        for (int t = lmax >> 1; t > 0; t >>= 1)
            if (c.delta(i, i + (l + t) * d) > deltaMin) l += t;
        int j = i + l * d;
        int first = std::min(i, j), last = std::max(i, j);
        int deltaNode = c.delta(first, last);
        int split = first, step = last - first;
        while (step > 1) {
            step = (step + 1) >> 1;
            int newSplit = split + step;
            if (newSplit < last && c.delta(first, newSplit) > deltaNode) split = newSplit;
        }
        uint32_t leafBase = internalCount;
        uint32_t leftChild = (split == first) ? leafBase + uint32_t(split) : uint32_t(split);
        int rightIndex = split + 1;
        uint32_t rightChild = (rightIndex == last) ? leafBase + uint32_t(rightIndex) : uint32_t(rightIndex);
        uint32_t base = 1u + iU * NODE2_STRIDE;   // writeInternal2, :116-122
        bvh2[base + 3] = leftChild; bvh2[base + 4] = rightChild; bvh2[base + 5] = 0u;
        parent[leftChild] = iU; parent[rightChild] = iU;
        if (iU == 0u) parent[0] = INVALID;
        (void)n;
    }
    // Pass 2: buildLeaves + propagateUp, BVHBuilder.wgsl:242-306
    for (uint32_t leafId = 0; leafId < numTris; leafId++) {
        uint32_t nodeIndex = internalCount + leafId;
        uint32_t ti = triIdx[leafId];
        const float* b = tris + size_t(ti) * 9;   // getTriangleBoundsByTriIndex, :36-58
        float mn[3], mx[3];
        for (int k = 0; k < 3; k++) { mn[k] = fmin3v(b[k], b[3 + k], b[6 + k]); mx[k] = fmax3v(b[k], b[3 + k], b[6 + k]); }
        write_bounds2(bvh2, nodeIndex, mn, mx);   // writeLeaf2, :124-132
        uint32_t base = 1u + nodeIndex * NODE2_STRIDE;
        bvh2[base + 3] = 0u; bvh2[base + 4] = 0u; bvh2[base + 5] = LEAF_FLAG | (ti & 0x7FFFFFFFu);
        if (internalCount == 0) { parent[0] = INVALID; continue; }
        uint32_t node = nodeIndex;
        for (;;) {                                // propagateUp
            uint32_t p = parent[node];
            if (p == INVALID || p >= internalCount) break;
            uint32_t old = flags[p]++;
            if (old == 0u) break;
            uint32_t pBase = 1u + p * NODE2_STRIDE;
            float lmn[3], lmx[3], rmn[3], rmx[3], umn[3], umx[3];
            read_bounds2(bvh2, bvh2[pBase + 3], lmn, lmx);
            read_bounds2(bvh2, bvh2[pBase + 4], rmn, rmx);
            for (int k = 0; k < 3; k++) { umn[k] = min_oz(lmn[k], rmn[k]); umx[k] = max_oz(lmx[k], rmx[k]); }
            write_bounds2(bvh2, p, umn, umx);
            node = p;
        }
    }
}

// ---------------------------------------------------------------------------
// Scene build: greedy collapse LBVH2 -> BVH4 (PathTracer.js:506-667)
// ---------------------------------------------------------------------------

// JS Math.min / Math.max (PathTracer.js:644-645): -0 orders below +0.
inline double js_math_min(double a, double b) { if (a < b) return a; if (b < a) return b; return std::signbit(a) ? a : b; }
inline double js_math_max(double a, double b) { if (a > b) return a; if (b > a) return b; return std::signbit(a) ? b : a; }

struct Collapse {
    const uint32_t* b2; std::vector<uint32_t> out;
    bool is_leaf2(uint32_t i) const { return (b2[1 + i * NODE2_STRIDE + 5] & LEAF_FLAG) != 0; }
    uint32_t emit() {                             // emitNode4, :572-576
        uint32_t idx = uint32_t((out.size() - 1) / NODE4_STRIDE);
        out.insert(out.end(), NODE4_STRIDE, 0u);
        return idx;
    }
    uint32_t build4(uint32_t node2) {             // :590-658
        uint32_t idx4 = emit();
        uint32_t off = 1 + node2 * NODE2_STRIDE;
        if (is_leaf2(node2)) {
            uint32_t base = 1 + idx4 * NODE4_STRIDE;
            out[base + 0] = b2[off + 0]; out[base + 1] = b2[off + 1]; out[base + 2] = b2[off + 2];
            out[base + 3] = out[base + 4] = out[base + 5] = out[base + 6] = INVALID;
            out[base + 7] = b2[off + 5];
            return idx4;
        }
        uint32_t kids[5]; int nk = 2;
        kids[0] = b2[off + 3]; kids[1] = b2[off + 4];
        bool changed = true;                      // greedy treelet collapse, :609-621
        while (nk < 4 && changed) {
            changed = false;
            for (int i = 0; i < nk; i++) {
                uint32_t k = kids[i];
                if (k != INVALID && !is_leaf2(k)) {
                    uint32_t ko = 1 + k * NODE2_STRIDE;
                    for (int m = nk; m > i + 1; m--) kids[m] = kids[m - 1]; // splice(i,1,l,r)
                    kids[i] = b2[ko + 3]; kids[i + 1] = b2[ko + 4];
                    nk++; changed = true; break;
                }
            }
        }
        uint32_t cIdx[4] = {INVALID, INVALID, INVALID, INVALID};
        double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int i = 0; i < 4; i++) {
            if (i >= nk) continue;
            uint32_t ci = build4(kids[i]);
            cIdx[i] = ci;
            uint32_t cb = 1 + ci * NODE4_STRIDE;
            uint32_t c0 = out[cb], c1 = out[cb + 1], c2 = out[cb + 2];
            double bmn[3] = {f16_to_f32(c0 & 0xFFFF), f16_to_f32(c0 >> 16), f16_to_f32(c1 & 0xFFFF)};
            double bmx[3] = {f16_to_f32(c1 >> 16), f16_to_f32(c2 & 0xFFFF), f16_to_f32(c2 >> 16)};
            for (int k = 0; k < 3; k++) { mn[k] = js_math_min(mn[k], bmn[k]); mx[k] = js_math_max(mx[k], bmx[k]); }
        }
        uint32_t base = 1 + idx4 * NODE4_STRIDE;  // encodeBounds (truncating), :560-566, 651
        out[base + 0] = f32_to_f16_trunc(float(mn[0])) | (f32_to_f16_trunc(float(mn[1])) << 16);
        out[base + 1] = f32_to_f16_trunc(float(mn[2])) | (f32_to_f16_trunc(float(mx[0])) << 16);
        out[base + 2] = f32_to_f16_trunc(float(mx[1])) | (f32_to_f16_trunc(float(mx[2])) << 16);
        out[base + 3] = cIdx[0]; out[base + 4] = cIdx[1]; out[base + 5] = cIdx[2]; out[base + 6] = cIdx[3];
        out[base + 7] = 0u;
        return idx4;
    }
};

// ---------------------------------------------------------------------------
// Hot path: renderer.wgsl
// ---------------------------------------------------------------------------

struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline V3 normalize(V3 v) { float inv = 1.0f / std::sqrt(dot(v, v)); return v * inv; }
inline float wmin(float a, float b) { return (b < a) ? b : a; }   // WGSL min
inline float wmax(float a, float b) { return (b > a) ? b : a; }   // WGSL max

inline V3 rotate_by_quat(V3 v, const float q[4]) {   // renderer.wgsl:66-72
    V3 u = v3(q[0], q[1], q[2]); float s = q[3];
    V3 uv = cross(u, v), uuv = cross(u, uv);
    return v3(std::fmaf(2.0f, std::fmaf(s, uv.x, uuv.x), v.x),
              std::fmaf(2.0f, std::fmaf(s, uv.y, uuv.y), v.y),
              std::fmaf(2.0f, std::fmaf(s, uv.z, uuv.z), v.z));
}
inline V3 safe_inv_dir(V3 d) {                       // renderer.wgsl:74-80
    return v3(std::fabs(d.x) > 1e-8f ? 1.0f / d.x : INF_T,
              std::fabs(d.y) > 1e-8f ? 1.0f / d.y : INF_T,
              std::fabs(d.z) > 1e-8f ? 1.0f / d.z : INF_T);
}

struct Node4 { V3 mn, mx; uint32_t c[4]; uint32_t triIndex; bool isLeaf; };

struct SceneView {
    const float* tris; const uint32_t* bvh; uint32_t numTris; uint32_t numNodes;
    // BUILD-DEFINED brute-force scene (config C1: triangles + analytic spheres, no BVH)
    const float* spheres = nullptr; uint32_t numSpheres = 0; bool brute = false;
    Node4 node(uint32_t index) const {               // getBVHNode4, renderer.wgsl:91-111
        uint32_t base = 1u + index * NODE4_STRIDE;
        uint32_t a = bvh[base], b = bvh[base + 1], c = bvh[base + 2];
        Node4 n;
        n.mn = v3(f16_to_f32(a & 0xFFFF), f16_to_f32(a >> 16), f16_to_f32(b & 0xFFFF));
        n.mx = v3(f16_to_f32(b >> 16), f16_to_f32(c & 0xFFFF), f16_to_f32(c >> 16));
        for (int k = 0; k < 4; k++) n.c[k] = bvh[base + 3 + k];
        uint32_t meta = bvh[base + 7];
        n.isLeaf = (meta & LEAF_FLAG) != 0u; n.triIndex = meta & 0x7FFFFFFFu;
        return n;
    }
};
inline bool degenerate(const Node4& n) { return n.mn.x > n.mx.x || n.mn.y > n.mx.y || n.mn.z > n.mx.z; }

struct Stats {   // mirrored in tests as ctypes.Structure
    uint64_t rays_closest, rays_shadow;
    uint64_t nodes_examined;   // read-once count
    uint64_t tris_tested;
    uint64_t node_fetches_ref; // getBVHNode4 calls as the reference performs them (double fetch)
    uint64_t stack_drops;      // pushes dropped at STACK_MAX (renderer.wgsl:337)
    uint64_t max_stack;        // deepest sp+1 seen
    uint64_t samples;
};

struct Packet { V3 origin[PACKET_SIZE], dir[PACKET_SIZE], invdir[PACKET_SIZE]; };
struct HitPacket { float t[PACKET_SIZE]; V3 normal[PACKET_SIZE]; bool hit[PACKET_SIZE]; uint32_t tri[PACKET_SIZE]; };
struct Mask { bool m[PACKET_SIZE]; };
inline bool any_lane(const Mask& m) { return m.m[0] || m.m[1] || m.m[2] || m.m[3]; }

// intersectAABBPacketMask, renderer.wgsl:121-169
void aabb_packet(const Packet& p, V3 mn, V3 mx, const Mask& in, const float bestT[PACKET_SIZE], Mask& out, float& outMinT) {
    float minT = INF_T; bool anyHit = false;
    if (mn.x > mx.x || mn.y > mx.y || mn.z > mx.z) { for (int i = 0; i < PACKET_SIZE; i++) out.m[i] = false; outMinT = INF_T; return; }
    for (int i = 0; i < PACKET_SIZE; i++) {
        if (!in.m[i]) { out.m[i] = false; continue; }
        V3 t1 = (mn - p.origin[i]) * p.invdir[i], t2 = (mx - p.origin[i]) * p.invdir[i];
        float tmin = wmax(wmax(wmin(t1.x, t2.x), wmin(t1.y, t2.y)), wmin(t1.z, t2.z));
        float tmax = wmin(wmin(wmax(t1.x, t2.x), wmax(t1.y, t2.y)), wmax(t1.z, t2.z));
        bool hit = (tmax >= wmax(tmin, 0.0f)) && (tmin < bestT[i]);
        out.m[i] = hit;
        if (hit) { minT = wmin(minT, tmin); anyHit = true; }
    }
    outMinT = anyHit ? minT : INF_T;
}

// intersectTrianglePacket, renderer.wgsl:171-208.  anyhit != nullptr -> BUILD-DEFINED early-out flag.
void tri_packet(const Packet& p, V3 v0, V3 v1, V3 v2, V3 triN, uint32_t ti, const Mask& lanes, HitPacket& out) {
    const float eps = 1e-7f;
    V3 e1 = v1 - v0, e2 = v2 - v0;
    for (int i = 0; i < PACKET_SIZE; i++) {
        if (!lanes.m[i]) continue;
        V3 pv = cross(p.dir[i], e2);
        float det = dot(e1, pv);
        if (std::fabs(det) < eps) continue;
        float invDet = 1.0f / det;
        V3 s = p.origin[i] - v0;
        float u = invDet * dot(s, pv);
        if (u < 0.0f || u > 1.0f) continue;
        V3 q = cross(s, e1);
        float v = invDet * dot(p.dir[i], q);
        if (v < 0.0f || (u + v) > 1.0f) continue;
        float t = invDet * dot(e2, q);
        if (t > eps && t < out.t[i]) { out.t[i] = t; out.normal[i] = triN; out.hit[i] = true; out.tri[i] = ti; }
    }
}

// traverseBVH4Packet, renderer.wgsl:210-346 -- literal restatement, incl. the double node
// fetch, the nearest-child swap and the silent push drop at STACK_MAX.
// anyhit=true is the BUILD-DEFINED shadow-ray variant: identical order, returns as soon as
// every initially active lane has a hit (with 1 active lane: at the first accepted hit).
HitPacket traverse_packet(const SceneView& sc, const Packet& packet, const Mask& initMask, bool anyhit, Stats* st) {
    HitPacket out;
    for (int i = 0; i < PACKET_SIZE; i++) { out.t[i] = INF_T; out.normal[i] = v3(0, 0, 0); out.hit[i] = false; out.tri[i] = INVALID; }
    if (sc.numNodes == 0u || sc.numTris == 0u || !any_lane(initMask)) return out;
    uint32_t stack[STACK_MAX]; Mask stackMask[STACK_MAX]; int sp = 0;
    stack[0] = 0u; stackMask[0] = initMask;
    if (st) { st->nodes_examined += 1; if (st->max_stack < 1) st->max_stack = 1; }
    for (;;) {
        if (sp < 0) break;
        uint32_t nodeIndex = stack[sp]; Mask laneMask = stackMask[sp]; sp -= 1;
        Node4 node = sc.node(nodeIndex);
        if (st) st->node_fetches_ref++;
        if (degenerate(node)) continue;
        Mask hitMask; float nodeMinT;
        aabb_packet(packet, node.mn, node.mx, laneMask, out.t, hitMask, nodeMinT);
        if (!any_lane(hitMask)) continue;
        if (node.isLeaf) {
            uint32_t ti = node.triIndex;
            if (ti < sc.numTris) {
                const float* b = sc.tris + size_t(ti) * 9;   // getTriangle, :82-89
                V3 v0 = v3(b[0], b[1], b[2]), v1 = v3(b[3], b[4], b[5]), v2 = v3(b[6], b[7], b[8]);
                V3 triN = normalize(cross(v1 - v0, v2 - v0));  // :269
                if (st) st->tris_tested++;
                tri_packet(packet, v0, v1, v2, triN, ti, hitMask, out);
                if (anyhit) {
                    bool all = true;
                    for (int i = 0; i < PACKET_SIZE; i++) if (initMask.m[i] && !out.hit[i]) all = false;
                    if (all) return out;
                }
            }
            continue;
        }
        uint32_t childIdx[4] = {node.c[0], node.c[1], node.c[2], node.c[3]};
        float childDist[4]; Mask childMasks[4]; uint32_t childCount = 0;
        for (uint32_t c = 0; c < 4; c++) {
            uint32_t ci = childIdx[c];
            if (ci == INVALID || ci >= sc.numNodes) continue;
            Node4 child = sc.node(ci);
            if (st) { st->node_fetches_ref++; st->nodes_examined++; }
            if (degenerate(child)) continue;
            Mask cmask; float cminT;
            aabb_packet(packet, child.mn, child.mx, hitMask, out.t, cmask, cminT);
            if (any_lane(cmask)) { childIdx[childCount] = ci; childDist[childCount] = cminT; childMasks[childCount] = cmask; childCount++; }
        }
        uint32_t best = 0;
        for (uint32_t i = 1; i < childCount; i++) best = (childDist[i] < childDist[best]) ? i : best;
        if (best != 0u) {
            std::swap(childIdx[0], childIdx[best]); std::swap(childDist[0], childDist[best]); std::swap(childMasks[0], childMasks[best]);
        }
        for (int i = int(childCount) - 1; i >= 0; i--) {
            if (sp + 1 < STACK_MAX) { sp++; stack[sp] = childIdx[i]; stackMask[sp] = childMasks[i]; }
            else if (st) st->stack_drops++;
        }
        if (st && uint64_t(sp + 1) > st->max_stack) st->max_stack = uint64_t(sp + 1);
    }
    return out;
}

struct Hit1 { float t; V3 normal; bool hit; uint32_t tri; };

constexpr uint32_t SPHERE_FLAG = 0x40000000u;

// BUILD-DEFINED (parity unpinned): closest / any hit without a BVH.  Triangles in index order with the
// reference's Moller-Trumbore (renderer.wgsl:171-208), then spheres (x,y,z,r) in index order:
// a = d.d, hb = oc.d, disc = hb*hb - a*(oc.oc - r*r); t = nearer root if > eps else farther; strict t < best.
Hit1 brute_single(const SceneView& sc, V3 o, V3 d, bool anyhit, Stats* st) {
    Hit1 h{INF_T, v3(0, 0, 0), false, INVALID};
    const float eps = 1e-7f;
    for (uint32_t ti = 0; ti < sc.numTris; ti++) {
        const float* b = sc.tris + size_t(ti) * 9;
        V3 v0 = v3(b[0], b[1], b[2]), e1 = v3(b[3], b[4], b[5]) - v0, e2 = v3(b[6], b[7], b[8]) - v0;
        if (st) st->tris_tested++;
        V3 pv = cross(d, e2); float det = dot(e1, pv);
        if (std::fabs(det) < eps) continue;
        float invDet = 1.0f / det; V3 s = o - v0;
        float u = invDet * dot(s, pv); if (u < 0.0f || u > 1.0f) continue;
        V3 q = cross(s, e1); float v = invDet * dot(d, q); if (v < 0.0f || (u + v) > 1.0f) continue;
        float t = invDet * dot(e2, q);
        if (t > eps && t < h.t) { h.t = t; h.normal = normalize(cross(e1, e2)); h.hit = true; h.tri = ti; if (anyhit) return h; }
    }
    for (uint32_t si = 0; si < sc.numSpheres; si++) {
        const float* sp = sc.spheres + size_t(si) * 4;
        V3 c = v3(sp[0], sp[1], sp[2]); float r = sp[3];
        V3 oc = o - c;
        float a = dot(d, d), hb = dot(oc, d), cc = dot(oc, oc) - r * r;
        float disc = hb * hb - a * cc;
        if (disc < 0.0f) continue;
        float sq = std::sqrt(disc);
        float t0 = (-hb - sq) / a, t1 = (-hb + sq) / a;
        float t = (t0 > eps) ? t0 : t1;
        if (t > eps && t < h.t) {
            h.t = t; V3 p = o + d * t; h.normal = normalize(p - c); h.hit = true; h.tri = SPHERE_FLAG | si;
            if (anyhit) return h;
        }
    }
    return h;
}

// Single-ray traversal == traverse_packet with exactly one active lane (lane 0).
Hit1 traverse_single(const SceneView& sc, V3 o, V3 d, V3 inv, bool anyhit, Stats* st) {
    if (sc.brute) return brute_single(sc, o, d, anyhit, st);
    Packet p; Mask m;
    for (int i = 0; i < PACKET_SIZE; i++) { p.origin[i] = v3(0, 0, 0); p.dir[i] = v3(0, 0, -1); p.invdir[i] = v3(INF_T, INF_T, INF_T); m.m[i] = false; }
    p.origin[0] = o; p.dir[0] = d; p.invdir[0] = inv; m.m[0] = true;
    HitPacket h = traverse_packet(sc, p, m, anyhit, st);
    return Hit1{h.t[0], h.normal[0], h.hit[0], h.tri[0]};
}

inline V3 light_dir() { return normalize(v3(1.0f, 1.5f, 1.0f)); }     // renderer.wgsl:349
inline V3 shade_ref(V3 n) {                                           // renderer.wgsl:348-353
    V3 base = v3(0.9f, 0.7f, 0.3f);
    float ndotl = wmax(dot(n, light_dir()), 0.0f);
    return base * (0.15f + ndotl);
}

struct Params {   // mirrored in tests as ctypes.Structure
    uint32_t width, height;
    float focal, aspect;                // UBO resolution.zw (PathTracer.js:761-770)
    float cam_pos[3]; uint32_t num_tris;
    float cam_quat[4];
    uint32_t frame;                     // frameCounter.x (unused by the reference shader)
    uint32_t mode;                      // 0 literal 2x2 packet; 1 single-ray reference shading; 2 path tracing
    uint32_t spp, max_bounces, seed;
    uint32_t x0, y0, x1, y1;            // half-open pixel rectangle to render
    uint32_t step_x, step_y;            // pixel subsampling stride (modes 1,2)
    uint32_t accum_frames;              // mode 2: number of consecutive frames accumulated (0 or 1 = single frame)
};

// primary ray for pixel position (fx, fy) in pixels, renderer.wgsl:387-395
inline void primary_ray(const Params& P, float fx, float fy, V3& o, V3& d, V3& inv) {
    float resx = float(P.width), resy = float(P.height);
    float uvx = fx / resx, uvy = fy / resy;
    float px = std::fmaf(uvx, 2.0f, -1.0f), py = std::fmaf(uvy, 2.0f, -1.0f);
    d = normalize(v3(px * P.aspect, py, -P.focal));
    d = rotate_by_quat(d, P.cam_quat);
    o = v3(P.cam_pos[0], P.cam_pos[1], P.cam_pos[2]);
    inv = safe_inv_dir(d);
}

// ---- BUILD-DEFINED (parity unpinned): RNG + sampling, DESIGN.md section 4 ----------------
inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
inline uint32_t sample_key(uint32_t seed, uint32_t pixel, uint32_t sidx) {
    uint32_t h = mix32(seed + 0x9E3779B9u);
    h = mix32(h ^ pixel);
    h = mix32(h ^ sidx);
    return h;
}
inline float rnd(uint32_t key, uint32_t bounce, uint32_t dim) {
    uint32_t h = mix32(key ^ (bounce * 8u + dim + 1u) * 0x9E3779B1u);
    return float(h >> 8) * (1.0f / 16777216.0f);
}
// cos/sin of 2*pi*u, u in [0,1): quadrant reduction + fixed fmaf Horner polynomials.
inline void sincos_2pi(float u, float& c, float& s) {
    float q = u * 4.0f;
    float kf = std::floor(q + 0.5f);
    float f = q - kf;                         // [-0.5, 0.5]
    float y = f * 1.57079632679489662f;
    float y2 = y * y;
    float sp = std::fmaf(y2, 2.7557319e-6f, -1.9841270e-4f);
    sp = std::fmaf(y2, sp, 8.3333333e-3f);
    sp = std::fmaf(y2, sp, -1.6666667e-1f);
    sp = std::fmaf(y2, sp, 1.0f);
    float sy = y * sp;
    float cp = std::fmaf(y2, -2.7557319e-7f, 2.4801587e-5f);
    cp = std::fmaf(y2, cp, -1.3888889e-3f);
    cp = std::fmaf(y2, cp, 4.1666667e-2f);
    cp = std::fmaf(y2, cp, -0.5f);
    float cy = std::fmaf(y2, cp, 1.0f);
    int k = int(kf) & 3;
    if (k == 0)      { c = cy;  s = sy;  }
    else if (k == 1) { c = -sy; s = cy;  }
    else if (k == 2) { c = -cy; s = -sy; }
    else             { c = sy;  s = -cy; }
}
inline V3 cosine_dir(V3 n, float u1, float u2) {
    float c, s; sincos_2pi(u2, c, s);
    float r = std::sqrt(u1);
    float lx = r * c, ly = r * s, lz = std::sqrt(1.0f - u1);
    float sign = std::copysign(1.0f, n.z);         // Duff et al. 2017 orthonormal basis
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    V3 t  = v3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    V3 bt = v3(b, sign + n.y * n.y * a, -n.y);
    return (t * lx + bt * ly) + n * lz;
}

constexpr float EPS_ORIGIN = 1e-4f;   // ray-origin offset along the face-forwarded normal
constexpr float BG_PRIMARY = 0.01f;   // renderer.wgsl:410 miss colour
constexpr float SKY_AMBIENT = 0.15f;  // the 0.15 ambient term of renderer.wgsl:352, as sky radiance
constexpr uint32_t RR_START = 2;      // first bounce index at which Russian roulette applies

V3 path_sample(const SceneView& sc, const Params& P, uint32_t px, uint32_t py, uint32_t sidx, Stats* st) {
    const V3 base = v3(0.9f, 0.7f, 0.3f);
    const V3 L = light_dir();
    uint32_t pixel = py * P.width + px;
    uint32_t key = sample_key(P.seed, pixel, sidx);
    V3 o, d, inv;
    primary_ray(P, float(px) + rnd(key, 0, 0), float(py) + rnd(key, 0, 1), o, d, inv);
    V3 rad = v3(0, 0, 0), T = v3(1, 1, 1);
    for (uint32_t bounce = 0;; bounce++) {
        if (st) st->rays_closest++;
        Hit1 h = traverse_single(sc, o, d, inv, false, st);
        if (!h.hit) { float e = (bounce == 0) ? BG_PRIMARY : SKY_AMBIENT; rad = rad + T * e; break; }
        V3 hp = o + d * h.t;
        V3 nf = (dot(h.normal, d) < 0.0f) ? h.normal : v3(-h.normal.x, -h.normal.y, -h.normal.z);
        V3 so = hp + nf * EPS_ORIGIN;
        float ndl = dot(nf, L);
        if (ndl > 0.0f) {                                       // next-event estimation
            if (st) st->rays_shadow++;
            Hit1 sh = traverse_single(sc, so, L, safe_inv_dir(L), true, st);
            if (!sh.hit) rad = rad + (T * base) * ndl;
        }
        if (bounce >= P.max_bounces) break;
        T = T * base;
        if (bounce >= RR_START) {                               // Russian roulette
            float p = wmax(wmax(T.x, T.y), T.z);
            if (rnd(key, bounce, 4) >= p) break;
            T = T * (1.0f / p);
        }
        d = cosine_dir(nf, rnd(key, bounce, 2), rnd(key, bounce, 3));
        o = so; inv = safe_inv_dir(d);
    }
    return rad;
}

} // namespace

// ===========================================================================
// C ABI (ctypes)
// ===========================================================================
extern "C" {

uint32_t orc_f32_to_f16_trunc(float v) { return f32_to_f16_trunc(v); }
uint32_t orc_f32_to_f16_rtne(float v) { return f32_to_f16_rtne(v); }
float    orc_f16_to_f32(uint32_t h) { return f16_to_f32(h & 0xFFFFu); }
float    orc_increment_f16(float v, int up) { return increment_f16(v, up != 0); }

void orc_morton_sort(const float* tris, uint32_t n, uint32_t* mortonSorted, uint32_t* triIndexSorted) {
    morton_sort(tris, n, mortonSorted, triIndexSorted);
}

// bvh2 must hold 1 + 6*(2n-1) words (PathTracer.js:227-232)
void orc_build_lbvh2(const float* tris, uint32_t n, const uint32_t* morton, const uint32_t* triIdx, uint32_t* bvh2) {
    build_lbvh2(tris, n, morton, triIdx, bvh2);
}

// out must hold 1 + 8*(2n-1) words; returns numNodes4 (PathTracer.js:506-667)
uint32_t orc_collapse_bvh4(const uint32_t* bvh2, uint32_t numTris, uint32_t* out) {
    if (numTris == 0) { out[0] = 0; return 0; }
    Collapse c; c.b2 = bvh2; c.out.reserve(1 + size_t(2 * numTris - 1) * NODE4_STRIDE); c.out.push_back(0u);
    c.build4(0);
    uint32_t n4 = uint32_t((c.out.size() - 1) / NODE4_STRIDE);
    c.out[0] = n4;
    std::memcpy(out, c.out.data(), c.out.size() * 4);
    return n4;
}

// tests/test.cpp:106-196: one-level grandchild promotion, node indices preserved.
// out must hold 1 + 8*numNodes2 words.
void orc_bvh4_wide(const uint32_t* bvh2, uint32_t* out) {
    uint32_t numNodes2 = bvh2[0];
    out[0] = numNodes2;
    auto is_leaf2 = [&](uint32_t n) { return n >= numNodes2 ? true : (bvh2[1 + n * NODE2_STRIDE + 5] & LEAF_FLAG) != 0; };
    for (uint32_t n = 0; n < numNodes2; n++) {
        size_t o2 = 1 + size_t(n) * NODE2_STRIDE, o4 = 1 + size_t(n) * NODE4_STRIDE;
        out[o4] = bvh2[o2]; out[o4 + 1] = bvh2[o2 + 1]; out[o4 + 2] = bvh2[o2 + 2];
        uint32_t meta = bvh2[o2 + 5];
        if (meta & LEAF_FLAG) {
            out[o4 + 3] = out[o4 + 4] = out[o4 + 5] = out[o4 + 6] = INVALID; out[o4 + 7] = meta;
        } else {
            uint32_t kids[4]; uint32_t k = 0;
            auto push = [&](uint32_t c) { if (k < 4) kids[k++] = c; };
            auto promote = [&](uint32_t c) {
                if (c == INVALID) return;
                if (is_leaf2(c)) push(c);
                else { size_t off = 1 + size_t(c) * NODE2_STRIDE; push(bvh2[off + 3]); push(bvh2[off + 4]); }
            };
            promote(bvh2[o2 + 3]); promote(bvh2[o2 + 4]);
            while (k < 4) kids[k++] = INVALID;
            out[o4 + 3] = kids[0]; out[o4 + 4] = kids[1]; out[o4 + 5] = kids[2]; out[o4 + 6] = kids[3]; out[o4 + 7] = 0;
        }
    }
}

// Renders into rgba (W*H*4 f32, row py, column px; py = 0 <-> p.y = -1, no flip).
// Pixels outside the requested rectangle / subsample grid are left untouched.
// tri_ids (optional, W*H u32): closest-hit triangle of the primary ray in modes 0/1.
int orc_render(const Params* Pp, const float* tris, const uint32_t* bvh4, float* rgba, uint32_t* tri_ids, Stats* st) {
    const Params& P = *Pp;
    SceneView sc{tris, bvh4, P.num_tris, bvh4[0]};
    if (st) std::memset(st, 0, sizeof(Stats));
    uint32_t x1 = std::min(P.x1, P.width), y1 = std::min(P.y1, P.height);
    if (P.mode == 0) {   // literal main(), renderer.wgsl:355-413, one "thread" per 2x2 packet
        for (uint32_t gy = P.y0 / PACKET_H; gy * PACKET_H < y1; gy++)
        for (uint32_t gx = P.x0 / PACKET_W; gx * PACKET_W < x1; gx++) {
            uint32_t bx = gx * PACKET_W, by = gy * PACKET_H;
            if (bx >= P.width || by >= P.height) continue;
            Packet packet; Mask lane;
            for (int i = 0; i < PACKET_SIZE; i++) {
                uint32_t px = bx + uint32_t(i % PACKET_W), py = by + uint32_t(i / PACKET_W);
                bool in = px < P.width && py < P.height;
                lane.m[i] = in;
                if (!in) { packet.origin[i] = v3(0, 0, 0); packet.dir[i] = v3(0, 0, -1); packet.invdir[i] = v3(INF_T, INF_T, INF_T); continue; }
                primary_ray(P, float(px) + 0.5f, float(py) + 0.5f, packet.origin[i], packet.dir[i], packet.invdir[i]);
                if (st) { st->rays_closest++; st->samples++; }
            }
            HitPacket hits = traverse_packet(sc, packet, lane, false, st);
            for (int i = 0; i < PACKET_SIZE; i++) {
                if (!lane.m[i]) continue;
                uint32_t px = bx + uint32_t(i % PACKET_W), py = by + uint32_t(i / PACKET_W);
                V3 col = hits.hit[i] ? shade_ref(hits.normal[i]) : v3(0.01f, 0.01f, 0.01f);
                float* o = rgba + (size_t(py) * P.width + px) * 4;
                o[0] = col.x; o[1] = col.y; o[2] = col.z; o[3] = 1.0f;
                if (tri_ids) tri_ids[size_t(py) * P.width + px] = hits.tri[i];
            }
        }
        return 0;
    }
    uint32_t sx = P.step_x ? P.step_x : 1, sy = P.step_y ? P.step_y : 1;
    for (uint32_t py = P.y0; py < y1; py += sy)
    for (uint32_t px = P.x0; px < x1; px += sx) {
        float* o = rgba + (size_t(py) * P.width + px) * 4;
        if (P.mode == 1) {
            V3 ro, rd, ri; primary_ray(P, float(px) + 0.5f, float(py) + 0.5f, ro, rd, ri);
            if (st) { st->rays_closest++; st->samples++; }
            Hit1 h = traverse_single(sc, ro, rd, ri, false, st);
            V3 col = h.hit ? shade_ref(h.normal) : v3(0.01f, 0.01f, 0.01f);
            o[0] = col.x; o[1] = col.y; o[2] = col.z; o[3] = 1.0f;
            if (tri_ids) tri_ids[size_t(py) * P.width + px] = h.tri;
        } else {
            // progressive accumulation (BUILD-DEFINED): each frame's samples are summed from zero in
            // sample order, frame sums are added to the running total in frame order
            const uint32_t frames = P.accum_frames ? P.accum_frames : 1u;
            V3 sum = v3(0, 0, 0); float count = 0.0f;
            for (uint32_t f = 0; f < frames; f++) {
                V3 fsum = v3(0, 0, 0);
                for (uint32_t s = 0; s < P.spp; s++) {
                    fsum = fsum + path_sample(sc, P, px, py, (P.frame + f) * P.spp + s, st);
                    if (st) st->samples++;
                }
                sum = sum + fsum; count = count + float(P.spp);
            }
            float invn = 1.0f / count;
            o[0] = sum.x * invn; o[1] = sum.y * invn; o[2] = sum.z * invn; o[3] = 1.0f;
        }
    }
    return 0;
}

// BUILD-DEFINED config C1: brute-force scene of triangles + spheres (no BVH), modes 1 and 2 only.
int orc_render_brute(const Params* Pp, const float* tris, const float* spheres, uint32_t numSpheres, float* rgba, Stats* st) {
    Params P = *Pp;
    if (P.mode == 0) return 1;
    const uint32_t one = 1;   // non-empty dummy BVH word so the shared render loop takes the traced path
    (void)one;
    if (st) std::memset(st, 0, sizeof(Stats));
    SceneView sc{tris, nullptr, P.num_tris, 1u};
    sc.spheres = spheres; sc.numSpheres = numSpheres; sc.brute = true;
    uint32_t x1 = std::min(P.x1, P.width), y1 = std::min(P.y1, P.height);
    uint32_t sx = P.step_x ? P.step_x : 1, sy = P.step_y ? P.step_y : 1;
    for (uint32_t py = P.y0; py < y1; py += sy)
    for (uint32_t px = P.x0; px < x1; px += sx) {
        float* o = rgba + (size_t(py) * P.width + px) * 4;
        if (P.mode == 1) {
            V3 ro, rd, ri; primary_ray(P, float(px) + 0.5f, float(py) + 0.5f, ro, rd, ri);
            if (st) { st->rays_closest++; st->samples++; }
            Hit1 h = traverse_single(sc, ro, rd, ri, false, st);
            V3 col = h.hit ? shade_ref(h.normal) : v3(0.01f, 0.01f, 0.01f);
            o[0] = col.x; o[1] = col.y; o[2] = col.z; o[3] = 1.0f;
        } else {
            V3 sum = v3(0, 0, 0);
            for (uint32_t s = 0; s < P.spp; s++) { sum = sum + path_sample(sc, P, px, py, P.frame * P.spp + s, st); if (st) st->samples++; }
            float invn = 1.0f / (0.0f + float(P.spp));
            o[0] = sum.x * invn; o[1] = sum.y * invn; o[2] = sum.z * invn; o[3] = 1.0f;
        }
    }
    return 0;
}

// single-ray probe for unit tests: returns hit flag, writes t / normal / tri
int orc_trace_ray(const float* tris, const uint32_t* bvh4, uint32_t numTris, const float o[3], const float d[3],
                  int anyhit, float* t, float* n, uint32_t* tri) {
    SceneView sc{tris, bvh4, numTris, bvh4[0]};
    V3 dd = v3(d[0], d[1], d[2]);
    Hit1 h = traverse_single(sc, v3(o[0], o[1], o[2]), dd, safe_inv_dir(dd), anyhit != 0, nullptr);
    *t = h.t; n[0] = h.normal.x; n[1] = h.normal.y; n[2] = h.normal.z; *tri = h.tri;
    return h.hit ? 1 : 0;
}

// single-lane probes of the two intersection routines (pinned against the reference's own Python statement of the same
// arithmetic, tests/test.py:64-99, by tests/test_ref_py_intersect.py)
// intersectAABBPacketMask with one active lane, renderer.wgsl:121-169: returns the lane's hit flag, writes the box's tmin (INF on a miss)
int orc_slab(const float o[3], const float inv[3], const float mn[3], const float mx[3], float best, float* tmin) {
    Packet p; Mask m, out; float bestT[PACKET_SIZE];
    for (int i = 0; i < PACKET_SIZE; i++) { p.origin[i] = v3(0, 0, 0); p.dir[i] = v3(0, 0, -1); p.invdir[i] = v3(INF_T, INF_T, INF_T); m.m[i] = false; bestT[i] = INF_T; }
    p.origin[0] = v3(o[0], o[1], o[2]); p.invdir[0] = v3(inv[0], inv[1], inv[2]); m.m[0] = true; bestT[0] = best;
    aabb_packet(p, v3(mn[0], mn[1], mn[2]), v3(mx[0], mx[1], mx[2]), m, bestT, out, *tmin);
    return out.m[0] ? 1 : 0;
}
// intersectTrianglePacket with one active lane, renderer.wgsl:171-208: returns the hit flag, writes t (unchanged `best` on a miss)
int orc_moller_trumbore(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3], float best, float* t) {
    Packet p; Mask m; HitPacket h;
    for (int i = 0; i < PACKET_SIZE; i++) { p.origin[i] = v3(0, 0, 0); p.dir[i] = v3(0, 0, -1); p.invdir[i] = v3(INF_T, INF_T, INF_T); m.m[i] = false; h.t[i] = best; h.hit[i] = false; h.tri[i] = INVALID; h.normal[i] = v3(0, 0, 0); }
    p.origin[0] = v3(o[0], o[1], o[2]); p.dir[0] = v3(d[0], d[1], d[2]); m.m[0] = true;
    tri_packet(p, v3(v0[0], v0[1], v0[2]), v3(v1[0], v1[1], v1[2]), v3(v2[0], v2[1], v2[2]), v3(0, 0, 1), 0u, m, h);
    *t = h.t[0];
    return h.hit[0] ? 1 : 0;
}
// rotateVectorByQuat, renderer.wgsl:66-72 (quaternion xyzw)
void orc_rotate_by_quat(const float v[3], const float q[4], float out[3]) { V3 r = rotate_by_quat(v3(v[0], v[1], v[2]), q); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_safe_inv_dir(const float d[3], float out[3]) { V3 r = safe_inv_dir(v3(d[0], d[1], d[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; }

float orc_rnd(uint32_t seed, uint32_t pixel, uint32_t sidx, uint32_t bounce, uint32_t dim) { return rnd(sample_key(seed, pixel, sidx), bounce, dim); }
void orc_sincos_2pi(float u, float* c, float* s) { sincos_2pi(u, *c, *s); }
void orc_cosine_dir(const float n[3], float u1, float u2, float out[3]) { V3 r = cosine_dir(v3(n[0], n[1], n[2]), u1, u2); out[0] = r.x; out[1] = r.y; out[2] = r.z; }

// tonemapper.wgsl:24-41: Reinhard + gamma 1/2.2 with the vertical flip of the full-screen
// triangle (uv.y = 0 at the bottom of the canvas; row 0 of the canvas is its top).
// in: W*H*4 f32 (the renderer's rgba8unorm-quantised output if quantize != 0), out: W*H*4 u8
// x^(1/2.2) for x in [0, 1], BUILD-DEFINED pinned f32 evaluation (WGSL leaves pow's precision implementation-defined, tonemapper.wgsl:36):
// x = m 2^e, m in [sqrt(1/2), sqrt(2)); ln m = 2 atanh((m-1)/(m+1)) as an odd polynomial (fmaf); z = log2(x) / 2.2; 2^z = 2^floor(z) times a
// degree-7 polynomial in the fraction (fmaf).  The HIP tonemap kernel evaluates the same sequence.
static float pow_1_2_2(float x) {
    if (!(x > 1.17549435e-38f)) return 0.0f;
    if (x >= 1.0f) return 1.0f;
    uint32_t bits; std::memcpy(&bits, &x, 4);
    int e = int(bits >> 23) - 127;
    uint32_t mb = (bits & 0x007fffffu) | 0x3f800000u; float m; std::memcpy(&m, &mb, 4);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float s = (m - 1.0f) / (m + 1.0f), s2 = s * s;
    float p = std::fmaf(s2, 0.11111111f, 0.14285715f);
    p = std::fmaf(s2, p, 0.2f);
    p = std::fmaf(s2, p, 0.33333334f);
    p = std::fmaf(s2, p, 1.0f);
    const float ln_m = 2.0f * s * p;
    const float z = (float(e) + ln_m * 1.44269504f) * 0.45454547f;
    const float kf = std::floor(z), r = z - kf;
    float q = std::fmaf(r, 1.5252734e-5f, 1.5403530e-4f);
    q = std::fmaf(r, q, 1.3333558e-3f);
    q = std::fmaf(r, q, 9.6181291e-3f);
    q = std::fmaf(r, q, 5.5504109e-2f);
    q = std::fmaf(r, q, 2.4022651e-1f);
    q = std::fmaf(r, q, 6.9314718e-1f);
    q = std::fmaf(r, q, 1.0f);
    const int k = int(kf);
    if (k < -126) return 0.0f;
    uint32_t sb = uint32_t(k + 127) << 23; float sc; std::memcpy(&sc, &sb, 4);
    return q * sc;
}
float orc_pow_1_2_2(float x) { return pow_1_2_2(x); }

void orc_tonemap(const float* rgba, uint32_t W, uint32_t H, int quantize, uint8_t* out) {
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) {
        uint32_t sy = H - 1 - y;
        const float* p = rgba + (size_t(sy) * W + x) * 4;
        uint8_t* o = out + (size_t(y) * W + x) * 4;
        for (int k = 0; k < 3; k++) {
            float c = p[k];
            if (quantize) { float q = c < 0.f ? 0.f : (c > 1.f ? 1.f : c); c = std::floor(q * 255.0f + 0.5f) / 255.0f; }
            float m = c / (c + 1.0f);
            float g = pow_1_2_2(m);
            float q = g < 0.f ? 0.f : (g > 1.f ? 1.f : g);
            o[k] = uint8_t(std::floor(q * 255.0f + 0.5f));
        }
        o[3] = 255;
    }
}

} // extern "C"
