// pt_host.cpp -- host-side scene build of libmi355pt: the steps the reference runs in
// JavaScript on the CPU (Morton codes + sort, LBVH2 -> BVH4 collapse) or as an offline tool
// (BVH4_wide), plus the re-layout of the reference's buffers into the device formats the
// HIP kernels read.  Strict f32/f64 (-ffp-contract=off); no HIP calls.
#include "pt_host.h"

#include <cmath>
#include <cstring>
#include <algorithm>

namespace pt {

// ------------------------------------------------------------------------------------
// f16
// ------------------------------------------------------------------------------------
static inline uint32_t bits_of(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float float_of(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

float half_to_float(uint32_t h) {
    const uint32_t sign = (h & 0x8000u) << 16;
    const uint32_t mag = h & 0x7fffu;
    if (mag >= 0x7c00u) return float_of(sign | 0x7f800000u | ((mag & 0x3ffu) << 13));   // inf / nan
    if (mag >= 0x0400u) return float_of(sign | ((mag + (112u << 10)) << 13));            // normal: rebias 15 -> 127
    // zero / subnormal: value = mag * 2^-24, exact in f32
    const float v = float(mag) * 5.9604644775390625e-8f;
    return float_of(bits_of(v) | sign);
}

uint32_t float_to_half_trunc(float v) {   // PathTracer.js:42-51
    const uint32_t u = bits_of(v);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const int32_t e = int32_t((u >> 23) & 0xffu) - 112;
    if (e <= 0) return sign;                       // below the f16 normal range -> signed zero
    if (e >= 31) return sign | 0x7c00u;            // saturate (also inf / nan)
    return sign | (uint32_t(e) << 10) | ((u >> 13) & 0x3ffu);
}

uint32_t float_to_half_rtne(float v) {
    const uint32_t u = bits_of(v);
    const uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return sign | 0x7e00u;    // nan
    if (a >= 0x477ff000u) return sign | 0x7c00u;   // rounds to / is infinity
    if (a < 0x38800000u) {                         // below 2^-14: subnormal half
        if (a < 0x33000000u) return sign;          // < 2^-25
        // align the 24-bit significand so that bit 0 of the result is 2^-24
        const uint32_t sig = (a & 0x007fffffu) | 0x00800000u;
        const uint32_t shift = 126u - (a >> 23);   // 14..24
        const uint32_t q = sig >> shift;
        const uint32_t rest = sig & ((1u << shift) - 1u);
        const uint32_t halfway = 1u << (shift - 1u);
        return sign | (q + ((rest > halfway) || (rest == halfway && (q & 1u)) ? 1u : 0u));
    }
    // normal: add rounding bias in the f32 domain, then rebias the exponent
    const uint32_t lsb = (a >> 13) & 1u;
    a += 0x0fffu + lsb;
    return sign | ((a - (112u << 23)) >> 13);
}

// ------------------------------------------------------------------------------------
// Morton codes + sort (PathTracer.js:411-481).  Doubles throughout, like the JS.
// Sort key is (code, triangle index); triangles start in index order, so a stable sort on
// the 30-bit code alone gives the same permutation -> 3-pass LSD radix sort.
// ------------------------------------------------------------------------------------
static inline uint32_t spread10(uint32_t v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
static inline uint32_t quantize1023(double x) {    // max(0, min(1023, (x*1023)|0))
    const double s = x * 1023;
    if (!(s == s)) return 0;                       // NaN | 0 == 0
    if (s >= 1023.0) return 1023;
    if (s <= 0.0) return 0;
    return uint32_t(int32_t(s));                   // truncation toward zero
}

void morton_codes_sorted(const float* tris, uint32_t n, uint32_t* morton, uint32_t* tri_index) {
    if (n == 0) return;
    std::vector<double> cen(size_t(n) * 3);
    double lo[3] = {1e30, 1e30, 1e30}, hi[3] = {-1e30, -1e30, -1e30};
    for (uint32_t t = 0; t < n; ++t) {
        const float* p = tris + size_t(t) * 9;
        for (int k = 0; k < 3; ++k) {
            const double c = (double(p[k]) + double(p[3 + k]) + double(p[6 + k])) / 3;
            cen[size_t(t) * 3 + k] = c;
            if (c < lo[k]) lo[k] = c;
            if (c > hi[k]) hi[k] = c;
        }
    }
    double ext[3];
    for (int k = 0; k < 3; ++k) { const double d = hi[k] - lo[k]; ext[k] = d > 1e-20 ? d : 1e-20; }
    std::vector<uint32_t> code(n), idxA(n), idxB(n);
    for (uint32_t t = 0; t < n; ++t) {
        const uint32_t qx = quantize1023((cen[size_t(t) * 3 + 0] - lo[0]) / ext[0]);
        const uint32_t qy = quantize1023((cen[size_t(t) * 3 + 1] - lo[1]) / ext[1]);
        const uint32_t qz = quantize1023((cen[size_t(t) * 3 + 2] - lo[2]) / ext[2]);
        code[t] = (spread10(qx) << 2) | (spread10(qy) << 1) | spread10(qz);
        idxA[t] = t;
    }
    uint32_t* src = idxA.data(); uint32_t* dst = idxB.data();
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass * 10;
        uint32_t hist[1025] = {0};
        for (uint32_t i = 0; i < n; ++i) hist[((code[src[i]] >> shift) & 0x3ffu) + 1]++;
        for (int b = 0; b < 1024; ++b) hist[b + 1] += hist[b];
        for (uint32_t i = 0; i < n; ++i) dst[hist[(code[src[i]] >> shift) & 0x3ffu]++] = src[i];
        std::swap(src, dst);
    }
    for (uint32_t i = 0; i < n; ++i) { tri_index[i] = src[i]; morton[i] = code[src[i]]; }
}

// ------------------------------------------------------------------------------------
// Greedy collapse LBVH2 -> BVH4 (PathTracer.js:506-667), iterative:
//   pass 1 numbers the BVH4 nodes in DFS pre-order (children in slot order) and records the
//          greedy child sets; pass 2 walks the indices backwards (children always have larger
//          pre-order indices than their parent) and unions the already-final child boxes.
// ------------------------------------------------------------------------------------
static inline float js_min_f(float a, float b) { if (a < b) return a; if (b < a) return b; return std::signbit(a) ? a : b; }
static inline float js_max_f(float a, float b) { if (a > b) return a; if (b > a) return b; return std::signbit(a) ? b : a; }

bool collapse_to_bvh4(const uint32_t* bvh2, uint32_t num_tris, std::vector<uint32_t>& out, std::string& err) {
    out.clear();
    if (num_tris == 0) { out.push_back(0u); return true; }
    const uint32_t nn2 = 2 * num_tris - 1;
    auto word = [&](uint32_t node, uint32_t k) { return bvh2[1 + size_t(node) * kNode2Stride + k]; };
    auto leaf = [&](uint32_t node) { return (word(node, 5) & kLeafFlag) != 0u; };
    out.reserve(1 + size_t(nn2) * kNode4Stride);
    out.push_back(0u);
    struct Todo { uint32_t node2; uint32_t parent4; uint32_t slot; };
    std::vector<Todo> todo;
    todo.push_back({0u, kInvalid, 0u});
    uint32_t count4 = 0;
    while (!todo.empty()) {
        const Todo cur = todo.back(); todo.pop_back();
        if (cur.node2 >= nn2) { err = "BVH2 child index out of range"; return false; }
        if (count4 >= nn2) { err = "BVH2 is not a tree (more BVH4 nodes than BVH2 nodes)"; return false; }
        const uint32_t id = count4++;
        const size_t base = out.size();
        out.resize(base + kNode4Stride, 0u);
        if (cur.parent4 != kInvalid) out[1 + size_t(cur.parent4) * kNode4Stride + 3 + cur.slot] = id;
        if (leaf(cur.node2)) {
            out[base + 0] = word(cur.node2, 0); out[base + 1] = word(cur.node2, 1); out[base + 2] = word(cur.node2, 2);
            out[base + 3] = out[base + 4] = out[base + 5] = out[base + 6] = kInvalid;
            out[base + 7] = word(cur.node2, 5);
            continue;
        }
        // greedy: repeatedly replace the first internal entry by its two children until 4 entries
        uint32_t kid[4]; uint32_t nk = 2;
        kid[0] = word(cur.node2, 3); kid[1] = word(cur.node2, 4);
        for (;;) {
            if (nk >= 4) break;
            uint32_t pos = nk;
            for (uint32_t i = 0; i < nk; ++i) {
                if (kid[i] >= nn2) { err = "BVH2 child index out of range"; return false; }
                if (!leaf(kid[i])) { pos = i; break; }
            }
            if (pos == nk) break;
            const uint32_t k = kid[pos];
            for (uint32_t m = nk; m > pos + 1; --m) kid[m] = kid[m - 1];
            kid[pos] = word(k, 3); kid[pos + 1] = word(k, 4);
            ++nk;
        }
        out[base + 3] = out[base + 4] = out[base + 5] = out[base + 6] = kInvalid;
        out[base + 7] = 0u;
        for (uint32_t i = nk; i-- > 0;) todo.push_back({kid[i], id, i});   // slot 0 is numbered first
    }
    out[0] = count4;
    for (uint32_t id = count4; id-- > 0;) {
        const size_t base = 1 + size_t(id) * kNode4Stride;
        if (out[base + 7] & kLeafFlag) continue;
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int s = 0; s < 4; ++s) {
            const uint32_t c = out[base + 3 + s];
            if (c == kInvalid) continue;
            const size_t cb = 1 + size_t(c) * kNode4Stride;
            const uint32_t w0 = out[cb], w1 = out[cb + 1], w2 = out[cb + 2];
            const float cmn[3] = {half_to_float(w0 & 0xffffu), half_to_float(w0 >> 16), half_to_float(w1 & 0xffffu)};
            const float cmx[3] = {half_to_float(w1 >> 16), half_to_float(w2 & 0xffffu), half_to_float(w2 >> 16)};
            for (int k = 0; k < 3; ++k) { mn[k] = js_min_f(mn[k], cmn[k]); mx[k] = js_max_f(mx[k], cmx[k]); }
        }
        out[base + 0] = float_to_half_trunc(mn[0]) | (float_to_half_trunc(mn[1]) << 16);
        out[base + 1] = float_to_half_trunc(mn[2]) | (float_to_half_trunc(mx[0]) << 16);
        out[base + 2] = float_to_half_trunc(mx[1]) | (float_to_half_trunc(mx[2]) << 16);
    }
    return true;
}

// ------------------------------------------------------------------------------------
// BVH4_wide (tests/test.cpp:106-196): each internal node adopts its grandchildren (or the
// child itself when that child is a leaf); node indices and bounds are those of the BVH2.
// ------------------------------------------------------------------------------------
bool promote_to_bvh4_wide(const uint32_t* bvh2, uint64_t words, std::vector<uint32_t>& out, std::string& err) {
    if (words < 1) { err = "empty BVH2 buffer"; return false; }
    const uint32_t nn2 = bvh2[0];
    if (words < 1 + uint64_t(nn2) * kNode2Stride) { err = "BVH2 buffer shorter than its node count"; return false; }
    out.assign(1 + size_t(nn2) * kNode4Stride, 0u);
    out[0] = nn2;
    for (uint32_t n = 0; n < nn2; ++n) {
        const uint32_t* s = bvh2 + 1 + size_t(n) * kNode2Stride;
        uint32_t* d = out.data() + 1 + size_t(n) * kNode4Stride;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        if (s[5] & kLeafFlag) { d[3] = d[4] = d[5] = d[6] = kInvalid; d[7] = s[5]; continue; }
        uint32_t got = 0;
        for (int side = 0; side < 2; ++side) {
            const uint32_t c = s[3 + side];
            if (c == kInvalid) continue;
            const bool cleaf = (c >= nn2) || (bvh2[1 + size_t(c) * kNode2Stride + 5] & kLeafFlag);
            if (cleaf) { if (got < 4) d[3 + got++] = c; }
            else {
                const uint32_t* cs = bvh2 + 1 + size_t(c) * kNode2Stride;
                if (got < 4) d[3 + got++] = cs[3];
                if (got < 4) d[3 + got++] = cs[4];
            }
        }
        while (got < 4) d[3 + got++] = kInvalid;
        d[7] = 0u;
    }
    return true;
}

// ------------------------------------------------------------------------------------
// Device layouts
// ------------------------------------------------------------------------------------
static inline bool box_degenerate(uint32_t w0, uint32_t w1, uint32_t w2) {
    // any(mn > mx), renderer.wgsl:133,244,291 (false when a NaN is involved, like WGSL's >)
    return half_to_float(w0 & 0xffffu) > half_to_float(w1 >> 16) ||
           half_to_float(w0 >> 16) > half_to_float(w2 & 0xffffu) ||
           half_to_float(w1 & 0xffffu) > half_to_float(w2 >> 16);
}

bool build_wide_bvh(const uint32_t* bvh4, uint64_t words, uint32_t num_tris, uint32_t node_base16, WideBvh& out, std::string& err) {
    out = WideBvh();
    if (words < 1) { err = "empty BVH buffer"; return false; }
    const uint32_t m = bvh4[0];
    out.num_nodes4 = m;
    if (m == 0) return true;
    if (words < 1 + uint64_t(m) * kNode4Stride) { err = "BVH buffer shorter than its node count"; return false; }
    auto rec = [&](uint32_t i) { return bvh4 + 1 + size_t(i) * kNode4Stride; };
    // Reachability walk from the root; assigns wide indices to reachable internal nodes in
    // DFS pre-order so that a parent's first child record follows it in memory.
    std::vector<uint32_t> wide_index(m, kInvalid);
    std::vector<uint8_t> seen(m, 0);
    std::vector<uint32_t> stack;
    std::vector<uint32_t> order;   // reachable internal nodes in pre-order
    stack.push_back(0u); seen[0] = 1;
    while (!stack.empty()) {
        const uint32_t i = stack.back(); stack.pop_back();
        const uint32_t* r = rec(i);
        if (r[7] & kLeafFlag) continue;
        wide_index[i] = uint32_t(order.size());
        order.push_back(i);
        for (int s = 3; s >= 0; --s) {
            const uint32_t c = r[3 + s];
            if (c == kInvalid || c >= m) continue;          // skipped by every ray (renderer.wgsl:288)
            if (seen[c]) { err = "BVH node reachable twice (not a tree)"; return false; }
            seen[c] = 1;
            stack.push_back(c);
        }
    }
    const uint32_t* r0 = rec(0);
    out.root_box[0] = r0[0]; out.root_box[1] = r0[1]; out.root_box[2] = r0[2];
    out.root_degenerate = box_degenerate(r0[0], r0[1], r0[2]);
    out.root_ref = (r0[7] & kLeafFlag) ? packed_leaf_ref(r0[7] & 0x7fffffffu, num_tris) : node_base16;
    out.nodes.resize(order.size());
    for (size_t w = 0; w < order.size(); ++w) {
        const uint32_t* r = rec(order[w]);
        WideNode& wn = out.nodes[w];
        for (int s = 0; s < 4; ++s) {
            const uint32_t c = r[3 + s];
            WideNode::Child& ch = wn.child[s];
            ch.box[0] = kEmptyBox0; ch.box[1] = kEmptyBox1; ch.box[2] = kEmptyBox2;   // the inverted box (+inf, -inf) no ray enters
            ch.ref = kInvalid;
            if (c == kInvalid || c >= m) continue;
            const uint32_t* cr = rec(c);
            if (box_degenerate(cr[0], cr[1], cr[2])) { ch.ref = kDegenerate; continue; }   // renderer.wgsl:291: fetched, then skipped
            ch.box[0] = cr[0]; ch.box[1] = cr[1]; ch.box[2] = cr[2];
            ch.ref = (cr[7] & kLeafFlag) ? packed_leaf_ref(cr[7] & 0x7fffffffu, num_tris) : node_base16 + 4u * wide_index[c];
        }
    }
    return true;
}

void build_tri_records(const float* tris, uint32_t n, TriRecord* out) {
    for (uint32_t t = 0; t < n; ++t) {
        const float* p = tris + size_t(t) * 9;
        TriRecord& r = out[t];
        float e1[3], e2[3];
        for (int k = 0; k < 3; ++k) { e1[k] = p[3 + k] - p[k]; e2[k] = p[6 + k] - p[k]; }
        const float cx = e1[1] * e2[2] - e1[2] * e2[1];
        const float cy = e1[2] * e2[0] - e1[0] * e2[2];
        const float cz = e1[0] * e2[1] - e1[1] * e2[0];
        const float inv = 1.0f / std::sqrt((cx * cx + cy * cy) + cz * cz);
        const float n[3] = {cx * inv, cy * inv, cz * inv};
        for (int k = 0; k < 3; ++k) { r.axis[k][0] = p[k]; r.axis[k][1] = e1[k]; r.axis[k][2] = e2[k]; r.axis[k][3] = 0.0f; r.n[k] = n[k]; }
        r.n[3] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------
// Tiles: 8x8 pixels; tile (tx,ty) belongs to rank (tx + ty) % count.  A rank's tiles are
// listed row-major; the list index is the tile's slot in the rank's compact buffer.
// ------------------------------------------------------------------------------------
void tile_list(uint32_t width, uint32_t height, uint32_t rank, uint32_t count, std::vector<uint32_t>& tiles) {
    tiles.clear();
    if (count == 0) count = 1;
    const uint32_t tx_n = (width + kTile - 1) / kTile, ty_n = (height + kTile - 1) / kTile;
    for (uint32_t ty = 0; ty < ty_n; ++ty)
        for (uint32_t tx = 0; tx < tx_n; ++tx)
            if ((tx + ty) % count == rank) tiles.push_back(ty * tx_n + tx);
}

uint32_t tile_count_of(uint32_t width, uint32_t height, uint32_t rank, uint32_t count) {
    if (count == 0) count = 1;
    const uint32_t tx_n = (width + kTile - 1) / kTile, ty_n = (height + kTile - 1) / kTile;
    uint32_t n = 0;
    for (uint32_t ty = 0; ty < ty_n; ++ty) {
        const uint32_t first = (rank + count - ty % count) % count;
        if (first < tx_n) n += (tx_n - first + count - 1) / count;
    }
    return n;
}

// the rank's tiles inside the tile rectangle rect = {tx0, ty0, tx1, ty1} (packed tile shares, pt_kernels.hip)
uint32_t rect_tile_count_of(uint32_t rank, uint32_t count, const uint32_t rect[4]) {
    if (count == 0) count = 1;
    if (rect[2] <= rect[0] || rect[3] <= rect[1]) return 0;
    uint32_t n = 0;
    for (uint32_t ty = rect[1]; ty < rect[3]; ++ty) {
        const uint32_t first = rect[0] + (rank + count - ((ty + rect[0]) % count)) % count;
        if (first < rect[2]) n += (rect[2] - first + count - 1) / count;
    }
    return n;
}

// ------------------------------------------------------------------------------------
// Procedural stand-in scenes
// ------------------------------------------------------------------------------------
namespace {

inline uint32_t hash_u32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
inline double hash01(uint32_t a, uint32_t b, uint32_t c, uint32_t seed) {
    return double(hash_u32(hash_u32(hash_u32(a + 0x9E3779B9u * seed) ^ b) ^ c) >> 8) / 16777216.0;
}
// periodic value noise on a ku x kv lattice, u,v in [0,1)
double value_noise(double u, double v, uint32_t ku, uint32_t kv, uint32_t oct, uint32_t seed) {
    const double fu = u * ku, fv = v * kv;
    const double iu = std::floor(fu), iv = std::floor(fv);
    double a = fu - iu, b = fv - iv;
    a = a * a * (3 - 2 * a); b = b * b * (3 - 2 * b);
    const uint32_t u0 = uint32_t(iu) % ku, u1 = (u0 + 1) % ku, v0 = uint32_t(iv) % kv, v1 = (v0 + 1) % kv;
    const double h00 = hash01(u0, v0, oct, seed), h10 = hash01(u1, v0, oct, seed);
    const double h01 = hash01(u0, v1, oct, seed), h11 = hash01(u1, v1, oct, seed);
    return (h00 * (1 - a) + h10 * a) * (1 - b) + (h01 * (1 - a) + h11 * a) * b;
}

struct D3 { double x, y, z; };
inline D3 operator+(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline D3 operator-(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline D3 operator*(D3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline D3 crossd(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double dotd(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 normd(D3 a) { const double l = std::sqrt(dotd(a, a)); return l > 0 ? a * (1.0 / l) : D3{0, 0, 1}; }

// "normalize cube" of Scene.js:104-165: centre on the bounding-box centre, scale 2/maxDim
void normalize_cube(std::vector<D3>& v) {
    D3 lo = v[0], hi = v[0];
    for (const D3& p : v) {
        lo.x = std::min(lo.x, p.x); lo.y = std::min(lo.y, p.y); lo.z = std::min(lo.z, p.z);
        hi.x = std::max(hi.x, p.x); hi.y = std::max(hi.y, p.y); hi.z = std::max(hi.z, p.z);
    }
    const D3 c = {(lo.x + hi.x) * 0.5, (lo.y + hi.y) * 0.5, (lo.z + hi.z) * 0.5};
    const double scale = 2.0 / std::max(hi.x - lo.x, std::max(hi.y - lo.y, hi.z - lo.z));
    for (D3& p : v) p = (p - c) * scale;
}

struct TriList {
    float* out; uint32_t cap; uint32_t n = 0;
    void add(D3 a, D3 b, D3 c) {
        if (n >= cap) return;
        float* p = out + size_t(n) * 9;
        p[0] = float(a.x); p[1] = float(a.y); p[2] = float(a.z);
        p[3] = float(b.x); p[4] = float(b.y); p[5] = float(b.z);
        p[6] = float(c.x); p[7] = float(c.y); p[8] = float(c.z);
        ++n;
    }
};

// Dragon-class: a thick trefoil-knot tube with a swelling body and multi-octave "scale"
// bumps: closed, concave, self-occluding.  a x b quad grid -> 2ab triangles; the remainder up
// to num_tris is made by centroid-splitting evenly spaced triangles (+2 each), a last odd
// triangle by one edge-midpoint split (+1).
bool dragon_class(uint32_t seed, uint32_t num_tris, float* out, std::string& err) {
    if (num_tris < 24) { err = "dragon-class scene needs at least 24 triangles"; return false; }
    uint32_t b = uint32_t(std::floor(std::sqrt(double(num_tris) / 2.0 / 12.0)));
    if (b < 3) b = 3;
    uint32_t a = num_tris / 2 / b;
    if (a < 4) { a = 4; b = num_tris / 2 / a; if (b < 3) b = 3; }
    const uint32_t base_tris = 2 * a * b;
    if (base_tris > num_tris) { err = "internal: grid larger than request"; return false; }
    std::vector<D3> vert(size_t(a) * b);
    const double two_pi = 6.283185307179586476925286766559;
    for (uint32_t i = 0; i < a; ++i) {
        const double u = double(i) / a, t = two_pi * u;
        // trefoil centre line and its frame (analytic tangent, numeric-free normal via second derivative)
        const double c3 = std::cos(3 * t), s3 = std::sin(3 * t), c2 = std::cos(2 * t), s2 = std::sin(2 * t);
        const D3 pos = {(2 + c3) * c2, (2 + c3) * s2, s3 * 1.3};
        const D3 d1 = {-3 * s3 * c2 - 2 * (2 + c3) * s2, -3 * s3 * s2 + 2 * (2 + c3) * c2, 3 * c3 * 1.3};
        const D3 d2 = {-9 * c3 * c2 + 12 * s3 * s2 - 4 * (2 + c3) * c2, -9 * c3 * s2 - 12 * s3 * c2 - 4 * (2 + c3) * s2, -9 * s3 * 1.3};
        const D3 T = normd(d1);
        const D3 N = normd(d2 - T * dotd(d2, T));
        const D3 B = crossd(T, N);
        const double swell = 0.62 + 0.22 * std::sin(5 * t + 0.7) + 0.10 * std::sin(11 * t);
        for (uint32_t j = 0; j < b; ++j) {
            const double v = double(j) / b, phi = two_pi * v;
            double bump = 0.0;
            bump += 0.090 * (value_noise(u, v, 96, 12, 0, seed) - 0.5);
            bump += 0.050 * (value_noise(u, v, 288, 36, 1, seed) - 0.5);
            bump += 0.025 * (value_noise(u, v, 864, 108, 2, seed) - 0.5);
            const double ridge = 0.06 * std::fabs(std::sin(40 * t)) * (0.5 + 0.5 * std::cos(phi));   // dorsal scales
            const double r = swell * (1.0 + 0.18 * std::cos(2 * phi)) + bump + ridge;
            vert[size_t(i) * b + j] = pos + (N * std::cos(phi) + B * std::sin(phi)) * r;
        }
    }
    normalize_cube(vert);
    TriList tl{out, num_tris};
    uint32_t extra = num_tris - base_tris;
    const uint32_t splits = extra / 2; const bool odd = (extra & 1u) != 0;
    // evenly spaced triangle ids get the centroid split
    uint64_t acc = 0; uint32_t next_split = 0, done_splits = 0; bool odd_done = !odd;
    uint32_t tri_id = 0;
    auto emit = [&](D3 p0, D3 p1, D3 p2) {
        bool split = false;
        if (done_splits < splits) {
            acc += splits;
            if (acc >= base_tris) { acc -= base_tris; split = true; }
        }
        (void)next_split;
        if (split) {
            const D3 c = (p0 + p1 + p2) * (1.0 / 3.0);
            tl.add(p0, p1, c); tl.add(p1, p2, c); tl.add(p2, p0, c);
            ++done_splits;
        } else if (!odd_done) {
            const D3 m = (p0 + p1) * 0.5;
            tl.add(p0, m, p2); tl.add(m, p1, p2);
            odd_done = true;
        } else {
            tl.add(p0, p1, p2);
        }
        ++tri_id;
    };
    for (uint32_t i = 0; i < a; ++i) {
        const uint32_t i1 = (i + 1) % a;
        for (uint32_t j = 0; j < b; ++j) {
            const uint32_t j1 = (j + 1) % b;
            const D3 p00 = vert[size_t(i) * b + j], p10 = vert[size_t(i1) * b + j];
            const D3 p01 = vert[size_t(i) * b + j1], p11 = vert[size_t(i1) * b + j1];
            emit(p00, p10, p11);
            emit(p00, p11, p01);
        }
    }
    if (tl.n != num_tris) { err = "internal: dragon-class generator produced a wrong triangle count"; return false; }
    return true;
}

// Sponza-class: an atrium seen from inside -- floor, four walls, a roof open along the middle, two
// colonnades of fluted columns (long thin triangles) and hanging wavy drapes.  Nearly every camera ray hits.
bool sponza_class(uint32_t seed, uint32_t num_tris, float* out, std::string& err) {
    if (num_tris < 12000) { err = "sponza-class scene needs at least 12000 triangles"; return false; }
    TriList tl{out, num_tris};
    std::vector<D3> quads;   // 4 corners per quad, later normalised together
    std::vector<D3> all;
    auto quad = [&](D3 p00, D3 p10, D3 p11, D3 p01) { all.push_back(p00); all.push_back(p10); all.push_back(p11); all.push_back(p01); };
    // budget: columns+drapes+walls take ~55 %, the floor/ceiling grids absorb the rest
    const double X = 3.0, Y = 1.2, Z = 1.4;   // half extents: long hall along x
    const uint32_t ncol = 10;                 // per side
    const uint32_t flutes = 48, rings = std::max(2u, uint32_t(num_tris * 0.30 / (2.0 * ncol * 2 * flutes)));
    const double two_pi = 6.283185307179586476925286766559;
    for (uint32_t side = 0; side < 2; ++side)
        for (uint32_t c = 0; c < ncol; ++c) {
            const double cx = -X + (c + 0.5) * (2 * X / ncol), cz = (side ? 0.75 : -0.75);
            for (uint32_t r = 0; r < rings; ++r) {
                const double y0 = -Y + 2 * Y * double(r) / rings, y1 = -Y + 2 * Y * double(r + 1) / rings;
                for (uint32_t f = 0; f < flutes; ++f) {
                    const double a0 = two_pi * f / flutes, a1 = two_pi * (f + 1) / flutes;
                    const double r0 = 0.11 + 0.012 * std::cos(a0 * 12) , r1 = 0.11 + 0.012 * std::cos(a1 * 12);
                    quad({cx + r0 * std::cos(a0), y0, cz + r0 * std::sin(a0)}, {cx + r1 * std::cos(a1), y0, cz + r1 * std::sin(a1)},
                         {cx + r1 * std::cos(a1), y1, cz + r1 * std::sin(a1)}, {cx + r0 * std::cos(a0), y1, cz + r0 * std::sin(a0)});
                }
            }
        }
    // drapes: wavy sheets hanging between columns
    const uint32_t ndrape = 8, du = 40, dv = std::max(4u, uint32_t(num_tris * 0.20 / (2.0 * ndrape * du)));
    for (uint32_t d = 0; d < ndrape; ++d) {
        const double x0 = -X + 0.4 + d * (2 * X - 0.8) / ndrape, x1 = x0 + (2 * X - 0.8) / ndrape * 0.8;
        const double z = (d & 1) ? 0.35 : -0.35;
        for (uint32_t i = 0; i < du; ++i) for (uint32_t j = 0; j < dv; ++j) {
            auto P = [&](uint32_t ii, uint32_t jj) {
                const double u = double(ii) / du, v = double(jj) / dv;
                const double w = 0.08 * std::sin(u * 18 + d) * (0.3 + v) + 0.05 * (value_noise(u, v, 8, 8, d, seed) - 0.5);
                return D3{x0 + (x1 - x0) * u, Y - 0.1 - 1.3 * v, z + w};
            };
            quad(P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1));
        }
    }
    // walls (coarse long thin strips) : 4 walls x strips
    const uint32_t strips = 64;
    for (uint32_t s = 0; s < strips; ++s) {
        const double xa = -X + 2 * X * double(s) / strips, xb = -X + 2 * X * double(s + 1) / strips;
        quad({xa, -Y, -Z}, {xb, -Y, -Z}, {xb, Y, -Z}, {xa, Y, -Z});
        quad({xb, -Y, Z}, {xa, -Y, Z}, {xa, Y, Z}, {xb, Y, Z});
    }
    for (uint32_t s = 0; s < strips / 4; ++s) {
        const double za = -Z + 2 * Z * double(s) / (strips / 4), zb = -Z + 2 * Z * double(s + 1) / (strips / 4);
        quad({-X, -Y, zb}, {-X, -Y, za}, {-X, Y, za}, {-X, Y, zb});
        quad({X, -Y, za}, {X, -Y, zb}, {X, Y, zb}, {X, Y, za});
    }
    // floor + ceiling grids absorb the remaining budget
    const uint64_t used = all.size() / 4 * 2;
    if (used + 64 > num_tris) { err = "internal: sponza-class fixed parts exceed the request"; return false; }
    const uint64_t rest_quads = (num_tris - used) / 2;
    uint32_t gz = std::max(2u, uint32_t(std::floor(std::sqrt(double(rest_quads) / 2.0 / 2.2))));
    uint32_t gx = std::max(2u, uint32_t(rest_quads / 2 / gz));
    for (uint32_t pass = 0; pass < 2; ++pass) {
        const double y = pass ? Y : -Y;
        for (uint32_t i = 0; i < gx; ++i) for (uint32_t j = 0; j < gz; ++j) {
            // the roof is open along the middle of the hall (an atrium): the ceiling grid covers two side
            // strips, columns j < gz/2 the one at -z, the others the one at +z
            const uint32_t half = gz / 2;
            auto P = [&](uint32_t ii, uint32_t jj) {
                const double u = double(ii) / gx, v = double(jj) / gz;
                if (!pass) {
                    const double h = 0.01 * (value_noise(u, v, 64, 32, 9, seed) - 0.5);
                    return D3{-X + 2 * X * u, y + h, -Z + 2 * Z * v};
                }
                const bool left = j < half;
                const double w = left ? double(jj) / half : double(jj - half) / (gz - half);   // 0..1 across the strip
                const double z = left ? (-Z + 0.5 * Z * w) : (0.5 * Z + 0.5 * Z * w);
                return D3{-X + 2 * X * u, y + 0.06 * std::sin(u * 40) * std::sin(w * 9), z};
            };
            if (pass) quad(P(i, j), P(i + 1, j), P(i + 1, j + 1), P(i, j + 1));
            else      quad(P(i, j), P(i, j + 1), P(i + 1, j + 1), P(i + 1, j));
        }
    }
    normalize_cube(all);
    const uint64_t nq = all.size() / 4;
    uint64_t base_tris = nq * 2;
    if (base_tris > num_tris) { err = "internal: sponza-class generator overshoot"; return false; }
    uint64_t extra = num_tris - base_tris;   // made up by splitting the first `extra` triangles at an edge midpoint (+1 each)
    for (uint64_t q = 0; q < nq; ++q) {
        const D3 p00 = all[q * 4], p10 = all[q * 4 + 1], p11 = all[q * 4 + 2], p01 = all[q * 4 + 3];
        for (int h = 0; h < 2; ++h) {
            const D3 a0 = p00, a1 = h ? p11 : p10, a2 = h ? p01 : p11;
            if (extra > 0) { const D3 m = (a0 + a1) * 0.5; tl.add(a0, m, a2); tl.add(m, a1, a2); --extra; }
            else tl.add(a0, a1, a2);
        }
    }
    if (tl.n != num_tris) { err = "internal: sponza-class generator produced a wrong triangle count"; return false; }
    return true;
}

} // namespace

bool procedural_scene(uint32_t kind, uint32_t seed, uint32_t num_tris, float* out, std::string& err) {
    if (!out) { err = "null output"; return false; }
    if (kind == 0) return dragon_class(seed, num_tris, out, err);
    if (kind == 1) return sponza_class(seed, num_tris, out, err);
    err = "unknown procedural scene kind";
    return false;
}

} // namespace pt
