// pt_host.h -- host-side scene-build helpers of libmi355pt (no HIP calls in here).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace pt {

constexpr uint32_t kNode2Stride = 6;            // BVHBuilder.wgsl:5
constexpr uint32_t kNode4Stride = 8;            // renderer.wgsl:10
constexpr uint32_t kLeafFlag    = 0x80000000u;  // renderer.wgsl:11
constexpr uint32_t kInvalid     = 0xFFFFFFFFu;  // renderer.wgsl:12
constexpr uint32_t kEmptyBox0 = 0x7C007C00u, kEmptyBox1 = 0xFC007C00u, kEmptyBox2 = 0xFC00FC00u;   // wide layout: box words of an empty or degenerate child slot, mn = +inf, mx = -inf (f16)
constexpr uint32_t kDegenerate  = 0xFFFFFFFEu;  // wide layout only: a child the reference fetches and then rejects for every ray (renderer.wgsl:291)

// ---- f16 codec -------------------------------------------------------------------
float    half_to_float(uint32_t h);             // exact widening (PathTracer.js:16-40)
uint32_t float_to_half_trunc(float v);          // PathTracer.js:42-51 semantics
uint32_t float_to_half_rtne(float v);           // pinned rounding of WGSL pack2x16float

// ---- scene build -----------------------------------------------------------------
void morton_codes_sorted(const float* tris, uint32_t n, uint32_t* morton, uint32_t* tri_index);
// returns false on a malformed BVH2 (message in err)
bool collapse_to_bvh4(const uint32_t* bvh2, uint32_t num_tris, std::vector<uint32_t>& out, std::string& err);
bool promote_to_bvh4_wide(const uint32_t* bvh2, uint64_t words, std::vector<uint32_t>& out, std::string& err);

// ---- device layouts ----------------------------------------------
// One 64-byte record per INTERNAL BVH4 node, child-major: four 16-byte pieces, piece k = child k's packed f16 box
// (3 words, reference packing, renderer.wgsl:94-99) + its reference.  A lane that fetches the whole record issues four
// dwordx4 loads (one ray per lane); the four lanes of a quad that share one ray fetch one piece each -- one 16-byte
// request per lane, the quad covers the 64-byte line (the drain's quad mode, pt_megakernel.hip).
struct WideNode {
    struct Child {
        uint32_t box[3];
        uint32_t ref;  // kInvalid = empty slot; kDegenerate = examined, never entered; else a packed reference (below)
    } child[4];
};
static_assert(sizeof(WideNode) == 64, "WideNode must be 64 bytes");

// Packed references: the traversal gathers triangle records (64 B) and wide nodes (64 B) from ONE arena, so a child reference
// is the record's position in it in 16-byte units -- `ref << 4` is the byte offset, and the shift drops the leaf flag:
//   leaf        kLeafFlag | 4 * tri                 (triangle record `tri` at byte 64 * tri)
//   wide node   node_base16 + 4 * index             (node `index` at byte 16 * node_base16 + 64 * index)
// A leaf whose triangle index is >= num_tris (the reference enters such a leaf and tests nothing, renderer.wgsl:262) points at
// the all-zero record behind the last triangle (index num_tris: never hit, |det| < eps).
inline uint32_t packed_leaf_ref(uint32_t tri, uint32_t num_tris) { return kLeafFlag | (4u * (tri < num_tris ? tri : num_tris)); }
struct WideBvh {
    std::vector<WideNode> nodes;
    uint32_t root_ref = kInvalid;      // same encoding as WideNode::ref; kInvalid = empty BVH
    uint32_t root_box[3] = {0, 0, 0};  // the root's own packed bounds
    bool     root_degenerate = false;  // any(mn > mx) on the root (renderer.wgsl:244)
    uint32_t num_nodes4 = 0;
};
// Validates the reference-layout BVH4 and builds the wide layout.  Rejects: short buffer,
// child reachable twice / cycles.  Children that the reference skips for every ray without
// fetching them (INVALID, index >= numNodes: renderer.wgsl:288) become empty slots; a child with a
// degenerate box (fetched, then rejected: renderer.wgsl:289-291) becomes a kDegenerate slot, which
// no ray enters but which counts as an examined record, exactly as in the reference.
bool build_wide_bvh(const uint32_t* bvh4, uint64_t words, uint32_t num_tris, uint32_t node_base16, WideBvh& out, std::string& err);

// 64-byte triangle record (one cache line, like a wide node): v0, e1 = v1-v0, e2 = v2-v0, n = normalize(cross(e1,e2)) -- the same
// f32 operations renderer.wgsl:179-180,269 performs per visit, done once at upload.  Pieces 0..2 are axis-major: piece a (16 bytes)
// holds component a of the three vectors of the intersection test, (v0[a], e1[a], e2[a], 0) -- three lanes of a quad that fetch one
// piece each hold the triangle in structure-of-arrays form for the quad's Moller-Trumbore (pt_megakernel.hip); piece 3 is the
// normal (n, 0), which the shade pass fetches with ONE 16-byte request (as three components of three pieces it cost three, and the
// vector L1's request rate is what the dense traversal is closest to: +2..3 % frame time, profiles/r04_q2_ab.txt).
struct TriRecord { float axis[3][4]; float n[4]; };
static_assert(sizeof(TriRecord) == 64, "TriRecord must be 64 bytes");
void build_tri_records(const float* tris, uint32_t n, TriRecord* out);

// ---- procedural stand-in scenes ---------------------------------------------------
bool procedural_scene(uint32_t kind, uint32_t seed, uint32_t num_tris, float* out, std::string& err);

// ---- tiles ------------------------------------------------------------------------
constexpr uint32_t kTile = 8;   // 8x8-pixel tiles, one wavefront each
void tile_list(uint32_t width, uint32_t height, uint32_t rank, uint32_t count, std::vector<uint32_t>& tiles);
// tile_list(...).size() without building the list: per tile row, the rank's tiles are tx = first, first + count, ... with first = (rank - ty) mod count
uint32_t tile_count_of(uint32_t width, uint32_t height, uint32_t rank, uint32_t count);
// the rank's tiles inside the tile rectangle rect = {tx0, ty0, tx1, ty1} (half-open): what a packed share holds
uint32_t rect_tile_count_of(uint32_t rank, uint32_t count, const uint32_t rect[4]);

} // namespace pt
