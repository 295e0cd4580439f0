// pt_group.cpp -- pixel-tile sharding across the GPUs of one node inside ONE process (include/mi355pt.h, "pt_group_*").
//
// The reference renders one image per PathTracer.render() call on one GPUDevice (src/main.js:54-76).  A group is the same
// call shape over N GPUs: every member context renders the 8x8 tiles with (tx + ty) % N == rank into a compact buffer, RCCL
// gathers the compact buffers on rank 0 over xGMI (ncclGather inside one ncclGroupStart/End: on the full-mesh node every
// sender uses its own link to the root), rank 0 de-interleaves them into the row-major frame.  The scene is replicated: each
// member builds it on its own GPU (1.3 ms for 871k triangles -- cheaper than broadcasting the 180 MB it expands to).  Images are
// bit-identical for every N: the RNG is keyed by global pixel and sample.
//
// Built on the public C ABI only (pt_render with tile_rank / tile_count, pt_set_compact_buffer, pt_deinterleave, ...): a group
// is exactly what a caller could write by hand, plus the collective.  librccl is opened when the first group with the RCCL
// transport is created; a process that never creates one never loads it.
//
// Stream order per member (its context's own in-order stream): resolve of batch b -> gather of batch b -> (rank 0) de-interleave.
// The trace of batch b+1 runs on the context's side streams meanwhile; two sets of compact / gathered buffers alternate.
#include "mi355pt.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

// the few RCCL entry points used, bound at run time (rccl.h: ncclCommInitAll :236, ncclGather :745)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
constexpr int kNcclFloat = 7;   // ncclFloat32
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string& err) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); if (lib) break; }
        if (!lib) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart"); GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Gather = (decltype(Gather))dlsym(lib, "ncclGather"); GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !GroupStart || !GroupEnd || !Gather || !GetErrorString) { err = "librccl lacks ncclCommInitAll / ncclGather / ncclGroupStart"; return false; }
        return true;
    }
};
Rccl g_rccl;
thread_local std::string g_group_error;

}  // namespace

struct PtGroup {
    uint32_t n = 0;
    uint32_t transport = PT_GROUP_TRANSPORT_RCCL;
    std::vector<int> devices;
    std::vector<PtContext*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    std::string err;
    // frame shape the buffers are sized for
    uint32_t width = 0, height = 0, batch = 1;
    uint64_t stride = 0;                          // floats per frame and rank (the largest share)
    std::vector<float*> compact[2];               // [set][rank]: batch * stride floats on that rank's GPU
    std::vector<float*> packed[2];                // [set][rank]: the same frames as packed shares (tiles inside the traced rectangle, 12 B per pixel): what travels
    float* gathered[2] = {nullptr, nullptr};      // rank 0's GPU: n * batch * stride floats (packed shares need three quarters of it at most)
    uint32_t rect[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};   // [set]: union of the traced tile rectangles of the frames submitted into the set
    uint32_t spp[2] = {1, 1};                     // [set]: samples per pixel of those frames (the camera-miss mean outside the rectangle)
    hipEvent_t consumed[2] = {nullptr, nullptr};  // copy transport: rank 0 has de-interleaved what was last copied into gathered[set] (the peers' next copy into it waits for this)
    bool consumed_valid[2] = {false, false};
    std::vector<hipEvent_t> ready;                // copy transport: member r's share of the batch has arrived on rank 0
    uint32_t set = 0, queued = 0;                 // current buffer set, frames submitted into it
    bool accumulating = false, dirty = false;     // an accumulating sequence gathers only when an image is asked for
    bool have_frame = false;
};

namespace {

int gfail(PtGroup* g, int code, const std::string& msg) { if (g) g->err = msg; else g_group_error = msg; return code; }
int gfail_ctx(PtGroup* g, uint32_t r, int code, const char* what) {
    return gfail(g, code, std::string(what) + " (rank " + std::to_string(r) + "): " + pt_last_error(g->ctx[r]));
}
#define G_PT(g, r, call) do { int rc__ = (call); if (rc__ != PT_OK) return gfail_ctx((g), (r), rc__, #call); } while (0)
#define G_HIP(g, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return gfail((g), PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); } while (0)
#define G_NCCL(g, call) do { ncclResult_t r__ = (call); if (r__ != 0) return gfail((g), PT_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r__)); } while (0)

void free_buffers(PtGroup* g) {
    for (int s = 0; s < 2; ++s) {
        for (uint32_t r = 0; r < g->compact[s].size(); ++r)
            if (g->compact[s][r]) { (void)hipSetDevice(g->devices[r]); (void)hipFree(g->compact[s][r]); }
        g->compact[s].clear();
        for (uint32_t r = 0; r < g->packed[s].size(); ++r)
            if (g->packed[s][r]) { (void)hipSetDevice(g->devices[r]); (void)hipFree(g->packed[s][r]); }
        g->packed[s].clear();
        if (g->gathered[s]) { (void)hipSetDevice(g->devices[0]); (void)hipFree(g->gathered[s]); g->gathered[s] = nullptr; }
        g->consumed_valid[s] = false;
    }
}

// (re)allocate the compact / gathered buffers for a frame shape; drains what is in flight first
int ensure_buffers(PtGroup* g, uint32_t width, uint32_t height) {
    if (g->width == width && g->height == height && !g->compact[0].empty()) return PT_OK;
    for (uint32_t r = 0; r < g->n; ++r) G_PT(g, r, pt_synchronize(g->ctx[r]));
    free_buffers(g);
    uint64_t stride = 0;
    for (uint32_t r = 0; r < g->n; ++r) {
        uint64_t fl = 0; uint32_t nt = 0;
        if (int rc = pt_tile_layout(width, height, r, g->n, &nt, &fl)) return gfail(g, rc, pt_last_error(nullptr));
        if (fl > stride) stride = fl;
    }
    if (stride == 0) stride = 4;
    g->stride = stride; g->width = width; g->height = height;
    const size_t bytes = size_t(g->batch) * stride * sizeof(float);
    for (int s = 0; s < 2; ++s) {
        g->compact[s].assign(g->n, nullptr); g->packed[s].assign(g->n, nullptr);
        for (uint32_t r = 0; r < g->n; ++r) {
            G_HIP(g, hipSetDevice(g->devices[r]));
            G_HIP(g, hipMalloc((void**)&g->compact[s][r], bytes));
            G_HIP(g, hipMemset(g->compact[s][r], 0, bytes));
            G_HIP(g, hipMalloc((void**)&g->packed[s][r], bytes));          // a packed share is at most 3/4 of the compact one (every tile inside the rectangle)
        }
        G_HIP(g, hipSetDevice(g->devices[0]));
        G_HIP(g, hipMalloc((void**)&g->gathered[s], bytes * g->n));
    }
    g->set = 0; g->queued = 0;
    return PT_OK;
}

// Gather the `frames` frames submitted into buffer set `s` on rank 0 and de-interleave the last one (the result the read-backs see).  Only the frames that were submitted travel: rank r's share of frame j lands at gathered[s] + (r * frames + j) * stride.
int gather_set(PtGroup* g, uint32_t s, uint32_t frames) {
    if (frames == 0) return PT_OK;
    for (uint32_t r = 0; r < g->n; ++r) G_PT(g, r, pt_flush(g->ctx[r]));          // a partly filled batch is launched now
    // Plain frames travel as PACKED shares: only the tiles inside the traced rectangle, 12 bytes per pixel (mi355pt.h, pt_pack_shares) -- a
    // quarter of the compact buffers for the dragon-class frame.  An accumulating sequence ships its compact buffer as it is (what a pixel
    // outside the rectangle holds depends on the frames before).
    const bool pack = !g->accumulating;
    uint64_t pstride = g->stride;                                                 // floats per frame and rank that travel
    if (pack) {
        uint32_t mt = 0;
        if (int rc = pt_packed_layout(g->width, g->height, g->n, g->rect[s], &mt, &pstride)) return gfail(g, rc, pt_last_error(nullptr));
        for (uint32_t r = 0; r < g->n && pstride != 0; ++r)
            G_PT(g, r, pt_pack_shares(g->ctx[r], g->compact[s][r], g->stride, frames, g->width, g->height, r, g->n, g->rect[s], g->packed[s][r], pstride));
    }
    std::vector<float*>& send = pack ? g->packed[s] : g->compact[s];
    const size_t count = size_t(frames) * pstride;                                // floats per rank
    if (count != 0) {
    if (g->transport == PT_GROUP_TRANSPORT_RCCL) {
        // one collective per batch: every member sends its compact buffer over its own xGMI link to the root.  (Rank 0's de-interleave of
        // the previous use of gathered[s] precedes this gather on its stream, and no peer's data lands before the root has posted it.)
        G_NCCL(g, g_rccl.GroupStart());
        for (uint32_t r = 0; r < g->n; ++r) {
            (void)hipSetDevice(g->devices[r]);                 // the documented single-process pattern: the rank's device is current when its call is issued
            const ncclResult_t rc = g_rccl.Gather(send[r], r == 0 ? g->gathered[s] : nullptr, count, kNcclFloat, 0, g->comm[r], g->stream[r]);
            if (rc != 0) { (void)g_rccl.GroupEnd(); return gfail(g, PT_ERR_HIP, std::string("ncclGather: ") + g_rccl.GetErrorString(rc)); }
        }
        G_NCCL(g, g_rccl.GroupEnd());
        G_HIP(g, hipSetDevice(g->devices[0]));
    } else {
        // diagnostic transport: peer copies instead of the collective (members may share one GPU, which RCCL refuses) --
        // same buffers, same order on every member's stream, so everything but the ncclGather call itself is exercised.
        // Nothing but an event orders a peer's copy into gathered[s] behind rank 0's de-interleave of what the set held two batches ago.
        for (uint32_t r = 0; r < g->n; ++r) {
            G_HIP(g, hipSetDevice(g->devices[r]));
            if (g->consumed_valid[s]) G_HIP(g, hipStreamWaitEvent(g->stream[r], g->consumed[s], 0));
            G_HIP(g, hipMemcpyPeerAsync(g->gathered[s] + size_t(r) * count, g->devices[0], send[r], g->devices[r], count * sizeof(float), g->stream[r]));
            G_HIP(g, hipEventRecord(g->ready[r], g->stream[r]));
        }
        G_HIP(g, hipSetDevice(g->devices[0]));
        for (uint32_t r = 1; r < g->n; ++r) G_HIP(g, hipStreamWaitEvent(g->stream[0], g->ready[r], 0));
    }
    }       // count != 0 (an empty rectangle: nothing travels, rank 0 fills the frame with the camera-miss value)
    // The group's read-backs deliver the LAST frame of a batch (one image per render() is the reference's call shape, src/main.js:54-76;
    // a batch exists to fill the GPUs): only that frame is scattered, into rank 0's own frame buffer -- the result then lives in memory the
    // context owns, whatever happens to the group's buffers afterwards (pt_group_set_batch, a change of resolution).
    if (pack) G_PT(g, 0, pt_unpack_batch(g->ctx[0], g->gathered[s], uint64_t(frames) * pstride, pstride, frames, g->width, g->height, g->n, g->rect[s], g->spp[s], nullptr, 0));
    else G_PT(g, 0, pt_deinterleave_batch(g->ctx[0], g->gathered[s], uint64_t(frames) * g->stride, g->stride, frames, g->width, g->height, g->n, nullptr, 0));
    if (g->transport != PT_GROUP_TRANSPORT_RCCL) {
        G_HIP(g, hipSetDevice(g->devices[0]));
        G_HIP(g, hipEventRecord(g->consumed[s], g->stream[0])); g->consumed_valid[s] = true;
    }
    g->have_frame = true;
    return PT_OK;
}

int flush_group(PtGroup* g) {
    if (g->queued == 0 && !g->dirty) return PT_OK;
    const uint32_t frames = g->accumulating ? 1u : g->queued;
    const int rc = gather_set(g, g->set, frames);
    g->set ^= 1u; g->queued = 0; g->dirty = false;
    return rc;
}

}  // namespace

extern "C" {

const char* pt_group_last_error(const PtGroup* g) { return g ? g->err.c_str() : g_group_error.c_str(); }

int pt_group_create(const int* device_ordinals, uint32_t num_devices, uint32_t transport, PtGroup** out) {
    if (!out) return gfail(nullptr, PT_ERR_INVALID_ARG, "pt_group_create: null out pointer");
    *out = nullptr;
    if (transport != PT_GROUP_TRANSPORT_RCCL && transport != PT_GROUP_TRANSPORT_COPY) return gfail(nullptr, PT_ERR_INVALID_ARG, "pt_group_create: unknown transport");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) return gfail(nullptr, PT_ERR_NO_DEVICE, "pt_group_create: no HIP device available; libmi355pt has no CPU path");
    if (num_devices == 0) num_devices = uint32_t(have);
    if (num_devices > 64) return gfail(nullptr, PT_ERR_INVALID_ARG, "pt_group_create: more than 64 members");
    PtGroup* g = new PtGroup();
    g->n = num_devices; g->transport = transport;
    for (uint32_t r = 0; r < num_devices; ++r) {
        const int d = device_ordinals ? device_ordinals[r] : int(r);
        if (d < 0 || d >= have) {
            delete g;
            return gfail(nullptr, PT_ERR_INVALID_ARG, "pt_group_create: device ordinal " + std::to_string(d) + " out of range (member " + std::to_string(r) + " of " + std::to_string(num_devices) +
                                                      "; this process sees " + std::to_string(have) + " HIP device" + (have == 1 ? "" : "s") + ")");
        }
        g->devices.push_back(d);
    }
    if (transport == PT_GROUP_TRANSPORT_RCCL) {
        for (uint32_t a = 0; a < num_devices; ++a)
            for (uint32_t b = a + 1; b < num_devices; ++b)
                if (g->devices[a] == g->devices[b]) { delete g; return gfail(nullptr, PT_ERR_INVALID_ARG, "pt_group_create: RCCL needs one distinct GPU per member"); }
        std::string err;
        if (!g_rccl.load(err)) { delete g; return gfail(nullptr, PT_ERR_NO_DEVICE, "pt_group_create: " + err); }
    }
    for (uint32_t r = 0; r < num_devices; ++r) {
        PtContext* c = nullptr;
        const int rc = pt_create(g->devices[r], &c);
        if (rc != PT_OK) { const std::string m = pt_last_error(nullptr); pt_group_destroy(g); return gfail(nullptr, rc, "pt_group_create: " + m); }
        g->ctx.push_back(c);
        void* s = nullptr; (void)pt_get_stream(c, &s);
        g->stream.push_back((hipStream_t)s);
    }
    if (transport == PT_GROUP_TRANSPORT_RCCL) {
        g->comm.assign(num_devices, nullptr);
        const ncclResult_t rc = g_rccl.CommInitAll(g->comm.data(), int(num_devices), g->devices.data());     // single process, all devices
        if (rc != 0) { const std::string m = g_rccl.GetErrorString(rc); g->comm.clear(); pt_group_destroy(g); return gfail(nullptr, PT_ERR_HIP, "ncclCommInitAll: " + m); }
    } else {
        g->ready.assign(num_devices, nullptr);
        for (uint32_t r = 0; r < num_devices; ++r) {
            (void)hipSetDevice(g->devices[r]);
            if (hipEventCreateWithFlags(&g->ready[r], hipEventDisableTiming) != hipSuccess) { pt_group_destroy(g); return gfail(nullptr, PT_ERR_HIP, "pt_group_create: hipEventCreate"); }
            if (g->devices[r] != g->devices[0]) {
                int can = 0; (void)hipDeviceCanAccessPeer(&can, g->devices[r], g->devices[0]);
                if (can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(g->devices[0], 0);
                    (void)hipGetLastError();      // "already enabled" (a second group over these GPUs) is success; never leave it behind for the next launch check
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { pt_group_destroy(g); return gfail(nullptr, PT_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe)); }
                }
            }
        }
        (void)hipSetDevice(g->devices[0]);
        for (int k = 0; k < 2; ++k)
            if (hipEventCreateWithFlags(&g->consumed[k], hipEventDisableTiming) != hipSuccess) { pt_group_destroy(g); return gfail(nullptr, PT_ERR_HIP, "pt_group_create: hipEventCreate"); }
    }
    *out = g;
    return PT_OK;
}

void pt_group_destroy(PtGroup* g) {
    if (!g) return;
    for (uint32_t r = 0; r < g->ctx.size(); ++r) if (g->ctx[r]) (void)pt_synchronize(g->ctx[r]);
    for (ncclComm_t c : g->comm) if (c) (void)g_rccl.CommDestroy(c);
    free_buffers(g);
    for (uint32_t r = 0; r < g->ready.size(); ++r) if (g->ready[r]) { (void)hipSetDevice(g->devices[r]); (void)hipEventDestroy(g->ready[r]); }
    for (int k = 0; k < 2; ++k) if (g->consumed[k]) { (void)hipSetDevice(g->devices[0]); (void)hipEventDestroy(g->consumed[k]); }
    for (PtContext* c : g->ctx) if (c) pt_destroy(c);
    delete g;
}

int pt_group_size(const PtGroup* g, uint32_t* n) {
    if (!g || !n) return gfail(nullptr, PT_ERR_INVALID_ARG, "pt_group_size: null argument");
    *n = g->n;
    return PT_OK;
}

int pt_group_context(PtGroup* g, uint32_t rank, PtContext** ctx) {
    if (!g || !ctx || rank >= g->n) return gfail(g, PT_ERR_INVALID_ARG, "pt_group_context: bad rank");
    *ctx = g->ctx[rank];
    return PT_OK;
}

// ---- scene: replicated ------------------------------------------------------------------
int pt_group_set_triangles(PtGroup* g, const float* tris, uint32_t num_tris) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (int rc = flush_group(g)) return rc;
    for (uint32_t r = 0; r < g->n; ++r) G_PT(g, r, pt_set_triangles(g->ctx[r], tris, num_tris));
    return PT_OK;
}
int pt_group_build_bvh(PtGroup* g) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (int rc = flush_group(g)) return rc;
    for (uint32_t r = 0; r < g->n; ++r) G_PT(g, r, pt_build_bvh(g->ctx[r]));      // deterministic: every member ends up with the same BVH2 / BVH4, bit for bit
    return PT_OK;
}
int pt_group_set_bvh2(PtGroup* g, const uint32_t* bvh2, uint64_t words) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (int rc = flush_group(g)) return rc;
    for (uint32_t r = 0; r < g->n; ++r) G_PT(g, r, pt_set_bvh2(g->ctx[r], bvh2, words));
    return PT_OK;
}
int pt_group_set_bvh4(PtGroup* g, const uint32_t* bvh4, uint64_t words) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (int rc = flush_group(g)) return rc;
    for (uint32_t r = 0; r < g->n; ++r) G_PT(g, r, pt_set_bvh4(g->ctx[r], bvh4, words));
    return PT_OK;
}

// ---- the hot path ---------------------------------------------------------------------------
int pt_group_set_batch(PtGroup* g, uint32_t frames_per_launch) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (frames_per_launch < 1u || frames_per_launch > 256u) return gfail(g, PT_ERR_INVALID_ARG, "pt_group_set_batch: 1..256 frames per launch");
    if (int rc = flush_group(g)) return rc;
    for (uint32_t r = 0; r < g->n; ++r) G_PT(g, r, pt_synchronize(g->ctx[r]));
    for (uint32_t r = 0; r < g->n; ++r) G_PT(g, r, pt_set_batch(g->ctx[r], frames_per_launch));
    g->batch = frames_per_launch;
    free_buffers(g); g->width = g->height = 0;
    return PT_OK;
}

int pt_group_render(PtGroup* g, const PtRenderParams* p) {
    if (!g || !p) return gfail(g, PT_ERR_INVALID_ARG, "pt_group_render: null argument");
    const bool accum = p->mode == PT_MODE_PATH && p->accumulate != 0;
    const uint32_t spp = p->mode == PT_MODE_PATH ? p->spp : 1u;
    if ((g->width != p->width || g->height != p->height || accum != g->accumulating || (g->queued && spp != g->spp[g->set])) && (g->queued || g->dirty)) { if (int rc = flush_group(g)) return rc; }
    if (int rc = ensure_buffers(g, p->width, p->height)) return rc;
    g->accumulating = accum;
    {   // the set's traced rectangle: union over the frames submitted into it (every member computes the same one for its launch)
        uint32_t rc4[4];
        G_PT(g, 0, pt_traced_tile_rect(g->ctx[0], p, rc4));
        uint32_t* u = g->rect[g->set];
        if (g->queued == 0) { u[0] = rc4[0]; u[1] = rc4[1]; u[2] = rc4[2]; u[3] = rc4[3]; }
        else if (rc4[2] > rc4[0] && rc4[3] > rc4[1]) {
            if (u[2] <= u[0] || u[3] <= u[1]) { u[0] = rc4[0]; u[1] = rc4[1]; u[2] = rc4[2]; u[3] = rc4[3]; }
            else { u[0] = rc4[0] < u[0] ? rc4[0] : u[0]; u[1] = rc4[1] < u[1] ? rc4[1] : u[1]; u[2] = rc4[2] > u[2] ? rc4[2] : u[2]; u[3] = rc4[3] > u[3] ? rc4[3] : u[3]; }
        }
        g->spp[g->set] = spp;
    }
    // an accumulating sequence keeps its running sum on each member and re-delivers the running mean into slot 0 of the set
    const uint32_t j = accum ? 0u : g->queued;
    PtRenderParams q = *p;
    q.tile_count = g->n;
    q.flags |= PT_FLAG_COMPACT;                 // a one-member group goes through the same compact buffer / gather / de-interleave path
    for (uint32_t r = 0; r < g->n; ++r) {
        q.tile_rank = r;
        G_PT(g, r, pt_set_compact_buffer(g->ctx[r], g->compact[g->set][r] + size_t(j) * g->stride, g->stride));
        G_PT(g, r, pt_render(g->ctx[r], &q));
    }
    if (accum) { g->dirty = true; return PT_OK; }          // gathered when an image is asked for
    if (++g->queued >= g->batch) return flush_group(g);    // the members have just launched this batch: gather it behind their resolve passes
    return PT_OK;
}

int pt_group_flush(PtGroup* g) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    return flush_group(g);
}

int pt_group_synchronize(PtGroup* g) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (int rc = flush_group(g)) return rc;
    for (uint32_t r = g->n; r-- > 0u;) G_PT(g, r, pt_synchronize(g->ctx[r]));     // rank 0 last: its de-interleave follows the others' sends
    return PT_OK;
}

int pt_group_read_radiance(PtGroup* g, float* dst, uint64_t dst_floats) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (int rc = pt_group_synchronize(g)) return rc;
    if (!g->have_frame) return gfail(g, PT_ERR_NO_SCENE, "pt_group_read_radiance: nothing rendered yet");
    G_PT(g, 0, pt_read_radiance(g->ctx[0], dst, dst_floats));
    return PT_OK;
}

int pt_group_read_rgba8(PtGroup* g, uint8_t* dst, uint64_t dst_bytes) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (int rc = pt_group_synchronize(g)) return rc;
    if (!g->have_frame) return gfail(g, PT_ERR_NO_SCENE, "pt_group_read_rgba8: nothing rendered yet");
    G_PT(g, 0, pt_read_rgba8(g->ctx[0], dst, dst_bytes));
    return PT_OK;
}

int pt_group_read_tonemapped(PtGroup* g, int from_rgba8, uint8_t* dst, uint64_t dst_bytes) {
    if (!g) return gfail(nullptr, PT_ERR_INVALID_ARG, "null group");
    if (int rc = pt_group_synchronize(g)) return rc;
    if (!g->have_frame) return gfail(g, PT_ERR_NO_SCENE, "pt_group_read_tonemapped: nothing rendered yet");
    G_PT(g, 0, pt_read_tonemapped(g->ctx[0], from_rgba8, dst, dst_bytes));
    return PT_OK;
}

}  // extern "C"
