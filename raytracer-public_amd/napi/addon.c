/* addon.c -- thin N-API binding of the C ABI in include/mi355pt.h.
 *
 * This is the reference-side stub INTEGRATION.md describes: the reference's host is JavaScript
 * (src/libs/PathTracer.js drives WebGPU); here the same JS class drives libmi355pt through
 * these synchronous entry points.  Typed arrays in, typed arrays out, a non-zero PtStatus
 * becomes a thrown Error (the reference's error convention: exceptions / rejected Promises).
 * Pure N-API (ABI-stable, no V8 / nan), C only.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#include "mi355pt.h"

#define NAPI_OK(call) do { if ((call) != napi_ok) { napi_throw_error(env, NULL, "N-API call failed: " #call); return NULL; } } while (0)

static napi_value throw_pt(napi_env env, PtContext* ctx, int rc, const char* where) {
    char msg[512];
    const char* e = pt_last_error(ctx);
    snprintf(msg, sizeof msg, "%s: libmi355pt error %d: %s", where, rc, e ? e : "");
    char code[16]; snprintf(code, sizeof code, "PT%d", rc);
    napi_throw_error(env, code, msg);
    return NULL;
}
#define PT_CALL(ctx, call, where) do { int rc__ = (call); if (rc__ != 0) return throw_pt(env, (ctx), rc__, (where)); } while (0)

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value* argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}

static int get_typed(napi_env env, napi_value v, napi_typedarray_type want, void** data, size_t* length) {
    bool is = false;
    if (napi_is_typedarray(env, v, &is) != napi_ok || !is) { napi_throw_type_error(env, NULL, "expected a typed array"); return 0; }
    napi_typedarray_type t; napi_value ab; size_t off;
    if (napi_get_typedarray_info(env, v, &t, length, data, &ab, &off) != napi_ok) { napi_throw_type_error(env, NULL, "bad typed array"); return 0; }
    if (t != want) { napi_throw_type_error(env, NULL, "typed array of the wrong element type"); return 0; }
    return 1;
}

static napi_value make_typed(napi_env env, napi_typedarray_type t, size_t elem_size, size_t length, void** data) {
    napi_value ab, ta;
    if (napi_create_arraybuffer(env, length * elem_size, data, &ab) != napi_ok) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    if (napi_create_typedarray(env, t, length, ab, 0, &ta) != napi_ok) { napi_throw_error(env, NULL, "typed array creation failed"); return NULL; }
    return ta;
}

static PtContext* get_ctx(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) { napi_throw_type_error(env, NULL, "expected a context handle"); return NULL; }
    return *(PtContext**)p;
}

static void finalize_ctx(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    PtContext** box = (PtContext**)data;
    if (*box) pt_destroy(*box);
    free(box);
}

static uint32_t get_u32(napi_env env, napi_value v) { uint32_t x = 0; napi_get_value_uint32(env, v, &x); return x; }
static double get_f64(napi_env env, napi_value v) { double x = 0; napi_get_value_double(env, v, &x); return x; }
static uint32_t prop_u32(napi_env env, napi_value obj, const char* name, uint32_t dflt) {
    napi_value v; bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return dflt;
    napi_get_named_property(env, obj, name, &v);
    napi_valuetype t; napi_typeof(env, v, &t);
    if (t == napi_boolean) { bool b = false; napi_get_value_bool(env, v, &b); return b ? 1u : 0u; }
    if (t != napi_number) return dflt;
    double d = 0; napi_get_value_double(env, v, &d);
    return (uint32_t)d;
}
static double prop_f64(napi_env env, napi_value obj, const char* name, double dflt) {
    napi_value v; bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) return dflt;
    napi_get_named_property(env, obj, name, &v);
    napi_valuetype t; napi_typeof(env, v, &t);
    if (t != napi_number) return dflt;
    double d = 0; napi_get_value_double(env, v, &d);
    return d;
}
static void set_num(napi_env env, napi_value obj, const char* name, double v) {
    napi_value n; napi_create_double(env, v, &n); napi_set_named_property(env, obj, name, n);
}

/* ---- context --------------------------------------------------------------------- */

static napi_value fn_create(napi_env env, napi_callback_info info) {          /* PathTracer.initialize(), PathTracer.js:97 */
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    int32_t dev = -1; napi_get_value_int32(env, argv[0], &dev);
    PtContext** box = (PtContext**)calloc(1, sizeof(PtContext*));
    int rc = pt_create(dev, box);
    if (rc != 0) { free(box); return throw_pt(env, NULL, rc, "pt_create"); }
    napi_value ext;
    NAPI_OK(napi_create_external(env, box, finalize_ctx, NULL, &ext));
    return ext;
}

static napi_value fn_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    void* p = NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) { PtContext** box = (PtContext**)p; if (*box) { pt_destroy(*box); *box = NULL; } }
    return NULL;
}

static napi_value fn_version(napi_env env, napi_callback_info info) {
    (void)info; napi_value s; NAPI_OK(napi_create_string_utf8(env, pt_version(), NAPI_AUTO_LENGTH, &s)); return s;
}

/* ---- host-side scene build -------------------------------------------------------- */

static napi_value fn_bvh2_sizing(napi_env env, napi_callback_info info) {     /* computeBVH2Sizing, PathTracer.js:227 */
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    double n = get_f64(env, argv[0]);
    uint32_t nn = 0; uint64_t bytes = 4;
    pt_compute_bvh2_sizing(n > 0 ? (uint32_t)n : 0u, &nn, &bytes);
    napi_value o; NAPI_OK(napi_create_object(env, &o));
    set_num(env, o, "numNodes2", nn); set_num(env, o, "bytes", (double)bytes);
    return o;
}
static napi_value fn_bvh4_sizing(napi_env env, napi_callback_info info) {     /* computeBVH4Sizing, PathTracer.js:234 */
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    double n = get_f64(env, argv[0]);
    uint64_t bytes = 4;
    pt_compute_bvh4_sizing(n > 0 ? (uint32_t)n : 0u, &bytes);
    napi_value o; NAPI_OK(napi_create_object(env, &o));
    set_num(env, o, "bytes", (double)bytes);
    return o;
}

static napi_value fn_morton_sort(napi_env env, napi_callback_info info) {     /* buildMortonAndSort, PathTracer.js:427 */
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    void* tris; size_t len; if (!get_typed(env, argv[0], napi_float32_array, &tris, &len)) return NULL;
    uint32_t n = (uint32_t)(len / 9);
    void *m, *t;
    napi_value ms = make_typed(env, napi_uint32_array, 4, n, &m); if (!ms) return NULL;
    napi_value ts = make_typed(env, napi_uint32_array, 4, n, &t); if (!ts) return NULL;
    PT_CALL(NULL, pt_morton_sort((const float*)tris, n, (uint32_t*)m, (uint32_t*)t), "pt_morton_sort");
    napi_value o; NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "mortonSorted", ms); napi_set_named_property(env, o, "triIndexSorted", ts);
    return o;
}

static napi_value fn_collapse(napi_env env, napi_callback_info info) {        /* collapseLBVH2ToBVH4, PathTracer.js:506 */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    void* b2; size_t len; if (!get_typed(env, argv[0], napi_uint32_array, &b2, &len)) return NULL;
    uint32_t n = get_u32(env, argv[1]);
    uint64_t need2 = n ? 1ull + 6ull * (2ull * n - 1ull) : 1ull;
    if (len < need2) { napi_throw_range_error(env, NULL, "BVH2 buffer shorter than 1 + 6*(2N-1) words"); return NULL; }
    uint64_t cap = n ? 1ull + 8ull * (2ull * n - 1ull) : 1ull;
    uint32_t* tmp = (uint32_t*)malloc(cap * 4);
    if (!tmp) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    uint32_t n4 = 0;
    int rc = pt_collapse_lbvh2_to_bvh4((const uint32_t*)b2, n, tmp, cap, &n4);
    if (rc != 0) { free(tmp); return throw_pt(env, NULL, rc, "pt_collapse_lbvh2_to_bvh4"); }
    size_t words = n ? 1 + 8 * (size_t)n4 : 1;
    void* out; napi_value ta = make_typed(env, napi_uint32_array, 4, words, &out);
    if (!ta) { free(tmp); return NULL; }
    memcpy(out, tmp, words * 4); free(tmp);
    napi_value o; NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "bvh4U32", ta); set_num(env, o, "numNodes4", n4);
    return o;
}

static napi_value fn_bvh4_wide(napi_env env, napi_callback_info info) {       /* tests/test.cpp:106-196 */
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    void* b2; size_t len; if (!get_typed(env, argv[0], napi_uint32_array, &b2, &len)) return NULL;
    if (len < 1) { napi_throw_range_error(env, NULL, "empty BVH2 buffer"); return NULL; }
    size_t words = 1 + 8 * (size_t)((const uint32_t*)b2)[0];
    void* out; napi_value ta = make_typed(env, napi_uint32_array, 4, words, &out); if (!ta) return NULL;
    PT_CALL(NULL, pt_bvh2_to_bvh4_wide((const uint32_t*)b2, len, (uint32_t*)out, words), "pt_bvh2_to_bvh4_wide");
    return ta;
}

static napi_value fn_write_u32(napi_env env, napi_callback_info info) {       /* data/BVH2.bin writer, src/server/api.js:27-31 */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    char path[4096]; size_t pl = 0; NAPI_OK(napi_get_value_string_utf8(env, argv[0], path, sizeof path, &pl));
    void* d; size_t len; if (!get_typed(env, argv[1], napi_uint32_array, &d, &len)) return NULL;
    PT_CALL(NULL, pt_file_write_u32(path, (const uint32_t*)d, len), "pt_file_write_u32");
    return NULL;
}
static napi_value fn_read_u32(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    char path[4096]; size_t pl = 0; NAPI_OK(napi_get_value_string_utf8(env, argv[0], path, sizeof path, &pl));
    uint64_t words = 0;
    PT_CALL(NULL, pt_file_read_u32(path, NULL, 0, &words), "pt_file_read_u32");
    void* out; napi_value ta = make_typed(env, napi_uint32_array, 4, (size_t)words, &out); if (!ta) return NULL;
    PT_CALL(NULL, pt_file_read_u32(path, (uint32_t*)out, words, &words), "pt_file_read_u32");
    return ta;
}

static napi_value fn_procedural(napi_env env, napi_callback_info info) {
    napi_value argv[3]; if (!get_args(env, info, 3, argv)) return NULL;
    uint32_t kind = get_u32(env, argv[0]), n = get_u32(env, argv[1]), seed = get_u32(env, argv[2]);
    void* out; napi_value ta = make_typed(env, napi_float32_array, 4, (size_t)n * 9, &out); if (!ta) return NULL;
    PT_CALL(NULL, pt_scene_procedural(kind, seed, n, (float*)out), "pt_scene_procedural");
    return ta;
}

/* ---- device scene state ----------------------------------------------------------- */

static napi_value fn_set_triangles(napi_env env, napi_callback_info info) {   /* writeBuffer(triangles), PathTracer.js:679 */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    void* d; size_t len; if (!get_typed(env, argv[1], napi_float32_array, &d, &len)) return NULL;
    PT_CALL(ctx, pt_set_triangles(ctx, (const float*)d, (uint32_t)(len / 9)), "pt_set_triangles");
    return NULL;
}
static napi_value fn_build_bvh(napi_env env, napi_callback_info info) {       /* buildBVH, PathTracer.js:671-749 */
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    PT_CALL(ctx, pt_build_bvh(ctx), "pt_build_bvh");
    return NULL;
}
static napi_value fn_read_bvh2(napi_env env, napi_callback_info info) {       /* readBVH2, PathTracer.js:485 */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    double bytes = get_f64(env, argv[1]);
    size_t size = bytes < 4 ? 4 : (size_t)bytes;                                /* Math.max(4, bytes), :486 */
    void* out; napi_value ta = make_typed(env, napi_uint32_array, 4, size / 4, &out); if (!ta) return NULL;
    PT_CALL(ctx, pt_read_bvh2(ctx, (uint32_t*)out, (size / 4) * 4), "pt_read_bvh2");
    return ta;
}
static napi_value fn_read_bvh4(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    uint32_t n4 = 0; pt_scene_info(ctx, NULL, NULL, &n4);
    size_t words = 1 + 8 * (size_t)n4;
    void* out; napi_value ta = make_typed(env, napi_uint32_array, 4, words, &out); if (!ta) return NULL;
    PT_CALL(ctx, pt_read_bvh4(ctx, (uint32_t*)out, words * 4), "pt_read_bvh4");
    return ta;
}
static napi_value fn_set_bvh4(napi_env env, napi_callback_info info) {        /* writeBuffer(BVH), PathTracer.js:739-740 */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    void* d; size_t len; if (!get_typed(env, argv[1], napi_uint32_array, &d, &len)) return NULL;
    PT_CALL(ctx, pt_set_bvh4(ctx, (const uint32_t*)d, len), "pt_set_bvh4");
    return NULL;
}
static napi_value fn_set_bvh2(napi_env env, napi_callback_info info) {
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    void* d; size_t len; if (!get_typed(env, argv[1], napi_uint32_array, &d, &len)) return NULL;
    PT_CALL(ctx, pt_set_bvh2(ctx, (const uint32_t*)d, len), "pt_set_bvh2");
    return NULL;
}
static napi_value fn_set_spheres(napi_env env, napi_callback_info info) {     /* config C1 extension: (x,y,z,r) per sphere */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    void* d; size_t len; if (!get_typed(env, argv[1], napi_float32_array, &d, &len)) return NULL;
    PT_CALL(ctx, pt_set_spheres(ctx, (const float*)d, (uint32_t)(len / 4)), "pt_set_spheres");
    return NULL;
}
static napi_value fn_scene_info(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    uint32_t a = 0, b = 0, c = 0; pt_scene_info(ctx, &a, &b, &c);
    napi_value o; NAPI_OK(napi_create_object(env, &o));
    set_num(env, o, "numTris", a); set_num(env, o, "numNodes2", b); set_num(env, o, "numNodes4", c);
    return o;
}

/* ---- the hot path ------------------------------------------------------------------ */

/* argv: the 16-float UBO exactly as PathTracer.js:764-787 packs it + the extension options */
static int params_from_ubo(napi_env env, napi_value ubo_v, napi_value opt, PtRenderParams* p) {
    void* u; size_t len; if (!get_typed(env, ubo_v, napi_float32_array, &u, &len)) return 0;
    if (len < 16) { napi_throw_range_error(env, NULL, "UBO must hold 16 floats"); return 0; }
    const float* ubo = (const float*)u;
    memset(p, 0, sizeof *p);
    p->width = (uint32_t)ubo[0]; p->height = (uint32_t)ubo[1];                  /* u32(ubo.resolution.xy), renderer.wgsl:357 */
    p->focal = ubo[2]; p->aspect = ubo[3];
    p->cam_pos[0] = ubo[4]; p->cam_pos[1] = ubo[5]; p->cam_pos[2] = ubo[6];
    p->num_tris = (uint32_t)ubo[7];                                             /* u32(camPosNumTris.w), renderer.wgsl:398 */
    p->cam_quat[0] = ubo[8]; p->cam_quat[1] = ubo[9]; p->cam_quat[2] = ubo[10]; p->cam_quat[3] = ubo[11];
    p->frame = (uint32_t)ubo[12];
    p->mode = prop_u32(env, opt, "mode", PT_MODE_REFERENCE);
    p->spp = prop_u32(env, opt, "spp", 1);
    p->max_bounces = prop_u32(env, opt, "maxBounces", 0);
    p->seed = prop_u32(env, opt, "seed", 1);
    p->accumulate = prop_u32(env, opt, "accumulate", 0);
    p->tile_rank = prop_u32(env, opt, "tileRank", 0);
    p->tile_count = prop_u32(env, opt, "tileCount", 1);
    p->flags = (prop_u32(env, opt, "stats", 0) ? PT_FLAG_STATS : 0u) | (prop_u32(env, opt, "simpleKernel", 0) ? PT_FLAG_SIMPLE_KERNEL : 0u) |
               (prop_u32(env, opt, "bruteForce", 0) ? PT_FLAG_BRUTE_FORCE : 0u);
    (void)prop_f64;
    return 1;
}

static napi_value fn_render(napi_env env, napi_callback_info info) {          /* PathTracer.render() compute pass, PathTracer.js:756-802 */
    napi_value argv[3]; if (!get_args(env, info, 3, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    PtRenderParams p; if (!params_from_ubo(env, argv[1], argv[2], &p)) return NULL;
    PT_CALL(ctx, pt_render(ctx, &p), "pt_render");
    return NULL;
}

static napi_value fn_set_batch(napi_env env, napi_callback_info info) {       /* frames per persistent launch (1..256) */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    PT_CALL(ctx, pt_set_batch(ctx, get_u32(env, argv[1])), "pt_set_batch");
    return NULL;
}
static napi_value fn_flush(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    PT_CALL(ctx, pt_flush(ctx), "pt_flush");
    return NULL;
}
static napi_value fn_last_ms(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    float ms = 0; PT_CALL(ctx, pt_last_render_ms(ctx, &ms), "pt_last_render_ms");
    napi_value v; NAPI_OK(napi_create_double(env, ms, &v)); return v;
}
static napi_value fn_sync(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    PT_CALL(ctx, pt_synchronize(ctx), "pt_synchronize");
    return NULL;
}
static napi_value fn_stats(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    PtStats st; PT_CALL(ctx, pt_get_stats(ctx, &st), "pt_get_stats");
    napi_value o; NAPI_OK(napi_create_object(env, &o));
    set_num(env, o, "raysClosest", (double)st.rays_closest); set_num(env, o, "raysShadow", (double)st.rays_shadow);
    set_num(env, o, "nodesExamined", (double)st.nodes_examined); set_num(env, o, "trisTested", (double)st.tris_tested);
    set_num(env, o, "stackDrops", (double)st.stack_drops); set_num(env, o, "maxStack", (double)st.max_stack);
    set_num(env, o, "samples", (double)st.samples);
    return o;
}
static napi_value fn_read_radiance(napi_env env, napi_callback_info info) {
    napi_value argv[3]; if (!get_args(env, info, 3, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    size_t n = (size_t)get_u32(env, argv[1]) * get_u32(env, argv[2]) * 4;
    void* out; napi_value ta = make_typed(env, napi_float32_array, 4, n, &out); if (!ta) return NULL;
    PT_CALL(ctx, pt_read_radiance(ctx, (float*)out, n), "pt_read_radiance");
    return ta;
}
static napi_value read_u8(napi_env env, napi_callback_info info, int which) {
    napi_value argv[4]; if (!get_args(env, info, which == 2 ? 4 : 3, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    size_t n = (size_t)get_u32(env, argv[1]) * get_u32(env, argv[2]) * 4;
    void* out; napi_value ta = make_typed(env, napi_uint8_array, 1, n, &out); if (!ta) return NULL;
    if (which == 1) PT_CALL(ctx, pt_read_rgba8(ctx, (uint8_t*)out, n), "pt_read_rgba8");
    else { bool q = true; napi_get_value_bool(env, argv[3], &q); PT_CALL(ctx, pt_read_tonemapped(ctx, q ? 1 : 0, (uint8_t*)out, n), "pt_read_tonemapped"); }
    return ta;
}
/* checkpoint of a progressive accumulation (pt_read_accum): { width, height, tileRank, tileCount, compact, samples, data: Float32Array } */
static napi_value fn_read_accum(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    PtAccumInfo ai; PT_CALL(ctx, pt_accum_info(ctx, &ai), "pt_accum_info");
    if (ai.floats == 0) { napi_throw_error(env, "PT4", "readAccumulation: no running accumulation (render with accumulate first)"); return NULL; }
    void* out; napi_value ta = make_typed(env, napi_float32_array, 4, (size_t)ai.floats, &out); if (!ta) return NULL;
    PT_CALL(ctx, pt_read_accum(ctx, (float*)out, ai.floats), "pt_read_accum");
    napi_value o; NAPI_OK(napi_create_object(env, &o));
    set_num(env, o, "width", ai.width); set_num(env, o, "height", ai.height); set_num(env, o, "tileRank", ai.tile_rank); set_num(env, o, "tileCount", ai.tile_count);
    set_num(env, o, "compact", ai.compact); set_num(env, o, "samples", ai.samples);
    napi_set_named_property(env, o, "data", ta);
    return o;
}
static napi_value fn_set_accum(napi_env env, napi_callback_info info) {          /* (ctx, the object readAccumulation returned) */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtContext* ctx = get_ctx(env, argv[0]); if (!ctx) return NULL;
    PtAccumInfo ai; memset(&ai, 0, sizeof ai);
    ai.width = prop_u32(env, argv[1], "width", 0); ai.height = prop_u32(env, argv[1], "height", 0); ai.tile_rank = prop_u32(env, argv[1], "tileRank", 0);
    ai.tile_count = prop_u32(env, argv[1], "tileCount", 1); ai.compact = prop_u32(env, argv[1], "compact", 0); ai.samples = prop_u32(env, argv[1], "samples", 0);
    napi_value dv; bool has = false;
    if (napi_has_named_property(env, argv[1], "data", &has) != napi_ok || !has) { napi_throw_type_error(env, NULL, "restoreAccumulation: no `data`"); return NULL; }
    napi_get_named_property(env, argv[1], "data", &dv);
    void* d; size_t len; if (!get_typed(env, dv, napi_float32_array, &d, &len)) return NULL;
    ai.floats = len;
    PT_CALL(ctx, pt_set_accum(ctx, &ai, (const float*)d), "pt_set_accum");
    return NULL;
}
static napi_value fn_read_rgba8(napi_env env, napi_callback_info info) { return read_u8(env, info, 1); }       /* outputTex, PathTracer.js:163-172 */
static napi_value fn_read_tonemapped(napi_env env, napi_callback_info info) { return read_u8(env, info, 2); }  /* tonemapper.wgsl */


/* ---- several GPUs, one image per render(): pt_group_* (include/mi355pt.h) --------------------------------------- */

static void finalize_group(napi_env env, void* data, void* hint) {
    (void)env; (void)hint;
    PtGroup** box = (PtGroup**)data;
    if (*box) pt_group_destroy(*box);
    free(box);
}
static PtGroup* get_group(napi_env env, napi_value v) {
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || !*(PtGroup**)p) { napi_throw_type_error(env, NULL, "expected a group handle"); return NULL; }
    return *(PtGroup**)p;
}
static napi_value throw_group(napi_env env, PtGroup* g, int rc, const char* where) {
    char msg[640];
    const char* e = pt_group_last_error(g);
    snprintf(msg, sizeof msg, "%s: libmi355pt error %d: %s", where, rc, e ? e : "");
    char code[16]; snprintf(code, sizeof code, "PT%d", rc);
    napi_throw_error(env, code, msg);
    return NULL;
}
#define PTG_CALL(g, call, where) do { int rc__ = (call); if (rc__ != 0) return throw_group(env, (g), rc__, (where)); } while (0)

static napi_value fn_group_create(napi_env env, napi_callback_info info) {    /* (Int32Array devices | number of devices, transport) */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    bool is_ta = false; napi_is_typedarray(env, argv[0], &is_ta);
    const int* devs = NULL; uint32_t n = 0;
    if (is_ta) { void* d; size_t len; if (!get_typed(env, argv[0], napi_int32_array, &d, &len)) return NULL; devs = (const int*)d; n = (uint32_t)len; }
    else n = get_u32(env, argv[0]);
    PtGroup** box = (PtGroup**)calloc(1, sizeof(PtGroup*));
    int rc = pt_group_create(devs, n, get_u32(env, argv[1]), box);
    if (rc != 0) { free(box); return throw_group(env, NULL, rc, "pt_group_create"); }
    napi_value ext;
    NAPI_OK(napi_create_external(env, box, finalize_group, NULL, &ext));
    return ext;
}
static napi_value fn_group_destroy(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    void* p = NULL;
    if (napi_get_value_external(env, argv[0], &p) == napi_ok && p) { PtGroup** box = (PtGroup**)p; if (*box) { pt_group_destroy(*box); *box = NULL; } }
    return NULL;
}
static napi_value fn_group_size(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    uint32_t n = 0; PTG_CALL(g, pt_group_size(g, &n), "pt_group_size");
    napi_value v; NAPI_OK(napi_create_uint32(env, n, &v)); return v;
}
static napi_value fn_group_set_triangles(napi_env env, napi_callback_info info) {
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    void* d; size_t len; if (!get_typed(env, argv[1], napi_float32_array, &d, &len)) return NULL;
    PTG_CALL(g, pt_group_set_triangles(g, (const float*)d, (uint32_t)(len / 9)), "pt_group_set_triangles");
    return NULL;
}
static napi_value fn_group_build_bvh(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    PTG_CALL(g, pt_group_build_bvh(g), "pt_group_build_bvh");
    return NULL;
}
static napi_value group_set_bvh(napi_env env, napi_callback_info info, int four) {
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    void* d; size_t len; if (!get_typed(env, argv[1], napi_uint32_array, &d, &len)) return NULL;
    if (four) PTG_CALL(g, pt_group_set_bvh4(g, (const uint32_t*)d, len), "pt_group_set_bvh4");
    else PTG_CALL(g, pt_group_set_bvh2(g, (const uint32_t*)d, len), "pt_group_set_bvh2");
    return NULL;
}
static napi_value fn_group_set_bvh4(napi_env env, napi_callback_info info) { return group_set_bvh(env, info, 1); }
static napi_value fn_group_set_bvh2(napi_env env, napi_callback_info info) { return group_set_bvh(env, info, 0); }
static napi_value fn_group_read_bvh2(napi_env env, napi_callback_info info) {  /* readBVH2 of rank 0 (every member holds the same buffer) */
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    PtContext* c0 = NULL; PTG_CALL(g, pt_group_context(g, 0, &c0), "pt_group_context");
    double bytes = get_f64(env, argv[1]);
    size_t size = bytes < 4 ? 4 : (size_t)bytes;
    void* out; napi_value ta = make_typed(env, napi_uint32_array, 4, size / 4, &out); if (!ta) return NULL;
    PT_CALL(c0, pt_read_bvh2(c0, (uint32_t*)out, (size / 4) * 4), "pt_read_bvh2");
    return ta;
}
static napi_value fn_group_set_batch(napi_env env, napi_callback_info info) {
    napi_value argv[2]; if (!get_args(env, info, 2, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    PTG_CALL(g, pt_group_set_batch(g, get_u32(env, argv[1])), "pt_group_set_batch");
    return NULL;
}
static napi_value fn_group_render(napi_env env, napi_callback_info info) {
    napi_value argv[3]; if (!get_args(env, info, 3, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    PtRenderParams p; if (!params_from_ubo(env, argv[1], argv[2], &p)) return NULL;
    PTG_CALL(g, pt_group_render(g, &p), "pt_group_render");
    return NULL;
}
static napi_value fn_group_flush(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    PTG_CALL(g, pt_group_flush(g), "pt_group_flush");
    return NULL;
}
static napi_value fn_group_sync(napi_env env, napi_callback_info info) {
    napi_value argv[1]; if (!get_args(env, info, 1, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    PTG_CALL(g, pt_group_synchronize(g), "pt_group_synchronize");
    return NULL;
}
static napi_value fn_group_read_radiance(napi_env env, napi_callback_info info) {
    napi_value argv[3]; if (!get_args(env, info, 3, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    size_t n = (size_t)get_u32(env, argv[1]) * get_u32(env, argv[2]) * 4;
    void* out; napi_value ta = make_typed(env, napi_float32_array, 4, n, &out); if (!ta) return NULL;
    PTG_CALL(g, pt_group_read_radiance(g, (float*)out, n), "pt_group_read_radiance");
    return ta;
}
static napi_value group_read_u8(napi_env env, napi_callback_info info, int which) {
    napi_value argv[4]; if (!get_args(env, info, which == 2 ? 4 : 3, argv)) return NULL;
    PtGroup* g = get_group(env, argv[0]); if (!g) return NULL;
    size_t n = (size_t)get_u32(env, argv[1]) * get_u32(env, argv[2]) * 4;
    void* out; napi_value ta = make_typed(env, napi_uint8_array, 1, n, &out); if (!ta) return NULL;
    if (which == 1) PTG_CALL(g, pt_group_read_rgba8(g, (uint8_t*)out, n), "pt_group_read_rgba8");
    else { bool q = true; napi_get_value_bool(env, argv[3], &q); PTG_CALL(g, pt_group_read_tonemapped(g, q ? 1 : 0, (uint8_t*)out, n), "pt_group_read_tonemapped"); }
    return ta;
}
static napi_value fn_group_read_rgba8(napi_env env, napi_callback_info info) { return group_read_u8(env, info, 1); }
static napi_value fn_group_read_tonemapped(napi_env env, napi_callback_info info) { return group_read_u8(env, info, 2); }

static napi_value init(napi_env env, napi_value exports) {
    static const struct { const char* name; napi_callback fn; } fns[] = {
        {"create", fn_create}, {"destroy", fn_destroy}, {"version", fn_version},
        {"computeBVH2Sizing", fn_bvh2_sizing}, {"computeBVH4Sizing", fn_bvh4_sizing},
        {"mortonSort", fn_morton_sort}, {"collapse", fn_collapse}, {"bvh4Wide", fn_bvh4_wide},
        {"writeU32File", fn_write_u32}, {"readU32File", fn_read_u32}, {"proceduralScene", fn_procedural},
        {"setTriangles", fn_set_triangles}, {"buildBVH", fn_build_bvh}, {"readBVH2", fn_read_bvh2}, {"readBVH4", fn_read_bvh4},
        {"setBVH4", fn_set_bvh4}, {"setBVH2", fn_set_bvh2}, {"setSpheres", fn_set_spheres}, {"sceneInfo", fn_scene_info},
        {"render", fn_render}, {"setBatch", fn_set_batch}, {"flush", fn_flush}, {"lastRenderMs", fn_last_ms}, {"synchronize", fn_sync}, {"getStats", fn_stats},
        {"readRadiance", fn_read_radiance}, {"readRGBA8", fn_read_rgba8}, {"readTonemapped", fn_read_tonemapped},
        {"readAccumulation", fn_read_accum}, {"restoreAccumulation", fn_set_accum},
        {"groupCreate", fn_group_create}, {"groupDestroy", fn_group_destroy}, {"groupSize", fn_group_size},
        {"groupSetTriangles", fn_group_set_triangles}, {"groupBuildBVH", fn_group_build_bvh}, {"groupSetBVH4", fn_group_set_bvh4}, {"groupSetBVH2", fn_group_set_bvh2},
        {"groupReadBVH2", fn_group_read_bvh2}, {"groupSetBatch", fn_group_set_batch}, {"groupRender", fn_group_render}, {"groupFlush", fn_group_flush},
        {"groupSynchronize", fn_group_sync}, {"groupReadRadiance", fn_group_read_radiance}, {"groupReadRGBA8", fn_group_read_rgba8}, {"groupReadTonemapped", fn_group_read_tonemapped},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; ++i) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        if (napi_set_named_property(env, exports, fns[i].name, f) != napi_ok) return NULL;
    }
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
