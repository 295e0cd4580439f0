"""raytracer-public_amd -- MI355X-native drop-in for the per-pixel-sample hot path of
31415Hacker/RayTracer-public (src/shaders/renderer.wgsl) behind the reference's own
PathTracer / Scene API.

This module is the ctypes binding of the C ABI in include/mi355pt.h (libmi355pt.so, built
in-tree by raytracer-public_amd/csrc/Makefile) plus a small Python mirror of the reference's
``PathTracer`` class (src/libs/PathTracer.js) used by tests and bench.py.  The production
host is the Node side in raytracer-public_amd/js/ over the N-API addon.

There is no CPU fallback: if the shared library is missing the import fails, and creating a
context without a HIP device raises ``PtError``.
"""
import ctypes as C
import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmi355pt.so")

PT_MODE_REFERENCE_PACKET, PT_MODE_REFERENCE, PT_MODE_PATH = 0, 1, 2
PT_FLAG_STATS = 1
PT_FLAG_SIMPLE_KERNEL = 2
PT_FLAG_BRUTE_FORCE = 4
PT_FLAG_COMPACT = 8
SCENE_DRAGON_CLASS, SCENE_SPONZA_CLASS = 0, 1


class PtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libmi355pt error %d: %s" % (code, message))
        self.code = code


class PtRenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("focal", C.c_float), ("aspect", C.c_float),
        ("cam_pos", C.c_float * 3), ("num_tris", C.c_uint32),
        ("cam_quat", C.c_float * 4),
        ("frame", C.c_uint32), ("mode", C.c_uint32),
        ("spp", C.c_uint32), ("max_bounces", C.c_uint32), ("seed", C.c_uint32),
        ("accumulate", C.c_uint32),
        ("tile_rank", C.c_uint32), ("tile_count", C.c_uint32),
        ("flags", C.c_uint32),
    ]


class PtStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_closest", "rays_shadow", "nodes_examined", "tris_tested", "stack_drops", "max_stack", "samples")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PtAccumInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("tile_rank", C.c_uint32), ("tile_count", C.c_uint32),
                ("compact", C.c_uint32), ("samples", C.c_uint32), ("floats", C.c_uint64)]


# every symbol include/mi355pt.h declares (tests/test_host_build.py::test_library_exports_every_declared_symbol checks the header against this)
EXPORTS = [
    "pt_create", "pt_destroy", "pt_last_error", "pt_version", "pt_set_stream", "pt_get_stream", "pt_synchronize",
    "pt_compute_bvh2_sizing", "pt_compute_bvh4_sizing", "pt_morton_sort", "pt_collapse_lbvh2_to_bvh4",
    "pt_bvh2_to_bvh4_wide", "pt_file_write_u32", "pt_file_read_u32", "pt_scene_procedural",
    "pt_set_triangles", "pt_build_bvh", "pt_build_lbvh2", "pt_read_bvh2", "pt_set_bvh4", "pt_set_bvh2",
    "pt_read_bvh4", "pt_set_spheres", "pt_scene_info", "pt_render", "pt_last_render_ms", "pt_set_batch", "pt_flush", "pt_timing_begin", "pt_timing_collect", "pt_timing_collect_spans", "pt_set_compact_buffer", "pt_set_output_buffer", "pt_get_stats", "pt_read_radiance",
    "pt_read_rgba8", "pt_read_tonemapped", "pt_tile_layout", "pt_tile_ids", "pt_compact_radiance", "pt_deinterleave", "pt_deinterleave_batch", "pt_buffer_busy",
    "pt_accum_info", "pt_read_accum", "pt_set_accum",
    "pt_traced_tile_rect", "pt_packed_layout", "pt_packed_tile_ids", "pt_pack_shares", "pt_unpack_batch",
    "pt_group_create", "pt_group_destroy", "pt_group_last_error", "pt_group_size", "pt_group_context", "pt_group_set_triangles", "pt_group_build_bvh",
    "pt_group_set_bvh2", "pt_group_set_bvh4", "pt_group_set_batch", "pt_group_render", "pt_group_flush", "pt_group_synchronize", "pt_group_read_radiance",
    "pt_group_read_rgba8", "pt_group_read_tonemapped",
    "pt_debug_set_tune", "pt_debug_counters", "pt_debug_wave_times", "pt_debug_launch_plan",     # diagnostics section of the header
]


def _load():
    # the trace phases of consecutive frames overlap on side streams; ROCm's default of 4 hardware
    # queues would serialise them (read by the HIP runtime when it initialises)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libmi355pt.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C raytracer-public_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_last_error.argtypes = [C.c_void_p]
    lib.pt_version.restype = C.c_char_p
    lib.pt_destroy.restype = None
    lib.pt_destroy.argtypes = [C.c_void_p]
    return lib


lib = _load()


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _check(rc, ctx=None):
    if rc != 0:
        msg = lib.pt_last_error(ctx)
        raise PtError(rc, msg.decode() if msg else "")


def focal_aspect(width, height):
    """PathTracer.js:761-769: fov 70 degrees; doubles on the host, f32 in the UBO."""
    fov = (70.0 * math.pi) / 180
    return float(np.float32(1.0 / math.tan(0.5 * fov))), float(np.float32(width / height))


# ---- host-side functions (no GPU) ---------------------------------------------------------

def compute_bvh2_sizing(num_tris):
    nn, by = C.c_uint32(), C.c_uint64()
    _check(lib.pt_compute_bvh2_sizing(C.c_uint32(num_tris), C.byref(nn), C.byref(by)))
    return {"numNodes2": nn.value, "bytes": by.value}


def compute_bvh4_sizing(num_nodes4):
    by = C.c_uint64()
    _check(lib.pt_compute_bvh4_sizing(C.c_uint32(num_nodes4), C.byref(by)))
    return {"bytes": by.value}


def morton_sort(tris):
    tris = np.ascontiguousarray(tris, np.float32).reshape(-1)
    n = tris.size // 9
    m = np.zeros(n, np.uint32); t = np.zeros(n, np.uint32)
    _check(lib.pt_morton_sort(_p(tris, C.c_float), C.c_uint32(n), _p(m, C.c_uint32), _p(t, C.c_uint32)))
    return m, t


def collapse_lbvh2_to_bvh4(bvh2, num_tris):
    bvh2 = np.ascontiguousarray(bvh2, np.uint32)
    cap = 1 + 8 * max(2 * num_tris - 1, 0)
    out = np.zeros(cap, np.uint32)
    n4 = C.c_uint32()
    _check(lib.pt_collapse_lbvh2_to_bvh4(_p(bvh2, C.c_uint32), C.c_uint32(num_tris), _p(out, C.c_uint32), C.c_uint64(cap), C.byref(n4)))
    return out[: 1 + 8 * n4.value].copy(), n4.value


def bvh2_to_bvh4_wide(bvh2):
    bvh2 = np.ascontiguousarray(bvh2, np.uint32)
    out = np.zeros(1 + 8 * int(bvh2[0]), np.uint32)
    _check(lib.pt_bvh2_to_bvh4_wide(_p(bvh2, C.c_uint32), C.c_uint64(bvh2.size), _p(out, C.c_uint32), C.c_uint64(out.size)))
    return out


def write_u32_file(path, words):
    words = np.ascontiguousarray(words, np.uint32)
    _check(lib.pt_file_write_u32(path.encode(), _p(words, C.c_uint32), C.c_uint64(words.size)))


def read_u32_file(path):
    n = C.c_uint64()
    _check(lib.pt_file_read_u32(path.encode(), None, C.c_uint64(0), C.byref(n)))
    out = np.zeros(n.value, np.uint32)
    _check(lib.pt_file_read_u32(path.encode(), _p(out, C.c_uint32), C.c_uint64(out.size), C.byref(n)))
    return out


def procedural_scene(kind, num_tris, seed=20260109):
    out = np.zeros(num_tris * 9, np.float32)
    _check(lib.pt_scene_procedural(C.c_uint32(kind), C.c_uint32(seed), C.c_uint32(num_tris), _p(out, C.c_float)))
    return out


def tile_layout(width, height, rank, count):
    nt, fl = C.c_uint32(), C.c_uint64()
    _check(lib.pt_tile_layout(C.c_uint32(width), C.c_uint32(height), C.c_uint32(rank), C.c_uint32(count), C.byref(nt), C.byref(fl)))
    return nt.value, fl.value


def packed_layout(width, height, count, rect):
    """(largest packed share over the ranks in tiles, floats per frame of the gather) for the tile rectangle rect = (tx0, ty0, tx1, ty1)."""
    r = (C.c_uint32 * 4)(*rect); mt, fl = C.c_uint32(), C.c_uint64()
    _check(lib.pt_packed_layout(C.c_uint32(width), C.c_uint32(height), C.c_uint32(count), r, C.byref(mt), C.byref(fl)))
    return mt.value, fl.value


def packed_tile_ids(width, height, rank, count, rect):
    """Tile ids of the rank's packed share (its tiles inside rect), in packed-buffer order."""
    r = (C.c_uint32 * 4)(*rect); n = C.c_uint32()
    _check(lib.pt_packed_tile_ids(C.c_uint32(width), C.c_uint32(height), C.c_uint32(rank), C.c_uint32(count), r, None, C.c_uint32(0), C.byref(n)))
    ids = np.zeros(max(n.value, 1), np.uint32)
    _check(lib.pt_packed_tile_ids(C.c_uint32(width), C.c_uint32(height), C.c_uint32(rank), C.c_uint32(count), r, _p(ids, C.c_uint32), C.c_uint32(ids.size), C.byref(n)))
    return ids[: n.value]


def tile_ids(width, height, rank, count):
    """Tile ids (ty * ceil(W/8) + tx) of the rank's share, in compact-buffer order."""
    n = C.c_uint32()
    _check(lib.pt_tile_ids(C.c_uint32(width), C.c_uint32(height), C.c_uint32(rank), C.c_uint32(count), None, C.c_uint32(0), C.byref(n)))
    ids = np.zeros(max(n.value, 1), np.uint32)
    _check(lib.pt_tile_ids(C.c_uint32(width), C.c_uint32(height), C.c_uint32(rank), C.c_uint32(count), ids.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint32(ids.size), C.byref(n)))
    return ids[: n.value]


# ---- device context -----------------------------------------------------------------------

class Context:
    """One GPU.  Thin, explicit wrapper over the pt_* entry points."""

    def __init__(self, device=-1):
        h = C.c_void_p()
        _check(lib.pt_create(C.c_int(device), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib.pt_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _check(rc, self.h)

    def set_stream(self, stream_handle):
        self._ck(lib.pt_set_stream(self.h, C.c_void_p(stream_handle)))

    def get_stream(self):
        p = C.c_void_p()
        self._ck(lib.pt_get_stream(self.h, C.byref(p)))
        return p.value

    def synchronize(self):
        self._ck(lib.pt_synchronize(self.h))

    def set_triangles(self, tris):
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1)
        self._ck(lib.pt_set_triangles(self.h, _p(tris, C.c_float), C.c_uint32(tris.size // 9)))
        self.num_tris = tris.size // 9

    def set_spheres(self, xyzr):
        xyzr = np.ascontiguousarray(xyzr, np.float32).reshape(-1)
        self._ck(lib.pt_set_spheres(self.h, _p(xyzr, C.c_float), C.c_uint32(xyzr.size // 4)))

    def build_bvh(self):
        self._ck(lib.pt_build_bvh(self.h))

    def build_lbvh2(self, morton, tri_idx):
        morton = np.ascontiguousarray(morton, np.uint32); tri_idx = np.ascontiguousarray(tri_idx, np.uint32)
        self._ck(lib.pt_build_lbvh2(self.h, _p(morton, C.c_uint32), _p(tri_idx, C.c_uint32)))

    def read_bvh2(self):
        out = np.zeros(compute_bvh2_sizing(self.num_tris)["bytes"] // 4, np.uint32)
        self._ck(lib.pt_read_bvh2(self.h, _p(out, C.c_uint32), C.c_uint64(out.size * 4)))
        return out

    def set_bvh4(self, bvh4):
        bvh4 = np.ascontiguousarray(bvh4, np.uint32)
        self._ck(lib.pt_set_bvh4(self.h, _p(bvh4, C.c_uint32), C.c_uint64(bvh4.size)))

    def set_bvh2(self, bvh2):
        bvh2 = np.ascontiguousarray(bvh2, np.uint32)
        self._ck(lib.pt_set_bvh2(self.h, _p(bvh2, C.c_uint32), C.c_uint64(bvh2.size)))

    def scene_info(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(lib.pt_scene_info(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"numTris": a.value, "numNodes2": b.value, "numNodes4": c.value}

    def read_bvh4(self):
        out = np.zeros(1 + 8 * self.scene_info()["numNodes4"], np.uint32)
        self._ck(lib.pt_read_bvh4(self.h, _p(out, C.c_uint32), C.c_uint64(out.size * 4)))
        return out

    def make_params(self, width, height, cam_pos=(0, 0, 2.5), cam_quat=(0, 0, 0, 1), mode=PT_MODE_REFERENCE, spp=1,
                    max_bounces=0, seed=1, frame=0, accumulate=False, tile_rank=0, tile_count=1, stats=False, num_tris=None, simple_kernel=False, brute_force=False):
        p = PtRenderParams()
        p.width, p.height = width, height
        p.focal, p.aspect = focal_aspect(width, height)
        p.cam_pos[:] = [float(np.float32(v)) for v in cam_pos]
        p.cam_quat[:] = [float(np.float32(v)) for v in cam_quat]
        p.num_tris = self.num_tris if num_tris is None else num_tris
        p.frame, p.mode, p.spp, p.max_bounces, p.seed = frame, mode, spp, max_bounces, seed
        p.accumulate = 1 if accumulate else 0
        p.tile_rank, p.tile_count = tile_rank, tile_count
        p.flags = (PT_FLAG_STATS if stats else 0) | (PT_FLAG_SIMPLE_KERNEL if simple_kernel else 0) | (PT_FLAG_BRUTE_FORCE if brute_force else 0)
        return p

    def render(self, params):
        self._ck(lib.pt_render(self.h, C.byref(params)))
        self._last = (params.width, params.height)

    def last_render_ms(self):
        ms = C.c_float()
        self._ck(lib.pt_last_render_ms(self.h, C.byref(ms)))
        return ms.value

    def set_batch(self, frames_per_launch):
        self._ck(lib.pt_set_batch(self.h, C.c_uint32(frames_per_launch)))

    def flush(self):
        self._ck(lib.pt_flush(self.h))

    def timing_begin(self, capacity):
        self._ck(lib.pt_timing_begin(self.h, C.c_uint32(capacity)))

    def timing_collect(self, capacity):
        ms = np.zeros(capacity, np.float32); n = C.c_uint32()
        self._ck(lib.pt_timing_collect(self.h, _p(ms, C.c_float), C.c_uint32(capacity), C.byref(n)))
        return ms[: n.value].copy()

    def debug_set_tune(self, name, value=None):
        """Diagnostics: override one launch heuristic of the megakernel on this context (None restores the default)."""
        self._ck(lib.pt_debug_set_tune(self.h, name.encode(), C.c_uint32(0xFFFFFFFF if value is None else value)))

    def timing_collect_spans(self, capacity):
        """(start_ms, dur_ms) of the launches recorded since timing_begin; starts are relative to the first launch."""
        st = np.zeros(capacity, np.float32); ms = np.zeros(capacity, np.float32); n = C.c_uint32()
        self._ck(lib.pt_timing_collect_spans(self.h, _p(st, C.c_float), _p(ms, C.c_float), C.c_uint32(capacity), C.byref(n)))
        return st[: n.value].copy(), ms[: n.value].copy()

    def set_compact_buffer(self, device_ptr, floats):
        self._ck(lib.pt_set_compact_buffer(self.h, C.c_void_p(device_ptr), C.c_uint64(floats)))

    def set_output_buffer(self, device_ptr, floats):
        self._ck(lib.pt_set_output_buffer(self.h, C.c_void_p(device_ptr), C.c_uint64(floats)))

    def stats(self):
        st = PtStats()
        self._ck(lib.pt_get_stats(self.h, C.byref(st)))
        return st.as_dict()

    def read_radiance(self, width=None, height=None):
        w, h = (width, height) if width else self._last
        out = np.zeros((h, w, 4), np.float32)
        self._ck(lib.pt_read_radiance(self.h, _p(out, C.c_float), C.c_uint64(out.size)))
        return out

    def read_rgba8(self):
        w, h = self._last
        out = np.zeros((h, w, 4), np.uint8)
        self._ck(lib.pt_read_rgba8(self.h, _p(out, C.c_uint8), C.c_uint64(out.size)))
        return out

    def read_tonemapped(self, from_rgba8=True):
        w, h = self._last
        out = np.zeros((h, w, 4), np.uint8)
        self._ck(lib.pt_read_tonemapped(self.h, C.c_int(int(from_rgba8)), _p(out, C.c_uint8), C.c_uint64(out.size)))
        return out

    def compact_radiance(self):
        ptr, fl = C.c_void_p(), C.c_uint64()
        self._ck(lib.pt_compact_radiance(self.h, C.byref(ptr), C.byref(fl)))
        return ptr.value, fl.value

    def deinterleave(self, gathered_device_ptr, stride_floats, width, height, tile_count):
        self._ck(lib.pt_deinterleave(self.h, C.c_void_p(gathered_device_ptr), C.c_uint64(stride_floats),
                                     C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile_count)))
        self._last = (width, height)

    def deinterleave_batch(self, gathered_device_ptr, rank_stride_floats, frame_stride_floats, num_frames, width, height, tile_count, frames_out_ptr=0, out_stride_floats=0):
        self._ck(lib.pt_deinterleave_batch(self.h, C.c_void_p(gathered_device_ptr), C.c_uint64(rank_stride_floats), C.c_uint64(frame_stride_floats), C.c_uint32(num_frames),
                                           C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile_count), C.c_void_p(frames_out_ptr or None), C.c_uint64(out_stride_floats)))
        self._last = (width, height)

    # ---- packed tile shares (what a sharded frame ships) ----
    def traced_tile_rect(self, params):
        r = (C.c_uint32 * 4)()
        self._ck(lib.pt_traced_tile_rect(self.h, C.byref(params), r))
        return tuple(int(v) for v in r)

    def pack_shares(self, compact_ptr, frame_stride_floats, num_frames, width, height, tile_rank, tile_count, rect, packed_ptr, packed_frame_stride_floats):
        self._ck(lib.pt_pack_shares(self.h, C.c_void_p(compact_ptr), C.c_uint64(frame_stride_floats), C.c_uint32(num_frames), C.c_uint32(width), C.c_uint32(height),
                                    C.c_uint32(tile_rank), C.c_uint32(tile_count), (C.c_uint32 * 4)(*rect), C.c_void_p(packed_ptr), C.c_uint64(packed_frame_stride_floats)))

    def unpack_batch(self, gathered_ptr, rank_stride_floats, frame_stride_floats, num_frames, width, height, tile_count, rect, spp, frames_out_ptr=0, out_stride_floats=0):
        self._ck(lib.pt_unpack_batch(self.h, C.c_void_p(gathered_ptr), C.c_uint64(rank_stride_floats), C.c_uint64(frame_stride_floats), C.c_uint32(num_frames),
                                     C.c_uint32(width), C.c_uint32(height), C.c_uint32(tile_count), (C.c_uint32 * 4)(*rect), C.c_uint32(spp),
                                     C.c_void_p(frames_out_ptr or None), C.c_uint64(out_stride_floats)))
        self._last = (width, height)

    # ---- checkpoint / resume of a progressive accumulation ----
    def accum_info(self):
        info = PtAccumInfo()
        self._ck(lib.pt_accum_info(self.h, C.byref(info)))
        return info

    def read_accum(self):
        """(PtAccumInfo, raw f32 dump: per-pixel sums in xyz, sample count in w) of the running accumulation."""
        info = self.accum_info()
        out = np.zeros(int(info.floats), np.float32)
        self._ck(lib.pt_read_accum(self.h, _p(out, C.c_float), C.c_uint64(out.size)))
        return info, out

    def set_accum(self, info, data):
        data = np.ascontiguousarray(data, np.float32).reshape(-1)
        assert data.size == info.floats
        self._ck(lib.pt_set_accum(self.h, C.byref(info), _p(data, C.c_float)))

    def buffer_busy(self, device_ptr, nbytes):
        b = C.c_int()
        self._ck(lib.pt_buffer_busy(self.h, C.c_void_p(device_ptr), C.c_uint64(nbytes), C.byref(b)))
        return bool(b.value)


PT_GROUP_TRANSPORT_RCCL, PT_GROUP_TRANSPORT_COPY = 0, 1


class Group:
    """Several GPUs, one image per render(): the pt_group_* entry points (one context per device inside this process, tile shares
    gathered on rank 0 over RCCL, de-interleaved there).  `devices=None` takes every visible device."""

    def __init__(self, devices=None, transport=PT_GROUP_TRANSPORT_RCCL):
        lib.pt_group_last_error.restype = C.c_char_p
        lib.pt_group_last_error.argtypes = [C.c_void_p]
        lib.pt_group_destroy.restype = None
        lib.pt_group_destroy.argtypes = [C.c_void_p]
        h = C.c_void_p()
        if devices is None:
            rc = lib.pt_group_create(None, C.c_uint32(0), C.c_uint32(transport), C.byref(h))
        else:
            arr = (C.c_int * len(devices))(*devices)
            rc = lib.pt_group_create(arr, C.c_uint32(len(devices)), C.c_uint32(transport), C.byref(h))
        if rc != 0:
            msg = lib.pt_group_last_error(None)
            raise PtError(rc, msg.decode() if msg else "")
        self.h = h
        self.num_tris = 0

    def _ck(self, rc):
        if rc != 0:
            msg = lib.pt_group_last_error(self.h)
            raise PtError(rc, msg.decode() if msg else "")

    def close(self):
        if getattr(self, "h", None):
            lib.pt_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        n = C.c_uint32()
        self._ck(lib.pt_group_size(self.h, C.byref(n)))
        return n.value

    def set_triangles(self, tris):
        tris = np.ascontiguousarray(tris, np.float32).reshape(-1)
        self._ck(lib.pt_group_set_triangles(self.h, _p(tris, C.c_float), C.c_uint32(tris.size // 9)))
        self.num_tris = tris.size // 9

    def build_bvh(self):
        self._ck(lib.pt_group_build_bvh(self.h))

    def set_bvh4(self, bvh4):
        bvh4 = np.ascontiguousarray(bvh4, np.uint32)
        self._ck(lib.pt_group_set_bvh4(self.h, _p(bvh4, C.c_uint32), C.c_uint64(bvh4.size)))

    def set_bvh2(self, bvh2):
        bvh2 = np.ascontiguousarray(bvh2, np.uint32)
        self._ck(lib.pt_group_set_bvh2(self.h, _p(bvh2, C.c_uint32), C.c_uint64(bvh2.size)))

    def set_batch(self, frames_per_launch):
        self._ck(lib.pt_group_set_batch(self.h, C.c_uint32(frames_per_launch)))

    def make_params(self, *a, **kw):
        kw.setdefault("num_tris", self.num_tris)
        return Context.make_params(self, *a, **kw)

    def render(self, params):
        self._ck(lib.pt_group_render(self.h, C.byref(params)))
        self._last = (params.width, params.height)

    def flush(self):
        self._ck(lib.pt_group_flush(self.h))

    def synchronize(self):
        self._ck(lib.pt_group_synchronize(self.h))

    def read_radiance(self):
        w, h = self._last
        out = np.zeros((h, w, 4), np.float32)
        self._ck(lib.pt_group_read_radiance(self.h, _p(out, C.c_float), C.c_uint64(out.size)))
        return out

    def read_rgba8(self):
        w, h = self._last
        out = np.zeros((h, w, 4), np.uint8)
        self._ck(lib.pt_group_read_rgba8(self.h, _p(out, C.c_uint8), C.c_uint64(out.size)))
        return out


class PathTracer:
    """Python mirror of the reference's PathTracer class (src/libs/PathTracer.js): same method
    names, argument meaning and return shapes; `canvas` is any object with width/height."""

    def __init__(self, canvas, device=-1):
        self.canvas = canvas
        self.cameraPosition = [0.0, 0.0, 3.5]            # PathTracer.js:67
        self.cameraQuaternion = [0.0, 0.0, 0.0, 1.0]
        self.frameCount = 0
        self.trianglesData = np.array([                  # default tetrahedron, PathTracer.js:79-84
            1, 1, 1, -1, -1, 1, -1, 1, -1,
            1, 1, 1, -1, 1, -1, 1, -1, -1,
            1, 1, 1, 1, -1, -1, -1, -1, 1,
            -1, -1, 1, 1, -1, -1, -1, 1, -1], np.float32)
        self.options = {"mode": PT_MODE_REFERENCE, "spp": 1, "maxBounces": 0, "seed": 1, "accumulate": False}
        self._device = device
        self.ctx = None

    def initialize(self):                                 # PathTracer.js:97-102
        self.ctx = Context(self._device)

    def computeBVH2Sizing(self, numTris):                 # :227
        return compute_bvh2_sizing(numTris)

    def computeBVH4Sizing(self, numNodes4):               # :234
        return compute_bvh4_sizing(numNodes4)

    def buildMortonAndSort(self, trianglesData):          # :427
        m, t = morton_sort(trianglesData)
        return {"mortonSorted": m, "triIndexSorted": t}

    def collapseLBVH2ToBVH4(self, bvh2U32, numTris):      # :506
        b, n = collapse_lbvh2_to_bvh4(bvh2U32, numTris)
        return {"bvh4U32": b, "numNodes4": n}

    def readBVH2(self, nbytes=None):                      # :485
        out = self.ctx.read_bvh2()
        return out if nbytes is None else out[: max(4, nbytes) // 4]

    def buildBVH(self, trianglesData):                    # :671
        if self.ctx is None:
            return                                        # `if (!device) return`, :673
        self.ctx.set_triangles(trianglesData)
        self.ctx.build_bvh()

    def setScene(self, scene):                            # :751
        self.trianglesData = np.ascontiguousarray(scene.getTrianglesFloat32(), np.float32)
        self.buildBVH(self.trianglesData)

    def render(self):                                     # :756
        if self.ctx is None or not self.ctx.scene_info()["numNodes4"] and self.trianglesData.size:
            return
        o = self.options
        p = self.ctx.make_params(self.canvas.width, self.canvas.height, self.cameraPosition, self.cameraQuaternion,
                                 mode=o["mode"], spp=o["spp"], max_bounces=o["maxBounces"], seed=o["seed"],
                                 frame=self.frameCount, accumulate=o["accumulate"], num_tris=self.trianglesData.size // 9)
        self.ctx.render(p)

    def readRadiance(self):
        return self.ctx.read_radiance()

    def setCameraPosition(self, x, y, z):                 # :824
        self.cameraPosition = [x, y, z]

    def setCameraQuaternion(self, x, y, z, w):            # :828
        self.cameraQuaternion = [x, y, z, w]

    def setFrameCount(self, frameCount):                  # :832
        self.frameCount = frameCount
