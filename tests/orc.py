"""ctypes view of oracle/liboracle.so -- the CPU oracle (test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")


class Params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("focal", C.c_float), ("aspect", C.c_float),
        ("cam_pos", C.c_float * 3), ("num_tris", C.c_uint32),
        ("cam_quat", C.c_float * 4),
        ("frame", C.c_uint32), ("mode", C.c_uint32),
        ("spp", C.c_uint32), ("max_bounces", C.c_uint32), ("seed", C.c_uint32),
        ("x0", C.c_uint32), ("y0", C.c_uint32), ("x1", C.c_uint32), ("y1", C.c_uint32),
        ("step_x", C.c_uint32), ("step_y", C.c_uint32),
        ("accum_frames", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_closest", "rays_shadow", "nodes_examined", "tris_tested",
        "node_fetches_ref", "stack_drops", "max_stack", "samples")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


MODE_PACKET, MODE_SINGLE, MODE_PATH = 0, 1, 2


def focal_aspect(width, height):
    """PathTracer.js:761-769: fov 70 deg, computed in double, stored as f32."""
    import math
    fov = (70.0 * math.pi) / 180
    return np.float32(1.0 / math.tan(0.5 * fov)), np.float32(width / height)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_f32_to_f16_trunc.restype = C.c_uint32; L.orc_f32_to_f16_trunc.argtypes = [C.c_float]
        L.orc_f32_to_f16_rtne.restype = C.c_uint32; L.orc_f32_to_f16_rtne.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float; L.orc_f16_to_f32.argtypes = [C.c_uint32]
        L.orc_increment_f16.restype = C.c_float; L.orc_increment_f16.argtypes = [C.c_float, C.c_int]
        L.orc_collapse_bvh4.restype = C.c_uint32
        L.orc_render.restype = C.c_int
        L.orc_trace_ray.restype = C.c_int
        L.orc_rnd.restype = C.c_float; L.orc_rnd.argtypes = [C.c_uint32] * 5
        L.orc_sincos_2pi.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_cosine_dir.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float)]
        F3 = C.POINTER(C.c_float)
        L.orc_slab.restype = C.c_int; L.orc_slab.argtypes = [F3, F3, F3, F3, C.c_float, F3]
        L.orc_moller_trumbore.restype = C.c_int; L.orc_moller_trumbore.argtypes = [F3, F3, F3, F3, F3, C.c_float, F3]
        L.orc_safe_inv_dir.argtypes = [F3, F3]
        L.orc_rotate_by_quat.argtypes = [F3, F3, F3]

    # ---- scene build -------------------------------------------------
    def morton_sort(self, tris):
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1)
        n = tris.size // 9
        m = np.zeros(n, np.uint32); t = np.zeros(n, np.uint32)
        self.lib.orc_morton_sort(_p(tris, C.c_float), C.c_uint32(n), _p(m, C.c_uint32), _p(t, C.c_uint32))
        return m, t

    def build_lbvh2(self, tris, morton=None, tri_idx=None):
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1)
        n = tris.size // 9
        if morton is None:
            morton, tri_idx = self.morton_sort(tris)
        out = np.zeros(1 + 6 * max(2 * n - 1, 0), np.uint32)
        self.lib.orc_build_lbvh2(_p(tris, C.c_float), C.c_uint32(n), _p(morton, C.c_uint32), _p(tri_idx, C.c_uint32), _p(out, C.c_uint32))
        return out

    def collapse_bvh4(self, bvh2, num_tris):
        bvh2 = np.ascontiguousarray(bvh2, dtype=np.uint32)
        out = np.zeros(1 + 8 * max(2 * num_tris - 1, 0), np.uint32)
        n4 = self.lib.orc_collapse_bvh4(_p(bvh2, C.c_uint32), C.c_uint32(num_tris), _p(out, C.c_uint32))
        return out[: 1 + 8 * n4].copy(), int(n4)

    def bvh4_wide(self, bvh2):
        bvh2 = np.ascontiguousarray(bvh2, dtype=np.uint32)
        out = np.zeros(1 + 8 * int(bvh2[0]), np.uint32)
        self.lib.orc_bvh4_wide(_p(bvh2, C.c_uint32), _p(out, C.c_uint32))
        return out

    def build_bvh4(self, tris):
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1)
        n = tris.size // 9
        bvh2 = self.build_lbvh2(tris)
        bvh4, n4 = self.collapse_bvh4(bvh2, n)
        return bvh2, bvh4

    # ---- render ------------------------------------------------------
    def make_params(self, width, height, num_tris, cam_pos=(0, 0, 2.5), cam_quat=(0, 0, 0, 1), mode=MODE_SINGLE,
                    spp=1, max_bounces=0, seed=1, frame=0, rect=None, step=(1, 1), accum_frames=1):
        focal, aspect = focal_aspect(width, height)
        p = Params()
        p.width, p.height, p.focal, p.aspect = width, height, float(focal), float(aspect)
        p.cam_pos[:] = [float(np.float32(v)) for v in cam_pos]
        p.cam_quat[:] = [float(np.float32(v)) for v in cam_quat]
        p.num_tris = num_tris
        p.frame, p.mode, p.spp, p.max_bounces, p.seed = frame, mode, spp, max_bounces, seed
        x0, y0, x1, y1 = rect if rect else (0, 0, width, height)
        p.x0, p.y0, p.x1, p.y1 = x0, y0, x1, y1
        p.step_x, p.step_y = step
        p.accum_frames = accum_frames
        return p

    def render(self, params, tris, bvh4, want_tri_ids=False):
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1)
        bvh4 = np.ascontiguousarray(bvh4, dtype=np.uint32)
        img = np.zeros((params.height, params.width, 4), np.float32)
        ids = np.full((params.height, params.width), 0xFFFFFFFF, np.uint32) if want_tri_ids else None
        st = Stats()
        rc = self.lib.orc_render(C.byref(params), _p(tris, C.c_float), _p(bvh4, C.c_uint32), _p(img, C.c_float),
                                 _p(ids, C.c_uint32) if want_tri_ids else None, C.byref(st))
        assert rc == 0
        return img, ids, st.as_dict()

    def render_mt(self, params, tris, bvh4, threads=None, band=8):
        """The same render over `band`-row strips on several host threads (the C call releases the GIL; strips write
        disjoint rows).  Returns (image, stats) -- the counters summed over the strips, max_stack as a maximum."""
        from concurrent.futures import ThreadPoolExecutor
        import copy
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1)
        bvh4 = np.ascontiguousarray(bvh4, dtype=np.uint32)
        img = np.zeros((params.height, params.width, 4), np.float32)
        threads = threads or max(1, min(16, os.cpu_count() or 1))
        sy = max(1, params.step_y)
        band = ((band + sy - 1) // sy) * sy          # strips start on the subsample grid
        y0, y1 = params.y0, min(params.y1, params.height)
        strips = [(y, min(y + band, y1)) for y in range(y0, y1, band)]

        def work(strip):
            p = Params.from_buffer_copy(params)
            p.y0, p.y1 = strip
            st = Stats()
            rc = self.lib.orc_render(C.byref(p), _p(tris, C.c_float), _p(bvh4, C.c_uint32), _p(img, C.c_float), None, C.byref(st))
            assert rc == 0
            return st.as_dict()

        with ThreadPoolExecutor(max_workers=threads) as ex:
            parts = list(ex.map(work, strips))
        total = {k: (max(d[k] for d in parts) if k == "max_stack" else sum(d[k] for d in parts)) for k in parts[0]}
        return img, total

    def render_brute(self, params, tris, spheres):
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1)
        spheres = np.ascontiguousarray(spheres, dtype=np.float32).reshape(-1)
        img = np.zeros((params.height, params.width, 4), np.float32)
        st = Stats()
        rc = self.lib.orc_render_brute(C.byref(params), _p(tris, C.c_float), _p(spheres, C.c_float), C.c_uint32(spheres.size // 4), _p(img, C.c_float), C.byref(st))
        assert rc == 0
        return img, st.as_dict()

    def trace_ray(self, tris, bvh4, o, d, anyhit=False):
        tris = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1)
        bvh4 = np.ascontiguousarray(bvh4, dtype=np.uint32)
        o = np.asarray(o, np.float32); d = np.asarray(d, np.float32)
        t = C.c_float(); n = (C.c_float * 3)(); tri = C.c_uint32()
        hit = self.lib.orc_trace_ray(_p(tris, C.c_float), _p(bvh4, C.c_uint32), C.c_uint32(tris.size // 9),
                                     _p(o, C.c_float), _p(d, C.c_float), C.c_int(int(anyhit)), C.byref(t), n, C.byref(tri))
        return bool(hit), t.value, np.array(list(n), np.float32), tri.value

    # ---- single-lane probes of the intersection routines --------------
    def slab(self, o, inv, mn, mx, best=1e30):
        """intersectAABBPacketMask with one lane (renderer.wgsl:121-169) -> (hit, tmin)."""
        a = [np.ascontiguousarray(v, np.float32) for v in (o, inv, mn, mx)]
        t = C.c_float()
        hit = self.lib.orc_slab(*[_p(v, C.c_float) for v in a], C.c_float(best), C.byref(t))
        return bool(hit), np.float32(t.value)

    def moller_trumbore(self, o, d, v0, v1, v2, best=1e30):
        """intersectTrianglePacket with one lane (renderer.wgsl:171-208) -> (hit, t)."""
        a = [np.ascontiguousarray(v, np.float32) for v in (o, d, v0, v1, v2)]
        t = C.c_float()
        hit = self.lib.orc_moller_trumbore(*[_p(v, C.c_float) for v in a], C.c_float(best), C.byref(t))
        return bool(hit), np.float32(t.value)

    def rotate_by_quat(self, v, q):
        """rotateVectorByQuat (renderer.wgsl:66-72), quaternion xyzw -> rotated vector (f32)."""
        v = np.ascontiguousarray(v, np.float32); q = np.ascontiguousarray(q, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.orc_rotate_by_quat(_p(v, C.c_float), _p(q, C.c_float), _p(out, C.c_float))
        return out

    def safe_inv_dir(self, d):
        d = np.ascontiguousarray(d, np.float32); out = np.zeros(3, np.float32)
        self.lib.orc_safe_inv_dir(_p(d, C.c_float), _p(out, C.c_float))
        return out

    def tonemap(self, rgba, quantize=True):
        rgba = np.ascontiguousarray(rgba, np.float32)
        h, w = rgba.shape[:2]
        out = np.zeros((h, w, 4), np.uint8)
        self.lib.orc_tonemap(_p(rgba, C.c_float), C.c_uint32(w), C.c_uint32(h), C.c_int(int(quantize)), _p(out, C.c_uint8))
        return out


def build():
    """Compile the oracle (and oracle/_ref when the reference checkout is present)."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "all", "ref"])


def load():
    if not os.path.exists(LIB):
        build()
    return Oracle(C.CDLL(LIB))
