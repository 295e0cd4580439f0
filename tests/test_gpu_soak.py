"""Randomised differential test of the hot path (tools/soak.py): the persistent megakernel under random scenes,
cameras, resolutions, spp / bounces, tile shares, batch sizes and accumulation, bit for bit against the
one-pixel-per-lane kernel that test_gpu_parity.py pins to the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_configurations_short_soak():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "20", "7"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "soak ok" in out.stdout


def test_open_batch_is_launched_before_its_buffers_change(rt, gpu_ctx):
    """Frames queued by pt_set_batch must not be affected by a later frame of another shape: a different tile share
    rewrites the tile list, a larger resolution reallocates the output -- both launch the open batch first."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    tris = rt.procedural_scene(0, 15000)
    gpu_ctx.set_triangles(tris); gpu_ctx.build_bvh()
    w, h, count = 128, 80, 2
    kw = dict(mode=rt.PT_MODE_PATH, spp=2, max_bounces=3, seed=4)
    gpu_ctx.render(gpu_ctx.make_params(w, h, **kw)); full = gpu_ctx.read_radiance().copy()
    stride = max(rt.tile_layout(w, h, r, count)[1] for r in range(count))
    buf = C.c_void_p(); assert hip.hipMalloc(C.byref(buf), C.c_size_t(stride * 4 * count)) == 0
    gpu_ctx.set_batch(8)
    for r in range(count):                      # rank 1's frame arrives while rank 0's is still queued (same tile count)
        gpu_ctx.set_compact_buffer(buf.value + r * stride * 4, stride)
        gpu_ctx.render(gpu_ctx.make_params(w, h, tile_rank=r, tile_count=count, **kw))
    gpu_ctx.synchronize()
    gpu_ctx.set_compact_buffer(0, 0)
    gpu_ctx.deinterleave(buf.value, stride, w, h, count)
    got = gpu_ctx.read_radiance().copy()
    hip.hipFree(buf)
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))
    # a queued small frame, then a frame large enough to reallocate the output buffer
    gpu_ctx.render(gpu_ctx.make_params(w, h, **kw))
    big = gpu_ctx.make_params(2048, 1536, **kw)
    gpu_ctx.render(big)
    img = gpu_ctx.read_radiance()
    gpu_ctx.set_batch(1)
    gpu_ctx.render(big)
    assert np.array_equal(img.view(np.uint32), gpu_ctx.read_radiance().view(np.uint32))
