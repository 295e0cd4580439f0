import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
rt = importlib.import_module("raytracer-public_amd")
N = 871414
t0 = time.time(); tris = rt.procedural_scene(0, N); print("scene gen s", time.time() - t0)
ctx = rt.Context(0)
t0 = time.time(); ctx.set_triangles(tris); print("set_triangles s", time.time() - t0)
t0 = time.time(); ctx.build_bvh(); print("build_bvh s", time.time() - t0, ctx.scene_info())
for mode, spp, b in [(rt.PT_MODE_REFERENCE, 1, 0), (rt.PT_MODE_REFERENCE_PACKET, 1, 0), (rt.PT_MODE_PATH, 4, 8)]:
    p = ctx.make_params(1920, 1080, mode=mode, spp=spp, max_bounces=b)
    for i in range(3):
        ctx.render(p); ms = ctx.last_render_ms()
    print("mode", mode, "ms", ms, "Msamples/s", 1920 * 1080 * spp / ms / 1e3)
p = ctx.make_params(1920, 1080, mode=rt.PT_MODE_PATH, spp=4, max_bounces=8, stats=True)
ctx.render(p); print("stats ms", ctx.last_render_ms(), ctx.stats())
img = ctx.read_radiance()
print("mean radiance", img[..., :3].mean(), "hit frac", (img[..., 0] > 0.011).mean())
np.save("gpurun_out/first_frame.npy", img[::4, ::4, :3].astype(np.float16))
