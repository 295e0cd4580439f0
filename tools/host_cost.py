import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
rt = importlib.import_module("raytracer-public_amd")
ctx = rt.Context(0); ctx.set_triangles(rt.procedural_scene(0, 871414)); ctx.build_bvh()
for name, mode, kw in (("reference", rt.PT_MODE_REFERENCE, {}), ("path", rt.PT_MODE_PATH, dict(spp=4, max_bounces=8))):
    p = ctx.make_params(1920, 1080, mode=mode, **kw)
    for _ in range(30): ctx.render(p)
    ctx.synchronize()
    n = 400
    t0 = time.perf_counter()
    for i in range(n):
        p.frame = i; ctx.render(p)
    t1 = time.perf_counter(); ctx.synchronize(); t2 = time.perf_counter()
    print("%s: submit %.1f us per render() call, total %.3f ms per frame" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e3))
