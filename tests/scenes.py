"""Small seeded scenes shared by the tests."""
import numpy as np

TETRA = np.array([   # the reference's default mesh, PathTracer.js:79-84
    1, 1, 1, -1, -1, 1, -1, 1, -1,
    1, 1, 1, -1, 1, -1, 1, -1, -1,
    1, 1, 1, 1, -1, -1, -1, -1, 1,
    -1, -1, 1, 1, -1, -1, -1, 1, -1], np.float32)


def random_soup(n, seed, size=0.15):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 1, 3), dtype=np.float32) * 1.6 - 0.8
    return (c + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * size).astype(np.float32).reshape(-1)


def quat_yaw_pitch(yaw, pitch):
    """q = yaw(Y) * pitch(X), xyzw (input-handler.js:101-104)."""
    cy, sy, cp, sp = np.cos(yaw / 2), np.sin(yaw / 2), np.cos(pitch / 2), np.sin(pitch / 2)
    # (0,sy,0,cy) * (sp,0,0,cp)
    return (cy * sp, sy * cp, -sy * sp, cy * cp)


def cornell():
    """Config C1 (build-defined): a box open towards the camera (5 quads = 10 triangles) and 2 spheres."""
    def quad(a, b, c, d):
        return [a, b, c, a, c, d]
    v = []
    v += quad((-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1))     # floor
    v += quad((-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1))         # ceiling
    v += quad((-1, -1, -1), (-1, 1, -1), (1, 1, -1), (1, -1, -1))     # back wall
    v += quad((-1, -1, -1), (-1, -1, 1), (-1, 1, 1), (-1, 1, -1))     # left wall
    v += quad((1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1))         # right wall
    tris = np.array(v, np.float32).reshape(-1)
    spheres = np.array([-0.45, -0.6, -0.3, 0.4, 0.5, -0.65, 0.25, 0.35], np.float32)
    assert tris.size == 90
    return tris, spheres
