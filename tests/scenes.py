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
