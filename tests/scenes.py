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


def _f16_down(v):
    h = np.float16(v)
    return h if np.float32(h) <= np.float32(v) else np.nextafter(h, np.float16(-np.inf))


def _f16_up(v):
    h = np.float16(v)
    return h if np.float32(h) >= np.float32(v) else np.nextafter(h, np.float16(np.inf))


def pack_box(mn, mx):
    """Three words of a node record (renderer.wgsl:94-99): f16 bounds rounded outwards."""
    lo = [int(np.array(_f16_down(v), np.float16).view(np.uint16)) for v in mn]
    hi = [int(np.array(_f16_up(v), np.float16).view(np.uint16)) for v in mx]
    return [lo[0] | (lo[1] << 16), lo[2] | (hi[0] << 16), hi[1] | (hi[2] << 16)]


INVALID, LEAF = 0xFFFFFFFF, 0x80000000


def comb_bvh4(levels, seed, all_hit=True):
    """A hand-made BVH4 that is one long chain: every internal node has three leaf children and (at a random slot) the
    next chain node; the last one has four leaves.  Seen from +z all four child boxes of a node overlap on screen and the
    chain child's box is the nearest (its leaves lie in front of this level's), so a ray that hits everything carries
    three stacked leaves per level and overruns the reference's 64-entry stack (renderer.wgsl:337) from level 21 on --
    where the push that is dropped first is the NEAREST child's, i.e. the rest of the chain.
    Returns (tris f32[9N], bvh4 u32[1+8M])."""
    rng = np.random.default_rng(seed)
    tris, recs = [], []          # recs: [w0,w1,w2,c0,c1,c2,c3,meta]
    z_front = 0.9

    def z_of(l, k):
        return -0.9 + 1.7 * (l * 4 + k) / (levels * 4)

    def leaf(l, k):
        z = np.float32(z_of(l, k))
        if all_hit:
            x0, x1, y0, y1 = -0.95, 0.95, -0.95, 0.95
        else:
            x0, x1 = sorted(rng.uniform(-0.95, 0.95, 2)); y0, y1 = sorted(rng.uniform(-0.95, 0.95, 2))
            x1 = max(x1, x0 + 0.3); y1 = max(y1, y0 + 0.3)
        xm = rng.uniform(x0, x1)
        ti = len(tris)
        tris.append([x0, y0, z, x1, y0, z, xm, y1, z])
        box = pack_box((x0, y0, z - 1e-3), (x1, y1, z + 1e-3))
        recs.append(box + [INVALID] * 4 + [LEAF | ti])
        return len(recs) - 1

    chain = []
    for l in range(levels):
        recs.append(None); chain.append(len(recs) - 1)
    for l in range(levels):
        kids = [leaf(l, k) for k in range(3)]
        if l + 1 < levels:
            kids.insert(int(rng.integers(0, 4)), chain[l + 1])
        else:
            kids.append(leaf(l, 3))
        box = pack_box((-0.95, -0.95, z_of(l, 0) - 1e-3), (0.95, 0.95, z_front))
        recs[chain[l]] = box + kids + [0]
    bvh4 = np.array([len(recs)] + [w for r in recs for w in r], np.uint32)
    return np.array(tris, np.float32).reshape(-1), bvh4


def spoil_bvh4(bvh4, seed, count=40):
    """Copies a valid BVH4 and plants the children the reference skips (renderer.wgsl:288-291): child indices >= numNodes
    (never fetched) and children whose box has min > max on one axis (fetched, then rejected for every ray)."""
    rng = np.random.default_rng(seed)
    b = bvh4.copy()
    m = int(b[0])
    internal = [i for i in range(m) if not (b[1 + 8 * i + 7] & LEAF)]
    picks = rng.choice(internal, size=min(count, len(internal)), replace=False)
    n_oob = n_deg = 0
    for j, i in enumerate(picks):
        base = 1 + 8 * int(i)
        slots = [s for s in range(4) if b[base + 3 + s] != INVALID and b[base + 3 + s] < m]
        if not slots:
            continue
        s = int(rng.choice(slots)); c = int(b[base + 3 + s])
        if j % 2 == 0:
            b[base + 3 + s] = m + int(rng.integers(0, 1000)) if j % 4 == 0 else 0xFFFFFFF0
            n_oob += 1
        else:
            cb = 1 + 8 * c
            w0, w1, w2 = int(b[cb]), int(b[cb + 1]), int(b[cb + 2])
            mnx, mxx = w0 & 0xFFFF, w1 >> 16
            if np.array(mnx, np.uint16).view(np.float16) < np.array(mxx, np.uint16).view(np.float16):
                b[cb] = (w0 & 0xFFFF0000) | mxx; b[cb + 1] = (w1 & 0xFFFF) | (mnx << 16)      # min.x <-> max.x
                n_deg += 1
    return b, n_oob, n_deg
