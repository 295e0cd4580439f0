"""Scene.loadGLB (raytracer-public_amd/js/Scene.js, the dependency-free restatement of the reference's Scene.js:15-165)
on synthetic GLBs that exercise what the two bundled fixtures do not: nested nodes with TRS and matrix transforms,
u8 / u16 / u32 indices, non-indexed primitives, interleaved buffer views (byteStride; indexed AND non-indexed), integer positions
(KHR_mesh_quantization), skipped point / line primitives, several root nodes.  Expected triangles come from an
independent numpy implementation of the glTF 2.0 rules (float64 until the final f32 store), with ONE rule taken from the reference
instead of the specification: the reference reads the position attribute's raw array (Scene.js:66-85), so a `normalized` integer
accessor yields the integers, not integer / 32767 -- tests/test_js_gltf_golden.py pins that against three's own GLTFLoader.
The realistic 262,144-triangle file of tests/glb_writer.py (what tests/test_gpu_ingest.py renders) is loaded here too."""
import json
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def trs(t, q, s):
    m = np.eye(4)
    m[:3, :3] = quat_to_mat(q) @ np.diag(s)
    m[:3, 3] = t
    return m


class Glb:
    def __init__(self):
        self.bin = bytearray(); self.views = []; self.accessors = []

    def view(self, data, stride=None):
        while len(self.bin) % 4: self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride: v["byteStride"] = stride
        self.bin += data
        self.views.append(v)
        return len(self.views) - 1

    def accessor(self, view, ctype, count, typ, offset=0, normalized=False):
        a = {"bufferView": view, "componentType": ctype, "count": count, "type": typ, "byteOffset": offset}
        if normalized: a["normalized"] = True
        self.accessors.append(a)
        return len(self.accessors) - 1

    def write(self, path, doc):
        doc = dict(doc, asset={"version": "2.0"}, buffers=[{"byteLength": len(self.bin)}], bufferViews=self.views, accessors=self.accessors)
        js = json.dumps(doc).encode()
        js += b" " * ((4 - len(js) % 4) % 4)
        b = bytes(self.bin) + b"\0" * ((4 - len(self.bin) % 4) % 4)
        with open(path, "wb") as f:
            f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(b)))
            f.write(struct.pack("<II", len(js), 0x4E4F534A)); f.write(js)
            f.write(struct.pack("<II", len(b), 0x004E4942)); f.write(b)


def build_case(tmp_path, seed):
    rng = np.random.default_rng(seed)
    g = Glb()
    prims = []          # (positions float64 [n,3] as the accessor decodes them, index list)

    def positions_f32(n):
        return rng.uniform(-2, 2, (n, 3)).astype(np.float32)

    # mesh 0 / primitive 0: interleaved position + normal (stride 24), u16 indices
    n = 40; p = positions_f32(n); inter = np.zeros((n, 6), np.float32); inter[:, :3] = p; inter[:, 3:] = 1
    v = g.view(inter.tobytes(), stride=24); ap = g.accessor(v, 5126, n, "VEC3")
    idx = rng.integers(0, n, 3 * 50).astype(np.uint16); ai = g.accessor(g.view(idx.tobytes()), 5123, idx.size, "SCALAR")
    prims.append((p.astype(np.float64), idx.astype(np.int64)))
    m0p0 = {"attributes": {"POSITION": ap}, "indices": ai}
    # mesh 0 / primitive 1: non-indexed, with an accessor byteOffset into a shared view; explicit mode 4
    n = 33; p = positions_f32(n + 2); v = g.view(p.tobytes()); ap = g.accessor(v, 5126, n, "VEC3", offset=24)
    prims.append((p[2:].astype(np.float64), np.arange(n)))
    m0p1 = {"attributes": {"POSITION": ap}, "mode": 4}
    # mesh 0 / primitive 2: a line primitive (skipped)
    m0p2 = {"attributes": {"POSITION": ap}, "mode": 1}
    # mesh 1 / primitive 0: normalised int16 positions (KHR_mesh_quantization), u32 indices
    n = 25; q = rng.integers(-32767, 32768, (n, 3)).astype(np.int16); q[0] = [-32768, 32767, 0]
    pad = np.zeros((n, 4), np.int16); pad[:, :3] = q
    v = g.view(pad.tobytes(), stride=8); ap = g.accessor(v, 5122, n, "VEC3", normalized=True)
    idx = rng.integers(0, n, 3 * 30).astype(np.uint32); ai = g.accessor(g.view(idx.tobytes()), 5125, idx.size, "SCALAR")
    prims.append((q.astype(np.float64), idx.astype(np.int64)))              # RAW integers, as the reference reads them (module docstring)
    m1p0 = {"attributes": {"POSITION": ap}, "indices": ai}
    # mesh 1 / primitive 1: u8 indices, normalised uint8 positions
    n = 12; q8 = rng.integers(0, 256, (n, 3)).astype(np.uint8); pad8 = np.zeros((n, 4), np.uint8); pad8[:, :3] = q8
    v = g.view(pad8.tobytes(), stride=4); ap = g.accessor(v, 5121, n, "VEC3", normalized=True)
    idx = rng.integers(0, n, 3 * 9).astype(np.uint8); ai = g.accessor(g.view(idx.tobytes()), 5121, idx.size, "SCALAR")
    prims.append((q8.astype(np.float64), idx.astype(np.int64)))
    m1p1 = {"attributes": {"POSITION": ap}, "indices": ai}
    # mesh 1 / primitive 2: NON-indexed over an interleaved view (position + uv, stride 20).  The one documented difference from the reference, which
    # reads this layout without its stride (js/Scene.js header): the glTF rules are what is expected here
    n = 21; p = positions_f32(n); inter = np.zeros((n, 5), np.float32); inter[:, :3] = p; inter[:, 3:] = 0.5
    ap = g.accessor(g.view(inter.tobytes(), stride=20), 5126, n, "VEC3")
    prims.append((p.astype(np.float64), np.arange(n)))
    m1p2 = {"attributes": {"POSITION": ap}}

    qa = rng.normal(size=4); qa /= np.linalg.norm(qa)
    qb = rng.normal(size=4); qb /= np.linalg.norm(qb)
    t_a, s_a = rng.uniform(-1, 1, 3), rng.uniform(0.5, 2, 3)
    m_c = trs(rng.uniform(-1, 1, 3), qb, rng.uniform(0.5, 1.5, 3))        # child given as a column-major matrix
    nodes = [
        {"translation": t_a.tolist(), "rotation": qa.tolist(), "scale": s_a.tolist(), "mesh": 0, "children": [1, 2]},   # 0: root A
        {"matrix": m_c.T.reshape(-1).tolist(), "mesh": 1},                                                              # 1: child with a matrix
        {"translation": [0.5, 0.0, -0.25], "children": [3]},                                                            # 2: empty node
        {"scale": [2.0, 2.0, 2.0], "mesh": 0},                                                                          # 3: grandchild re-using mesh 0
        {"rotation": qb.tolist(), "mesh": 1},                                                                           # 4: second root
    ]
    doc = {"scene": 0, "scenes": [{"nodes": [0, 4]}], "nodes": nodes,
           "meshes": [{"primitives": [m0p0, m0p1, m0p2]}, {"primitives": [m1p0, m1p1, m1p2]}],
           "extensionsUsed": ["KHR_mesh_quantization"]}
    path = os.path.join(str(tmp_path), "case%d.glb" % seed)
    g.write(path, doc)

    # expected: depth-first, node's own mesh before its children (Scene.js:47-99 / Object3D.traverse)
    mesh_prims = {0: [prims[0], prims[1]], 1: [prims[2], prims[3], prims[4]]}
    world = {}
    m_a = trs(t_a, qa, s_a)
    world[0] = m_a; world[1] = m_a @ m_c; world[2] = m_a @ trs([0.5, 0.0, -0.25], [0, 0, 0, 1], [1, 1, 1])
    world[3] = world[2] @ trs([0, 0, 0], [0, 0, 0, 1], [2, 2, 2]); world[4] = trs([0, 0, 0], qb, [1, 1, 1])
    order = [(0, 0), (1, 1), (3, 0), (4, 1)]          # (node, mesh) in traversal order: 0, its children 1 then 2 -> 3, then root 4
    tris = []
    for node, mesh in order:
        for pos, idx in mesh_prims[mesh]:
            pw = (world[node] @ np.concatenate([pos, np.ones((len(pos), 1))], axis=1).T).T[:, :3]
            nt = len(idx) // 3
            tris.append(pw[idx[: nt * 3]].reshape(nt, 9))
    return path, np.concatenate(tris).astype(np.float32)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_synthetic_glb_matches_numpy_gltf_rules(tmp_path, seed):
    path, want = build_case(tmp_path, seed)
    out = os.path.join(str(tmp_path), "tris.f32")
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js_glb_dump.js"), path, out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float32).reshape(-1, 9)
    assert int(r.stdout.strip()) == len(want) == len(got)
    # same rules, different operation order (numpy matmul vs three's compose/multiply): equal to f32 rounding
    assert np.allclose(got, want, rtol=2e-6, atol=2e-6)
    # normalisation: cube mode maps the bounding box to [-1, 1] on its longest axis, centred (Scene.js:104-165)
    r = subprocess.run([NODE, os.path.join(ROOT, "tests", "js_glb_dump.js"), path, out, "normalize"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    nrm = np.fromfile(out, np.float32).reshape(-1, 3)
    lo, hi = nrm.min(axis=0), nrm.max(axis=0)
    assert np.allclose((lo + hi) / 2, 0, atol=1e-5) and abs((hi - lo).max() - 2.0) < 1e-5


def _dump(path, out, *more):
    return subprocess.run([NODE, os.path.join(ROOT, "tests", "js_glb_dump.js"), path, out] + list(more), capture_output=True, text=True, timeout=120)


def test_unsupported_features_fail_loudly(tmp_path):
    """What the reference's loader refuses is refused here, loudly: Draco (no DRACOLoader set, Scene.js:6), required meshopt compression (no decoder
    set), primitive modes beyond 6, files that are neither GLB nor JSON; plus what a file on disk can get wrong (a missing external buffer, an
    accessor or an index that reads beyond its data, remote uris)."""
    out = os.path.join(str(tmp_path), "o")
    p = np.zeros((3, 3), np.float32)

    def one(name, prim=None, patch=None):
        g = Glb()
        ap = g.accessor(g.view(p.tobytes()), 5126, 3, "VEC3")
        ai = g.accessor(g.view(np.array([0, 1, 5], np.uint16).tobytes()), 5123, 3, "SCALAR")
        doc = {"scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}], "meshes": [{"primitives": [dict({"attributes": {"POSITION": ap}}, **(prim(ai) if prim else {}))]}]}
        path = os.path.join(str(tmp_path), name)
        g.write(path, doc)
        if patch:
            patch(g, path, doc)
        return path

    r = _dump(one("draco.glb", lambda ai: {"extensions": {"KHR_draco_mesh_compression": {"bufferView": 0, "attributes": {"POSITION": 0}}}}), out)
    assert r.returncode != 0 and "Draco" in (r.stderr + r.stdout)
    r = _dump(one("mode7.glb", lambda ai: {"mode": 7}), out)
    assert r.returncode != 0 and "Primitive mode unsupported" in (r.stderr + r.stdout)
    r = _dump(one("badindex.glb", lambda ai: {"indices": ai}), out)
    assert r.returncode != 0 and "outside its POSITION accessor" in (r.stderr + r.stdout)

    def meshopt(g, path, doc):
        g.views[0]["extensions"] = {"EXT_meshopt_compression": {"buffer": 0, "byteLength": 4, "byteStride": 12, "count": 3, "mode": "ATTRIBUTES"}}
        g.write(path, dict(doc, extensionsUsed=["EXT_meshopt_compression"], extensionsRequired=["EXT_meshopt_compression"]))
    r = _dump(one("meshopt.glb", None, meshopt), out)
    assert r.returncode != 0 and "meshopt" in (r.stderr + r.stdout)

    def short(g, path, doc):
        g.accessors[0]["count"] = 30
        g.write(path, doc)
    r = _dump(one("short.glb", None, short), out)
    assert r.returncode != 0 and "beyond the end" in (r.stderr + r.stdout)
    # .gltf JSON whose external buffer is not there / is remote
    base = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}], "meshes": [{"primitives": [{"attributes": {"POSITION": 0}}]}],
            "bufferViews": [{"buffer": 0, "byteLength": 36}], "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}]}
    for name, uri, word in (("missing.gltf", "nowhere.bin", "ENOENT"), ("remote.gltf", "https://example.com/a.bin", "remote")):
        path = os.path.join(str(tmp_path), name)
        json.dump(dict(base, buffers=[{"byteLength": 36, "uri": uri}]), open(path, "w"))
        r = _dump(path, out)
        assert r.returncode != 0 and word in (r.stderr + r.stdout), (name, r.stderr)
    with open(os.path.join(str(tmp_path), "junk.glb"), "wb") as f:
        f.write(b"not a glb at all")
    r = _dump(os.path.join(str(tmp_path), "junk.glb"), out)
    assert r.returncode != 0
    # an old asset version
    path = os.path.join(str(tmp_path), "v1.gltf")
    json.dump(dict(base, asset={"version": "1.0"}, buffers=[{"byteLength": 36, "uri": "data:application/octet-stream;base64," + "A" * 48}]), open(path, "w"))
    r = _dump(path, out)
    assert r.returncode != 0 and "version" in (r.stderr + r.stdout)


def test_realistic_multi_mesh_glb_matches_the_gltf_rules(tmp_path):
    """tests/glb_writer.py on the 262,144-triangle sponza-class mesh of C4 (the GPU suite renders this very file, tests/test_gpu_ingest.py): several
    meshes and primitives, TRS and matrix nodes three levels deep, u16 and u32 index buffers, interleaved views, a non-indexed primitive."""
    import importlib, sys
    sys.path.insert(0, ROOT)
    from glb_writer import write_realistic_glb, normalize_cube
    rt = importlib.import_module("raytracer-public_amd")
    tris = rt.procedural_scene(rt.SCENE_SPONZA_CLASS, 262144)             # C4's mesh: large enough for one primitive to need u32 indices
    path = os.path.join(str(tmp_path), "realistic.glb")
    expect, stats = write_realistic_glb(path, tris, seed=9)
    assert stats["u16"] >= 2 and stats["u32"] >= 1 and stats["interleaved"] >= 2 and stats["nonindexed"] == 1
    out = os.path.join(str(tmp_path), "t.f32")
    r = _dump(path, out)
    assert r.returncode == 0, r.stderr
    got = np.fromfile(out, np.float32).reshape(-1, 3, 3)
    assert len(got) == 262144
    assert np.allclose(got, expect.astype(np.float32), rtol=2e-6, atol=2e-6)
    assert np.allclose(got.reshape(-1), tris, rtol=0, atol=1e-5)             # and that is the soup the file was made from
    r = _dump(path, out, "normalize")
    assert r.returncode == 0, r.stderr
    want, _, _ = normalize_cube(expect)
    assert np.allclose(np.fromfile(out, np.float32).reshape(-1, 3, 3), want.astype(np.float32), rtol=2e-6, atol=2e-6)
