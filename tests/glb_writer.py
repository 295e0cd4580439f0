"""A GLB writer that turns a triangle soup into a *realistic* file -- what an exporter would make of a scene -- plus the independent numpy
statement of the glTF 2.0 rules that says which world-space triangles a loader has to get out of it (float64 until the final f32 store).

The file (write_realistic_glb): several meshes with several primitives each; a node hierarchy three levels deep that mixes TRS nodes and
`matrix` nodes (non-uniform scale, a mirrored node); every primitive's vertices are stored in the LOCAL space of its node (so the loader's
world matrices matter), de-duplicated and indexed -- u16 index buffers where a primitive has at most 65,535 distinct vertices, u32 where it
has more, one primitive left non-indexed; position data tightly packed for some primitives and in interleaved `byteStride` views (position +
normal, position + normal + uv) for others; one mesh referenced by no node, one line primitive (not a mesh), one empty node.

Test infrastructure only (tests/test_gpu_ingest.py, tests/test_js_glb_synthetic.py)."""
import json
import struct

import numpy as np


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


class _Bin:
    def __init__(self):
        self.bin = bytearray(); self.views = []; self.accessors = []

    def view(self, data, stride=None, target=None):
        while len(self.bin) % 4:
            self.bin.append(0)
        v = {"buffer": 0, "byteOffset": len(self.bin), "byteLength": len(data)}
        if stride:
            v["byteStride"] = stride
        if target:
            v["target"] = target
        self.bin += data
        self.views.append(v)
        return len(self.views) - 1

    def accessor(self, view, ctype, count, typ, offset=0, **extra):
        a = {"bufferView": view, "componentType": ctype, "count": count, "type": typ}
        if offset:
            a["byteOffset"] = offset
        a.update(extra)
        self.accessors.append(a)
        return len(self.accessors) - 1

    def write(self, path, doc):
        doc = dict(doc, asset={"version": "2.0", "generator": "tests/glb_writer.py"}, buffers=[{"byteLength": len(self.bin)}], bufferViews=self.views, accessors=self.accessors)
        js = json.dumps(doc).encode()
        js += b" " * ((4 - len(js) % 4) % 4)
        b = bytes(self.bin) + b"\0" * ((4 - len(self.bin) % 4) % 4)
        with open(path, "wb") as f:
            f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(b)))
            f.write(struct.pack("<II", len(js), 0x4E4F534A)); f.write(js)
            f.write(struct.pack("<II", len(b), 0x004E4942)); f.write(b)


def normalize_cube(tris64):
    """Scene.normalizeMesh (Scene.js:104-165), cube mode, on float64 world triangles [n, 3, 3] -> (normalised float64, center, scale)."""
    v = tris64.reshape(-1, 3)
    lo, hi = v.min(axis=0), v.max(axis=0)
    center = (lo + hi) * 0.5
    scale = 2.0 / float((hi - lo).max())
    return (tris64 - center) * scale, center, scale


def write_realistic_glb(path, tris9, seed=1):
    """tris9: f32[9N] world-space soup.  Writes `path`; returns the world triangles the glTF rules give for the file, float64 [N, 3, 3], in the
    loader's traversal order (which is the soup's order: the primitives take consecutive runs of it)."""
    rng = np.random.default_rng(seed)
    world = np.asarray(tris9, np.float32).reshape(-1, 3, 3).astype(np.float64)
    n = len(world)

    def uq():
        q = rng.normal(size=4)
        return q / np.linalg.norm(q)

    # ---- the node hierarchy (index: node) and its world matrices
    qa, qb, qc, qd, qf = uq(), uq(), uq(), uq(), uq()
    local = {
        0: trs(rng.uniform(-0.5, 0.5, 3), qa, rng.uniform(0.6, 1.6, 3)),            # root A: TRS, mesh 0
        1: trs(rng.uniform(-1, 1, 3), qb, [1.25, -0.8, 1.1]),                       # child of A: `matrix`, mirrored (negative determinant), mesh 1
        2: trs([0.5, 0.0, -0.25], [0, 0, 0, 1], [1, 1, 1]),                         # child of A: translation only, no mesh
        3: trs([0, 0, 0], qc, [2.0, 2.0, 2.0]),                                     # child of 2: rotation + uniform scale, mesh 2
        4: trs(rng.uniform(-2, 2, 3), qd, rng.uniform(0.5, 2, 3)),                  # child of 3: `matrix`, mesh 3
        5: trs([0, 0, 0], qf, [1, 1, 1]),                                           # second root: rotation only, mesh 4
        6: np.eye(4),                                                               # third root: nothing at all
    }
    parent = {0: None, 1: 0, 2: 0, 3: 2, 4: 3, 5: None, 6: None}
    wm = {}
    for i in range(7):
        wm[i] = local[i] if parent[i] is None else wm[parent[i]] @ local[i]
    node_json = [
        {"name": "A", "translation": local[0][:3, 3].tolist(), "rotation": qa.tolist(), "scale": None, "mesh": 0, "children": [1, 2]},
        {"name": "B", "matrix": local[1].T.reshape(-1).tolist(), "mesh": 1},
        {"name": "C", "translation": [0.5, 0.0, -0.25], "children": [3]},
        {"name": "D", "rotation": qc.tolist(), "scale": [2.0, 2.0, 2.0], "mesh": 2, "children": [4]},
        {"name": "E", "matrix": local[4].T.reshape(-1).tolist(), "mesh": 3},
        {"name": "F", "rotation": qf.tolist(), "mesh": 4},
        {"name": "G"},
    ]
    # TRS members are written from the matrices actually used, so that file and expectation cannot drift apart
    sa = np.linalg.norm(local[0][:3, :3], axis=0); node_json[0]["scale"] = sa.tolist()
    # DFS pre-order, a node's own primitives before its children: A(mesh0), B(mesh1), C, D(mesh2), E(mesh3), F(mesh4)
    mesh_node = {0: 0, 1: 1, 2: 3, 3: 4, 4: 5}
    # ---- primitives: consecutive runs of the soup; (mesh, layout) per primitive, sizes as fractions of N
    plan = [(0, "packed"), (0, "pn24"), (1, "packed"), (1, "pnu32"), (1, "nonindexed"), (2, "packed"), (2, "pn24"), (3, "packed"), (3, "packed"), (4, "pnu32"), (4, "packed")]
    frac = np.array([0.10, 0.04, 0.52, 0.03, 0.01, 0.05, 0.08, 0.02, 0.08, 0.03, 0.04])      # the third one is large enough for > 65,535 distinct vertices at 262,144 triangles: u32 indices
    cuts = np.concatenate([[0], np.minimum(n, np.round(np.cumsum(frac) / frac.sum() * n).astype(np.int64))]); cuts[-1] = n
    g = _Bin()
    meshes = [{"primitives": []} for _ in range(6)]
    expect = np.empty((n, 3, 3), np.float64)
    stats = {"u16": 0, "u32": 0, "nonindexed": 0, "interleaved": 0}
    for k, (mesh, layout) in enumerate(plan):
        a, b = int(cuts[k]), int(cuts[k + 1])
        if b <= a:
            continue
        w = wm[mesh_node[mesh]]
        inv = np.linalg.inv(w)
        v = world[a:b].reshape(-1, 3)
        loc = (np.concatenate([v, np.ones((len(v), 1))], axis=1) @ inv.T)[:, :3].astype(np.float32)          # what the file stores
        if layout == "nonindexed":
            verts, idx = loc, None
            stats["nonindexed"] += 1
        else:
            verts, idx = np.unique(loc, axis=0, return_inverse=True)
            idx = idx.reshape(-1)
        prim = {"attributes": {}}
        if layout in ("packed", "nonindexed"):
            vv = g.view(verts.tobytes(), target=34962)
            prim["attributes"]["POSITION"] = g.accessor(vv, 5126, len(verts), "VEC3", min=verts.min(axis=0).tolist(), max=verts.max(axis=0).tolist())
        else:
            width = 6 if layout == "pn24" else 8
            inter = np.zeros((len(verts), width), np.float32)
            inter[:, :3] = verts; inter[:, 3:6] = [0.0, 1.0, 0.0]
            if width == 8:
                inter[:, 6:] = rng.uniform(0, 1, (len(verts), 2))
            vv = g.view(inter.tobytes(), stride=4 * width, target=34962)
            prim["attributes"]["POSITION"] = g.accessor(vv, 5126, len(verts), "VEC3", min=verts.min(axis=0).tolist(), max=verts.max(axis=0).tolist())
            prim["attributes"]["NORMAL"] = g.accessor(vv, 5126, len(verts), "VEC3", offset=12)
            if width == 8:
                prim["attributes"]["TEXCOORD_0"] = g.accessor(vv, 5126, len(verts), "VEC2", offset=24)
            stats["interleaved"] += 1
        if idx is not None:
            if len(verts) <= 65535:
                prim["indices"] = g.accessor(g.view(idx.astype(np.uint16).tobytes(), target=34963), 5123, idx.size, "SCALAR"); stats["u16"] += 1
            else:
                prim["indices"] = g.accessor(g.view(idx.astype(np.uint32).tobytes(), target=34963), 5125, idx.size, "SCALAR"); stats["u32"] += 1
            used = verts[idx]
        else:
            used = verts
        meshes[mesh]["primitives"].append(prim)
        u = used.astype(np.float64)
        expect[a:b] = (np.concatenate([u, np.ones((len(u), 1))], axis=1) @ w.T)[:, :3].reshape(-1, 3, 3)
    # a line primitive on mesh 0 (not a mesh object: skipped by the loader) and a mesh no node uses
    meshes[0]["primitives"].append({"attributes": {"POSITION": meshes[0]["primitives"][0]["attributes"]["POSITION"]}, "mode": 1})
    meshes[5]["primitives"].append({"attributes": {"POSITION": meshes[0]["primitives"][0]["attributes"]["POSITION"]}})
    doc = {"scene": 0, "scenes": [{"name": "Scene", "nodes": [0, 5, 6]}], "nodes": node_json, "meshes": meshes,
           "materials": [{"name": "stone", "pbrMetallicRoughness": {"baseColorFactor": [0.9, 0.7, 0.3, 1.0]}}]}
    g.write(path, doc)
    return expect, stats
