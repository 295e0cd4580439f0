"""The camera convention the renderer's UBO carries, pinned to the dependency the reference builds it with.

FPSCamera (src/libs/controls/input-handler.js:99-110) makes `q = normalize(yaw(Y) * pitch(X))` with gl-matrix 3.4.4 and moves along
`transformQuat((0,0,-1) / (1,0,0) / (0,1,0), q)`; src/main.js:72-73 hands q (xyzw) to PathTracer.setCameraQuaternion and the shader
rotates its camera-space ray directions by it (rotateVectorByQuat, renderer.wgsl:66-72, :391).  The fixture
tests/golden/glmatrix_camera_golden.json was produced by requiring the reference's own node_modules/gl-matrix
(tests/golden/gen_golden_glmatrix.js, build container only) and holds inputs and outputs as f64.

Checked here: tests/scenes.py::quat_yaw_pitch (what every test camera is built with) is gl-matrix's quaternion, and the oracle's
rotateVectorByQuat agrees with gl-matrix's transformQuat on the camera axes and on arbitrary vectors -- to 1e-6 (f32 arithmetic on
the oracle's side, doubles in the fixture): multiplication order, handedness and xyzw layout are the reference's."""
import json
import os

import numpy as np
import pytest

import scenes

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "glmatrix_camera_golden.json")))


def test_fixture_shape(golden):
    assert "gl-matrix 3.4.4" in golden["source"]
    assert len(golden["cases"]) == 64
    c = golden["cases"][1]                      # yaw = +90 degrees: the camera looks down -x, its right is -z
    assert np.allclose(c["fwd"], [-1, 0, 0], atol=1e-12) and np.allclose(c["right"], [0, 0, -1], atol=1e-12)


def test_quat_yaw_pitch_is_gl_matrix(golden):
    for c in golden["cases"]:
        q = np.array(scenes.quat_yaw_pitch(c["yaw"], c["pitch"]), np.float64)
        assert np.allclose(q, c["q"], atol=1e-12), (c["yaw"], c["pitch"])
        assert abs(np.dot(q, q) - 1.0) < 1e-12


def test_oracle_rotation_is_gl_matrix_transform_quat(orc, golden):
    worst = 0.0
    for c in golden["cases"]:
        q32 = np.array(c["q"], np.float32)
        for v, want in [((0, 0, -1), c["fwd"]), ((1, 0, 0), c["right"]), ((0, 1, 0), c["up"])] + [(p["v"], p["out"]) for p in c["probes"]]:
            got = orc.rotate_by_quat(v, q32).astype(np.float64)
            worst = max(worst, float(np.abs(got - np.array(want)).max()))
    assert worst < 1e-6, worst
