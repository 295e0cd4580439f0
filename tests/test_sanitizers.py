"""Host-side code of the product (pt_host.cpp) and the oracle under AddressSanitizer + UBSan (CPU build only;
GPU sanitizers are not available on the pool)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_host_build_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_sanitize")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           os.path.join(HERE, "native", "host_sanitize.cpp"), os.path.join(ROOT, "raytracer-public_amd", "csrc", "pt_host.cpp"),
           os.path.join(ROOT, "oracle", "pt_oracle.cpp"), "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.check_output([exe], text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert "host_sanitize ok" in out, out
