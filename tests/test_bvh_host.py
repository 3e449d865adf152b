"""rustray_amd/csrc/rr_bvh.cpp (the host-side BVH builder and BVH2 -> BVH4 collapse) under AddressSanitizer + UBSan on
the CPU: empty meshes / zero-item top levels, single primitives, coincident boxes, deep median splits."""
import os
import subprocess

from tests.helpers import ROOT


def test_bvh_builder_degenerate_inputs_under_asan(tmp_path):
    exe = str(tmp_path / "bvh_host_test")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-D__HIP_PLATFORM_AMD__", f"-I{rocm}/include", "-o", exe,
           os.path.join(ROOT, "tests", "native", "bvh_host_test.cpp"), os.path.join(ROOT, "rustray_amd", "csrc", "rr_bvh.cpp")]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bvh host test OK" in out.stdout, out.stdout + out.stderr
