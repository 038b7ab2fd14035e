"""The C ABI from plain C (no Python, no torch): tests/c_abi/abi_smoke.c is compiled with gcc against
include/bmi_tfhe.h, linked with libbmi_tfhe.so and the system HIP runtime, and run on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_program_drives_the_abi(tmp_path):
    lib_dir = os.path.join(REPO, "bounty-matrix-inversion_amd", "lib")
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(REPO, "tests", "c_abi", "abi_smoke.c"),
                           "-I", os.path.join(REPO, "include"), "-L", lib_dir, "-lbmi_tfhe",
                           "-L/opt/rocm/lib", "-lamdhip64", "-lstdc++", "-lm",
                           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke OK" in out.stdout
