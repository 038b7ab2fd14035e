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


def test_error_behaviour_of_the_host_entry_points():
    """Errors come back as negative codes with a message, never as a device fault: unregistered look-up ids, calls
    before keygen, secret-key operations on an evaluation-only context, wrong-sized key arrays."""
    import numpy as np
    from bmi_amd import tfhe
    e = tfhe.Engine(tfhe.default_params(q_bits=49))
    try:
        with pytest.raises(tfhe.BmiError, match="no keys"):
            e.encrypt(np.zeros(1, dtype=np.int64), e.delta_log())
        e.keygen(5)
        dl = e.delta_log()
        lid = e.lut_register(np.arange(-8, 8), 4, dl)
        ct = e.encrypt(np.array([3, -2]), dl)
        with pytest.raises(tfhe.BmiError, match="not registered"):
            e.pbs_host(ct, np.array([lid, lid + 1], dtype=np.uint32))
        with pytest.raises(tfhe.BmiError, match="not registered"):
            e.blind_rotate_host(e.keyswitch_host(ct), np.array([7, lid], dtype=np.uint32))
        assert list(e.decrypt(e.pbs_host(ct, np.array([lid, lid], dtype=np.uint32)), dl)) == [3, -2]   # still usable
        with pytest.raises(tfhe.BmiError):
            e.import_keys(None, None, np.zeros(5, dtype=np.uint64), np.zeros(5, dtype=np.uint64))
    finally:
        e.close()
