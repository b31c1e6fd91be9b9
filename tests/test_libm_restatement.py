"""CPU: csrc/device_libm.h -- the C library functions the path calls (sinf, cosf, acosf, tanf, powf, atan2f), restated from glibc 2.35 so
that the device returns the reference build's bits -- compiled for the HOST and compared with this machine's libm, bit for bit.

tests/tools/libm_check.c walks every stride-th float32 bit pattern (NaNs, infinities and subnormals included); stride 1 (all 2^32
arguments of each function, 0 differ) is what round 5 ran once (about 4 CPU-minutes on 8 cores); here a prime stride
keeps it to a second.  The same functions as the DEVICE computes them are checked in tests/test_hip_parity.py
(test_device_libm_is_the_c_librarys)."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _glibc():
    f = ctypes.CDLL(None).gnu_get_libc_version
    f.restype = ctypes.c_char_p
    return f().decode()


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    out = tmp_path_factory.mktemp("libm") / "libm_check"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-fopenmp", "-x", "c++", os.path.join(ROOT, "tests", "tools", "libm_check.c"), "-o", str(out), "-lm"],
                   check=True, cwd=os.path.join(ROOT, "tests", "tools"))
    return str(out)


@pytest.mark.parametrize("stride", [4099, 65537 * 3 + 2])
def test_restated_libm_has_this_machines_bits(checker, stride):
    if _glibc() != "2.35":
        pytest.skip(f"device_libm.h restates glibc 2.35 (the image the reference build was made in); this machine has {_glibc()}")
    r = subprocess.run([checker, str(stride)], capture_output=True, text=True, env=dict(os.environ, OMP_NUM_THREADS="4"))
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if "arguments" in ln]
    assert len(lines) == 13 and all(" 0 differ" in ln for ln in lines), r.stdout
