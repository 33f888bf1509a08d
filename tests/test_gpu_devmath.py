"""csrc/devmath.hpp restates the device library's atan2 / exp with the polynomial coefficients held in scalar registers (it frees ~60
VGPRs in K2c).  The claim is bit-identity with atan2() / exp() on the device: checked here over 2^26 arguments per routine, special
values included (tools/check_devmath.hip).  Round 6: dense_kernels.hpp's chol_rsqrt -- rsqrt(double) without its closing special-case
select, on the chain of every pivot of the exact step's elimination -- against rsqrt() over every positive finite binade."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.gpu
def test_scalar_constant_atan2_and_exp_equal_the_device_library_bit_for_bit(tmp_path):
    exe = str(tmp_path / "check_devmath")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "globalsfmpy_amd", "csrc"),
                           "-o", exe, os.path.join(ROOT, "tools", "check_devmath.hip")])
    r = subprocess.run([exe, str(1 << 26)], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" 0 mismatches") == 4   # atan2_q1 (two argument sets), exp_sc, chol_rsqrt
