"""The C-ABI library loads on a CPU-only box and exports every symbol include/gsfm_rot.h declares.
No compute call is made here (the product has no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import have_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    txt = open(os.path.join(ROOT, "include", "gsfm_rot.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(gsfm_[a-z0-9_]+)\s*\(", txt))
    names.discard("gsfm_loss_callback")
    return sorted(names)


def test_library_exports_every_declared_entry_point():
    from globalsfmpy_amd import _abi
    lib = _abi.load_library()
    names = _declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libgsfm_rot.so does not export %s" % n
    assert lib.gsfm_rot_abi_version() == 4


def test_struct_layout_matches_header_defaults():
    from globalsfmpy_amd import _abi
    lib = _abi.load_library()
    o = _abi.Options()
    lib.gsfm_rot_options_default(C.byref(o))
    # Ceres 1.14 defaults + the values hard-coded at reference estimator.cpp:72-74,176-179
    assert (o.max_num_iterations, o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (200, 1e-6, 1e-10, 1e-8)
    assert (o.initial_trust_region_radius, o.max_trust_region_radius, o.min_relative_decrease) == (1e4, 1e16, 1e-3)
    assert (o.min_lm_diagonal, o.max_lm_diagonal, o.jacobi_scaling) == (1e-6, 1e32, 1)
    assert o.cg_relative_tolerance == 1e-12 and o.verbose == 0
    assert C.sizeof(_abi.LossNode) == 32
    assert [lib.gsfm_rot_residual_dim(t) for t in range(9)] == [4, 9, 3, 3, 3, 3, 3, 3, 3]


def test_oracle_and_product_share_option_defaults(oracle):
    from globalsfmpy_amd import _abi
    a, b = _abi.Options(), _abi.Options()
    _abi.load_library().gsfm_rot_options_default(C.byref(a))
    oracle.lib().orc_options_default(C.byref(b))
    assert bytes(a) == bytes(b)


@pytest.mark.skipif(have_gpu(), reason="CPU-only behaviour")
def test_fails_loudly_without_a_gpu():
    from globalsfmpy_amd.solver import RotationProblem, SolverError
    with pytest.raises(SolverError, match="no HIP device"):
        RotationProblem(3, [0, 1], [1, 2], np.zeros((2, 3)))


def test_invalid_loss_programs_are_rejected_by_make_program():
    from globalsfmpy_amd import _abi
    with pytest.raises(ValueError):
        _abi.make_program([(0, 0.0, 0.0, 0.0)] * 17)


def test_header_is_valid_pedantic_c99_and_links(tmp_path):
    """include/gsfm_rot.h is a C header, not a C++ one in disguise: the plain-C example builds with -std=c99 -pedantic -Werror."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "c_abi_minimal")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_minimal.c"), "-L" + os.path.join(ROOT, "globalsfmpy_amd"), "-lgsfm_rot",
           "-Wl,-rpath," + os.path.join(ROOT, "globalsfmpy_amd"), "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_shared_libraries_depend_on_nothing_but_the_hip_runtime_and_libc():
    """The drop-in boundary carries no torch / python / rocSOLVER / RCCL link-time dependency (RCCL is dlopen'ed by the optional
    communicator library only)."""
    import shutil
    import subprocess
    if shutil.which("readelf") is None:
        pytest.skip("no readelf")
    allowed = ("libamdhip64.so", "libstdc++.so", "libm.so", "libgcc_s.so", "libc.so", "libdl.so", "ld-linux", "libpthread.so", "librt.so")
    for name, extra in (("libgsfm_rot.so", ()), ("libgsfm_rccl.so", ()), ("libgsfm_estimator.so", ("libgsfm_rot.so",))):
        path = os.path.join(ROOT, "globalsfmpy_amd", name)
        if not os.path.exists(path):
            pytest.skip(name + " not built")
        out = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
        needed = re.findall(r"\(NEEDED\)\s+Shared library: \[([^\]]+)\]", out)
        assert needed, name
        for lib in needed:
            assert lib.startswith(allowed + extra), (name, lib)


def test_summary_carries_the_abi_v4_fields_and_the_auxiliary_libraries_export_their_entry_points():
    """ABI v4: num_pcg_capped_steps, worst_accepted_cg_residual, num_forcing_restarts appended to gsfm_rot_summary (the C header, the ctypes
    mirror and the pybind summary agree); the peer-store library exports the error entry points of round 5, and bench.py's crash-line handler
    lives in its own bench-only library, not in a product one."""
    import ctypes as C
    from globalsfmpy_amd import _abi
    hdr = open(os.path.join(ROOT, "include", "gsfm_rot.h")).read()
    body = hdr[hdr.index("typedef struct {\n  int32_t termination;"):hdr.index("} gsfm_rot_summary;")]
    names_h = re.findall(r"^\s+(?:int32_t|uint64_t|double)\s+(\w+);", body, flags=re.M)
    names_py = [n for n, _ in _abi.Summary._fields_]
    assert names_h == names_py, (names_h, names_py)
    for n in ("num_pcg_capped_steps", "worst_accepted_cg_residual", "num_forcing_restarts"):
        assert n in names_py
    assert "#define GSFM_ROT_ABI_VERSION 4" in hdr
    mod = open(os.path.join(ROOT, "globalsfmpy_amd", "host", "module.cpp")).read()
    for n in ("num_pcg_capped_steps", "worst_accepted_cg_residual", "num_forcing_restarts", "num_inexact_steps"):
        assert 'd["%s"]' % n in mod
    peer = os.path.join(ROOT, "globalsfmpy_amd", "libgsfm_peer.so")
    guard = os.path.join(ROOT, "globalsfmpy_amd", "libgsfm_benchguard.so")
    if not (os.path.exists(peer) and os.path.exists(guard)):
        pytest.skip("auxiliary libraries not built")
    _abi.preload_hip_runtime()
    lp, lg = C.CDLL(peer), C.CDLL(guard)
    for n in ("gsfm_peer_create", "gsfm_peer_connect", "gsfm_peer_all_gather", "gsfm_peer_all_reduce_sum", "gsfm_peer_error", "gsfm_peer_error_take", "gsfm_peer_inject_error"):
        assert hasattr(lp, n), n
    assert not hasattr(lp, "gsfm_crash_line_arm")
    assert hasattr(lg, "gsfm_crash_line_arm") and hasattr(lg, "gsfm_crash_line_disarm")
