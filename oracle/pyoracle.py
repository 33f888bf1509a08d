"""TEST INFRASTRUCTURE -- ctypes binding of oracle/liborc.so (the CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from globalsfmpy_amd import _abi
from globalsfmpy_amd.solver import ProblemBase, _prep_edges, _dp, _u32p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liborc.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def usable_cores():
    """CPU threads this process can actually keep busy: the affinity mask capped by the cgroup CPU quota (the GPU boxes
    expose 256 hardware threads under a 16-CPU quota; 256 OpenMP threads on that are throttled into the ground)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]           # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_problem_create.argtypes = [C.c_uint32, C.c_uint64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                         C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_problem_create.restype = C.c_void_p
        _abi.declare_solver_signatures(L, "orc_")
        L.orc_options_default.argtypes = [C.POINTER(_abi.Options)]
        L.orc_set_linear_solver.argtypes = [C.c_void_p, C.c_int32]
        L.orc_residual_dim.argtypes = [C.c_int32]; L.orc_residual_dim.restype = C.c_int32
        L.orc_loss_eval.argtypes = [C.POINTER(_abi.LossNode), C.c_int32, C.c_double, C.POINTER(C.c_double)]
        L.orc_magsac_table.argtypes = [C.c_int32, C.POINTER(C.c_double), C.c_int32]; L.orc_magsac_table.restype = C.c_int32
        L.orc_magsac_constants.argtypes = [C.c_int32] + [C.POINTER(C.c_double)] * 3
        L.orc_whitening.argtypes = [C.c_int32, C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)]
        for name in ("orc_angle_axis_to_rotation_matrix", "orc_rotation_matrix_to_angle_axis",
                     "orc_angle_axis_to_quaternion", "orc_quaternion_to_angle_axis"):
            getattr(L, name).argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_pairwise_rotation_error.argtypes = [C.POINTER(C.c_double)] * 3 + [C.c_double, C.POINTER(C.c_double)]
        L.orc_edge_jacobians.argtypes = [C.c_void_p, C.c_uint64] + [C.POINTER(C.c_double)] * 4
        L.orc_capture_steps.argtypes = [C.c_void_p, C.c_int32]
        L.orc_captured_step.argtypes = [C.c_void_p, C.c_int32] + [C.POINTER(C.c_double)] * 7
        L.orc_cov_estimate.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)] + [C.POINTER(C.c_double)] * 4 + [C.c_int32] + [C.POINTER(C.c_double)] * 3 + [C.POINTER(C.c_int32)] * 2
        L.orc_sampson_residual.argtypes = [C.POINTER(C.c_double)] * 5
        L.orc_sampson_residual.restype = C.c_double
        L.orc_homogeneous_plus.argtypes = [C.POINTER(C.c_double)] * 3
        L.orc_homogeneous_jacobian.argtypes = [C.POINTER(C.c_double)] * 2
        L.orc_set_num_threads.argtypes = [C.c_int]; L.orc_set_num_threads.restype = C.c_int
        L.orc_set_num_threads(usable_cores())
        _lib = L
    return _lib


class OracleProblem(ProblemBase):
    _prefix = "orc_"

    def __init__(self, n_cams, edge_i, edge_j, rel_aa, error_type=_abi.ANGLE_AXIS, cov6=None, inlier_weight=None):
        L = lib()
        ei, ej, rel, c6, iw, n_edges = _prep_edges(n_cams, edge_i, edge_j, rel_aa, cov6, inlier_weight)
        h = L.orc_problem_create(int(n_cams), int(n_edges), _u32p(ei), _u32p(ej), _dp(rel), int(error_type), _dp(c6), _dp(iw))
        if not h:
            raise ValueError("orc_problem_create rejected the inputs")
        ProblemBase.__init__(self, L, C.c_void_p(h), n_cams, n_edges, error_type, L.orc_residual_dim(int(error_type)))

    def _options_default(self, o):
        self._lib.orc_options_default(C.byref(o))

    def set_linear_solver(self, kind):
        self._lib.orc_set_linear_solver(self._h, {"auto": 0, "dense": 1, "pcg": 2}[kind])

    def capture_steps(self, max_steps):
        """Keep the linear systems of the first `max_steps` LM iterations of the next solve (test hook)."""
        self._lib.orc_capture_steps(self._h, int(max_steps))

    def captured_step(self, k):
        """System k of the last solve as dict(Ji, Jj (E x R x 3), rt (E x R), D, rhs, y, scale (3N), cg): (J^T J + diag(D)^2) y = rhs."""
        E, R, n = self.n_edges, self.residual_dim, 3 * self.n_cams
        out = {"Ji": np.empty((E, R, 3)), "Jj": np.empty((E, R, 3)), "rt": np.empty((E, R)), "D": np.empty(n), "rhs": np.empty(n), "y": np.empty(n), "scale": np.empty(n)}
        cg = self._lib.orc_captured_step(self._h, int(k), *[_dp(out[key]) for key in ("Ji", "Jj", "rt", "D", "rhs", "y", "scale")])
        if cg < 0:
            raise IndexError("LM iteration %d was not captured" % k)
        out["cg"] = cg
        return out

    def edge_jacobians(self, e, rot_aa):
        rot = np.ascontiguousarray(rot_aa, dtype=np.float64)
        R = self.residual_dim
        r, Ji, Jj = np.empty(R), np.empty((R, 3)), np.empty((R, 3))
        st = self._lib.orc_edge_jacobians(self._h, int(e), _dp(rot), _dp(r), _dp(Ji), _dp(Jj))
        assert st == 0
        return r, Ji, Jj


def loss_eval(nodes, s):
    arr, n = _abi.make_program(nodes)
    out = (C.c_double * 3)()
    lib().orc_loss_eval(arr, n, float(s), out)
    return np.array([out[0], out[1], out[2]])


def magsac_table(nu):
    n = lib().orc_magsac_table(int(nu), None, 0)
    out = np.empty(n)
    lib().orc_magsac_table(int(nu), _dp(out), n)
    return out


def magsac_constants(nu):
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib().orc_magsac_constants(int(nu), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def whitening(error_type, cov6, inlier_w=1.0):
    c = np.ascontiguousarray(cov6, dtype=np.float64)
    W = np.empty(9)
    lib().orc_whitening(int(error_type), _dp(c), float(inlier_w), _dp(W))
    return W.reshape(3, 3)


def _vec_fn(name, x, nout):
    a = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty(nout)
    getattr(lib(), name)(_dp(a), _dp(out))
    return out


def angle_axis_to_rotation_matrix(aa):
    return _vec_fn("orc_angle_axis_to_rotation_matrix", aa, 9).reshape(3, 3)


def rotation_matrix_to_angle_axis(R):
    return _vec_fn("orc_rotation_matrix_to_angle_axis", np.asarray(R, dtype=np.float64).reshape(9), 3)


def angle_axis_to_quaternion(aa):
    return _vec_fn("orc_angle_axis_to_quaternion", aa, 4)


def quaternion_to_angle_axis(q):
    return _vec_fn("orc_quaternion_to_angle_axis", q, 3)


def pairwise_rotation_error(aa1, aa2, rel_aa, weight=1.0):
    a, b, c = [np.ascontiguousarray(v, dtype=np.float64) for v in (aa1, aa2, rel_aa)]
    out = np.empty(3)
    lib().orc_pairwise_rotation_error(_dp(a), _dp(b), _dp(c), float(weight), _dp(out))
    return out


def estimate_rotation_covariances(match_ptr, matches, intrinsics, rot, trans, max_iterations=500):
    """CPU oracle of globalsfmpy_amd.covariance.estimate_rotation_covariances (same outputs)."""
    from globalsfmpy_amd.covariance import _call
    code, out = _call(lib().orc_cov_estimate, match_ptr, matches, intrinsics, rot, trans, max_iterations, False)
    assert code == 0
    return out


def sampson_residual(match4, intr6, rot, t, want_jacobian=False):
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (match4, intr6, rot, t)]
    jac = np.zeros(6)
    r = lib().orc_sampson_residual(_dp(a[0]), _dp(a[1]), _dp(a[2]), _dp(a[3]), _dp(jac) if want_jacobian else None)
    return (r, jac) if want_jacobian else r


def homogeneous_plus(x3, d2):
    x, d, out = np.ascontiguousarray(x3, dtype=np.float64), np.ascontiguousarray(d2, dtype=np.float64), np.zeros(3)
    lib().orc_homogeneous_plus(_dp(x), _dp(d), _dp(out))
    return out


def homogeneous_jacobian(x3):
    x, out = np.ascontiguousarray(x3, dtype=np.float64), np.zeros((3, 2))
    lib().orc_homogeneous_jacobian(_dp(x), _dp(out))
    return out
