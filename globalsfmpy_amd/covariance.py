"""Batched per-edge rotation covariance (reference src/uncertainty.cpp:36-198) over the C-ABI `gsfm_cov_estimate`,
plus a seeded synthetic two-view generator for tests and benchmarks."""
import ctypes as C

import numpy as np

from . import _abi
from .solver import SolverError, _dp


def _call(fn, match_ptr, matches, intrinsics, rot, trans, max_iterations, with_ms):
    mp = np.ascontiguousarray(match_ptr, dtype=np.uint64)
    E = mp.shape[0] - 1
    m = np.ascontiguousarray(matches, dtype=np.float64).reshape(-1, 4)
    K = np.ascontiguousarray(intrinsics, dtype=np.float64).reshape(E, 6)
    r = np.ascontiguousarray(rot, dtype=np.float64).reshape(E, 3)
    t = np.ascontiguousarray(trans, dtype=np.float64).reshape(E, 3)
    cov = np.zeros((E, 3, 3)); ro = np.zeros((E, 3)); to = np.zeros((E, 3))
    st = np.zeros(E, dtype=np.int32); it = np.zeros(E, dtype=np.int32)
    ms = C.c_double(0)
    args = [C.c_uint64(E), mp.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(m), _dp(K), _dp(r), _dp(t), int(max_iterations), _dp(cov), _dp(ro), _dp(to),
            st.ctypes.data_as(C.POINTER(C.c_int32)), it.ctypes.data_as(C.POINTER(C.c_int32))]
    if with_ms:
        args.append(C.byref(ms))
    code = fn(*args)
    return code, {"cov": cov, "rotation": ro, "translation": to, "status": st, "iterations": it, "kernel_ms": ms.value}


def estimate_rotation_covariances(match_ptr, matches, intrinsics, rot, trans, max_iterations=500):
    """Device path. Returns dict(cov (E,3,3), rotation, translation, status, iterations, kernel_ms)."""
    lib = _abi.load_library()
    code, out = _call(lib.gsfm_cov_estimate, match_ptr, matches, intrinsics, rot, trans, max_iterations, True)
    if code != 0:
        raise SolverError("gsfm_cov_estimate failed with status %d: %s" % (code, lib.gsfm_last_error().decode("utf-8", "replace")))
    return out


def cov_to_cov6(cov):
    """(E,3,3) -> the C00 C11 C22 C01 C02 C12 order of covariance_rot.txt / gsfm_rot_problem_create."""
    c = np.asarray(cov)
    return np.stack([c[:, 0, 0], c[:, 1, 1], c[:, 2, 2], c[:, 0, 1], c[:, 0, 2], c[:, 1, 2]], axis=1)


def make_two_view_batch(n_edges, seed, matches_per_edge=(60, 400), noise_px=0.5, init_rot_noise=0.01, init_t_noise=0.02):
    """Synthetic view pairs under the functor's convention x2^T K2^-T R [t]x K1^-1 x1 = 0."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ptr = [0]
    ms, Ks, rs, ts, gt_r = [], [], [], [], []
    for _ in range(n_edges):
        n = int(rng.integers(matches_per_edge[0], matches_per_edge[1] + 1))
        f1, f2 = rng.uniform(800, 1600, 2)
        u1, v1, u2, v2 = rng.uniform(300, 900, 4)
        w = rng.uniform(-0.4, 0.4, 3)
        th = np.linalg.norm(w)
        k = w / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        Rm = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.standard_normal(3); t /= np.linalg.norm(t)
        X1 = np.c_[rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(4, 9, n)]
        X2 = (Rm @ (X1 + t).T).T                       # (X2)^T R (t x X1) = (X1 + t).(t x X1) = 0
        x1 = np.c_[f1 * X1[:, 0] / X1[:, 2] + u1, f1 * X1[:, 1] / X1[:, 2] + v1] + noise_px * rng.standard_normal((n, 2))
        x2 = np.c_[f2 * X2[:, 0] / X2[:, 2] + u2, f2 * X2[:, 1] / X2[:, 2] + v2] + noise_px * rng.standard_normal((n, 2))
        ms.append(np.c_[x1, x2]); Ks.append([f1, u1, v1, f2, u2, v2]); gt_r.append(w)
        rs.append(w + init_rot_noise * rng.standard_normal(3)); ts.append(t + init_t_noise * rng.standard_normal(3))
        ptr.append(ptr[-1] + n)
    return {"match_ptr": np.array(ptr, dtype=np.uint64), "matches": np.ascontiguousarray(np.vstack(ms)), "intrinsics": np.array(Ks),
            "rot": np.array(rs), "trans": np.array(ts), "gt_rot": np.array(gt_r)}
