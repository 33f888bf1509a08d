"""Flat-array host API over the C-ABI (include/gsfm_rot.h).

`RotationProblem` is the numpy-facing wrapper used by bench.py, the tests and the
GlobalSfMpy-compatible estimator classes.  It owns one `gsfm_rot_problem`.
"""
import ctypes as C
import numpy as np

from . import _abi


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


class SolverError(RuntimeError):
    pass


class ProblemBase(object):
    """Shared call sequence for anything exporting the gsfm_rot_* argument lists.
    (The CPU oracle under oracle/ mirrors them with an orc_ prefix for the parity tests.)"""

    _prefix = "gsfm_rot_"

    def __init__(self, lib, handle, n_cams, n_edges, error_type, residual_dim):
        self._lib = lib
        self._h = handle
        self.n_cams = int(n_cams)
        self.n_edges = int(n_edges)
        self.error_type = int(error_type)
        self.residual_dim = int(residual_dim)
        self._keep = []

    def _fn(self, name):
        return getattr(self._lib, self._prefix + name)

    def _check(self, st, what):
        # every C call may have run the host-callback loss: an exception parked by the trampoline surfaces HERE, from the call that caused
        # it, not from some later, unrelated one
        self._reraise_callback_error()
        if st != 0:
            raise SolverError("%s failed with status %d: %s" % (what, st, self._last_error()))
        comm = getattr(self, "_comm", None)
        if comm is not None and hasattr(comm, "take_error") and comm.take_error():
            # (peer-store exchange: a wait for a peer's slice timed out in the last calls of this solve -- what it returned is not a result;
            # the captured PCG chunks still hold the peer kernels: set_stream drops them, the next solve captures the fallback's)
            self._lib.gsfm_rot_set_stream(self._h, C.c_void_p(comm.stream_handle() or 0))
            raise SolverError("%s: the peer-store exchange timed out waiting for a peer; the result is invalid (later solves use the fallback collectives)" % what)

    def _last_error(self):
        return ""

    def close(self):
        if self._h is not None:
            self._fn("problem_destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- loss ---------------------------------------------------------------
    def set_loss(self, loss):
        """loss: None (Ceres NULL loss), a list of (kind, p0, p1, p2) nodes, or an object with
        .native_program() (globalsfmpy_amd.loss_functions classes); any other object with
        .Evaluate(s, out) is used through the host callback path."""
        if loss is None:
            nodes = []
        elif isinstance(loss, (list, tuple)):
            nodes = list(loss)
        elif hasattr(loss, "native_program") and loss.native_program() is not None:
            nodes = loss.native_program()
        elif hasattr(loss, "Evaluate"):
            return self.set_loss_callback(loss.Evaluate)
        else:
            raise TypeError("unsupported loss object %r" % (loss,))
        arr, n = _abi.make_program(nodes)
        self._check(self._fn("set_loss")(self._h, arr, n), "set_loss")

    def set_loss_callback(self, evaluate):
        """Host-callback loss.  An exception raised by `evaluate` cannot cross the C boundary: it is recorded, the edge gets NaN
        (the solve then ends with FAILURE / nonfinite instead of running on stale values) and solve() re-raises it."""
        self._cb_error = None

        def _cb(_user, s, out):
            try:
                buf = [0.0, 0.0, 0.0]
                evaluate(s, buf)
                out[0], out[1], out[2] = buf[0], buf[1], buf[2]
            except BaseException as e:  # noqa: BLE001
                if self._cb_error is None:
                    self._cb_error = e
                out[0] = out[1] = out[2] = float("nan")
        cb = _abi.LOSS_CALLBACK_FN(_cb)
        self._keep.append(cb)
        self._check(self._fn("set_loss_callback")(self._h, cb, None), "set_loss_callback")

    def _reraise_callback_error(self):
        e, self._cb_error = getattr(self, "_cb_error", None), None
        if e is not None:
            raise e

    def set_edge_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert w.shape == (self.n_edges,)
        self._check(self._fn("set_edge_weights")(self._h, _dp(w)), "set_edge_weights")

    # -- evaluation ---------------------------------------------------------
    def residuals(self, rot_aa, want_residuals=False):
        rot = np.ascontiguousarray(rot_aa, dtype=np.float64).reshape(self.n_cams, 3)
        s = np.empty(self.n_edges)
        rho = np.empty((self.n_edges, 3))
        r = np.empty((self.n_edges, self.residual_dim)) if want_residuals else None
        cost = C.c_double(0)
        st = self._fn("residuals")(self._h, _dp(rot), _dp(s), _dp(rho), _dp(r), C.byref(cost))
        self._reraise_callback_error()
        self._check(st, "residuals")
        return {"s": s, "rho": rho, "residuals": r, "cost": cost.value}

    def linearize(self, rot_aa):
        rot = np.ascontiguousarray(rot_aa, dtype=np.float64).reshape(self.n_cams, 3)
        g = np.empty((self.n_cams, 3))
        d = np.empty((self.n_cams, 3, 3))
        cost = C.c_double(0)
        self._check(self._fn("linearize")(self._h, _dp(rot), _dp(g), _dp(d), C.byref(cost)), "linearize")
        return {"gradient": g, "diag_blocks": d, "cost": cost.value}

    def normal_matvec(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64).reshape(self.n_cams, 3)
        y = np.empty_like(v)
        self._check(self._fn("normal_matvec")(self._h, _dp(v), _dp(y)), "normal_matvec")
        return y

    # -- solve --------------------------------------------------------------
    def default_options(self):
        o = _abi.Options()
        self._options_default(o)
        return o

    def _options(self, kw):
        o = self.default_options()
        for k, v in kw.items():
            if not hasattr(o, k):
                raise TypeError("unknown solver option %r" % k)
            setattr(o, k, v)
        return o

    def solve(self, rot_aa, **options):
        rot = np.array(rot_aa, dtype=np.float64, order="C").reshape(self.n_cams, 3)
        o = self._options(options)
        s = _abi.Summary()
        st = self._fn("solve")(self._h, _dp(rot), C.byref(o), C.byref(s))
        self._reraise_callback_error()
        self._check(st, "solve")
        return rot, s.as_dict()

    def solve_sigma_consensus(self, rot_aa, iters_num, sigma_max, **options):
        rot = np.array(rot_aa, dtype=np.float64, order="C").reshape(self.n_cams, 3)
        o = self._options(options)
        s = _abi.Summary()
        st = self._fn("solve_sigma_consensus")(self._h, _dp(rot), int(iters_num), float(sigma_max), C.byref(o), C.byref(s))
        self._reraise_callback_error()
        self._check(st, "solve_sigma_consensus")
        return rot, s.as_dict()

    def trace(self):
        rows = self._fn("get_trace")(self._h, None, 0)
        out = np.zeros((max(rows, 0), 8))
        if rows > 0:
            self._fn("get_trace")(self._h, _dp(out), rows)
        return out


def _device_rotations(rot_dev, n_cams):
    """Device address of a (n_cams, 3) float64 buffer: a contiguous torch CUDA tensor, anything with __cuda_array_interface__, or an int."""
    if isinstance(rot_dev, int):
        return rot_dev
    if hasattr(rot_dev, "data_ptr"):   # torch
        if not rot_dev.is_cuda or str(rot_dev.dtype) != "torch.float64" or not rot_dev.is_contiguous() or rot_dev.numel() != 3 * n_cams:
            raise ValueError("solve_resident needs a contiguous float64 CUDA tensor of %d x 3" % n_cams)
        return int(rot_dev.data_ptr())
    cai = getattr(rot_dev, "__cuda_array_interface__", None)
    if cai is None:
        raise TypeError("solve_resident needs device memory (a torch CUDA tensor, __cuda_array_interface__ or a raw address); host arrays go to solve()")
    if cai["typestr"] not in ("<f8", "=f8") or int(np.prod(cai["shape"])) != 3 * n_cams or cai.get("strides") is not None:
        raise ValueError("solve_resident needs a contiguous float64 device buffer of %d x 3" % n_cams)
    return int(cai["data"][0])


def _prep_edges(n_cams, edge_i, edge_j, rel_aa, cov6, inlier_weight):
    ei = np.ascontiguousarray(edge_i, dtype=np.uint32)
    ej = np.ascontiguousarray(edge_j, dtype=np.uint32)
    rel = np.ascontiguousarray(rel_aa, dtype=np.float64).reshape(-1, 3)
    n_edges = ei.shape[0]
    if ej.shape[0] != n_edges or rel.shape[0] != n_edges:
        raise ValueError("edge arrays disagree in length")
    c6 = None if cov6 is None else np.ascontiguousarray(cov6, dtype=np.float64).reshape(n_edges, 6)
    iw = None if inlier_weight is None else np.ascontiguousarray(inlier_weight, dtype=np.float64).reshape(n_edges)
    return ei, ej, rel, c6, iw, n_edges


class RotationProblem(ProblemBase):
    """The product: gsfm_rot_problem on the current HIP device."""

    def __init__(self, n_cams, edge_i, edge_j, rel_aa, error_type=_abi.ANGLE_AXIS, cov6=None,
                 inlier_weight=None, shard=None, stream=None):
        lib = _abi.load_library()
        ei, ej, rel, c6, iw, n_edges = _prep_edges(n_cams, edge_i, edge_j, rel_aa, cov6, inlier_weight)
        h = C.c_void_p()
        st = lib.gsfm_rot_problem_create(int(n_cams), int(n_edges), _u32p(ei), _u32p(ej), _dp(rel), int(error_type),
                                         _dp(c6), _dp(iw), C.byref(shard) if shard is not None else None, C.byref(h))
        if st != 0:
            raise SolverError("gsfm_rot_problem_create failed with status %d: %s"
                              % (st, lib.gsfm_last_error().decode("utf-8", "replace")))
        ProblemBase.__init__(self, lib, h, n_cams, n_edges, error_type, lib.gsfm_rot_residual_dim(int(error_type)))
        self._shard = shard
        if stream is not None:
            self._check(lib.gsfm_rot_set_stream(self._h, C.c_void_p(int(stream))), "set_stream")

    def _last_error(self):
        return self._lib.gsfm_last_error().decode("utf-8", "replace")

    def _options_default(self, o):
        self._lib.gsfm_rot_options_default(C.byref(o))

    def solve_resident(self, rot_dev, **options):
        """gsfm_rot_solve_resident: `rot_dev` is DEVICE memory on the problem's device ((n_cams, 3) float64, contiguous: a torch CUDA tensor,
        an object with __cuda_array_interface__, or a raw address), read as the start and overwritten with the result -- nothing but the
        summary crosses PCIe.  Returns the summary; the stream is synchronised on return."""
        ptr = _device_rotations(rot_dev, self.n_cams)
        o = self._options(options)
        s = _abi.Summary()
        st = self._lib.gsfm_rot_solve_resident(self._h, C.c_void_p(ptr), C.byref(o), C.byref(s))
        self._reraise_callback_error()
        self._check(st, "solve_resident")
        return s.as_dict()

    def time_sweep(self, rot_aa, reps=20):
        rot = np.ascontiguousarray(rot_aa, dtype=np.float64).reshape(self.n_cams, 3)
        ms = C.c_double(0)
        self._check(self._lib.gsfm_rot_time_sweep(self._h, _dp(rot), int(reps), C.byref(ms)), "time_sweep")
        return ms.value

    def time_kernels(self, rot_aa, reps=10):
        """Mean HIP-event time (ms) of k_cost, the linearisation and one mat-vec (row-major or column-sorted forms, incl. their finish kernels)."""
        rot = np.ascontiguousarray(rot_aa, dtype=np.float64).reshape(self.n_cams, 3)
        out = np.zeros(4)
        self._check(self._lib.gsfm_rot_time_kernels(self._h, _dp(rot), int(reps), _dp(out)), "time_kernels")
        return {"k_cost": out[0], "k_lin": out[1], "k_matvec": out[2]}

    def matvec_bytes(self):
        """(bytes one mat-vec streams as laid out, form: 0 general blocks, 1 Laplacian row-major, 2 Laplacian column-sorted)."""
        b, l, f = C.c_double(0), C.c_double(0), C.c_int32(0)
        self._check(self._lib.gsfm_rot_matvec_bytes(self._h, C.byref(b), C.byref(l), C.byref(f)), "matvec_bytes")
        return b.value, int(f.value)

    def linearize_bytes(self):
        b, l, f = C.c_double(0), C.c_double(0), C.c_int32(0)
        self._check(self._lib.gsfm_rot_matvec_bytes(self._h, C.byref(b), C.byref(l), C.byref(f)), "matvec_bytes")
        return l.value

    def loss_eval(self, s, with_fast_rho1=False):
        """(rho, rho', rho'')(s) and the cost-only rho(s) of the current native loss, evaluated by the device routines; with_fast_rho1 adds
        rho'(s) as the fast path of the linearisation K2 computes it (NaN for loss programs that have no fast path)."""
        s = np.ascontiguousarray(s, dtype=np.float64).ravel()
        rho3, val = np.empty((s.size, 3)), np.empty(s.size)
        fast = np.empty(s.size) if with_fast_rho1 else None
        self._check(self._lib.gsfm_rot_loss_eval(self._h, _dp(s), s.size, _dp(rho3), _dp(val), _dp(fast)), "loss_eval")
        return (rho3, val, fast) if with_fast_rho1 else (rho3, val)

    def edge_order(self):
        """order[u] = index (in the arrays given at creation) of the edge at position u of the device-side per-edge planes."""
        n = self._lib.gsfm_rot_edge_order(self._h, None, 0)
        out = np.empty(max(n, 0), dtype=np.uint32)
        if n > 0:
            self._lib.gsfm_rot_edge_order(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out

    def time_sweep_variants(self, rot_aa, reps=10):
        """Mean HIP-event time (ms) of the K1 variants (trial cost; s + rho triple stored; s only; rho' only) and of the sigma-consensus
        forms of K1 / K2 against their plain forms (zeros unless the problem is an ANGLE_AXIS one carrying scalar weights)."""
        rot = np.ascontiguousarray(rot_aa, dtype=np.float64).reshape(self.n_cams, 3)
        out = np.zeros(8)
        self._check(self._lib.gsfm_rot_time_sweep_variants(self._h, _dp(rot), int(reps), _dp(out)), "time_sweep_variants")
        return {"trial_cost": out[0], "full_reweight": out[1], "s_only": out[2], "rho1_only": out[3],
                "k1_sigma_fused": out[4], "k1_sigma_plain": out[5], "k2_sigma_fused": out[6], "k2_sigma_plain": out[7]}

    def sweep_bytes(self):
        a, b = C.c_double(0), C.c_double(0)
        self._check(self._lib.gsfm_rot_sweep_bytes(self._h, C.byref(a), C.byref(b)), "sweep_bytes")
        return a.value, b.value


def magsac_table(nu):
    lib = _abi.load_library()
    n = lib.gsfm_magsac_table(int(nu), None, 0)
    out = np.empty(n)
    lib.gsfm_magsac_table(int(nu), _dp(out), n)
    return out


def magsac_constants(nu):
    lib = _abi.load_library()
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    lib.gsfm_magsac_constants(int(nu), C.byref(a), C.byref(b), C.byref(c))
    return a.value, b.value, c.value


def edge_sq_norms(n_cams, edge_i, edge_j, rel_aa, rot_aa, cov6=None, max_sq_norm=None):
    """gsfm_rot_edge_sq_norms: per-edge squared (whitened) loop residual on the device, and the keep mask of the orientation filter."""
    lib = _abi.load_library()
    ei, ej, rel, c6, _, n_edges = _prep_edges(n_cams, edge_i, edge_j, rel_aa, cov6, None)
    rot = np.ascontiguousarray(rot_aa, dtype=np.float64).reshape(int(n_cams), 3)
    s = np.empty(n_edges)
    keep = np.empty(n_edges, dtype=np.uint8) if max_sq_norm is not None else None
    kept, ms = C.c_uint64(0), C.c_double(0)
    st = lib.gsfm_rot_edge_sq_norms(int(n_cams), int(n_edges), _u32p(ei), _u32p(ej), _dp(rel), _dp(c6), _dp(rot),
                                    float(max_sq_norm if max_sq_norm is not None else -1.0), _dp(s),
                                    keep.ctypes.data_as(C.POINTER(C.c_uint8)) if keep is not None else None, C.byref(kept), C.byref(ms))
    if st != 0:
        raise SolverError("gsfm_rot_edge_sq_norms failed with status %d: %s" % (st, lib.gsfm_last_error().decode("utf-8", "replace")))
    return {"s": s, "keep": None if keep is None else keep.astype(bool), "n_kept": int(kept.value), "kernel_ms": ms.value}
