"""ctypes mirror of include/gsfm_rot.h (the C-ABI drop-in boundary).

The shared library is the product: if libgsfm_rot.so is missing or does not load,
importing the solver fails loudly -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSFM_ROT_LIB") or os.path.join(_HERE, "libgsfm_rot.so")  # override: A/B builds of the kernels (tools/)

# gsfm_status
OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_EMPTY, ERR_COMM, ERR_UNSUPPORTED = range(7)

# gsfm_rot_error_type == RotationErrorType (reference include/pairwise_rotation_error_quat.hpp:50-61)
QUATERNION_NORM = 0
ROTATION_MAT_FNORM = 1
QUATERNION_COSINE = 2
ANGLE_AXIS_COVARIANCE = 3
ANGLE_AXIS = 4
ANGLE_AXIS_INLIERS = 5
ANGLE_AXIS_COV_INLIERS = 6
ANGLE_AXIS_COVTRACE = 7
ANGLE_AXIS_COVNORM = 8

# gsfm_loss_kind
LOSS_TRIVIAL, LOSS_HUBER, LOSS_SOFT_L1, LOSS_CAUCHY, LOSS_ARCTAN, LOSS_TOLERANT, LOSS_TUKEY, \
    LOSS_LONE_HALF, LOSS_LTWO, LOSS_GEMAN_MCCLURE, LOSS_MAGSAC = range(11)
LOSS_OP_SCALE, LOSS_OP_PUSH_ARG, LOSS_OP_COMPOSE = 32, 33, 34
LOSS_MAX_NODES = 16

TERMINATION_NAMES = ["FUNCTION_TOLERANCE", "GRADIENT_TOLERANCE", "PARAMETER_TOLERANCE", "NO_CONVERGENCE", "FAILURE"]


class LossNode(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("p", C.c_double * 3)]


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("num_threads", C.c_int32),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double), ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double), ("jacobi_scaling", C.c_int32),
        ("max_cg_iterations", C.c_int32), ("cg_relative_tolerance", C.c_double),
        ("cg_check_interval", C.c_int32), ("verbose", C.c_int32),
        ("pcg_single_reduction", C.c_int32), ("cg_stall_iterations", C.c_int32),
        ("dense_cholesky_max_cams", C.c_int32), ("pcg_hip_graph", C.c_int32),
        ("pcg_forcing", C.c_int32), ("dense_cholesky_auto_cams", C.c_int32), ("pcg_forcing_tolerance", C.c_double),
        ("lm_device_control", C.c_int32), ("component_rest", C.c_int32),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32), ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("num_residual_sweeps", C.c_int32), ("num_linearizations", C.c_int32),
        ("num_cg_iterations", C.c_int32), ("iters_to_1e6", C.c_int32),
        ("nonfinite", C.c_int32), ("outer_iterations", C.c_int32),
        ("num_edges_used", C.c_uint64),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("final_gradient_max_norm", C.c_double), ("final_radius", C.c_double),
        ("last_weight_change", C.c_double),
        ("t_total_ms", C.c_double), ("t_linearize_ms", C.c_double),
        ("t_sweep_ms", C.c_double), ("t_cg_ms", C.c_double),
        ("num_dense_solves", C.c_int32), ("num_graph_launches", C.c_int32),
        ("num_collectives", C.c_int32), ("num_pcg_collectives", C.c_int32),
        ("num_pcg_launched", C.c_int32), ("num_forcing_refinements", C.c_int32),
        ("num_inexact_steps", C.c_int32), ("num_pcg_capped_steps", C.c_int32),
        ("worst_accepted_cg_residual", C.c_double), ("num_forcing_restarts", C.c_int32), ("reserved_", C.c_int32),
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["termination_name"] = TERMINATION_NAMES[self.termination] if 0 <= self.termination < 5 else "?"
        return d


ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
LOSS_CALLBACK_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_double, C.POINTER(C.c_double))


SHARD_CAPTURABLE = 1
SHARD_DISCONNECTED = 2


class Shard(C.Structure):
    _fields_ = [
        ("rank", C.c_int32), ("world_size", C.c_int32),
        ("slice_width", C.c_uint32), ("flags", C.c_uint32),
        ("ctx", C.c_void_p), ("all_gather", ALL_GATHER_FN), ("all_reduce_sum", ALL_REDUCE_FN),
    ]


def make_program(nodes):
    """nodes: list of (kind, p0, p1, p2) -> (LossNode array, n)."""
    n = len(nodes)
    if n > LOSS_MAX_NODES:
        raise ValueError("loss program too long (%d > %d nodes)" % (n, LOSS_MAX_NODES))
    arr = (LossNode * max(n, 1))()
    for k, nd in enumerate(nodes):
        arr[k].kind = int(nd[0])
        arr[k].reserved = 0
        ps = list(nd[1:]) + [0.0] * (3 - len(nd[1:]))
        for c in range(3):
            arr[k].p[c] = float(ps[c])
    return arr, n


_DP = C.POINTER(C.c_double)
_U32P = C.POINTER(C.c_uint32)


def declare_solver_signatures(lib, prefix, problem_ptr=C.c_void_p):
    """Shared by the product library (prefix 'gsfm_rot_') and the oracle (prefix 'orc_'):
    the oracle deliberately mirrors the product's argument lists."""
    f = getattr(lib, prefix + "set_loss")
    f.argtypes = [problem_ptr, C.POINTER(LossNode), C.c_int32]; f.restype = C.c_int
    f = getattr(lib, prefix + "set_loss_callback")
    f.argtypes = [problem_ptr, LOSS_CALLBACK_FN, C.c_void_p]; f.restype = C.c_int
    f = getattr(lib, prefix + "set_edge_weights")
    f.argtypes = [problem_ptr, _DP]; f.restype = C.c_int
    f = getattr(lib, prefix + "residuals")
    f.argtypes = [problem_ptr, _DP, _DP, _DP, _DP, _DP]; f.restype = C.c_int
    f = getattr(lib, prefix + "linearize")
    f.argtypes = [problem_ptr, _DP, _DP, _DP, _DP]; f.restype = C.c_int
    f = getattr(lib, prefix + "normal_matvec")
    f.argtypes = [problem_ptr, _DP, _DP]; f.restype = C.c_int
    f = getattr(lib, prefix + "solve")
    f.argtypes = [problem_ptr, _DP, C.POINTER(Options), C.POINTER(Summary)]; f.restype = C.c_int
    f = getattr(lib, prefix + "solve_sigma_consensus")
    f.argtypes = [problem_ptr, _DP, C.c_int32, C.c_double, C.POINTER(Options), C.POINTER(Summary)]; f.restype = C.c_int
    f = getattr(lib, prefix + "get_trace")
    f.argtypes = [problem_ptr, _DP, C.c_int32]; f.restype = C.c_int32
    f = getattr(lib, prefix + "problem_destroy")
    f.argtypes = [problem_ptr]; f.restype = None


_lib = None


def preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (soname libamdhip64.so.7) next to libtorch.  If the system
    runtime under /opt/rocm is mapped first (by libgsfm_rot.so) and torch is imported afterwards, torch maps its own copy
    as well, and the runtime that initialises second finds no device (measured on the MI355X box: `hipGetDeviceCount`
    fails in libgsfm_rot.so after `torch.cuda.is_available()`).  The other order is fine because the loader matches
    libgsfm_rot.so's NEEDED libamdhip64.so.7 against the soname of torch's copy.  So: when torch is installed and no HIP
    runtime is mapped yet, map torch's copy first.  torch itself is not imported.  Processes without torch (the C++
    consumers, scripts/sfm_pipeline.py) just get the system runtime."""
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return None
    except OSError:
        pass
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(cand):
        return None
    C.CDLL(cand, mode=C.RTLD_GLOBAL)
    return cand


def load_library():
    """Load libgsfm_rot.so (built by __graft_entry__.build()). Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    preload_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "globalsfmpy_amd: %s not found. The HIP extension is the product and has no CPU fallback; "
            "build it with `python -c 'import __graft_entry__ as g; g.build()'`." % LIB_PATH)
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    if lib.gsfm_rot_abi_version() != 4:
        raise ImportError("globalsfmpy_amd: ABI version mismatch in %s" % LIB_PATH)
    lib.gsfm_last_error.restype = C.c_char_p
    lib.gsfm_rot_options_default.argtypes = [C.POINTER(Options)]
    lib.gsfm_rot_problem_create.argtypes = [C.c_uint32, C.c_uint64, _U32P, _U32P, _DP, C.c_int32, _DP, _DP,
                                            C.POINTER(Shard), C.POINTER(C.c_void_p)]
    lib.gsfm_rot_problem_create.restype = C.c_int
    declare_solver_signatures(lib, "gsfm_rot_")
    lib.gsfm_rot_set_stream.argtypes = [C.c_void_p, C.c_void_p]; lib.gsfm_rot_set_stream.restype = C.c_int
    lib.gsfm_rot_solve_resident.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Options), C.POINTER(Summary)]; lib.gsfm_rot_solve_resident.restype = C.c_int
    lib.gsfm_rot_residual_dim.argtypes = [C.c_int32]; lib.gsfm_rot_residual_dim.restype = C.c_int32
    lib.gsfm_rot_time_sweep.argtypes = [C.c_void_p, _DP, C.c_int32, _DP]; lib.gsfm_rot_time_sweep.restype = C.c_int
    lib.gsfm_rot_time_kernels.argtypes = [C.c_void_p, _DP, C.c_int32, _DP]; lib.gsfm_rot_time_kernels.restype = C.c_int
    lib.gsfm_rot_time_sweep_variants.argtypes = [C.c_void_p, _DP, C.c_int32, _DP]; lib.gsfm_rot_time_sweep_variants.restype = C.c_int
    lib.gsfm_rot_matvec_bytes.argtypes = [C.c_void_p, _DP, _DP, C.POINTER(C.c_int32)]; lib.gsfm_rot_matvec_bytes.restype = C.c_int
    lib.gsfm_rot_sweep_bytes.argtypes = [C.c_void_p, _DP, _DP]; lib.gsfm_rot_sweep_bytes.restype = C.c_int
    lib.gsfm_rot_loss_eval.argtypes = [C.c_void_p, _DP, C.c_uint64, _DP, _DP, _DP]; lib.gsfm_rot_loss_eval.restype = C.c_int
    lib.gsfm_rot_edge_order.argtypes = [C.c_void_p, _U32P, C.c_uint64]; lib.gsfm_rot_edge_order.restype = C.c_int64
    lib.gsfm_rot_locality_order.argtypes = [C.c_uint32, C.c_uint64, _U32P, _U32P, _U32P]; lib.gsfm_rot_locality_order.restype = C.c_int32
    lib.gsfm_rot_edge_sq_norms.argtypes = [C.c_uint32, C.c_uint64, _U32P, _U32P, _DP, _DP, _DP, C.c_double, _DP, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), _DP]
    lib.gsfm_rot_edge_sq_norms.restype = C.c_int
    lib.gsfm_rot_count_components.argtypes = [C.c_uint32, C.c_uint64, _U32P, _U32P]; lib.gsfm_rot_count_components.restype = C.c_int64
    lib.gsfm_cov_estimate.argtypes = [C.c_uint64, C.POINTER(C.c_uint64), _DP, _DP, _DP, _DP, C.c_int32, _DP, _DP, _DP,
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int32), _DP]
    lib.gsfm_cov_estimate.restype = C.c_int
    lib.gsfm_magsac_table.argtypes = [C.c_int32, _DP, C.c_int32]; lib.gsfm_magsac_table.restype = C.c_int32
    lib.gsfm_magsac_constants.argtypes = [C.c_int32, _DP, _DP, _DP]; lib.gsfm_magsac_constants.restype = C.c_int
    _lib = lib
    return lib
