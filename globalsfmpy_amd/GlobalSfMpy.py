"""`import GlobalSfMpy as sfm` -- the name scripts/sfm_pipeline.py:5-6 imports.  The bindings live in the compiled
`_GlobalSfMpy` extension beside this file; this shim only makes sure one HIP runtime serves the whole process before the
extension (and with it libgsfm_rot.so) is mapped -- see globalsfmpy_amd/_abi.py:preload_hip_runtime."""
import ctypes as _ctypes
import importlib.util as _ilu
import os as _os


def _preload_hip_runtime():
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return
    except OSError:
        pass
    try:
        spec = _ilu.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is not None and spec.origin:
        cand = _os.path.join(_os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if _os.path.exists(cand):
            _ctypes.CDLL(cand, mode=_ctypes.RTLD_GLOBAL)


_preload_hip_runtime()

import _GlobalSfMpy as _native  # noqa: E402
from _GlobalSfMpy import *  # noqa: E402,F401,F403

for _k in dir(_native):  # `import *` skips nothing we need, but keep dunder-free private helpers reachable too
    if not _k.startswith("__") and _k not in globals():
        globals()[_k] = getattr(_native, _k)
__doc__ = (_native.__doc__ or "") + "\n" + __doc__
