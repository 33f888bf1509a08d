#!/usr/bin/env python3
"""Builds the C++ host layer in-tree:
  globalsfmpy_amd/libgsfm_estimator.so  theia::GSfMNonlinearRotationEstimator + view-graph helpers
  globalsfmpy_amd/_GlobalSfMpy<ext>     the pybind11 module, imported as `GlobalSfMpy` through the GlobalSfMpy.py shim beside it
Both link libgsfm_rot.so (the HIP C-ABI library) through an $ORIGIN rpath."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)


def newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def main():
    force = "--force" in sys.argv
    import pybind11
    cxx = os.environ.get("CXX", "g++")
    inc = [os.path.join(ROOT, "include", "gsfm", f) for f in os.listdir(os.path.join(ROOT, "include", "gsfm"))] + [os.path.join(ROOT, "include", "gsfm_rot.h")]
    common = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-fvisibility=default"]
    link = ["-L" + PKG, "-lgsfm_rot", "-Wl,-rpath,$ORIGIN"]
    est = os.path.join(PKG, "libgsfm_estimator.so")
    est_src = [os.path.join(HERE, "rotation_estimator.cpp"), os.path.join(HERE, "view_graph.cpp"), os.path.join(HERE, "dataset_1dsfm.cpp"), os.path.join(HERE, "evaluation.cpp")]
    if force or newer(est, est_src + inc + [os.path.join(PKG, "libgsfm_rot.so")]):
        subprocess.check_call(common + ["-o", est] + est_src + link)
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    mod = os.path.join(PKG, "_GlobalSfMpy" + ext)
    stale = os.path.join(PKG, "GlobalSfMpy" + ext)  # pre-shim builds: an extension would shadow GlobalSfMpy.py
    if os.path.exists(stale):
        os.remove(stale)
    mod_src = [os.path.join(HERE, "module.cpp")]
    if force or newer(mod, mod_src + inc + [est]):
        subprocess.check_call(common + ["-fvisibility=hidden", "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"],
                                        "-o", mod] + mod_src + ["-L" + PKG, "-lgsfm_estimator", "-lgsfm_rot", "-Wl,-rpath,$ORIGIN"])
    print("host layer built:", os.path.basename(est), os.path.basename(mod))


if __name__ == "__main__":
    main()
