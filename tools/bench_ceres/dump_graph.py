#!/usr/bin/env python3
"""Writes the synthetic graph of bench.py's CPU sample in the flat binary layout bench_ceres.cpp reads.
usage: dump_graph.py <cams> <edges> <out.bin> [seed] [outlier_frac]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from globalsfmpy_amd import synth  # noqa: E402

n, e, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 2024
frac = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3
g = synth.make_graph(n, e, seed, outlier_frac=frac)
with open(out, "wb") as f:
    np.array([n, e], dtype=np.uint64).tofile(f)
    np.ascontiguousarray(g["edge_i"], dtype=np.uint32).tofile(f)
    np.ascontiguousarray(g["edge_j"], dtype=np.uint32).tofile(f)
    np.ascontiguousarray(g["rel_aa"], dtype=np.float64).tofile(f)
    np.ascontiguousarray(g["cov6"], dtype=np.float64).tofile(f)
    np.ascontiguousarray(g["init_aa"], dtype=np.float64).tofile(f)
print("wrote %s: %d cameras, %d edges" % (out, n, e))
