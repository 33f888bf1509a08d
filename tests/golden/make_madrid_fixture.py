#!/usr/bin/env python3
"""Generates, from the reference's datasets/Madrid_Metropolis (data files, runs only here):
  tests/golden/madrid_graph.npz      the whole rotation graph (C1): view ids, edge list and the
                                     angle-axis relative rotations under Theia's EGs convention
                                     (io/read_1dsfm.cc:299-372), parsed with numpy + scipy;
  tests/golden/1dsfm_sample/         the first 120 EGs.txt rows restricted to cc.txt ids, + cc.txt:
                                     a small real input for the text reader.
covariance_rot.txt / tracks.txt / coords.txt are missing from the reference checkout (large blobs)."""
import os

import numpy as np
from scipy.spatial.transform import Rotation as R

SRC = "/root/reference/datasets/Madrid_Metropolis"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    cc = np.loadtxt(os.path.join(SRC, "cc.txt"), dtype=np.int64)
    eg = np.loadtxt(os.path.join(SRC, "EGs.txt"))
    a, b = eg[:, 0].astype(np.int64), eg[:, 1].astype(np.int64)
    keep = np.isin(a, cc) & np.isin(b, cc)
    Rm = eg[:, 2:11].reshape(-1, 3, 3)
    S = np.diag([1.0, -1.0, -1.0])
    Rp = S @ np.transpose(Rm, (0, 2, 1)) @ S                      # R' = S R^T S
    # ceres::RotationMatrixToAngleAxis without re-orthonormalisation: quaternion route; scipy normalises the
    # matrix first, which differs at the 1e-6 level of the file's precision -> use the same formula as Theia
    q = np.empty((Rp.shape[0], 4))
    tr = Rp[:, 0, 0] + Rp[:, 1, 1] + Rp[:, 2, 2]
    assert (tr >= 0).mean() > 0.5
    aa = np.empty((Rp.shape[0], 3))
    for k in range(Rp.shape[0]):
        M = Rp[k]
        t = M[0, 0] + M[1, 1] + M[2, 2]
        if t >= 0:
            s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
            v = np.array([(M[2, 1] - M[1, 2]) * s, (M[0, 2] - M[2, 0]) * s, (M[1, 0] - M[0, 1]) * s])
        else:
            i = 0
            if M[1, 1] > M[0, 0]: i = 1
            if M[2, 2] > M[i, i]: i = 2
            j, l = (i + 1) % 3, (i + 2) % 3
            s = np.sqrt(M[i, i] - M[j, j] - M[l, l] + 1.0)
            v = np.zeros(3); v[i] = 0.5 * s; s = 0.5 / s
            w = (M[l, j] - M[j, l]) * s; v[j] = (M[j, i] + M[i, j]) * s; v[l] = (M[l, i] + M[i, l]) * s
        n = np.linalg.norm(v)
        th = 2.0 * (np.arctan2(-n, -w) if w < 0 else np.arctan2(n, w))
        aa[k] = v * (th / n) if n > 0 else 2 * v
    # sanity against scipy on the orthonormalised matrix (file has ~6 significant digits)
    assert np.abs(R.from_matrix(Rp[:50]).as_rotvec() - aa[:50]).max() < 1e-4
    ids = np.unique(np.concatenate([a[keep], b[keep]]))
    np.savez_compressed(os.path.join(HERE, "madrid_graph.npz"), view_ids=ids.astype(np.uint32), edge_a=a[keep].astype(np.uint32),
                        edge_b=b[keep].astype(np.uint32), rel_aa=aa[keep])
    out = os.path.join(HERE, "1dsfm_sample")
    os.makedirs(out, exist_ok=True)
    lines = open(os.path.join(SRC, "EGs.txt")).read().splitlines()
    with open(os.path.join(out, "EGs.txt"), "w") as f:
        f.write("\n".join(lines[:120]) + "\n")
    with open(os.path.join(out, "cc.txt"), "w") as f:
        f.write(open(os.path.join(SRC, "cc.txt")).read())
    print("edges kept", int(keep.sum()), "of", len(a), "views", len(ids))


if __name__ == "__main__":
    main()
