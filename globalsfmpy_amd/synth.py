"""Seeded synthetic pose graphs and the evaluation metric.

numpy only (PCG64 generator: platform independent); no device code.  Conventions are the
reference's: world->camera angle-axis per camera, R_ij = R_j R_i^T on edge (i, j), i < j.
"""
import numpy as np


# ---- small vectorised SO(3) toolkit (quaternions as [..., (x, y, z, w)]) ----
def aa_to_quat(aa):
    aa = np.asarray(aa, dtype=np.float64)
    th = np.linalg.norm(aa, axis=-1, keepdims=True)
    small = th < 1e-12
    k = np.where(small, 0.5, np.sin(0.5 * th) / np.where(small, 1.0, th))
    return np.concatenate([aa * k, np.where(small, 1.0, np.cos(0.5 * th))], axis=-1)


def quat_to_aa(q):
    q = np.asarray(q, dtype=np.float64)
    v, w = q[..., :3], q[..., 3:4]
    s = np.linalg.norm(v, axis=-1, keepdims=True)
    sign = np.where(w < 0, -1.0, 1.0)
    two_theta = 2.0 * np.arctan2(sign * s, sign * w)
    k = np.where(s > 0, two_theta / np.where(s > 0, s, 1.0), 2.0)
    return v * k


def quat_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def quat_conj(a):
    return a * np.array([-1.0, -1.0, -1.0, 1.0])


def quat_to_matrix(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def random_unit_quat(rng, n):
    q = rng.standard_normal((n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def make_edges(rng, n_cams, n_edges):
    """Spanning chain (k-1, k) plus uniformly random distinct pairs i < j."""
    if n_edges < n_cams - 1:
        raise ValueError("need at least n_cams - 1 edges")
    max_edges = n_cams * (n_cams - 1) // 2
    if n_edges > max_edges:
        raise ValueError("more edges than distinct pairs")
    chain_i = np.arange(0, n_cams - 1, dtype=np.int64)
    chain_j = chain_i + 1
    key_chain = chain_i * n_cams + chain_j
    need = n_edges - (n_cams - 1)
    keys = np.empty(0, dtype=np.int64)
    while keys.size < need:
        m = int((need - keys.size) * 1.2) + 16
        a = rng.integers(0, n_cams, m)
        b = rng.integers(0, n_cams, m)
        ok = (a != b) & (np.abs(a - b) != 1)
        lo, hi = np.minimum(a, b)[ok], np.maximum(a, b)[ok]
        keys = np.unique(np.concatenate([keys, lo * n_cams + hi]))
        if keys.size > need:
            keys = rng.permutation(keys)[:need]
    keys = np.concatenate([key_chain, np.sort(keys)])
    return (keys // n_cams).astype(np.uint32), (keys % n_cams).astype(np.uint32)


def make_local_edges(rng, n_cams, n_edges, window, shuffle=True):
    """A spatially coherent view graph with arbitrary ids: in a hidden ordering every camera is linked to its successor
    and to random cameras at most window/2 places away; the ids are then shuffled (shuffle=True), which is how real
    datasets arrive.  Returns (edge_i, edge_j) with edge_i < edge_j."""
    chain_i = np.arange(0, n_cams - 1, dtype=np.int64)
    need = n_edges - (n_cams - 1)
    # pairs (a, a + d) with 2 <= d <= window // 2 (d >= 2: the chain holds d = 1) and a + d < n_cams
    reach = max(2, window // 2)
    available = sum(max(0, n_cams - d) for d in range(2, reach + 1))
    if need > available:
        raise ValueError("a window of %d admits only %d edges beside the chain on %d cameras, %d asked for" % (window, available, n_cams, need))
    keys = np.empty(0, dtype=np.int64)
    while keys.size < need:
        m = int((need - keys.size) * 1.3) + 16
        a = rng.integers(0, n_cams, m)
        b = a + rng.integers(2, max(3, window // 2 + 1), m)
        ok = b < n_cams
        keys = np.unique(np.concatenate([keys, a[ok] * n_cams + b[ok]]))
        if keys.size > need:
            keys = rng.permutation(keys)[:need]
    keys = np.concatenate([chain_i * n_cams + chain_i + 1, np.sort(keys)])
    i, j = keys // n_cams, keys % n_cams
    if shuffle:
        sigma = rng.permutation(n_cams)
        i, j = sigma[i], sigma[j]
    return np.minimum(i, j).astype(np.uint32), np.maximum(i, j).astype(np.uint32)


def make_graph(n_cams, n_edges, seed, outlier_frac=0.0, scale=0.2, full_so3=False,
               sigma_deg=(0.2, 2.0), kappa=3e-4, init_noise_deg=2.0, noise=True, local_window=0):
    """Returns dict(n_cams, edge_i, edge_j, rel_aa, cov6, inlier_weight, gt_aa, init_aa, is_outlier).

    Inlier measurement R_ij = Exp(n) R_j R_i^T with n ~ N(0, Sigma_ij), Sigma_ij = A diag(s^2) A^T,
    s ~ logU[sigma_deg]; the covariance handed to the solver is kappa * Sigma_ij so that the whitened
    inlier s_e = 1e-8 n^T (kappa Sigma)^-1 n averages 1e-4 (MAGSAC sigma_max = 0.02 scale).
    Outliers (only among the non-chain edges) are uniform random rotations.  The initial guess is the
    ground truth perturbed by init_noise_deg (stand-in for the spanning-tree initialisation).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    if full_so3:
        q_gt = random_unit_quat(rng, n_cams)
        gt_aa = quat_to_aa(q_gt)
    else:
        gt_aa = scale * rng.uniform(-1.0, 1.0, (n_cams, 3))
    q_gt = aa_to_quat(gt_aa)
    ei, ej = make_local_edges(rng, n_cams, n_edges, local_window) if local_window else make_edges(rng, n_cams, n_edges)
    E = ei.shape[0]
    q_rel = quat_mul(q_gt[ej], quat_conj(q_gt[ei]))
    lo, hi = np.log(np.deg2rad(sigma_deg[0])), np.log(np.deg2rad(sigma_deg[1]))
    sig = np.exp(rng.uniform(lo, hi, (E, 3)))
    q_axes = random_unit_quat(rng, E)
    z = rng.standard_normal((E, 3)) if noise else None
    # the 3x3 intermediates (axes, covariance) are the bulk of the memory: built in blocks of edges, so that the 80M-edge graph
    # of an 8-rank weak-scaling run peaks at ~12 GB per rank instead of ~30 GB
    cov6 = np.empty((E, 6))
    n = np.empty((E, 3)) if noise else None
    for b0 in range(0, E, 1 << 22):
        sl = slice(b0, min(E, b0 + (1 << 22)))
        A = quat_to_matrix(q_axes[sl])
        Sf = kappa * np.einsum("eij,ej,ekj->eik", A, sig[sl] * sig[sl], A)
        cov6[sl] = np.stack([Sf[:, 0, 0], Sf[:, 1, 1], Sf[:, 2, 2], Sf[:, 0, 1], Sf[:, 0, 2], Sf[:, 1, 2]], axis=1)
        if noise:
            n[sl] = np.einsum("eij,ej->ei", A, sig[sl] * z[sl])
    del q_axes, z
    if noise:
        q_rel = quat_mul(aa_to_quat(n), q_rel)
    is_out = np.zeros(E, dtype=bool)
    if outlier_frac > 0:
        cand = np.arange(n_cams - 1, E)   # (with local_window the chain is not a prefix of the shuffled list: any edge may be hit)
        k = int(round(outlier_frac * E))
        pick = rng.choice(cand, size=min(k, cand.size), replace=False)
        is_out[pick] = True
        q_rel[pick] = random_unit_quat(rng, pick.size)
    init_q = quat_mul(aa_to_quat(np.deg2rad(init_noise_deg) * rng.standard_normal((n_cams, 3))), q_gt)
    inlier_weight = rng.integers(30, 400, E).astype(np.float64) / 100.0
    return {"n_cams": int(n_cams), "edge_i": ei, "edge_j": ej, "rel_aa": np.ascontiguousarray(quat_to_aa(q_rel)),
            "cov6": np.ascontiguousarray(cov6), "inlier_weight": inlier_weight, "gt_aa": gt_aa,
            "init_aa": np.ascontiguousarray(quat_to_aa(init_q)), "is_outlier": is_out}


def spanning_tree_init(g, seed=0, inlier_matches=(150, 1500), outlier_matches=(16, 150)):
    """The initialisation the reference pipeline feeds the solver (OrientationsFromMaximumSpanningTree, Theia
    orientations_from_maximum_spanning_tree.cc:62-181): a maximum spanning tree of the view graph weighted by the number of verified
    matches, rotations composed from the root along the tree (R_j = R_ij R_i going up in id, R_i = R_ij^T R_j going down).  Match
    counts are drawn so that inlier edges dominate the tree, as they do in real view graphs (an outlier pair has few verified matches).
    Returns (init_aa, num_matches).  SURVEY 8(d): "Init = chain composition or BFS-tree composition"."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import breadth_first_order, minimum_spanning_tree
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    n, ei, ej = g["n_cams"], g["edge_i"].astype(np.int64), g["edge_j"].astype(np.int64)
    E = ei.size
    m = rng.integers(inlier_matches[0], inlier_matches[1], E)
    out = g["is_outlier"]
    m[out] = rng.integers(outlier_matches[0], outlier_matches[1], int(out.sum()))
    tree = minimum_spanning_tree(sp.coo_matrix((-m.astype(np.float64), (ei, ej)), shape=(n, n)).tocsr())
    order, pred = breadth_first_order(tree + tree.T, 0, directed=False)
    order, pred = order.astype(np.int64), pred.astype(np.int64)   # (scipy returns int32: lo * n + hi below would overflow from 46k cameras on)
    keys = ei * n + ej
    srt = np.argsort(keys, kind="stable")
    q_rel = aa_to_quat(g["rel_aa"])
    q = np.tile(np.array([0.0, 0.0, 0.0, 1.0]), (n, 1))      # cameras outside the root's component keep the identity
    depth = np.zeros(n, dtype=np.int64)
    nodes = order[1:]
    for v in nodes:                                            # (BFS order: a parent's depth is final before its children are visited)
        depth[v] = depth[pred[v]] + 1
    for d in range(1, int(depth.max()) + 1 if nodes.size else 1):
        v = nodes[depth[nodes] == d]
        p = pred[v]
        lo, hi = np.minimum(p, v), np.maximum(p, v)
        e = srt[np.searchsorted(keys, lo * n + hi, sorter=srt)]
        up = p < v                                             # the parent is `first`: R_v = R_ij R_p; else R_v = R_ij^T R_p
        q[v] = np.where(up[:, None], quat_mul(q_rel[e], q[p]), quat_mul(quat_conj(q_rel[e]), q[p]))
    return np.ascontiguousarray(quat_to_aa(q)), m


def angular_distance(aa_a, aa_b):
    """Per-camera geodesic distance (rad) between two sets of rotations, no alignment."""
    q = quat_mul(aa_to_quat(aa_a), quat_conj(aa_to_quat(aa_b)))
    return 2.0 * np.arctan2(np.linalg.norm(q[..., :3], axis=-1), np.abs(q[..., 3]))


def align_rotations(aa_est, aa_ref):
    """Global gauge alignment est * R = ref (AlignRotations semantics, Theia align_rotations.cc:132-154):
    chordal L2 mean of R_est^T R_ref projected on SO(3). Returns the aligned angle-axis set."""
    Re = quat_to_matrix(aa_to_quat(aa_est))
    Rr = quat_to_matrix(aa_to_quat(aa_ref))
    M = np.einsum("nji,njk->ik", Re, Rr)
    U, _, Vt = np.linalg.svd(M)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = U @ D @ Vt
    Ra = Re @ R
    # matrix -> quaternion via the robust trace method, vectorised through scipy-free eigen trick
    q = matrix_to_quat(Ra)
    return quat_to_aa(q)


def matrix_to_quat(R):
    R = np.asarray(R)
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    q = np.empty(R.shape[:-2] + (4,))
    tr = m00 + m11 + m22
    c0 = tr >= 0
    c1 = (~c0) & (m00 >= m11) & (m00 >= m22)
    c2 = (~c0) & (~c1) & (m11 >= m22)
    c3 = (~c0) & (~c1) & (~c2)
    t = np.sqrt(np.maximum(np.where(c0, 1 + tr, np.where(c1, 1 + m00 - m11 - m22, np.where(c2, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22))), 1e-300))
    s = 0.5 / t
    q[..., 3] = np.where(c0, 0.5 * t, np.where(c1, (R[..., 2, 1] - R[..., 1, 2]) * s, np.where(c2, (R[..., 0, 2] - R[..., 2, 0]) * s, (R[..., 1, 0] - R[..., 0, 1]) * s)))
    q[..., 0] = np.where(c0, (R[..., 2, 1] - R[..., 1, 2]) * s, np.where(c1, 0.5 * t, np.where(c2, (R[..., 0, 1] + R[..., 1, 0]) * s, (R[..., 0, 2] + R[..., 2, 0]) * s)))
    q[..., 1] = np.where(c0, (R[..., 0, 2] - R[..., 2, 0]) * s, np.where(c1, (R[..., 0, 1] + R[..., 1, 0]) * s, np.where(c2, 0.5 * t, (R[..., 1, 2] + R[..., 2, 1]) * s)))
    q[..., 2] = np.where(c0, (R[..., 1, 0] - R[..., 0, 1]) * s, np.where(c1, (R[..., 0, 2] + R[..., 2, 0]) * s, np.where(c2, (R[..., 1, 2] + R[..., 2, 1]) * s, 0.5 * t)))
    return q
