#!/usr/bin/env python3
"""Round-5 review item 5: candidate preconditioners for the first LM steps from the maximum-spanning-tree start, measured OFFLINE (CPU, scipy)
before any kernel is written.  The CPU oracle exports the linear systems of its first LM iterations exactly as its solver saw them
(orc_capture_steps: corrected, column-scaled Jacobian blocks, damping, right-hand side); this script rebuilds A = J^T J + D^2 as a sparse
matrix and counts PCG iterations to a relative (preconditioned) residual of 1e-12 under
  jacobi      3x3 block-Jacobi (what the device runs)
  tree        block diagonal + the off-diagonal blocks of the maximum-weight spanning tree, factorised exactly (the reviewer's round-5 probe)
  ml-add      ADDITIVE multilevel aggregation: z = M^-1 r + P1 (M1^-1 + P2 (M2^-1 + ... + Pk Ak^-1 Pk^T ...) P2^T) P1^T r, aggregates by heavy-edge
              matching on the block strengths (x8 per level), tentative prolongation fitted to three near-null-space candidates (smoothed-aggregation
              style: local QR of relaxed random vectors); NO extra fine mat-vec per application
  ml-add-sm   the same with the prolongations smoothed, P <- (I - w D^-1 A) P
  ml-V11      MULTIPLICATIVE V(1,1) cycle on the same hierarchy (two extra fine-level mat-vecs per application)
  defl        step 2 only: block-Jacobi + an additive coarse correction on the k lowest Ritz vectors harvested from step 1's PCG (recycling)
The bar: >= 6 x fewer iterations than block-Jacobi at an application cost that can be <= 2 mat-vecs.
usage: python tools/r06_far_start_precond.py [cams] [edges] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from scipy.sparse.csgraph import minimum_spanning_tree
from globalsfmpy_amd import _abi, synth
from globalsfmpy_amd.loss_functions import MAGSACWeightBasedLoss
from oracle import pyoracle

n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
n_edges = int(sys.argv[2]) if len(sys.argv) > 2 else 2000000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2023
TOL = 1e-12
FULL = n_cams <= 5000   # (the smoothed prolongations of the shallow hierarchies only at the small size: their sparse triple products take minutes here)


def build_system(st, ei, ej, n):
    """A = J^T J + diag(D)^2 as BSR(3x3), b = rhs (the oracle's own solve is st['y'])."""
    Ji, Jj = st["Ji"], st["Jj"]                       # E x R x 3
    Hii = np.einsum("erk,erl->ekl", Ji, Ji); Hjj = np.einsum("erk,erl->ekl", Jj, Jj); Hij = np.einsum("erk,erl->ekl", Ji, Jj)
    rows = np.concatenate([ei, ej, ei, ej]); cols = np.concatenate([ei, ej, ej, ei])
    blocks = np.concatenate([Hii, Hjj, Hij, np.transpose(Hij, (0, 2, 1))])
    order = np.lexsort((cols, rows))
    rows, cols, blocks = rows[order], cols[order], blocks[order]
    key = rows.astype(np.int64) * n + cols
    first = np.concatenate([[True], key[1:] != key[:-1]])
    idx = np.cumsum(first) - 1
    nb = int(idx[-1]) + 1
    data = np.zeros((nb, 3, 3)); np.add.at(data, idx, blocks)
    r_u, c_u = rows[first], cols[first]
    indptr = np.zeros(n + 1, dtype=np.int64); np.add.at(indptr, r_u + 1, 1); indptr = np.cumsum(indptr)
    A = sp.bsr_matrix((data, c_u, indptr), shape=(3 * n, 3 * n)).tocsr()
    A = A + sp.diags(st["D"] ** 2)
    return A.tocsr(), st["rhs"].copy()


def block_diag_inv(A, n):
    Ab = A.tobsr(blocksize=(3, 3))
    D = np.zeros((n, 3, 3))
    for r in range(n):
        lo, hi = Ab.indptr[r], Ab.indptr[r + 1]
        k = lo + np.searchsorted(Ab.indices[lo:hi], r)
        D[r] = Ab.data[k]
    return D, np.linalg.inv(D)


def apply_blocks(Minv, r):
    return np.einsum("nij,nj->ni", Minv, r.reshape(-1, 3)).ravel()


def pcg(A, b, prec, tol=TOL, maxit=20000, harvest=0):
    """Preconditioned CG from x = 0; stops on sqrt(r.z / r0.z0) <= tol (the device's rule).  harvest > 0: also returns that many lowest Ritz vectors."""
    x = np.zeros_like(b); r = b.copy(); z = prec(r); p = z.copy()
    rz = r @ z; rz0 = rz
    alphas, betas, Z = [], [], []
    it = 0
    while it < maxit:
        if harvest and it < 400: Z.append(z / np.sqrt(rz))
        Ap = A @ p
        a = rz / (p @ Ap)
        x += a * p; r -= a * Ap
        z = prec(r); rz_new = r @ z
        it += 1
        alphas.append(a)
        if np.sqrt(abs(rz_new) / rz0) <= tol: break
        bt = rz_new / rz; betas.append(bt)
        p = z + bt * p; rz = rz_new
    if not harvest: return x, it
    m = len(Z)
    T = np.zeros((m, m))
    for k in range(m):
        T[k, k] = 1.0 / alphas[k] + (betas[k - 1] / alphas[k - 1] if k else 0.0)
        if k + 1 < m: T[k, k + 1] = T[k + 1, k] = -np.sqrt(betas[k]) / alphas[k]
    w, V = np.linalg.eigh(T)
    W = np.stack(Z, axis=1) @ V[:, :harvest]          # Ritz vectors of M^-1 A (lowest), in the z basis
    return x, it, W


def strength_graph(A, n):
    Ab = A.tobsr(blocksize=(3, 3))
    rows = np.repeat(np.arange(n), np.diff(Ab.indptr)); cols = Ab.indices
    w = np.sqrt((Ab.data ** 2).sum(axis=(1, 2)))
    off = rows != cols
    return sp.coo_matrix((w[off], (rows[off], cols[off])), shape=(n, n)).tocsr()


def heavy_edge_aggregate(S, rounds=3):
    """Aggregates by `rounds` passes of heavy-edge matching (x2 per pass): returns agg id per node."""
    n = S.shape[0]
    agg = np.arange(n)
    cur = S.copy()
    for _ in range(rounds):
        m = cur.shape[0]
        cur = cur.tocsr(); cur.setdiag(0); cur.eliminate_zeros()
        match = -np.ones(m, dtype=np.int64)
        # visit nodes by descending strongest edge; match with the strongest unmatched neighbour
        best = np.zeros(m)
        for r in range(m):
            lo, hi = cur.indptr[r], cur.indptr[r + 1]
            if hi > lo: best[r] = cur.data[lo:hi].max()
        for r in np.argsort(-best):
            if match[r] >= 0: continue
            lo, hi = cur.indptr[r], cur.indptr[r + 1]
            nb, wv = cur.indices[lo:hi], cur.data[lo:hi]
            free = match[nb] < 0
            free &= nb != r
            if free.any():
                c = nb[free][np.argmax(wv[free])]
                match[r] = c; match[c] = r
            else:
                match[r] = r
        rep = np.minimum(np.arange(m), match)
        uniq, new = np.unique(rep, return_inverse=True)
        agg = new[agg]
        Pm = sp.coo_matrix((np.ones(m), (np.arange(m), new)), shape=(m, uniq.size)).tocsr()
        cur = (Pm.T @ cur @ Pm).tocsr()
    return agg


def tentative_P(agg, B):
    """Smoothed-aggregation tentative prolongation: per aggregate the local QR of the candidates B (3 n x 3) restricted to it."""
    n = agg.size; na = int(agg.max()) + 1
    order = np.argsort(agg, kind="stable"); bounds = np.searchsorted(agg[order], np.arange(na + 1))
    rows, cols, vals = [], [], []
    Bc = np.zeros((3 * na, B.shape[1]))
    for a in range(na):
        nodes = order[bounds[a]:bounds[a + 1]]
        idx = (3 * nodes[:, None] + np.arange(3)).ravel()
        Q, R = np.linalg.qr(B[idx])
        k = Q.shape[1]
        if k < 3:   # (an aggregate of a single camera has 3 rows: k = 3 always here; kept for safety)
            Q = np.pad(Q, ((0, 0), (0, 3 - k))); R = np.pad(R, ((0, 3 - k), (0, 0)))
        rows.append(np.repeat(idx, 3)); cols.append(np.tile(3 * a + np.arange(3), idx.size)); vals.append(Q.ravel())
        Bc[3 * a:3 * a + 3] = R
    P = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(3 * n, 3 * na)).tocsr()
    return P, Bc


def near_null_candidates(A, Minv, n, sweeps=60, seed=1):
    """Three candidates for the near-null space (the gauge rotations, damped): block-Jacobi relaxation of A x = 0 from random vectors."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((3 * n, 3))
    for _ in range(sweeps):
        R = A @ X
        for c in range(3): X[:, c] -= 0.7 * apply_blocks(Minv, R[:, c])
    X, _ = np.linalg.qr(X)
    return X


def build_hierarchy(A, n, levels, smooth):
    H = []
    Al, nl = A, n
    D, Minv = block_diag_inv(Al, nl)
    B = near_null_candidates(Al, Minv, nl)
    for _ in range(levels):
        S = strength_graph(Al, nl)
        agg = heavy_edge_aggregate(S, rounds=3)
        P, Bc = tentative_P(agg, B)
        if smooth:
            DinvA = sp.bsr_matrix((Minv, np.arange(nl), np.arange(nl + 1)), shape=(3 * nl, 3 * nl)).tocsr() @ Al
            P = (P - (2.0 / 3.0) * (DinvA @ P)).tocsr()
        Ac = (P.T @ Al @ P).tocsr()
        H.append({"A": Al, "Minv": Minv, "P": P, "n": nl})
        Al, nl, B = Ac, int(agg.max()) + 1, Bc
        D, Minv = block_diag_inv(Al, nl)
        if nl <= 64: break
    H.append({"A": Al, "Minv": Minv, "n": nl, "solve": spla.splu(Al.tocsc()).solve})
    return H


def prec_additive(H):
    def rec(l, r):
        if "solve" in H[l]: return H[l]["solve"](r)
        z = apply_blocks(H[l]["Minv"], r)
        return z + H[l]["P"] @ rec(l + 1, H[l]["P"].T @ r)
    return lambda r: rec(0, r)


def prec_vcycle(H):
    def rec(l, r):
        if "solve" in H[l]: return H[l]["solve"](r)
        A, Minv, P = H[l]["A"], H[l]["Minv"], H[l]["P"]
        x = 0.7 * apply_blocks(Minv, r)
        x = x + P @ rec(l + 1, P.T @ (r - A @ x))
        return x + 0.7 * apply_blocks(Minv, r - A @ x)
    return lambda r: rec(0, r)


def prec_tree(A, n, Minv_unused):
    S = strength_graph(A, n)
    T = minimum_spanning_tree(sp.csr_matrix((-S.data, S.indices, S.indptr), shape=S.shape))
    T = ((T + T.T) != 0).astype(np.float64)
    mask = sp.kron(T + sp.eye(n), np.ones((3, 3))).tocsr()
    At = A.multiply(mask).tocsc()
    lu = spla.splu(At)
    return lu.solve


def main():
    t0 = time.time()
    g = synth.make_graph(n_cams, n_edges, seed, outlier_frac=0.3)
    init, _ = synth.spanning_tree_init(g, seed)
    e0 = np.rad2deg(synth.angular_distance(synth.align_rotations(init, g["gt_aa"]), g["gt_aa"]).mean())
    o = pyoracle.OracleProblem(g["n_cams"], g["edge_i"], g["edge_j"], g["rel_aa"], _abi.ANGLE_AXIS_COVARIANCE, cov6=g["cov6"])
    o.set_loss(MAGSACWeightBasedLoss(0.02))
    o.capture_steps(2)
    _, s = o.solve(init, max_num_iterations=2)
    print("# %d cameras / %d edges / 30 %% outliers, seed %d, maximum-spanning-tree start (%.1f deg mean error), MAGSAC(0.02) + covariance whitening; oracle: 2 LM steps in %.0f s"
          % (n_cams, n_edges, seed, e0, time.time() - t0), flush=True)
    ei, ej = g["edge_i"].astype(np.int64), g["edge_j"].astype(np.int64)
    W1 = None
    for k in (0, 1):
        st = o.captured_step(k)
        A, b = build_system(st, ei, ej, n_cams)
        D, Minv = block_diag_inv(A, n_cams)
        jac = lambda r: apply_blocks(Minv, r)
        t = time.time()
        if k == 0: x, it_j, W1 = pcg(A, b, jac, harvest=32)
        else: x, it_j = pcg(A, b, jac)
        err = np.linalg.norm(x - st["y"]) / np.linalg.norm(st["y"])
        print("step %d: oracle's own PCG %d iterations; here block-Jacobi %d (solution vs the oracle's %.1e), %.0f s" % (k + 1, st["cg"], it_j, err, time.time() - t), flush=True)
        res = {}
        if os.environ.get("FAR_ONLY_DEFL"):
            if k == 1 and W1 is not None:
                for kk in (8, 32):
                    Wk = W1[:, :kk]
                    G = np.linalg.inv(Wk.T @ (A @ Wk))
                    def defl(r, Wk=Wk, G=G): return jac(r) + Wk @ (G @ (Wk.T @ r))
                    t = time.time(); _, it = pcg(A, b, defl)
                    print("   defl k=%-2d (recycled from step 1) %5d  (%.1f x; no extra mat-vec)  %.0f s" % (kk, it, it_j / it, time.time() - t), flush=True)
            continue
        t = time.time(); _, res["tree"] = pcg(A, b, prec_tree(A, n_cams, Minv)); print("   tree                 %5d  (%.1f x)  %.0f s" % (res["tree"], it_j / res["tree"], time.time() - t), flush=True)
        for levels in (1, 2, 3):
            for smooth in ((False, True) if (FULL or levels == 3) else (False,)):
                t = time.time()
                H = build_hierarchy(A, n_cams, levels, smooth)
                sizes = " -> ".join(str(h["n"]) for h in H)
                nnz = sum(h["A"].nnz for h in H[1:]) / A.nnz
                _, it = pcg(A, b, prec_additive(H))
                print("   ml-add%s L=%d        %5d  (%.1f x)  [%s; coarse nnz / fine nnz = %.2f]  %.0f s" % ("-sm" if smooth else "   ", levels, it, it_j / it, sizes, nnz, time.time() - t), flush=True)
                if levels == 3:
                    t = time.time(); _, it = pcg(A, b, prec_vcycle(H))
                    print("   ml-V11%s L=%d        %5d  (%.1f x; + 2 fine mat-vecs and %.2f of one on the coarse levels per application)  %.0f s" % ("-sm" if smooth else "   ", levels, it, it_j / it, 2 * nnz, time.time() - t), flush=True)
        if k == 1 and W1 is not None:
            for kk in (8, 32):
                Wk = W1[:, :kk]
                AW = A @ Wk; G = np.linalg.inv(Wk.T @ AW)
                def defl(r, Wk=Wk, G=G): return jac(r) + Wk @ (G @ (Wk.T @ r))   # additive coarse correction on the recycled vectors (symmetric positive definite: valid inside PCG)
                t = time.time(); _, it = pcg(A, b, defl)
                print("   defl k=%-2d (recycled from step 1) %5d  (%.1f x; no extra mat-vec)  %.0f s" % (kk, it, it_j / it, time.time() - t), flush=True)


if __name__ == "__main__":
    main()
