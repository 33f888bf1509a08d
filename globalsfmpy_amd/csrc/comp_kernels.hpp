// Kernels of the per-component exact step of a disconnected view graph (host side: solver_components.hpp).
#pragma once
#include "kernels.hpp"
#include "dense_kernels.hpp"

namespace gsfm {

// ---- block-diagonal systems (a disconnected view graph: BASELINE C4, the 14 scenes as one problem) ---------------------------------------
// The reference's Cholesky factorises the normal matrix block by block, every connected component exactly (estimator.cpp:299-305).  Here the
// components of at most dense_cholesky_max_cams cameras are assembled into their own tiled matrices by ONE launch and factorised side by side
// (dense_kernels.hpp, k_chol_*_batch); the larger ones stay with PCG, which is started on the right-hand side with the small components'
// entries zeroed -- their iterates then stay zero and they take no part in any of its scalars (solver_components.hpp).
struct CompMap {
  const int32_t* item;    // per camera: index of its component in the batch, -1: not factorised (a large component, a camera without edges)
  const uint32_t* loc;    // per camera: its index inside its component
  uint32_t row_base;      // camera of row 0 (a rank of a packed sharded problem holds the rows of its own cameras only)
};
// Which factorised components have anything to solve in this LM step: component c is skipped when sum_{k in c} b_k . Minv_k b_k <= floor^2 / B,
// i.e. when block-Jacobi's estimate of every one of its cameras' steps is below the absolute floor of the step (kernels.hpp, k_cam_bound) -- a
// scene that converged dozens of LM iterations ago while the batch iterates on.  One workgroup per component over its cameras in a fixed
// order: the same decision on every run.  (Its A tiles are then neither assembled nor factorised, its step is zero.)
// Under a SMOOTH loss a component is moreover put to rest for the remainder of the solve once its EXACT step -- the factorisation's, measured
// by k_comp_scatter: the largest camera update of the component in radians -- has been below `freeze_below` (1e-10 rad): the scenes of a batch
// are independent problems, a scene whose Newton step is 1e-10 rad has 1e-9 rad left to go at most (steps near convergence contract), three
// orders inside the parity bar, and factorising it 30 more times while another scene iterates on is what made C4 cost twice the slow scene
// alone.  Never under the MAGSAC losses (freeze_below = 0): there an iterate 1e-10 rad off can sit in another table cell.
__global__ void __launch_bounds__(GSFM_BLOCK) k_comp_activity(const uint32_t* __restrict__ cam_ptr, const uint32_t* __restrict__ cams, const double* __restrict__ b,
                                                              const double* __restrict__ Minv, const double* zbound, double floor2, int* active,
                                                              unsigned long long* stepmax, unsigned long long* stepprev, int* frozen, double freeze_below, const double* freeze_ok) {
  __shared__ double lds[8];
  const uint32_t c = blockIdx.x;
  double v = 0.0;
  for (uint32_t u = cam_ptr[c] + threadIdx.x; u < cam_ptr[c + 1]; u += GSFM_BLOCK) {
    const uint32_t k = cams[u];
    const double r[3] = {b[3 * (size_t)k], b[3 * (size_t)k + 1], b[3 * (size_t)k + 2]};
    double z[3];
    sym3_mulvec(Minv + 6 * (size_t)k, r, z);
    v += r[0] * z[0] + r[1] * z[1] + r[2] * z[2];
  }
  const double t = block_sum_bcast(v, lds);
  if (threadIdx.x == 0) {
    const double B = *zbound;
    int fr = frozen[c];
    // cur: last iteration's exact step of this component (+inf before the first and after an iteration in which it was idle -- an idle component
    // was not factorised, nothing was measured, and block-Jacobi's estimate, which idles it, is no bound on an ill-conditioned scene's true step);
    // prev: the one measured before it.  A component goes to rest when its step is below the threshold AND there is evidence that it is small
    // because the scene has converged, not because the shared trust region has collapsed under ANOTHER scene's rejections (round-5 advisor: a
    // damped step can fall below any threshold on an unconverged scene): either the step has at least halved against the previous measurement --
    // steps near convergence contract fast (C4: by 4-10 x per accepted step), a damping-limited scene's barely move, and a REJECTED iteration
    // re-solves the same point (ratio ~ 1) -- or the damping is no stronger than at the start of the solve (*freeze_ok: radius >= its initial value).
    const double cur = __longlong_as_double((long long)stepmax[c]), prev = __longlong_as_double((long long)stepprev[c]);
    const bool contracted = prev < __longlong_as_double(0x7ff0000000000000ll) && cur <= 0.5 * prev;
    if (!fr && freeze_below > 0.0 && cur <= freeze_below && (contracted || *freeze_ok != 0.0)) fr = 1;
    const int act = !fr && (!(B > 0.0) || t * B > floor2);
    frozen[c] = fr; stepprev[c] = stepmax[c]; stepmax[c] = act ? 0ull : 0x7ff0000000000000ull;
    active[c] = act;
  }
}
// (one workgroup per camera of a factorised component -- `cams`, the list k_comp_activity walks -- not per row of the problem: at C4 2 168 of 14 019)
__global__ void __launch_bounds__(GSFM_BLOCK) k_comp_assemble(DenseArgs a, CompMap cm, const CholBatchItem* items, const uint32_t* __restrict__ cams) {
  if (blockIdx.x == 0 && threadIdx.x == 3) *a.info_slot = 0.0;
  const uint32_t row = cams[blockIdx.x];            // the camera,
  const uint32_t lrow = row - cm.row_base;          // its row of the stored blocks
  if (row < cm.row_base || lrow >= a.n_rows) return;
  const int32_t ci = cm.item[row];
  if (ci < 0) return;
  const CholBatchItem it = items[ci];
  if (!*it.active) { if (threadIdx.x == 0 && cm.loc[row] == 0) *it.info = 0; return; }
  const uint32_t lr = cm.loc[row];
  if (threadIdx.x == 0) {
    const double* M = a.Mblk + 6 * (size_t)row;
    const double m[9] = {M[0], M[1], M[2], M[1], M[3], M[4], M[2], M[4], M[5]};
    for (int r = 0; r < 3; ++r) for (int c = 0; c <= r; ++c) *dense_elem(it.A, 3 * lr + r, 3 * lr + c) = m[3 * r + c];
    for (int c = 0; c < 3; ++c) it.A[(((size_t)it.T * (it.T + 1) / 2) + (3 * lr + c) / 32) * 1024 + (3 * lr + c) % 32] = a.b[3 * (size_t)row + c];
    if (lr == 0) { for (uint32_t g = it.n; g < it.T * 32; ++g) *dense_elem(it.A, g, g) = 1.0; *it.info = 0; }   // padding of the last tile: identity
  }
  for (uint32_t d = a.row_ptr[lrow] + threadIdx.x; d < a.row_ptr[lrow + 1]; d += GSFM_BLOCK) {
    const uint32_t m = a.col[d] & 0x7fffffffu;
    if (m >= row) continue;   // upper triangle; (m and row are in the same component: cm.item[m] == ci)
    const uint32_t lm = cm.loc[m];
    double H[9];
    if (a.lap) {
      const double2 A0 = a.h0[d], B0 = a.h1[d], C0 = a.h2[d];
      const double Gm[9] = {A0.x, A0.y, B0.x, A0.y, B0.y, C0.x, B0.x, C0.x, C0.y};
      double Rk[9], Rm[9], T[9];
      qmat(load_q(a.q, row), Rk);
      qmat(load_q(a.q, m), Rm);
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[3 * r + c] = Rk[3 * r] * Rm[3 * c] + Rk[3 * r + 1] * Rm[3 * c + 1] + Rk[3 * r + 2] * Rm[3 * c + 2];   // R_k R_m^T
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[3 * r + c] = -(Gm[3 * r] * T[c] + Gm[3 * r + 1] * T[3 + c] + Gm[3 * r + 2] * T[6 + c]);
    } else {
      const double2 A0 = a.h0[d], B0 = a.h1[d], C0 = a.h2[d], D0 = a.h3[d];
      H[0] = A0.x; H[1] = A0.y; H[2] = B0.x; H[3] = B0.y; H[4] = C0.x; H[5] = C0.y; H[6] = D0.x; H[7] = D0.y; H[8] = a.h4[d];
    }
    // the numbering inside a component follows the cameras' order, so lm < lr and the entry is in the lower triangle; several directed
    // entries of a repeated pair add up
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) atomicAdd(dense_elem(it.A, 3 * lr + r, 3 * lm + c), H[3 * r + c]);
  }
}
// b_pcg = b with the factorised components' entries zeroed (PCG's right-hand side)
__global__ void __launch_bounds__(GSFM_BLOCK) k_comp_mask_rhs(const double* b, CompMap cm, uint32_t n, double* b_pcg) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k >= n) return;
  const bool dense = cm.item[k] >= 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) b_pcg[3 * (size_t)k + c] = dense ? 0.0 : b[3 * (size_t)k + c];
}
// out = b on the cameras [lo, hi), zero elsewhere: the right-hand side of a rank's own PCG on a PACKED sharded problem (problem_create.hpp) -- x, r, z
// and p then stay exactly zero on the other ranks' cameras and every dot product is the rank's own
__global__ void __launch_bounds__(GSFM_BLOCK) k_mask_range(const double* b, uint32_t lo, uint32_t hi, uint32_t n, double* out) {
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k >= n) return;
  const bool own = k >= lo && k < hi;
#pragma unroll
  for (int c = 0; c < 3; ++c) out[3 * (size_t)k + c] = own ? b[3 * (size_t)k + c] : 0.0;
}
// the factorised components' solutions into the step vector (their PCG residual is exactly zero: PCG ran on a zero right-hand side there --
// or did not run at all: all_dense, then the large components' entries are cleared too), and the first failing factorisation, if any, into
// the scalar block's status word (the caller then solves the whole step by PCG)
__global__ void __launch_bounds__(GSFM_BLOCK) k_comp_scatter(CompMap cm, const CholBatchItem* items, uint32_t n_items, uint32_t n, int all_dense, double* eta, double* rcg, double* info_slot,
                                                             const double* __restrict__ Tinv, unsigned long long* stepmax, double* bad_flag) {
  // (the maxima first per workgroup in LDS -- 16 slots claimed by component number, a workgroup's 256 consecutive cameras belong to a few
  // components at most; a component that finds its slot taken goes to memory directly --, then one atomic per slot: 2 168 atomics on six words
  // were 15 of the kernel's 18 us at C4)
  __shared__ int tag[16];
  __shared__ unsigned long long smax[16];
  if (threadIdx.x < 16) { tag[threadIdx.x] = -1; smax[threadIdx.x] = 0ull; }
  __syncthreads();
  const uint32_t k = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (k == 0) {
    int bad = 0;
    for (uint32_t c = 0; c < n_items && !bad; ++c) bad = *items[c].info;
    __builtin_memcpy(info_slot, &bad, sizeof(int));
    if (bad_flag) *bad_flag = bad ? 1.0 : 0.0;   // (packed sharded problems: summed over the ranks, so that all of them fall back together)
  }
  const int32_t ci = k < n ? cm.item[k] : -1;
  if (ci >= 0) {
    const bool live = *items[ci].active != 0;
    const double* x = items[ci].x + 3 * (size_t)cm.loc[k];
#pragma unroll
    for (int c = 0; c < 3; ++c) { eta[3 * (size_t)k + c] = live ? x[c] : 0.0; rcg[3 * (size_t)k + c] = 0.0; }
    if (live) {   // the component's largest camera update (delta = Tinv eta, radians; half-angles for the quaternion state): positive doubles order like their bits
      const double* Ti = Tinv + 9 * (size_t)k;
      const double d0 = Ti[0] * x[0] + Ti[1] * x[1] + Ti[2] * x[2], d1 = Ti[3] * x[0] + Ti[4] * x[1] + Ti[5] * x[2], d2 = Ti[6] * x[0] + Ti[7] * x[1] + Ti[8] * x[2];
      const unsigned long long v = (unsigned long long)__double_as_longlong(2.0 * sqrt(d0 * d0 + d1 * d1 + d2 * d2));
      const int slot = ci & 15, prev = atomicCAS(&tag[slot], -1, ci);
      if (prev == -1 || prev == ci) atomicMax(&smax[slot], v);
      else atomicMax(stepmax + ci, v);
    }
  } else if (all_dense && k < n) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { eta[3 * (size_t)k + c] = 0.0; rcg[3 * (size_t)k + c] = 0.0; }
  }
  __syncthreads();
  if (threadIdx.x < 16 && tag[threadIdx.x] >= 0) atomicMax(stepmax + tag[threadIdx.x], smax[threadIdx.x]);
}

}  // namespace gsfm
