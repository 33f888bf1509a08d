// Host side, part 4: the exact LM step for small graphs -- dense tiled Cholesky of the damped normal matrix (dense_kernels.hpp), one hipGraph per problem.
#pragma once
#include "host_common.hpp"

namespace {

// Exact step for small graphs: dense Cholesky of (J^T J + Lambda) in the left-tangent space (dense_kernels.hpp).  Enqueues only: the
// factorisation's status lands in the scalar block (SC_DENSE_INFO) and is read together with the trial cost, one host synchronisation
// later; a non-positive pivot makes the caller solve the step again by PCG.  *used = false if nothing was enqueued (size, memory).
// plain = true: enqueue the kernels themselves (the caller is capturing them into a graph of its own: solver_lm.hpp, the LM iteration graph)
int run_dense(gsfm_rot_problem* P, bool* used, bool plain = false) {
  *used = false;
  const uint32_t n = 3 * P->n_cams, T = (n + GSFM_CB - 1) / GSFM_CB;
  if (T > GSFM_DENSE_MAX_T) return 0;
  const size_t elems = chol_num_tiles(T) * GSFM_TILE_ELEMS;
  if (!P->denseA.p) {
    if (P->denseA.alloc(elems) != hipSuccess || P->denseL.alloc(elems, true) != hipSuccess || P->dense_x.alloc((size_t)T * GSFM_CB, true) != hipSuccess) { P->denseA.release(); return 0; }
  }
  auto enqueue = [&]() {
    (void)hipMemsetAsync(P->denseA.p, 0, 8 * elems, P->stream);
    int* const info = (int*)(P->scal.p + SC_DENSE_INFO);
    DenseArgs a{};
    a.n_rows = P->n_rows; a.row_ptr = P->row_ptr.p; a.col = P->col.p; a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.h3 = P->h3.p; a.h4 = P->h4.p;
    a.Mblk = P->Mblk.p; a.b = P->b.p; a.A = P->denseA.p; a.n = n; a.T = T; a.q = P->q_lin; a.lap = P->lin_is_lap; a.info_slot = P->scal.p + SC_DENSE_INFO; a.rcg = P->r.p;
    if (P->cs.active) hipLaunchKernelGGL(k_dense_assemble_col, dim3(P->cs.n_wg), dim3(GSFM_BLOCK), 0, P->stream, a, P->cs.dev());
    else hipLaunchKernelGGL(k_dense_assemble, dim3(P->n_rows), dim3(GSFM_BLOCK), 0, P->stream, a);
    // One launch per TWO block columns, the panels of the columns c0, c0 + 1 beside the update with the two columns before them
    // (dense_kernels.hpp, k_chol_look2; round 6).  Up to 64 block columns it gives the bits of the fused step it replaces at one
    // elimination per block row and column -- Madrid's 37 columns 519 -> 444 us per factorisation + solve; GSFM_CHOL_FUSED=1: the
    // fused step (A/B, the bit-identity test).  Beyond, it replaces the panel + MFMA-update pairs of rounds 3-5 (another summation order
    // in the update: the last bits of those factors changed): 3N = 2400 / 4500 / 9000 1.25 -> 0.89, 3.11 -> 2.25, 13.2 -> 11.8 ms.
    const char* fused_env = getenv("GSFM_CHOL_FUSED");
    const bool fused = fused_env && fused_env[0] == '1' && T <= GSFM_CHOL_FUSED_MAX_T;
    if (!fused) {
      CholArgs c{P->denseA.p, P->denseL.p, T, 0, info};
      hipLaunchKernelGGL(k_chol_look2<0>, dim3(chol_look2_grid(T, 0, false)), dim3(256), 0, P->stream, c);
      for (c.k = 2; c.k < T; c.k += 2) hipLaunchKernelGGL(k_chol_look2<2>, dim3(chol_look2_grid(T, c.k, true)), dim3(256), 0, P->stream, c);
    } else {
      for (uint32_t k = 0; k < T; ++k) {
        CholArgs c{P->denseA.p, P->denseL.p, T, k, info};
        const uint32_t m = T - k, nt = chol_step_tiles_per_wg(m);
        const dim3 grid(chol_step_grid(m, nt));
        if (nt == 3) hipLaunchKernelGGL(k_chol_step<3>, grid, dim3(256), 0, P->stream, c); else if (nt == 2) hipLaunchKernelGGL(k_chol_step<2>, grid, dim3(256), 0, P->stream, c); else hipLaunchKernelGGL(k_chol_step<1>, grid, dim3(256), 0, P->stream, c);
      }
    }
    // backward substitution, L^T x = y (y = block row T of L), in groups of 8 block rows: one workgroup solves a group, one launch
    // folds its x into all block rows above it (dense_kernels.hpp).  (The forms it replaced -- one workgroup for everything, one launch per
    // block row -- and groups of 16 lost their A/B runs, Madrid 37.6 / 36.0 against 35.1 ms of linear solves, and were removed in round 5.)
    constexpr uint32_t GR = 8;
    for (uint32_t k1 = T; k1 > 0;) {
      const uint32_t k0 = k1 > GR ? k1 - GR : 0;
      CholBackGroupArgs b{P->denseL.p, P->dense_x.p, n, T, k0, k1};
      hipLaunchKernelGGL(k_chol_back_group<GR>, dim3(1), dim3(64 * GR), 0, P->stream, b);
      if (k0) hipLaunchKernelGGL(k_chol_back_update<GR>, dim3(k0), dim3(32 * GR), 0, P->stream, b);
      k1 = k0;
    }
    (void)hipMemcpyAsync(P->xcg.p, P->dense_x.p, 8 * (size_t)n, hipMemcpyDeviceToDevice, P->stream);
    // (exact solve: the PCG residual term of the model decrease is zero -- k_dense_assemble cleared it)
  };
  if (plain) {
    if (!P->denseA.p) return 0;   // (nothing may be allocated under a capture)
    enqueue();
    *used = true;
    return 0;
  }
  const int tk = P->timer.begin(T_CG);
  if (P->dense_graph && P->dense_graph_lap != P->lin_is_lap) { (void)hipGraphExecDestroy(P->dense_graph); P->dense_graph = nullptr; }
  if (!P->dense_graph && !P->pcg_graph.unusable) {   // one launch per 32 columns: replay them as one graph
    P->dense_graph_lap = P->lin_is_lap;
    hipGraph_t captured = nullptr;
    if (hipStreamBeginCapture(P->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      enqueue();
      if (hipStreamEndCapture(P->stream, &captured) != hipSuccess || !captured || hipGraphInstantiate(&P->dense_graph, captured, nullptr, nullptr, 0) != hipSuccess)
        P->dense_graph = nullptr;
      if (captured) (void)hipGraphDestroy(captured);
    }
    if (!P->dense_graph) (void)hipGetLastError();
  }
  if (P->dense_graph) { HIPCHK(hipGraphLaunch(P->dense_graph, P->stream)); P->graph_launches++; }
  else enqueue();
  P->timer.end(tk);
  *used = true;
  return 0;
}

}  // namespace
