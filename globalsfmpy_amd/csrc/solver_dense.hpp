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
  // schedule: one fused kernel per block column (shortest chain for tiny matrices), or panel + MFMA update + one backward launch per block
  // row.  Measured (tools/bench_chol.hip, profiles/r03_bench_chol.txt): 1.42 vs 2.7 ms at 3N = 2400, 3.45 vs 10.2 ms at 4500; at Madrid's
  // 1182 the fused schedule is the faster one inside the solver (37.7 vs 39.3 ms of linear solves per 63 LM iterations), and up to
  // ~68 block columns in the benchmark (2048: 1.06 vs 1.10 ms), so the switch is at 64 block columns (682 cameras).
  constexpr uint32_t split_T = GSFM_CHOL_SPLIT_DEFAULT;
  auto enqueue = [&]() {
    (void)hipMemsetAsync(P->denseA.p, 0, 8 * elems, P->stream);
    int* const info = (int*)(P->scal.p + SC_DENSE_INFO);
    DenseArgs a{};
    a.n_rows = P->n_rows; a.row_ptr = P->row_ptr.p; a.col = P->col.p; a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.h3 = P->h3.p; a.h4 = P->h4.p;
    a.Mblk = P->Mblk.p; a.b = P->b.p; a.A = P->denseA.p; a.n = n; a.T = T; a.q = P->q_lin; a.lap = P->lin_is_lap; a.info_slot = P->scal.p + SC_DENSE_INFO; a.rcg = P->r.p;
    if (P->cs.active) hipLaunchKernelGGL(k_dense_assemble_col, dim3(P->cs.n_wg), dim3(GSFM_BLOCK), 0, P->stream, a, P->cs.dev());
    else hipLaunchKernelGGL(k_dense_assemble, dim3(P->n_rows), dim3(GSFM_BLOCK), 0, P->stream, a);
    // Up to 64 block columns: one launch per TWO block columns, the panels of the columns c0, c0 + 1 beside the update with the two columns before
    // them (dense_kernels.hpp, k_chol_look2; round 6: the same bits as the fused step it replaces at one elimination per block row and
    // column -- Madrid's 37 columns 519 -> 444 us per factorisation + solve).  GSFM_CHOL_FUSED=1: the fused step (A/B, the bit-identity test).
    const char* fused_env = getenv("GSFM_CHOL_FUSED");
    const bool fused = fused_env && fused_env[0] == '1';
    if (T <= split_T && !fused) {
      CholArgs c{P->denseA.p, P->denseL.p, T, 0, info};
      hipLaunchKernelGGL(k_chol_look2<0>, dim3(chol_look2_grid(T, 0, false)), dim3(256), 0, P->stream, c);
      for (c.k = 2; c.k < T; c.k += 2) hipLaunchKernelGGL(k_chol_look2<2>, dim3(chol_look2_grid(T, c.k, true)), dim3(256), 0, P->stream, c);
    } else if (T <= split_T) {
      for (uint32_t k = 0; k < T; ++k) {
        CholArgs c{P->denseA.p, P->denseL.p, T, k, info};
        const uint64_t m = T - k;
        { const uint32_t nt = chol_step_tiles_per_wg((uint32_t)m); const dim3 grid(chol_step_grid((uint32_t)m, nt));
          if (nt == 3) hipLaunchKernelGGL(k_chol_step<3>, grid, dim3(256), 0, P->stream, c); else if (nt == 2) hipLaunchKernelGGL(k_chol_step<2>, grid, dim3(256), 0, P->stream, c); else hipLaunchKernelGGL(k_chol_step<1>, grid, dim3(256), 0, P->stream, c); }
      }
    } else {
      // larger matrices: panel (one wavefront per tile row), then the trailing update on the matrix cores -- block columns in GROUPS of two (four beyond 192 block columns):
      // inside a group the finished columns are folded into the NEXT block column alone, so that its panel can run, and after the group all
      // of them are folded into the rest in one pass (every trailing tile read and written once per group instead of once per column; same
      // launch count; per tile the columns are still applied in ascending order, so the factor is bit-identical to the column-by-column schedule)
      auto update = [&](uint32_t k, uint32_t ncol, uint32_t j0, bool col_only) {
        CholUpdArgs u{P->denseA.p, P->denseL.p, T, k, j0, col_only ? 1u : 0u};
        const uint64_t m = T - j0 + 1, tiles = col_only ? m : m * (m + 1) / 2;
        if (j0 > T || !tiles) return;
        const dim3 grid((uint32_t)((tiles + 3) / 4)), blk(256);
        if (ncol == 4) hipLaunchKernelGGL(k_chol_update_mfma<4>, grid, blk, 0, P->stream, u);
        else if (ncol == 3) hipLaunchKernelGGL(k_chol_update_mfma<3>, grid, blk, 0, P->stream, u);
        else if (ncol == 2) hipLaunchKernelGGL(k_chol_update_mfma<2>, grid, blk, 0, P->stream, u);
        else hipLaunchKernelGGL(k_chol_update_mfma<1>, grid, blk, 0, P->stream, u);
      };
      const uint32_t GROUP = T > 192 ? 4 : 2;   // (3N = 2400 / 4500 / 9000: pairs 1.42 / 3.44 / 14.7 ms, fours 1.48 / 3.50 / 13.8; column by column 1.48 / 3.75 / 17.4)
      for (uint32_t k = 0; k < T; k += GROUP) {
        const uint32_t g = std::min(GROUP, T - k);
        for (uint32_t c = 0; c < g; ++c) {
          CholArgs pc{P->denseA.p, P->denseL.p, T, k + c, info};
          hipLaunchKernelGGL(k_chol_panel, dim3(T - (k + c) + 1), dim3(64), 0, P->stream, pc);
          if (c + 1 < g) update(k, c + 1, k + c + 1, true);    // columns k .. k + c into block column k + c + 1 alone: the next panel's input
        }
        update(k, g, k + g, false);                            // all g columns into the rest (for the last group: the right-hand side row only)
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
