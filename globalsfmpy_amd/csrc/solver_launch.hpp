// Host side, part 2: kernel dispatch on (functor, whitening mode, loss shape), loss preparation, the collective callbacks of a sharded
// problem and the launchers of the per-edge / per-camera kernels (K1 cost sweep, K2 / K2c linearisation, K3 / K3c mat-vec, K5 camera kernels).
#pragma once
#include "host_common.hpp"

namespace {

// ---- kernel dispatch on (functor, whitening mode, loss shape) --------------------------------
int loss_mode(const gsfm_rot_problem* P) {
  if (P->cb) return LM_SIMPLE;  // rho comes from rho_ext; the in-kernel loss is never evaluated
  const DevLoss& L = P->h_loss;
  if (L.n == 0) return LM_SIMPLE;
  if (L.n == 1) {
    const int k = L.nodes[0].kind;
    if (k == GSFM_LOSS_MAGSAC) return (L.nodes[0].nu == 3 && !L.nodes[0].inverse) ? LM_MAGSAC : LM_PROGRAM;
    if (k == GSFM_LOSS_TRIVIAL || k == GSFM_LOSS_HUBER || k == GSFM_LOSS_SOFT_L1 || k == GSFM_LOSS_TUKEY || k == GSFM_LOSS_GEMAN_MCCLURE) return LM_SIMPLE;
  }
  return LM_PROGRAM;
}
template <typename ArgsT, template <int, int, int> class Launcher>
int dispatch(const gsfm_rot_problem* P, const ArgsT& args, int grid) {
  const int f = P->functor, w = (P->wmode == W_MATRIX && P->q3) ? W_MATRIX3 : P->wmode, l = loss_mode(P);   // (W_MATRIX3: the same kernels on three-component measurement planes)
#define GSFM_CASE3(F, W, L) if (f == F && w == W && l == L) { Launcher<F, W, L>::go(args, grid, P->stream); return 0; }
#define GSFM_CASE(F, W) GSFM_CASE3(F, W, LM_PROGRAM) GSFM_CASE3(F, W, LM_SIMPLE) GSFM_CASE3(F, W, LM_MAGSAC)
  GSFM_CASE(F_AA, W_NONE) GSFM_CASE(F_AA, W_SCALAR) GSFM_CASE(F_AA, W_MATRIX) GSFM_CASE(F_AA, W_MATRIX3)
  GSFM_CASE(F_QCOS, W_NONE) GSFM_CASE(F_QNORM, W_NONE) GSFM_CASE(F_RFNORM, W_NONE)
#undef GSFM_CASE
#undef GSFM_CASE3
  return 1;
}
template <int F, int W, int L> struct CostLauncher {
  static void go(const CostArgs& a, int grid, hipStream_t s) {
    const bool full = a.s_only || a.rho_ext || a.srho_out || a.rho12_out || a.rho1_out || a.r_out || a.sigma.on;
    // the reweight sweep proper (rho' out and nothing else asked for) has its own lean instantiation
    const bool reweight = a.rho1_out && !(a.s_only || a.rho_ext || a.srho_out || a.rho12_out || a.r_out || a.sigma.on);
    if (a.direct) {
      if (reweight) hipLaunchKernelGGL((k_cost_direct<F, W, L, 2>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
      else if (full) hipLaunchKernelGGL((k_cost_direct<F, W, L, 1>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
      else hipLaunchKernelGGL((k_cost_direct<F, W, L, 0>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
      return;
    }
    if (reweight) hipLaunchKernelGGL((k_cost<F, W, L, 2>), dim3(grid), dim3(GSFM_TILE_THREADS), 0, s, a);
    else if (full) hipLaunchKernelGGL((k_cost<F, W, L, 1>), dim3(grid), dim3(GSFM_TILE_THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_cost<F, W, L, 0>), dim3(grid), dim3(GSFM_TILE_THREADS), 0, s, a);
  }
};
// K2: GSFM_K2_FAST=0 switches the fast path (losses with rho'' <= 0) off, for A/B measurements.  Read at every launch.
bool k2_fast_enabled() {
  const char* e = getenv("GSFM_K2_FAST");
  return !(e && *e && atoi(e) <= 0);
}
// ... and the loss must really have rho'' <= 0 everywhere: true for the cheap leaves with ordinary parameters, but e.g. Geman-McClure with a
// negative sigma^2 has rho'' > 0 (t = s / a^2 + g2 < 0), where the reference applies the Corrector in full (round-3 advisor).  Decided per
// problem in prepare_loss (gsfm_rot_problem::fast_lin_ok); the launchers take the conjunction through LinArgs::fast_ok.
bool k2_fast_path(const gsfm_rot_problem* P) { return k2_fast_enabled() && P->fast_lin_ok; }
template <int F, int W, int L> struct LinLauncher {
  static void go(const LinArgs& a, int grid, hipStream_t s) {
    if constexpr (F == F_AA || F == F_QCOS) {   // functors of R_j R_i^T only: the Laplacian form exists (lin_rows)
      if (a.lap) {
        if constexpr (L != LM_PROGRAM) {
          // (a host-callback loss may have rho'' > 0: the general path applies the Corrector in full)
          if (!a.rho_ext && a.fast_ok) hipLaunchKernelGGL((k_lin_fast<F, W, L>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
          else hipLaunchKernelGGL((k_lin3<F, W, L, true>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
        } else hipLaunchKernelGGL((k_lin<F, W, L, true>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
        return;
      }
    }
    if constexpr (L != LM_PROGRAM && F != F_RFNORM) hipLaunchKernelGGL((k_lin3<F, W, L, false>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((k_lin<F, W, L, false>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
  }
};
template <int F, int W, int L> struct ColLinLauncher {   // K2c (column-sorted layout: Laplacian-capable functors only)
  static void go(const ColLinArgs& a, int grid, hipStream_t s) {
    if constexpr (F == F_AA || F == F_QCOS) {
      const dim3 g(grid), b(GSFM_COLLIN_THREADS);
      if constexpr (L != LM_PROGRAM) {
        if (!a.lin.rho_ext && a.lin.fast_ok) hipLaunchKernelGGL((k_lin_col<F, W, L, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_lin_col<F, W, L, false>), g, b, 0, s, a);
      } else hipLaunchKernelGGL((k_lin_col<F, W, L, false>), g, b, 0, s, a);
    }
  }
};

int sync_check(gsfm_rot_problem* P, const char* what) {
  hipError_t e = hipStreamSynchronize(P->stream);
  if (e != hipSuccess) return fail(GSFM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
  e = hipGetLastError();
  if (e != hipSuccess) return fail(GSFM_ERR_HIP, std::string(what) + " (launch): " + hipGetErrorString(e));
  P->timer.resolve();
  return 0;
}

// Read `bytes` (<= 256) from the device and wait.  The LM / PCG control reads ~100 bytes two to five times per iteration; a copy into
// pageable memory costs 22 us per read on this platform, into pinned memory 14 us (tools/archive/bench_sync.hip), which is what small graphs feel.
int read_back(gsfm_rot_problem* P, void* dst, const void* src_dev, size_t bytes, const char* what) {
  void* stage = (P->pin && bytes <= 256) ? P->pin : dst;
  HIPCHK(hipMemcpyAsync(stage, src_dev, bytes, hipMemcpyDeviceToHost, P->stream));
  if (int st = sync_check(P, what)) return st;
  if (stage != dst) std::memcpy(dst, stage, bytes);
  return 0;
}

// ---- loss preparation ---------------------------------------------------------------------
int prepare_loss(gsfm_rot_problem* P, const gsfm_loss_node* prog, int n) {
  if (n < 0 || n > GSFM_LOSS_MAX_NODES) return fail(GSFM_ERR_INVALID_ARG, "loss program length out of range");
  P->loss_epoch++;   // (captured LM iterations froze the kernels the old loss selected)
  DevLoss L;
  std::memset(&L, 0, sizeof(L));
  L.n = n;
  int nr = 0, na = 1;
  for (int k = 0; k < n; ++k) {
    const gsfm_loss_node& s = prog[k];
    DevLossNode& d = L.nodes[k];
    d.kind = s.kind; d.p[0] = s.p[0]; d.p[1] = s.p[1]; d.p[2] = s.p[2];
    switch (s.kind) {
      case GSFM_LOSS_OP_SCALE: if (nr < 1) return fail(GSFM_ERR_INVALID_ARG, "loss program: SCALE on empty stack"); break;
      case GSFM_LOSS_OP_PUSH_ARG:
        if (nr < 1 || na >= GSFM_LOSS_MAX_STACK) return fail(GSFM_ERR_INVALID_ARG, "loss program: bad PUSH_ARG");
        ++na; break;
      case GSFM_LOSS_OP_COMPOSE:
        if (nr < 2 || na < 2) return fail(GSFM_ERR_INVALID_ARG, "loss program: bad COMPOSE");
        --nr; --na; break;
      case GSFM_LOSS_TRIVIAL: case GSFM_LOSS_HUBER: case GSFM_LOSS_SOFT_L1: case GSFM_LOSS_CAUCHY: case GSFM_LOSS_ARCTAN:
      case GSFM_LOSS_TUKEY: case GSFM_LOSS_LONE_HALF: case GSFM_LOSS_LTWO: case GSFM_LOSS_GEMAN_MCCLURE:
        if (nr >= GSFM_LOSS_MAX_STACK) return fail(GSFM_ERR_INVALID_ARG, "loss program: stack overflow");
        ++nr; break;
      case GSFM_LOSS_TOLERANT:
        if (!(s.p[0] >= 0) || !(s.p[1] > 0)) return fail(GSFM_ERR_INVALID_ARG, "TolerantLoss needs a >= 0, b > 0");
        if (nr >= GSFM_LOSS_MAX_STACK) return fail(GSFM_ERR_INVALID_ARG, "loss program: stack overflow");
        d.p[2] = s.p[1] * std::log(1 + std::exp(-s.p[0] / s.p[1]));  // c (loss_functions.py:147)
        ++nr; break;
      case GSFM_LOSS_MAGSAC: {
        const int nu = (int)s.p[1];
        if (nu != 3 && nu != 4 && nu != 9) return fail(GSFM_ERR_INVALID_ARG, "MAGSAC loss: nu must be 3, 4 or 9");
        if (nr >= GSFM_LOSS_MAX_STACK) return fail(GSFM_ERR_INVALID_ARG, "loss program: stack overflow");
        const MagsacConst c = magsac_const(nu);
        const double sigma = s.p[0];
        // loss_functions.py:286-298 (constructor constants)
        const double squared_sigma = sigma * sigma;
        const double dof_minus_one_per_two = (c.nu - 1.0) / 2.0;
        const double C_times_two_ad_dof = c.C * std::pow(2.0, dof_minus_one_per_two);
        const double one_over_sigma = C_times_two_ad_dof / sigma;
        const double gamma_difference = std::tgamma(dof_minus_one_per_two) - c.gk;
        d.nu = nu; d.inverse = s.p[2] != 0.0;
        d.aux[0] = squared_sigma; d.aux[1] = 2.0 * squared_sigma; d.aux[2] = squared_sigma * sigma;
        d.aux[3] = C_times_two_ad_dof; d.aux[4] = one_over_sigma; d.aux[5] = one_over_sigma * gamma_difference;
        d.aux[6] = c.q * c.q * squared_sigma; d.aux[7] = c.gk;
        d.rho1_scale = C_times_two_ad_dof / (2.0 * squared_sigma * sigma);   // rho' = rho1_scale * exp(-x / 1000) for nu = 3 (loss_functions.py:311)
        d.rho2_scale = 2.0 * C_times_two_ad_dof / (squared_sigma * 8.0 * squared_sigma * sigma);   // -rho'' / exp(..) for nu = 3 (:319-321)
        d.e2_clamp = std::exp(-1e-7 / (2.0 * squared_sigma));
        d.x_clamp = 0;
        while (d.x_clamp < c.n && (double)d.x_clamp * (2.0 * squared_sigma) / 1000.0 < 1e-7) d.x_clamp++;   // cells whose s = x 2 sigma^2 / 1000 the reference lifts to 1e-7 (:317)
        const int ti = nu == 3 ? 0 : nu == 4 ? 1 : 2;
        if (!P->tables[ti].p) {
          if (P->tables[ti].upload(magsac_table(nu)) != hipSuccess) return fail(GSFM_ERR_HIP, "uploading MAGSAC table failed");
        }
        d.table = P->tables[ti].p; d.table_len = c.n;
        ++nr; break; }
      default: return fail(GSFM_ERR_INVALID_ARG, "loss program: unknown node kind");
    }
  }
  if (n > 0 && (nr != 1 || na != 1)) return fail(GSFM_ERR_INVALID_ARG, "loss program does not reduce to one value");
  // K2's fast path assumes rho'' <= 0 for every s (Ceres' Corrector then always takes its alpha = 0 branch).  That is a property of the leaf
  // kind AND of its parameters: Geman-McClure, rho'' = -g2^2 / (a^2 t^3) with t = s / a^2 + g2, turns positive for a negative sigma^2 (g2);
  // a MAGSAC weight loss with a negative sigma flips the sign of rho' and rho''.  Anything doubtful takes the general path (full Corrector).
  P->fast_lin_ok = true;
  if (n == 1) {
    const gsfm_loss_node& s0 = prog[0];
    if (s0.kind == GSFM_LOSS_GEMAN_MCCLURE && !(s0.p[1] > 0.0 && s0.p[0] != 0.0)) P->fast_lin_ok = false;
    if (s0.kind == GSFM_LOSS_MAGSAC && !(s0.p[0] > 0.0)) P->fast_lin_ok = false;
    if ((s0.kind == GSFM_LOSS_TUKEY || s0.kind == GSFM_LOSS_SOFT_L1 || s0.kind == GSFM_LOSS_HUBER) && !(s0.p[0] != 0.0)) P->fast_lin_ok = false;
  }
  // a loss that switches edges off altogether (rho' identically 0 beyond a cut-off: Tukey): see lm_solve, the forcing schedule
  P->loss_cuts_off = P->loss_staircase = false;
  for (int i = 0; i < n; ++i) {
    if (prog[i].kind == GSFM_LOSS_TUKEY) P->loss_cuts_off = true;
    if (prog[i].kind == GSFM_LOSS_MAGSAC) P->loss_staircase = true;   // rho and rho' piecewise constant in s: the table index is rounded (loss_functions.py:303)
  }
  P->h_loss = L;
  if (!P->d_loss.p && P->d_loss.alloc(1) != hipSuccess) return fail(GSFM_ERR_HIP, "alloc loss");
  HIPCHK(hipMemcpy(P->d_loss.p, &P->h_loss, sizeof(DevLoss), hipMemcpyHostToDevice));
  return 0;
}

// ---- collectives --------------------------------------------------------------------------
// A failing callback also drops the captured PCG chunks: they hold whatever kernels the communicator enqueued when they were captured (the peer-store
// exchange's, say), and a communicator that has just failed may answer differently from now on (csrc/gsfm_peer.hip: its fallback collectives) --
// the next solve captures afresh, through the callbacks.
int comm_failed(gsfm_rot_problem* P, const char* what) {
  P->pcg_graph.reset(); P->pcg2_graph.reset();
  P->shard.flags &= ~GSFM_SHARD_CAPTURABLE;   // ... and whatever it falls back to (host-staged collectives, say) is not assumed to be capturable: plain launches from now on
  return fail(GSFM_ERR_COMM, what);
}
int all_gather(gsfm_rot_problem* P, double* buf, size_t count_per_rank) {
  if (!P->sharded) return 0;
  P->n_collectives++;
  if (P->shard.all_gather(P->shard.ctx, buf, count_per_rank, (void*)P->stream) != 0) return comm_failed(P, "all_gather callback failed");
  return 0;
}
int all_reduce(gsfm_rot_problem* P, double* buf, size_t count) {
  if (!P->sharded) return 0;
  P->n_collectives++;
  if (P->shard.all_reduce_sum(P->shard.ctx, buf, count, (void*)P->stream) != 0) return comm_failed(P, "all_reduce callback failed");
  return 0;
}

// ---- launches -----------------------------------------------------------------------------
enum { SC_COST = 0, SC_GMAX = 1, SC_STEP = 2 /* ..6 */, SC_XNORM2 = 7, SC_TRIAL = 8, SC_ZL8 = 9 /* k_cam_step's sixth sum (loose steps) */, SC_ZBOUND = 10 /* k_cam_bound's B: the absolute floor of the PCG tolerance */, SC_COMPBAD = 11 /* packed sharded problems: ranks whose component factorisation broke down (all-reduced) */, SC_FREEZE_OK = 12 /* component step: 1.0 while a small exact step may put a component to rest (k_comp_activity) */, SC_DENSE_INFO = 15 /* an int: status of the Cholesky factorisation */, SC_N = 16,
       SC_CTL = 16 /* .. 31: the device-side LM control block (kernels.hpp, CT_*) */, SC_REC = 32 /* .. 95: ring of four per-iteration records of it */, SC_ALL = 96 };
enum { T_LIN = 0, T_SWEEP = 1, T_CG = 2 };

void launch_cache(gsfm_rot_problem* P, const double* x, double2* q) {
  hipLaunchKernelGGL(k_cam_cache, dim3(grid_for(P->n_cams)), dim3(GSFM_BLOCK), 0, P->stream, x, P->n_cams, P->param_dim, q);
}

// s of every edge this rank holds, by rows (sharded problems): see k_row_s
int launch_row_s(gsfm_rot_problem* P, const double2* q, double* s_out, bool unit_weights) {
  if (P->cs.active) {   // column-sorted layout (Laplacian-capable functors only); unit_weights is no longer asked for by any caller
    ColRowSArgs ca{};
    ca.L = P->cs.dev(); ca.row_base = P->own_begin; ca.n_rows = P->n_rows; ca.eid = P->dir.eid.p; ca.qr0 = P->dir.qr0.p; ca.qr1 = P->dir.qr1.p;
    ca.w0 = P->dir.w0.p; ca.w1 = P->dir.w1.p; ca.w2 = P->dir.w2.p; ca.ws = P->dir.ws.p; ca.q = q; ca.s_out = s_out;
    const dim3 grid(P->cs.n_wg), blk(GSFM_BLOCK);
    if (unit_weights) return fail(GSFM_ERR_UNSUPPORTED, "unit-weight row sweep on the column-sorted layout");
    if (P->functor == F_AA && P->wmode == W_NONE) hipLaunchKernelGGL((k_col_s<F_AA, W_NONE>), grid, blk, 0, P->stream, ca);
    else if (P->functor == F_AA && P->wmode == W_SCALAR) hipLaunchKernelGGL((k_col_s<F_AA, W_SCALAR>), grid, blk, 0, P->stream, ca);
    else if (P->functor == F_AA && P->wmode == W_MATRIX && P->q3) hipLaunchKernelGGL((k_col_s<F_AA, W_MATRIX3>), grid, blk, 0, P->stream, ca);
    else if (P->functor == F_AA && P->wmode == W_MATRIX) hipLaunchKernelGGL((k_col_s<F_AA, W_MATRIX>), grid, blk, 0, P->stream, ca);
    else if (P->functor == F_QCOS) hipLaunchKernelGGL((k_col_s<F_QCOS, W_NONE>), grid, blk, 0, P->stream, ca);
    else return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
    return 0;
  }
  RowSArgs ra{};
  ra.n_rows = P->n_rows; ra.row_base = P->own_begin; ra.G = P->G; ra.row_ptr = P->row_ptr.p; ra.col = P->col.p; ra.eid = P->dir.eid.p;
  ra.qr0 = P->dir.qr0.p; ra.qr1 = P->dir.qr1.p; ra.w0 = P->dir.w0.p; ra.w1 = P->dir.w1.p; ra.w2 = P->dir.w2.p; ra.ws = P->dir.ws.p; ra.q = q; ra.s_out = s_out;
  const dim3 grid(grid_for((size_t)P->n_rows * P->G)), blk(GSFM_BLOCK);
  const int f = P->functor, w = P->wmode;
#define GSFM_ROWS(F, W, U) hipLaunchKernelGGL((k_row_s<F, W, U>), grid, blk, 0, P->stream, ra)
  if (f == F_AA && w == W_SCALAR && unit_weights) GSFM_ROWS(F_AA, W_SCALAR, true);
  else if (f == F_AA && w == W_NONE) GSFM_ROWS(F_AA, W_NONE, false);
  else if (f == F_AA && w == W_SCALAR) GSFM_ROWS(F_AA, W_SCALAR, false);
  else if (f == F_AA && w == W_MATRIX && P->q3) GSFM_ROWS(F_AA, W_MATRIX3, false);
  else if (f == F_AA && w == W_MATRIX) GSFM_ROWS(F_AA, W_MATRIX, false);
  else if (f == F_QCOS) GSFM_ROWS(F_QCOS, W_NONE, false);
  else if (f == F_QNORM) GSFM_ROWS(F_QNORM, W_NONE, false);
  else if (f == F_RFNORM) GSFM_ROWS(F_RFNORM, W_NONE, false);
  else return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
#undef GSFM_ROWS
  return 0;
}

CostArgs cost_args(gsfm_rot_problem* P, const double2* q) {
  CostArgs a{};
  a.tiles = P->cost_tiles.p; a.direct = P->cost_direct; a.n_cams = P->n_cams; a.n = P->cost.n; a.idx = P->cost_idx.p; a.qr0 = P->cost.qr0.p; a.qr1 = P->cost.qr1.p;
  a.w0 = P->cost.w0.p; a.w1 = P->cost.w1.p; a.w2 = P->cost.w2.p; a.ws = P->cost.ws.p; a.ws_rw = P->cost.ws.p;
  a.q = q; a.loss = P->d_loss.p; a.eid = P->cost.eid.p; a.partials = P->part_cost.p;
  return a;
}

// host-callback loss: s per edge -> host -> rho triples per ORIGINAL edge -> device
int refresh_external_rho(gsfm_rot_problem* P, const double2* q) {
  const size_t E = P->n_edges_in;
  P->h_s.resize(E); P->h_rho.resize(3 * E);
  if (P->sharded) {   // the rows of this rank need rho for every edge it holds, not only for the ones it counts in the cost
    if (int st = launch_row_s(P, q, P->s_ext.p, false)) return st;
    HIPCHK(hipMemcpyAsync(P->h_s.data(), P->s_ext.p, 8 * E, hipMemcpyDeviceToHost, P->stream));
    if (int st = sync_check(P, "callback loss: read s")) return st;
    for (size_t e = 0; e < E; ++e) P->cb(P->cb_user, P->h_s[e], &P->h_rho[3 * e]);
  } else {            // K1's s-only mode writes in the problem's edge order; the callback's answers go back to the original numbering
    CostArgs a = cost_args(P, q);
    a.s_out = P->s_ext.p; a.s_only = 1;
    if (dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost)) return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
    HIPCHK(hipMemcpyAsync(P->h_s.data(), P->s_ext.p, 8 * P->cost.n, hipMemcpyDeviceToHost, P->stream));
    if (int st = sync_check(P, "callback loss: read s")) return st;
    for (size_t u = 0; u < P->cost.n; ++u) P->cb(P->cb_user, P->h_s[u], &P->h_rho[3 * (size_t)P->h_cost_eid[u]]);
  }
  HIPCHK(hipMemcpyAsync(P->rho_ext.p, P->h_rho.data(), 24 * E, hipMemcpyHostToDevice, P->stream));
  return 0;
}

// optional per-edge outputs of K1 (device pointers, problem edge order)
struct CostOutputs { double2* srho = nullptr; double2* rho12 = nullptr; double* rho1 = nullptr; double* r = nullptr; };

int launch_lin(gsfm_rot_problem* P, const double2* q, const double* go = nullptr);

// K1: cost at quaternion cache q -> scal[slot] (all-reduced when sharded)
// reduce = false: the caller sums the sweep's partials itself (k_lm_decide; unsharded problems only)
int launch_cost(gsfm_rot_problem* P, const double2* q, int slot, const CostOutputs& out = CostOutputs(), bool reduce = true) {
  const bool sig = P->sigma_pending_cost;
  P->sigma_pending_cost = false;
  if (P->cb) {
    if (sig) {   // the callback's s must already carry the new weights: weight-only passes of K1 and K2 first (rare path: the host loop dominates it)
      CostArgs a = cost_args(P, q);
      a.s_out = P->s_ext.p; a.s_only = 1; a.sigma = P->sigma; a.sigma.on = 1;
      if (dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost)) return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cost.p + P->nb_cost, P->nb_cost, P->sigma_sum.p);
      gsfm_loss_callback keep = P->cb;
      P->cb = nullptr;                       // (one linearisation with the in-kernel loss slot: only its weight stores matter)
      P->sigma_pending_lin = true;
      const int st = launch_lin(P, q);
      P->cb = keep;
      if (st) return st;
    }
    if (int st = refresh_external_rho(P, q)) return st;
  }
  CostArgs a = cost_args(P, q);
  a.rho_ext = P->cb ? P->rho_ext.p : nullptr;
  a.srho_out = out.srho; a.rho12_out = out.rho12; a.rho1_out = out.rho1; a.r_out = out.r; a.s_only = 0;
  if (sig && !P->cb) { a.sigma = P->sigma; a.sigma.on = 1; }
  const int tk = P->timer.begin(T_SWEEP);
  if (dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost)) return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
  P->timer.end(tk);
  if (reduce) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cost.p, P->nb_cost, P->scal.p + slot);
  if (sig && !P->cb) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cost.p + P->nb_cost, P->nb_cost, P->sigma_sum.p);
  return all_reduce(P, P->scal.p + slot, 1);
}

// K2: linearise at q -> gD (all-gathered), H blocks
int launch_lin(gsfm_rot_problem* P, const double2* q, const double* go) {
  if (P->cb) { if (int st = refresh_external_rho(P, q)) return st; }
  LinArgs a{};
  a.n_rows = P->n_rows; a.row_base = P->own_begin; a.G = P->G; a.row_ptr = P->row_ptr.p; a.col = P->col.p; a.eid = P->dir.eid.p;
  a.qr0 = P->dir.qr0.p; a.qr1 = P->dir.qr1.p; a.w0 = P->dir.w0.p; a.w1 = P->dir.w1.p; a.w2 = P->dir.w2.p; a.ws = P->dir.ws.p; a.ws_rw = P->dir.ws.p;
  a.q = q; a.loss = P->d_loss.p; a.rho_ext = P->cb ? P->rho_ext.p : nullptr; a.fast_ok = k2_fast_path(P) ? 1 : 0; a.go = go;
  if (P->sigma_pending_lin) { a.sigma = P->sigma; a.sigma.on = 1; P->sigma_pending_lin = false; }
  if (!P->lap && !P->h3.p && (P->h3.alloc(P->dir.n) != hipSuccess || P->h4.alloc(P->dir.n) != hipSuccess)) return fail(GSFM_ERR_HIP, "allocating the general normal-equation blocks failed");
  a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.h3 = P->h3.p; a.h4 = P->h4.p; a.gD = P->gD.p; a.lap = P->lap;
  P->lin_is_lap = P->lap; P->q_lin = q;
  const int tk = P->timer.begin(T_LIN);
  if (P->cs.active) {
    ColLinArgs ca{};
    ca.lin = a; ca.L = P->cs.dev(); ca.part = P->cs.part.p;
    if (dispatch<ColLinArgs, ColLinLauncher>(P, ca, (int)P->cs.n_wg)) return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
    // (the fast path of the angle-axis family sums in the rows' body frames: the finishing kernel rotates -- same predicate as ColLinLauncher's choice)
    const bool body = col_lin_body_frame(P->functor, loss_mode(P) != LM_PROGRAM && !a.rho_ext && a.fast_ok);
    hipLaunchKernelGGL(k_lin_col_finish, dim3(grid_for(P->n_rows)), dim3(GSFM_BLOCK), 0, P->stream, P->n_rows, P->own_begin, P->cs.nch, P->cs.n_wg, (const double*)P->cs.part.p, P->gD.p,
                       body ? q : (const double2*)nullptr);
  } else if (dispatch<LinArgs, LinLauncher>(P, a, grid_for((size_t)P->n_rows * P->G))) return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
  P->timer.end(tk);
  P->have_lin = true;
  return all_gather(P, P->gD.p, (size_t)P->shard.slice_width * 9);
}

void launch_prep(gsfm_rot_problem* P, const gsfm_rot_options& o, double radius, bool init_scale, const double* radius_dev = nullptr, bool reduce = true) {
  PrepArgs a{};
  a.radius_dev = radius_dev;
  a.n = P->n_cams; a.param_dim = P->param_dim; a.x = P->x.p; a.gD = P->gD.p; a.scale = P->scale.p;
  a.init_scale = init_scale; a.jacobi_scaling = o.jacobi_scaling; a.radius = radius; a.min_diag = o.min_lm_diagonal; a.max_diag = o.max_lm_diagonal;
  a.Mblk = P->Mblk.p; a.Minv = P->Minv.p; a.Lam = P->Lam.p; a.Tinv = P->Tinv.p; a.b = P->b.p; a.gmax_partials = P->part_cam.p;
  hipLaunchKernelGGL(k_cam_prep, dim3(P->nb_cam), dim3(GSFM_BLOCK), 0, P->stream, a);
  if (reduce) {
    hipLaunchKernelGGL(k_max_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cam.p, P->nb_cam, P->scal.p + SC_GMAX);
    // B of the PCG tolerance's absolute floor (kernels.hpp, k_cam_bound) for the damping just built -- host-controlled steps only: the
    // device-controlled exact pipeline runs no PCG (a PCG step behind it finds the bound of the last host-controlled prep, or none)
    double* part = P->part_cam.p + (size_t)5 * P->nb_cam;
    hipLaunchKernelGGL(k_cam_bound, dim3(P->nb_cam), dim3(GSFM_BLOCK), 0, P->stream, (const double*)P->Minv.p, (const double*)P->Tinv.p, (const double*)P->active.p, P->n_cams, part);
    hipLaunchKernelGGL(k_max_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, (const double*)part, P->nb_cam, P->scal.p + SC_ZBOUND);
  }
}

int launch_matvec(gsfm_rot_problem* P, const double* Mblk, const double* p, double* y, const int* done, double* dot_part = nullptr, bool* dot_done = nullptr) {
  if (dot_done) *dot_done = false;
  MatvecArgs a{};
  a.n_rows = P->n_rows; a.row_base = P->own_begin; a.G = P->G; a.row_ptr = P->row_ptr.p; a.col = P->col.p;
  a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.h3 = P->h3.p; a.h4 = P->h4.p; a.Mblk = Mblk; a.p = p; a.y = y; a.done = done;
  a.q = P->q_lin; a.u = P->u_rot.p;   // Laplacian form: the caller keeps u_rot = R^T p (PCG vector kernels, or k_cam_rotT)
  if (P->cs.active) {   // graphs without locality: the column-sorted form (always Laplacian)
    auto& c = P->cs;
    ColMatvecArgs m{};
    m.L = c.dev(); m.b0 = P->h0.p; m.b1 = P->h1.p; m.b2 = P->h2.p; m.u = P->u_rot.p; m.part = c.part.p; m.done = done;
    // (occupancy: four workgroups per CU; holding it at 3 / 2 / 1 with unused dynamic LDS measured 215 / 226 / 306 us against 196)
    if (c.k16_active) hipLaunchKernelGGL(k_mv_col<true>, dim3(c.n_wg), dim3(GSFM_K3C_THREADS), 0, P->stream, m);
    else hipLaunchKernelGGL(k_mv_col<false>, dim3(c.n_wg), dim3(GSFM_K3C_THREADS), 0, P->stream, m);
    ColFinishArgs f{};
    f.n_rows = P->n_rows; f.row_base = P->own_begin; f.nch = c.nch; f.n_wg = c.n_wg; f.part = c.part.p; f.Mblk = Mblk; f.p = p; f.q = P->q_lin; f.y = y; f.done = done;
    if (dot_part && !P->sharded) { f.dot_part = dot_part; *dot_done = true; }   // (one GPU: rows = cameras, the finish grid is the camera kernels' grid)
    hipLaunchKernelGGL(k_mv_col_finish, dim3(grid_for(P->n_rows)), dim3(GSFM_BLOCK), 0, P->stream, f);
    return P->pcg_local ? 0 : all_gather(P, y, (size_t)P->shard.slice_width * 3);
  }
  if (P->lin_is_lap) hipLaunchKernelGGL(k_matvec<true>, dim3(grid_for((size_t)P->n_rows * P->G)), dim3(GSFM_BLOCK), 0, P->stream, a);
  else hipLaunchKernelGGL(k_matvec<false>, dim3(grid_for((size_t)P->n_rows * P->G)), dim3(GSFM_BLOCK), 0, P->stream, a);
  return P->pcg_local ? 0 : all_gather(P, y, (size_t)P->shard.slice_width * 3);
}

}  // namespace
