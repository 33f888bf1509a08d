// C-ABI implementation (include/gsfm_rot.h): problem assembly, Levenberg-Marquardt control with
// Ceres 1.14 trust-region semantics, block-Jacobi PCG orchestration.  All arithmetic on the edges
// and cameras runs in the kernels of kernels.hpp; the host only sequences launches and reads a
// handful of scalars per LM iteration.  Built with hipcc --offload-arch=gfx950 into libgsfm_rot.so.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <sched.h>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

#include "../../include/gsfm_rot.h"
#include "kernels.hpp"
#include "cov_kernels.hpp"
#include "dense_kernels.hpp"
#include "colsort_kernels.hpp"

using namespace gsfm;

namespace {

thread_local std::string g_err;
int fail(gsfm_status st, const std::string& msg) { g_err = msg; return st; }

// NULL when a device is usable, else the message for GSFM_ERR_NO_DEVICE (kept in a thread-local buffer).
const char* no_device_reason(const char* who) {
  int ndev = 0;
  const hipError_t e = hipGetDeviceCount(&ndev);
  if (e == hipSuccess && ndev > 0) return nullptr;
  static thread_local std::string msg;
  msg = std::string("no HIP device: ") + who + " has no CPU fallback (hipGetDeviceCount: " + hipGetErrorString(e) + ", " + std::to_string(ndev) +
        " devices; if another HIP runtime copy, e.g. PyTorch's bundled one, initialised first in this process, load it before this library)";
  (void)hipGetLastError();
  return msg.c_str();
}

#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      return fail(GSFM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));               \
    }                                                                                             \
  } while (0)

#define HIPCHK_S(expr)                                                                            \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) {                                                                       \
      return (gsfm_status)fail(GSFM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));  \
    }                                                                                             \
  } while (0)

// Host threads for the one-off structure build of gsfm_rot_problem_create: the affinity mask capped by the cgroup CPU quota (the GPU
// boxes show 256 hardware threads under a quota of 16; oversubscribing that is far slower than one thread) and by 16.
int host_threads() {
  if (const char* e = getenv("GSFM_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return std::min(v, 64); }
  long n = (long)std::thread::hardware_concurrency();
  if (n <= 0) n = 1;
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min<long>(n, CPU_COUNT(&set));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[64]; long period = 0;
    if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) n = std::min<long>(n, std::max<long>(1, (atol(q) + period / 2) / period));
    fclose(f);
  }
  return (int)std::max<long>(1, std::min<long>(n, 16));
}
// body(t, T) on T threads (T - 1 spawned + the caller)
template <typename F> void parallel_run(int T, F body) {
  if (T <= 1) { body(0, 1); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back([&body, t, T] { body(t, T); });
  body(0, T);
  for (auto& x : th) x.join();
}

// counts[k + 1] += number of items with key(u) == k, u in [0, n): per-thread histograms over contiguous item ranges, summed in thread order
template <typename K> void parallel_count(int T, size_t n, size_t n_keys, K key, uint32_t* counts_plus_one) {
  // one histogram per thread: cap the threads so that T * n_keys stays below 64 M counters (256 MB) -- the tile buckets have up to
  // 16.7 M keys; the serial loop needs one table
  T = (int)std::min<size_t>((size_t)T, std::max<size_t>(1, ((size_t)64 << 20) / std::max<size_t>(1, n_keys)));
  if (T <= 1 || n < 200000) { for (size_t u = 0; u < n; ++u) counts_plus_one[key(u)]++; return; }
  std::vector<std::vector<uint32_t>> h((size_t)T);
  parallel_run(T, [&](int t, int TT) {
    h[t].assign(n_keys, 0);
    const size_t lo = n * t / TT, hi = n * (t + 1) / TT;
    for (size_t u = lo; u < hi; ++u) h[t][key(u)]++;
  });
  parallel_run(T, [&](int t, int TT) {
    const size_t lo = n_keys * t / TT, hi = n_keys * (t + 1) / TT;
    for (size_t k = lo; k < hi; ++k) { uint32_t c = 0; for (int w = 0; w < TT; ++w) c += h[w][k]; counts_plus_one[k] += c; }
  });
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- MAGSAC constants / tables (include/gamma_values.cpp; regenerated, see oracle/ref_loss.hpp) ----
struct MagsacConst { double nu, C, q, gk; int n; };
MagsacConst magsac_const(int nu) {
  switch (nu) {
    case 3: return {3.0, 4.029720004054876e-01, 3.368214175218727, 3.439485560754856e-03, 36843};
    case 4: return {4.0, 2.525252525252525e-01, 3.643721193503644e+00, 3.611260617758625e-03, 38683};
    default: return {9.0, 3.837828575290349e-03, 4.654674460524809e+00, 3.344206155099048e-02, 48553};
  }
}
double upper_gamma_closed_form(int nu, double x) {
  if (x == 0.0) return std::tgamma((nu - 1.0) / 2.0);  // Gamma(a, 0) = Gamma(a): keeps rho(0) == 0 exactly
  if (nu == 3) return std::exp(-x);
  if (nu == 4) return 0.5 * std::sqrt(M_PI) * std::erfc(std::sqrt(x)) + std::sqrt(x) * std::exp(-x);
  return 6.0 * std::exp(-x) * (1.0 + x + x * x / 2.0 + x * x * x / 6.0);
}
std::vector<double> make_magsac_table(int nu) {
  const MagsacConst c = magsac_const(nu);
  std::vector<double> t(c.n);
  for (int x = 0; x < c.n; ++x) t[x] = upper_gamma_closed_form(nu, x / 1000.0);
  return t;
}
const std::vector<double>& magsac_table(int nu) {
  static const std::vector<double> t3 = make_magsac_table(3);
  static const std::vector<double> t4 = make_magsac_table(4);
  static const std::vector<double> t9 = make_magsac_table(9);
  return (nu == 3) ? t3 : (nu == 4) ? t4 : t9;
}

// Host staging arrays of tens of millions of elements that are filled completely, in parallel, right after they are sized: an allocator whose
// default construction does nothing, so that sizing them does not zero (and fault in) hundreds of megabytes on one thread first.
template <typename T>
struct NoInitAlloc {
  using value_type = T;
  NoInitAlloc() = default;
  template <typename U> NoInitAlloc(const NoInitAlloc<U>&) {}
  T* allocate(size_t n) { return std::allocator<T>().allocate(n); }
  void deallocate(T* p, size_t n) { std::allocator<T>().deallocate(p, n); }
  template <typename U, typename... A> void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
  }
  template <typename U> bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <typename U> bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <typename T> using hvec = std::vector<T, NoInitAlloc<T>>;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count, bool zero = false) {
    release();
    n = count;
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
    if (e != hipSuccess) { p = nullptr; return e; }
    if (zero) { e = hipMemset(p, 0, count * sizeof(T)); if (e == hipSuccess) e = hipDeviceSynchronize(); }
    return e;
  }
  template <typename A> hipError_t upload(const std::vector<T, A>& h) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  ~DevBuf() { release(); }
};

// one set of per-edge planes (cost edges or directed entries)
struct EdgePlanes {
  size_t n = 0;
  DevBuf<uint32_t> eid;
  DevBuf<double2> qr0, qr1, w0, w1, w2;
  DevBuf<double> ws;
};

inline int grid_for(size_t n) { return (int)((n + GSFM_BLOCK - 1) / GSFM_BLOCK); }

struct EventTimer {  // GPU time per phase, resolved at host syncs
  static constexpr int NPAIR = 96;
  hipEvent_t ev[2 * NPAIR];
  int cat[NPAIR];
  int used = 0;
  bool ok = false;
  double acc[3] = {0, 0, 0};
  hipStream_t stream = nullptr;
  void init() { ok = true; for (int k = 0; k < 2 * NPAIR; ++k) if (hipEventCreate(&ev[k]) != hipSuccess) ok = false; }
  void destroy() { if (ok) for (int k = 0; k < 2 * NPAIR; ++k) (void)hipEventDestroy(ev[k]); ok = false; }
  int begin(int category) {
    if (!ok) return -1;
    if (used == NPAIR) { (void)hipStreamSynchronize(stream); resolve(); }
    const int k = used++;
    cat[k] = category;
    (void)hipEventRecord(ev[2 * k], stream);
    return k;
  }
  void end(int k) { if (k >= 0) (void)hipEventRecord(ev[2 * k + 1], stream); }
  void resolve() {  // only after a stream sync
    for (int k = 0; k < used; ++k) { float ms = 0; if (hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]) == hipSuccess) acc[cat[k]] += ms; }
    used = 0;
  }
};


}  // namespace

struct gsfm_rot_problem {
  uint32_t n_cams = 0;
  uint64_t n_edges_in = 0;
  int error_type = 0, functor = F_AA, wmode = W_NONE, param_dim = 3, res_dim = 3;
  bool sharded = false;
  gsfm_rot_shard shard{};
  uint32_t own_begin = 0, own_end = 0, n_rows = 0, n_pad = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // replayable chunk of PCG iterations (hipGraph), keyed on the by-value kernel arguments it froze
  struct PcgGraph {
    hipGraphExec_t exec = nullptr;
    double tol = 0; int max_iters = 0, stall = 0, chunk = 0, collectives = 0; uint32_t coarse = 0;
    bool unusable = false, lap = false;
    void reset() { if (exec) (void)hipGraphExecDestroy(exec); exec = nullptr; }
  } pcg_graph, pcg2_graph;

  EdgePlanes cost;            // cost-owned edges
  DevBuf<uint2> cost_idx;
  DevBuf<CostTile> cost_tiles;
  // Laplacian form of the normal matrix (kernels.hpp, lin_rows): chosen per linearisation; u_rot = R^T p for the mat-vec
  const double2* q_lin = nullptr;   // quaternions the current blocks were linearised at
  bool lap = false, lap_capable = false, lin_is_lap = false;   // lin_is_lap: what the stored blocks currently are
  DevBuf<double> u_rot;
  // locality relabelling adopted at create (empty = identity): internal id = perm[external id]
  std::vector<uint32_t> perm;
  std::vector<double> h_cam;   // staging for permuted per-camera transfers
  int cost_direct = 0;        // 1: K1 gathers the quaternions directly (thin tiles), 0: 2-D LDS tiles
  EdgePlanes dir;             // directed entries (rows = owned cameras)
  DevBuf<uint32_t> row_ptr, col;
  uint32_t G = 16;
  DevBuf<double2> h0, h1, h2, h3;
  DevBuf<double> h4;
  std::vector<uint32_t> h_cost_eid;  // host copies for weight re-upload

  // cameras
  DevBuf<double> x, x_trial, aa_io, active, scale, gD, Mblk, Minv, Lam, Tinv, b, D6;
  DevBuf<double2> q, q_trial;
  DevBuf<double> xcg, r, z, p, Ap, s_dir, part_g2, part_d2;
  DevBuf<double> w_gather;   // sharded single-reduction PCG: per rank [slice of A u | delta partials of its rows] (run_pcg2)
  uint32_t w_tail = 0;
  DevBuf<Cg2Scalars> cg2sc;
  // two-level preconditioner (kernels.hpp, k_coarse_*): aggregates wanted (0 = off, decided at create) / in use for the current LM step
  uint32_t coarse_want = 0, coarse_n = 0, coarse_chunk = 0;
  bool coarse_adaptive = false;     // use it only once a block-Jacobi PCG solve of the run has needed more than 150 iterations
  DevBuf<double> coarseA, coarseAinv, coarse_rc, coarse_xc, coarse_scale, coarse_part;
  std::vector<double> h_coarse, h_coarse_inv;
  void* pin = nullptr;              // 256 B of pinned host memory: staging for the small read-backs of the solve loop (read_back)
  DevBuf<double> denseA, denseL, dense_x;
  hipGraphExec_t dense_graph = nullptr;   // zero + assemble + blocked Cholesky + solve, captured once
  bool dense_graph_lap = false;           // form of the blocks the captured assemble kernel expects
  int nb_mv = 1, mv_reps = 1;
  uint32_t n_components = 1;  // connected components of the view graph (1 when sharded: a rank sees only its own edges)
  DevBuf<double> part_a, part_b, part_cost, part_cam, scal;
  DevBuf<CgScalars> cgsc;
  int nb_cam = 1, nb_cost = 1;

  // loss
  DevLoss h_loss{};
  DevBuf<DevLoss> d_loss;
  DevBuf<double> tables[3];
  gsfm_loss_callback cb = nullptr;
  void* cb_user = nullptr;
  DevBuf<double> rho_ext, s_ext, w_orig;
  std::vector<double> h_s, h_rho;

  // Column-sorted layout of the directed entries for large graphs without locality (colsort_kernels.hpp): when active it IS the order of
  // every per-entry plane (dir.*, col, h0..h2), and K2c / K3c replace the row-major K2 / K3
  struct ColSort {
    bool active = false;
    uint32_t nch = 0, n_wg = 0;
    size_t n_pos = 0;
    DevBuf<ColWg> wg;
    DevBuf<uint2> meta;
    DevBuf<uint32_t> kcol;
    DevBuf<uint16_t> kcnt;
    uint32_t cbits = 0, cmax = 0;
    DevBuf<double> part;      // 9 planes of [n_wg * RB] (K2c; K3c uses the first three)
    ColLayoutDev dev() const { return ColLayoutDev{wg.p, meta.p, kcol.p, kcnt.p, cbits, cmax, n_wg, nch}; }
  } cs;

  // sigma consensus (gsfm_rot_solve_sigma_consensus): the weights are computed inside the first cost sweep / linearisation of a solve
  SigmaDev sigma{};
  bool sigma_pending_cost = false, sigma_pending_lin = false;
  DevBuf<double> sigma_table, sigma_sum;   // nu = 3 table; [0] = sum |w - w_old| over this rank's cost edges

  bool have_lin = false;
  int graph_launches = 0;
  int n_collectives = 0, n_pcg_collectives = 0, n_pcg_launched = 0;   // issued (or replayed from a graph) since the solve started
  std::vector<double> trace;
  EventTimer timer;
};

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
  const int f = P->functor, w = P->wmode, l = loss_mode(P);
#define GSFM_CASE3(F, W, L) if (f == F && w == W && l == L) { Launcher<F, W, L>::go(args, grid, P->stream); return 0; }
#define GSFM_CASE(F, W) GSFM_CASE3(F, W, LM_PROGRAM) GSFM_CASE3(F, W, LM_SIMPLE) GSFM_CASE3(F, W, LM_MAGSAC)
  GSFM_CASE(F_AA, W_NONE) GSFM_CASE(F_AA, W_SCALAR) GSFM_CASE(F_AA, W_MATRIX)
  GSFM_CASE(F_QCOS, W_NONE) GSFM_CASE(F_QNORM, W_NONE) GSFM_CASE(F_RFNORM, W_NONE)
#undef GSFM_CASE
#undef GSFM_CASE3
  return 1;
}
template <int F, int W, int L> struct CostLauncher {
  static void go(const CostArgs& a, int grid, hipStream_t s) {
    const bool full = a.s_only || a.rho_ext || a.srho_out || a.rho12_out || a.rho1_out || a.r_out || a.sigma.on;
    if (a.direct) {
      if (full) hipLaunchKernelGGL((k_cost_direct<F, W, L, true>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
      else hipLaunchKernelGGL((k_cost_direct<F, W, L, false>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
      return;
    }
    if (full) hipLaunchKernelGGL((k_cost<F, W, L, true>), dim3(grid), dim3(GSFM_TILE_THREADS), 0, s, a);
    else hipLaunchKernelGGL((k_cost<F, W, L, false>), dim3(grid), dim3(GSFM_TILE_THREADS), 0, s, a);
  }
};
// K2: GSFM_K2_FAST=0 switches the fast path (losses with rho'' <= 0) off, for A/B measurements.  Read at every launch.
bool k2_fast_enabled() {
  const char* e = getenv("GSFM_K2_FAST");
  return !(e && *e && atoi(e) <= 0);
}
template <int F, int W, int L> struct LinLauncher {
  static void go(const LinArgs& a, int grid, hipStream_t s) {
    if constexpr (F == F_AA || F == F_QCOS) {   // functors of R_j R_i^T only: the Laplacian form exists (lin_rows)
      if (a.lap) {
        if constexpr (L != LM_PROGRAM) {
          // (a host-callback loss may have rho'' > 0: the general path applies the Corrector in full)
          if (!a.rho_ext && k2_fast_enabled()) hipLaunchKernelGGL((k_lin_fast<F, W, L>), dim3(grid), dim3(GSFM_BLOCK), 0, s, a);
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
        if (!a.lin.rho_ext && k2_fast_enabled()) hipLaunchKernelGGL((k_lin_col<F, W, L, true>), g, b, 0, s, a);
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
// pageable memory costs 22 us per read on this platform, into pinned memory 14 us (tools/bench_sync.hip), which is what small graphs feel.
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
  P->h_loss = L;
  if (!P->d_loss.p && P->d_loss.alloc(1) != hipSuccess) return fail(GSFM_ERR_HIP, "alloc loss");
  HIPCHK(hipMemcpy(P->d_loss.p, &P->h_loss, sizeof(DevLoss), hipMemcpyHostToDevice));
  return 0;
}

// ---- collectives --------------------------------------------------------------------------
int all_gather(gsfm_rot_problem* P, double* buf, size_t count_per_rank) {
  if (!P->sharded) return 0;
  P->n_collectives++;
  if (P->shard.all_gather(P->shard.ctx, buf, count_per_rank, (void*)P->stream) != 0) return fail(GSFM_ERR_COMM, "all_gather callback failed");
  return 0;
}
int all_reduce(gsfm_rot_problem* P, double* buf, size_t count) {
  if (!P->sharded) return 0;
  P->n_collectives++;
  if (P->shard.all_reduce_sum(P->shard.ctx, buf, count, (void*)P->stream) != 0) return fail(GSFM_ERR_COMM, "all_reduce callback failed");
  return 0;
}

// ---- launches -----------------------------------------------------------------------------
enum { SC_COST = 0, SC_GMAX = 1, SC_STEP = 2 /* ..6 */, SC_XNORM2 = 7, SC_TRIAL = 8, SC_DENSE_INFO = 15 /* an int: status of the Cholesky factorisation */, SC_N = 16 };
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

int launch_lin(gsfm_rot_problem* P, const double2* q);

// K1: cost at quaternion cache q -> scal[slot] (all-reduced when sharded)
int launch_cost(gsfm_rot_problem* P, const double2* q, int slot, const CostOutputs& out = CostOutputs()) {
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
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cost.p, P->nb_cost, P->scal.p + slot);
  if (sig && !P->cb) hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cost.p + P->nb_cost, P->nb_cost, P->sigma_sum.p);
  return all_reduce(P, P->scal.p + slot, 1);
}

// K2: linearise at q -> gD (all-gathered), H blocks
int launch_lin(gsfm_rot_problem* P, const double2* q) {
  if (P->cb) { if (int st = refresh_external_rho(P, q)) return st; }
  LinArgs a{};
  a.n_rows = P->n_rows; a.row_base = P->own_begin; a.G = P->G; a.row_ptr = P->row_ptr.p; a.col = P->col.p; a.eid = P->dir.eid.p;
  a.qr0 = P->dir.qr0.p; a.qr1 = P->dir.qr1.p; a.w0 = P->dir.w0.p; a.w1 = P->dir.w1.p; a.w2 = P->dir.w2.p; a.ws = P->dir.ws.p; a.ws_rw = P->dir.ws.p;
  a.q = q; a.loss = P->d_loss.p; a.rho_ext = P->cb ? P->rho_ext.p : nullptr;
  if (P->sigma_pending_lin) { a.sigma = P->sigma; a.sigma.on = 1; P->sigma_pending_lin = false; }
  if (!P->lap && !P->h3.p && (P->h3.alloc(P->dir.n) != hipSuccess || P->h4.alloc(P->dir.n) != hipSuccess)) return fail(GSFM_ERR_HIP, "allocating the general normal-equation blocks failed");
  a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.h3 = P->h3.p; a.h4 = P->h4.p; a.gD = P->gD.p; a.lap = P->lap;
  P->lin_is_lap = P->lap; P->q_lin = q;
  const int tk = P->timer.begin(T_LIN);
  if (P->cs.active) {
    ColLinArgs ca{};
    ca.lin = a; ca.L = P->cs.dev(); ca.part = P->cs.part.p;
    if (dispatch<ColLinArgs, ColLinLauncher>(P, ca, (int)P->cs.n_wg)) return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
    hipLaunchKernelGGL(k_lin_col_finish, dim3(grid_for(P->n_rows)), dim3(GSFM_BLOCK), 0, P->stream, P->n_rows, P->own_begin, P->cs.nch, P->cs.n_wg, (const double*)P->cs.part.p, P->gD.p);
  } else if (dispatch<LinArgs, LinLauncher>(P, a, grid_for((size_t)P->n_rows * P->G))) return fail(GSFM_ERR_UNSUPPORTED, "no kernel for this error type");
  P->timer.end(tk);
  P->have_lin = true;
  return all_gather(P, P->gD.p, (size_t)P->shard.slice_width * 9);
}

void launch_prep(gsfm_rot_problem* P, const gsfm_rot_options& o, double radius, bool init_scale) {
  PrepArgs a{};
  a.n = P->n_cams; a.param_dim = P->param_dim; a.x = P->x.p; a.gD = P->gD.p; a.scale = P->scale.p;
  a.init_scale = init_scale; a.jacobi_scaling = o.jacobi_scaling; a.radius = radius; a.min_diag = o.min_lm_diagonal; a.max_diag = o.max_lm_diagonal;
  a.Mblk = P->Mblk.p; a.Minv = P->Minv.p; a.Lam = P->Lam.p; a.Tinv = P->Tinv.p; a.b = P->b.p; a.gmax_partials = P->part_cam.p;
  hipLaunchKernelGGL(k_cam_prep, dim3(P->nb_cam), dim3(GSFM_BLOCK), 0, P->stream, a);
  hipLaunchKernelGGL(k_max_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cam.p, P->nb_cam, P->scal.p + SC_GMAX);
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
    hipLaunchKernelGGL(k_mv_col, dim3(c.n_wg), dim3(GSFM_COL_RB), 0, P->stream, m);
    ColFinishArgs f{};
    f.n_rows = P->n_rows; f.row_base = P->own_begin; f.nch = c.nch; f.n_wg = c.n_wg; f.part = c.part.p; f.Mblk = Mblk; f.p = p; f.q = P->q_lin; f.y = y; f.done = done;
    if (dot_part && !P->sharded) { f.dot_part = dot_part; *dot_done = true; }   // (one GPU: rows = cameras, the finish grid is the camera kernels' grid)
    hipLaunchKernelGGL(k_mv_col_finish, dim3(grid_for(P->n_rows)), dim3(GSFM_BLOCK), 0, P->stream, f);
    return all_gather(P, y, (size_t)P->shard.slice_width * 3);
  }
  if (P->lin_is_lap) hipLaunchKernelGGL(k_matvec<true>, dim3(grid_for((size_t)P->n_rows * P->G)), dim3(GSFM_BLOCK), 0, P->stream, a);
  else hipLaunchKernelGGL(k_matvec<false>, dim3(grid_for((size_t)P->n_rows * P->G)), dim3(GSFM_BLOCK), 0, P->stream, a);
  return all_gather(P, y, (size_t)P->shard.slice_width * 3);
}

// Inverse of a symmetric positive definite n x n matrix (row-major) through its Cholesky factor; false if a pivot is not positive.
bool spd_inverse(std::vector<double>& A, size_t n, std::vector<double>& inv) {
  for (size_t j = 0; j < n; ++j) {          // A <- L (lower), column by column
    double d = A[j * n + j];
    for (size_t k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double l = std::sqrt(d);
    A[j * n + j] = l;
    for (size_t i = j + 1; i < n; ++i) {
      double v = A[i * n + j];
      for (size_t k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = v / l;
    }
  }
  std::vector<double> Li(n * n, 0.0);       // L^-1 (lower), row by row: row_i = (e_i - sum_{k<i} L[i][k] row_k) / L[i][i]  (contiguous rows only)
  for (size_t i = 0; i < n; ++i) {
    double* ri = &Li[i * n];
    ri[i] = 1.0;
    for (size_t k = 0; k < i; ++k) {
      const double l = A[i * n + k];
      const double* rk = &Li[k * n];
      for (size_t c = 0; c <= k; ++c) ri[c] -= l * rk[c];
    }
    const double d = 1.0 / A[i * n + i];
    for (size_t c = 0; c <= i; ++c) ri[c] *= d;
  }
  inv.assign(n * n, 0.0);                   // A^-1 = L^-T L^-1 as a sum of rank-one updates of the lower triangle, then mirrored
  for (size_t k = 0; k < n; ++k) {
    const double* rk = &Li[k * n];
    for (size_t i = 0; i <= k; ++i) {
      const double a = rk[i];
      double* oi = &inv[i * n];
      for (size_t j = 0; j <= i; ++j) oi[j] += a * rk[j];
    }
  }
  for (size_t i = 0; i < n; ++i) for (size_t j = 0; j < i; ++j) inv[j * n + i] = inv[i * n + j];
  return true;
}

// Coarse matrix of the two-level preconditioner for the current linearisation and damping: assembled on the device, inverted on the host
// (3 n_agg <= 384 unknowns).  Leaves P->coarse_n = 0 (plain block-Jacobi for this step) if the matrix is not positive definite.
int coarse_build(gsfm_rot_problem* P, bool pcg_struggles) {
  P->coarse_n = 0;
  if (!P->coarse_want || !P->lin_is_lap || (P->coarse_adaptive && !pcg_struggles)) return 0;
  const uint32_t na = P->coarse_want, nc = 3 * na;   // (buffers: allocated at creation, before the ranks of a sharded problem vote)
  const int tk = P->timer.begin(T_CG);
  HIPCHK(hipMemsetAsync(P->coarseA.p, 0, 8 * (size_t)nc * nc, P->stream));
  CoarseAsmArgs a{};
  a.n_rows = P->n_rows; a.G = P->G; a.n_agg = na; a.chunk = P->coarse_chunk; a.row_ptr = P->row_ptr.p; a.col = P->col.p;
  a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.Mblk = P->Mblk.p; a.q = P->q_lin; a.Ac = P->coarseA.p;
  a.row_base = P->own_begin; a.scale = P->coarse_scale.p;
  {  // the fixed-point scale must be the same on every rank: the largest diagonal entry over ALL cameras (Mblk is complete everywhere)
    const uint32_t nb = (uint32_t)grid_for(P->n_cams);
    hipLaunchKernelGGL(k_coarse_scale, dim3(nb), dim3(GSFM_BLOCK), 0, P->stream, (const double*)P->Mblk.p, P->n_cams, 0u, P->part_a.p, P->coarse_scale.p, 0);
    hipLaunchKernelGGL(k_coarse_scale, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, (const double*)P->Mblk.p, nb, 0u, P->part_a.p, P->coarse_scale.p, 1);
  }
  if (P->n_rows) hipLaunchKernelGGL(k_coarse_assemble, dim3(grid_for((size_t)P->n_rows * P->G)), dim3(GSFM_BLOCK), 0, P->stream, a);
  hipLaunchKernelGGL(k_coarse_unscale, dim3(grid_for((size_t)nc * nc)), dim3(GSFM_BLOCK), 0, P->stream, P->coarseA.p, (size_t)nc * nc, (const double*)P->coarse_scale.p);
  P->timer.end(tk);
  // sharded: every rank summed the rows it owns; the vector side (restriction, coarse solve, prolongation) then runs replicated on the
  // replicated PCG vectors like every other O(N) step, so this all-reduce per LM step is the only collective the preconditioner adds
  if (int st = all_reduce(P, P->coarseA.p, (size_t)nc * nc)) return st;
  P->h_coarse.resize((size_t)nc * nc);
  HIPCHK(hipMemcpyAsync(P->h_coarse.data(), P->coarseA.p, 8 * (size_t)nc * nc, hipMemcpyDeviceToHost, P->stream));
  const double t_a = now_ms();
  if (int st = sync_check(P, "coarse matrix")) return st;
  const double t_b = now_ms();
  auto& A = P->h_coarse;
  for (size_t i = 0; i < nc; ++i) for (size_t j = 0; j < i; ++j) { const double v = 0.5 * (A[i * nc + j] + A[j * nc + i]); A[i * nc + j] = A[j * nc + i] = v; }
  for (size_t i = 0; i < nc; ++i) if (A[i * nc + i] == 0.0) A[i * nc + i] = 1.0;   // an aggregate of cameras without edges: decoupled, its correction stays zero
  if (!spd_inverse(A, nc, P->h_coarse_inv)) return 0;
  if (getenv("GSFM_COARSE_TIMING")) fprintf(stderr, "gsfm coarse: assemble + download (wait) %.2f ms, host inverse of %u unknowns %.2f ms\n", t_b - t_a, nc, now_ms() - t_b);
  HIPCHK(hipMemcpyAsync(P->coarseAinv.p, P->h_coarse_inv.data(), 8 * (size_t)nc * nc, hipMemcpyHostToDevice, P->stream));
  P->coarse_n = na;
  return 0;
}

// May a chunk of PCG iterations of a SHARDED problem be captured into a hipGraph together with its collectives?  Only if the
// communicator's callbacks do nothing but enqueue work on the solver's stream (GSFM_SHARD_CAPTURABLE: the native RCCL communicator).
// pcg_hip_graph = 1 (default) then captures; 2 is the old explicit opt-in and means the same; GSFM_PCG_GRAPH_COLLECTIVES=0 switches the
// capture of collectives off (plain launches), e.g. to isolate a communicator problem.
bool graph_collectives_ok(const gsfm_rot_problem* P, const gsfm_rot_options& o) {
  if (!(P->shard.flags & GSFM_SHARD_CAPTURABLE) || o.pcg_hip_graph < 1) return false;
  const char* e = getenv("GSFM_PCG_GRAPH_COLLECTIVES");
  return !(e && *e && atoi(e) == 0);
}

// block-Jacobi PCG on (J^T J + Lambda) eta = -g; returns iterations
int run_pcg(gsfm_rot_problem* P, const gsfm_rot_options& o, int* iters_out, double* rel_out) {
  CgArgs a{};
  a.n = P->n_cams; a.nb = P->nb_cam; a.par = 0; a.tol = o.cg_relative_tolerance; a.max_iters = o.max_cg_iterations; a.stall_limit = o.cg_stall_iterations;
  a.Minv = P->Minv.p; a.b = P->b.p; a.xcg = P->xcg.p; a.r = P->r.p; a.z = P->z.p; a.p = P->p.p; a.Ap = P->Ap.p;
  a.part_a = P->part_a.p; a.part_b = P->part_b.p; a.sc = P->cgsc.p;
  a.q = P->q_lin; a.u = P->lin_is_lap ? P->u_rot.p : nullptr;
  a.coarse_n = P->coarse_n; a.coarse_chunk = P->coarse_chunk; a.xc = P->coarse_xc.p; a.active = P->active.p;
  // aggregates at least as wide as a block of the camera kernels (always, unless forced narrower): the restriction rides along in k_cg_update
  const bool fused_restrict = P->coarse_n && P->coarse_chunk >= GSFM_BLOCK;
  a.rc_part = fused_restrict ? P->coarse_part.p : nullptr;
  CoarseArgs ca{};
  ca.n = P->n_cams; ca.n_agg = P->coarse_n; ca.chunk = P->coarse_chunk; ca.q = P->q_lin; ca.r = P->r.p; ca.rc = P->coarse_rc.p; ca.Ainv = P->coarseAinv.p;
  ca.xc = P->coarse_xc.p; ca.done = nullptr; ca.active = P->active.p; ca.rc_part = nullptr; ca.nb = (uint32_t)P->nb_cam;
  const dim3 g(P->nb_cam), blk(GSFM_BLOCK);
  const int tk0 = P->timer.begin(T_CG);
  hipLaunchKernelGGL(k_cg_init, g, blk, 0, P->stream, a);
  hipLaunchKernelGGL(k_cg_init_fin, dim3(1), blk, 0, P->stream, a);
  if (a.coarse_n) {   // z_0 = Minv r_0 + P Ac^-1 P^T r_0
    hipLaunchKernelGGL(k_coarse_restrict, dim3(a.coarse_n), blk, 0, P->stream, ca);
    hipLaunchKernelGGL(k_coarse_apply, dim3(1), dim3(1024), 0, P->stream, ca);
    hipLaunchKernelGGL(k_cg_init_coarse, g, blk, 0, P->stream, a);
    hipLaunchKernelGGL(k_cg_init_coarse_fin, dim3(1), dim3(1), 0, P->stream, a);
  }
  ca.done = &P->cgsc.p->done; ca.rc_part = a.rc_part;
  P->timer.end(tk0);
  CgScalars h{};
  const int chunk = std::max(1, o.cg_check_interval);
  auto enqueue_chunk = [&]() -> int {  // `chunk` iterations; leaves a.par where it found it when chunk is even
    for (int c = 0; c < chunk; ++c) {
      bool dotted = false;
      if (int st = launch_matvec(P, P->Mblk.p, P->p.p, P->Ap.p, &P->cgsc.p->done, a.part_a, &dotted)) return st;
      if (P->sharded) P->n_pcg_collectives++;
      if (!dotted) hipLaunchKernelGGL(k_cg_dot, g, blk, 0, P->stream, a);
      hipLaunchKernelGGL(k_cg_update, g, blk, 0, P->stream, a);
      if (a.coarse_n) {
        if (!fused_restrict) hipLaunchKernelGGL(k_coarse_restrict, dim3(a.coarse_n), blk, 0, P->stream, ca);
        hipLaunchKernelGGL(k_coarse_apply, dim3(1), dim3(1024), 0, P->stream, ca);
      }
      hipLaunchKernelGGL(k_cg_pupdate, g, blk, 0, P->stream, a);
      a.par ^= 1;
    }
    return 0;
  };
  // The chunk between two host checks as one hipGraph launch: 4 * chunk dependent kernels whose arguments never change.
  auto& G = P->pcg_graph;
  bool graph = o.pcg_hip_graph && (!P->sharded || graph_collectives_ok(P, o)) && chunk % 2 == 0 && !G.unusable;
  if (graph && (!G.exec || G.tol != a.tol || G.max_iters != a.max_iters || G.stall != a.stall_limit || G.chunk != chunk || G.lap != P->lin_is_lap || G.coarse != a.coarse_n)) {
    G.reset();
    hipGraph_t captured = nullptr;
    if (hipStreamBeginCapture(P->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      const int c0 = P->n_collectives, p0 = P->n_pcg_collectives;
      const int st = enqueue_chunk();
      const hipError_t e = hipStreamEndCapture(P->stream, &captured);
      G.collectives = P->n_collectives - c0;            // captured, not issued: counted per replay below
      P->n_collectives = c0; P->n_pcg_collectives = p0;
      if (st == 0 && e == hipSuccess && captured && hipGraphInstantiate(&G.exec, captured, nullptr, nullptr, 0) == hipSuccess) {
        G.tol = a.tol; G.max_iters = a.max_iters; G.stall = a.stall_limit; G.chunk = chunk; G.lap = P->lin_is_lap; G.coarse = a.coarse_n;
      } else { G.exec = nullptr; }
      if (captured) (void)hipGraphDestroy(captured);
    }
    if (!G.exec) { (void)hipGetLastError(); G.unusable = true; graph = false; }  // e.g. a stream that cannot be captured: plain launches
  }
  int launched = 0, chunks = 1;
  while (true) {
    const int tk = P->timer.begin(T_CG);
    for (int c = 0; c < chunks; ++c) {
      if (graph) { HIPCHK(hipGraphLaunch(G.exec, P->stream)); P->graph_launches++; P->n_collectives += G.collectives; P->n_pcg_collectives += G.collectives; }
      else if (int st = enqueue_chunk()) return st;
      launched += chunk; P->n_pcg_launched += chunk;
    }
    P->timer.end(tk);
    if (int st = read_back(P, &h, P->cgsc.p, sizeof(h), "pcg")) return st;
    if (h.done || launched >= o.max_cg_iterations + chunk) break;
    // Fewer host round trips: extrapolate the average convergence factor so far to the tolerance and enqueue that many
    // chunks before looking again (kernels past convergence return at their first instruction, so overshoot is cheap).
    chunks = 1;
    if (h.iters > 0 && h.last_rel > 0.0 && h.last_rel < 1.0 && a.tol > 0.0 && a.tol < h.last_rel) {
      const double per_iter = std::log(h.last_rel) / h.iters;
      const double remaining = std::log(a.tol / h.last_rel) / per_iter;
      chunks = (int)std::min(8.0, std::max(1.0, std::ceil(remaining / chunk)));
    }
    chunks = std::min(chunks, std::max(1, (o.max_cg_iterations + chunk - launched + chunk - 1) / chunk));
  }
  *iters_out = h.iters; *rel_out = h.last_rel;
  return 0;
}

// single-reduction PCG (Chronopoulos-Gear): 2 kernels per iteration (3 + one all-gather when sharded); the launch-latency regime's
// default (see use_single_reduction).  The chunk between two host checks replays as one hipGraph, like run_pcg's.
int run_pcg2(gsfm_rot_problem* P, const gsfm_rot_options& o, int* iters_out, double* rel_out) {
  Cg2Args c{};
  const int nb_mv = P->nb_mv, reps = P->mv_reps;
  c.n = P->n_cams; c.nb_cam = P->nb_cam; c.n_part_d = (P->sharded || P->cs.active) ? P->nb_cam : nb_mv; c.par = 0; c.first = 1; c.tol = o.cg_relative_tolerance; c.max_iters = o.max_cg_iterations;
  c.Minv = P->Minv.p; c.b = P->b.p; c.x = P->xcg.p; c.r = P->r.p; c.u = P->z.p; c.w = P->Ap.p; c.p = P->p.p; c.s = P->s_dir.p;
  c.part_g = P->part_g2.p; c.part_d = P->part_d2.p; c.sc = P->cg2sc.p;
  c.q = P->q_lin; c.urot = P->lin_is_lap ? P->u_rot.p : nullptr;
  // Sharded: A u and the delta partials of a rank's rows leave in ONE all-gather (slot = slice of w, then the partials); the mat-vec kernels
  // address y by global camera index, so they get the slot's base shifted back by the rank's first camera.
  double* w_own = P->Ap.p;          // what the mat-vec writes through (indexed 3 * global camera)
  double* dots_own = P->part_d2.p;  // where its delta partials go
  if (P->sharded) {
    const uint32_t slice = P->shard.slice_width, tail = P->w_tail, stride = 3 * slice + tail;
    if (!P->w_gather.p) return fail(GSFM_ERR_HIP, "the all-gather buffer of the sharded PCG was not allocated");   // (create allocates it, before the ranks agree)
    c.w = P->w_gather.p; c.w_stride = stride; c.w_slice = slice; c.w_tail = tail; c.n_part_d = (int)(tail * P->shard.world_size);
    double* slot = P->w_gather.p + (size_t)P->shard.rank * stride;
    w_own = slot - 3 * (size_t)P->own_begin; dots_own = slot + 3 * (size_t)slice;
  }
  MatvecCgArgs m{};
  m.mv.n_rows = P->n_rows; m.mv.row_base = P->own_begin; m.mv.G = P->G; m.mv.row_ptr = P->row_ptr.p; m.mv.col = P->col.p;
  m.mv.h0 = P->h0.p; m.mv.h1 = P->h1.p; m.mv.h2 = P->h2.p; m.mv.h3 = P->h3.p; m.mv.h4 = P->h4.p; m.mv.Mblk = P->Mblk.p;
  m.mv.p = P->z.p; m.mv.y = w_own; m.mv.done = nullptr; m.mv.q = P->q_lin; m.mv.u = P->u_rot.p; m.with_dots = 1; m.reps = (uint32_t)reps;
  const dim3 gcam(P->nb_cam), gmv(nb_mv), blk(GSFM_BLOCK);
  const int tk0 = P->timer.begin(T_CG);
  hipLaunchKernelGGL(k_cg2_init, gcam, blk, 0, P->stream, c);
  P->timer.end(tk0);
  Cg2Scalars h{};
  const int chunk = std::max(1, o.cg_check_interval);
  // One iteration = mat-vec (+ delta partials), vector step.  `first` / `par` are by-value kernel arguments: a captured chunk must
  // start at par == 0, first == 0, so the very first iteration is launched plainly and chunks have even length.
  auto enqueue_iter = [&]() -> int {
    m.cg = c; m.cg.part_d = dots_own;
    if (P->cs.active) {   // column-sorted layout: K3c with the same entry decision, delta partials from its finishing kernel (one per camera block)
      auto& L = P->cs;
      ColMatvecCgArgs cm{};
      cm.mv.L = L.dev(); cm.mv.b0 = P->h0.p; cm.mv.b1 = P->h1.p; cm.mv.b2 = P->h2.p; cm.mv.u = P->u_rot.p; cm.mv.part = L.part.p; cm.mv.done = nullptr; cm.cg = c;
      hipLaunchKernelGGL(k_mv_col_cg, dim3(L.n_wg), dim3(GSFM_COL_RB), 0, P->stream, cm);
      ColFinishArgs f{};
      f.n_rows = P->n_rows; f.row_base = P->own_begin; f.nch = L.nch; f.n_wg = L.n_wg; f.part = L.part.p; f.Mblk = P->Mblk.p; f.p = P->z.p; f.q = P->q_lin; f.y = w_own;
      f.done = &P->cg2sc.p->done; f.dot_part = dots_own;
      hipLaunchKernelGGL(k_mv_col_finish, dim3(grid_for(P->n_rows)), dim3(GSFM_BLOCK), 0, P->stream, f);
    }
    else if (P->lin_is_lap) hipLaunchKernelGGL(k_matvec_cg<true>, gmv, blk, 0, P->stream, m);
    else hipLaunchKernelGGL(k_matvec_cg<false>, gmv, blk, 0, P->stream, m);
    if (P->sharded) {
      if (int st = all_gather(P, P->w_gather.p, (size_t)c.w_stride)) return st;
      P->n_pcg_collectives++;
    }
    hipLaunchKernelGGL(k_cg2_step, gcam, blk, 0, P->stream, c);
    c.par ^= 1; c.first = 0;
    return 0;
  };
  int launched = 0;
  {  // iterations 0 and 1 (first = 1, then par = 1): plain launches; afterwards par == 0 at every chunk start
    const int tk = P->timer.begin(T_CG);
    for (int k = 0; k < 2; ++k) { if (int st = enqueue_iter()) return st; ++launched; P->n_pcg_launched++; }
    P->timer.end(tk);
  }
  auto& G = P->pcg2_graph;
  bool graph = o.pcg_hip_graph && (!P->sharded || graph_collectives_ok(P, o)) && chunk % 2 == 0 && !G.unusable && !P->pcg_graph.unusable;
  if (graph && (!G.exec || G.tol != c.tol || G.max_iters != c.max_iters || G.chunk != chunk || G.lap != P->lin_is_lap)) {
    G.reset();
    hipGraph_t captured = nullptr;
    if (hipStreamBeginCapture(P->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      int st = 0;
      const int c0 = P->n_collectives, p0 = P->n_pcg_collectives;
      for (int k = 0; k < chunk && st == 0; ++k) st = enqueue_iter();
      const hipError_t e = hipStreamEndCapture(P->stream, &captured);
      G.collectives = P->n_collectives - c0;
      P->n_collectives = c0; P->n_pcg_collectives = p0;
      if (st == 0 && e == hipSuccess && captured && hipGraphInstantiate(&G.exec, captured, nullptr, nullptr, 0) == hipSuccess) {
        G.tol = c.tol; G.max_iters = c.max_iters; G.chunk = chunk; G.lap = P->lin_is_lap;
      } else { G.exec = nullptr; }
      if (captured) (void)hipGraphDestroy(captured);
    }
    if (!G.exec) { (void)hipGetLastError(); G.unusable = true; graph = false; }
  }
  int chunks = 1;
  while (true) {
    const int tk = P->timer.begin(T_CG);
    for (int cc = 0; cc < chunks; ++cc) {
      if (graph) { HIPCHK(hipGraphLaunch(G.exec, P->stream)); P->graph_launches++; P->n_collectives += G.collectives; P->n_pcg_collectives += G.collectives; }
      else { for (int k = 0; k < chunk; ++k) if (int st = enqueue_iter()) return st; }
      launched += chunk; P->n_pcg_launched += chunk;
    }
    P->timer.end(tk);
    if (int st = read_back(P, &h, P->cg2sc.p, sizeof(h), "pcg")) return st;
    if (h.done || launched >= o.max_cg_iterations + chunk + 2) break;
    chunks = 1;   // same look-ahead as run_pcg: extrapolate the convergence factor, enqueue that many chunks before looking again
    if (h.iters > 0 && h.last_rel > 0.0 && h.last_rel < 1.0 && c.tol > 0.0 && c.tol < h.last_rel) {
      const double per_iter = std::log(h.last_rel) / h.iters;
      const double remaining = std::log(c.tol / h.last_rel) / per_iter;
      chunks = (int)std::min(8.0, std::max(1.0, std::ceil(remaining / chunk)));
    }
  }
  *iters_out = h.iters; *rel_out = h.last_rel;
  return 0;
}

// Which PCG: the single-reduction variant halves the dependent launches per iteration (2 instead of 4), which is what bounds small
// graphs (tools/small_graph_pcg.py: 22 -> 14 -> 8 us per iteration at C2 size); from ~1M directed entries on the kernels dominate
// and the textbook recurrence is kept (its residual is the recursively updated one of the reference description).
bool use_single_reduction(const gsfm_rot_problem* P, const gsfm_rot_options& o) {
  if (o.pcg_single_reduction >= 0) return o.pcg_single_reduction != 0;
  if (o.cg_stall_iterations > 0) return false;   // stagnation detection lives in the textbook variant's scalar kernel
  // (On the column-sorted layout the variant exists too -- k_mv_col_cg, one vector kernel instead of two -- and measures the same as the
  // textbook recurrence at C5: 30.69 against 30.63 ms per solve, 129 iterations both; the entry decision of its mat-vec costs what the
  // saved launch gains.  tools/r03_pcg_variants.py)
  // Sharded: always -- there every launch counts (the per-rank kernels shrink with the rank count, the launches do not), and the variant is
  // 3 kernels + 1 collective per iteration (mat-vec, finish, [all-gather of A u with the delta partials in its tail], vector step) against
  // 5 + 1 for the textbook recurrence; except at tolerances below 1e-13 (disconnected graphs, lm_solve), where the recursively updated
  // residual of the textbook form is the safer one.
  if (P->sharded) return o.cg_relative_tolerance >= 1e-13 && P->n_components <= 1;
  return P->dir.n <= (size_t)2000000;
}
bool single_reduction_possible(const gsfm_rot_problem*) { return true; }

// Exact step for small graphs: dense Cholesky of (J^T J + Lambda) in the left-tangent space (dense_kernels.hpp).  Enqueues only: the
// factorisation's status lands in the scalar block (SC_DENSE_INFO) and is read together with the trial cost, one host synchronisation
// later; a non-positive pivot makes the caller solve the step again by PCG.  *used = false if nothing was enqueued (size, memory).
int run_dense(gsfm_rot_problem* P, bool* used) {
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
  // ~68 block columns in the benchmark (2048: 1.06 vs 1.10 ms), so the switch is at 64 block columns (682 cameras).  GSFM_CHOL_SPLIT_T overrides the switch point (block columns; A/B measurements).
  static const uint32_t split_T = [] { const char* e = getenv("GSFM_CHOL_SPLIT_T"); const int v = e && *e ? atoi(e) : GSFM_CHOL_SPLIT_DEFAULT; return (uint32_t)std::max(0, std::min(v, GSFM_CHOL_SPLIT_T)); }();
  auto enqueue = [&]() {
    (void)hipMemsetAsync(P->denseA.p, 0, 8 * elems, P->stream);
    int* const info = (int*)(P->scal.p + SC_DENSE_INFO);
    DenseArgs a{};
    a.n_rows = P->n_rows; a.row_ptr = P->row_ptr.p; a.col = P->col.p; a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.h3 = P->h3.p; a.h4 = P->h4.p;
    a.Mblk = P->Mblk.p; a.b = P->b.p; a.A = P->denseA.p; a.n = n; a.T = T; a.q = P->q_lin; a.lap = P->lin_is_lap; a.info_slot = P->scal.p + SC_DENSE_INFO; a.rcg = P->r.p;
    if (P->cs.active) hipLaunchKernelGGL(k_dense_assemble_col, dim3(P->cs.n_wg), dim3(GSFM_BLOCK), 0, P->stream, a, P->cs.dev());
    else hipLaunchKernelGGL(k_dense_assemble, dim3(P->n_rows), dim3(GSFM_BLOCK), 0, P->stream, a);
    if (T <= split_T) {
      for (uint32_t k = 0; k < T; ++k) {
        CholArgs c{P->denseA.p, P->denseL.p, T, k, info};
        const uint64_t m = T - k;
        { const uint32_t nt = getenv("GSFM_CHOL_NT") ? (uint32_t)std::max(1, std::min(3, atoi(getenv("GSFM_CHOL_NT")))) : chol_step_tiles_per_wg((uint32_t)m); const dim3 grid(chol_step_grid((uint32_t)m, nt));
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
    // folds its x into all block rows above it (dense_kernels.hpp).  GSFM_CHOL_BACK_GROUPS=0: the forms it replaced (one workgroup for
    // everything up to 48 block rows, one launch per block row beyond), kept for A/B measurements.
    static const bool grouped = [] { const char* e = getenv("GSFM_CHOL_BACK_GROUPS"); return !(e && atoi(e) == 0); }();
    if (grouped) {
      static const uint32_t GR = [] { const char* e = getenv("GSFM_CHOL_BACK_GROUP"); return e && atoi(e) == 16 ? 16u : 8u; }();   // (Madrid, linear solves per solve: 8 rows per group 35.1 ms, 16: 36.0, the single workgroup it replaces 37.6)
      for (uint32_t k1 = T; k1 > 0;) {
        const uint32_t k0 = k1 > GR ? k1 - GR : 0;
        CholBackGroupArgs b{P->denseL.p, P->dense_x.p, n, T, k0, k1};
        if (GR == 16) {
          hipLaunchKernelGGL(k_chol_back_group<16>, dim3(1), dim3(1024), 0, P->stream, b);
          if (k0) hipLaunchKernelGGL(k_chol_back_update<16>, dim3(k0), dim3(512), 0, P->stream, b);
        } else {
          hipLaunchKernelGGL(k_chol_back_group<8>, dim3(1), dim3(512), 0, P->stream, b);
          if (k0) hipLaunchKernelGGL(k_chol_back_update<8>, dim3(k0), dim3(256), 0, P->stream, b);
        }
        k1 = k0;
      }
      (void)hipMemcpyAsync(P->xcg.p, P->dense_x.p, 8 * (size_t)n, hipMemcpyDeviceToDevice, P->stream);
    } else if (T <= split_T) hipLaunchKernelGGL(k_chol_back<GSFM_CHOL_SPLIT_T>, dim3(1), dim3(1024), 0, P->stream, (const double*)P->denseL.p, n, T, P->xcg.p);
    else {   // one launch per block row, all tiles of the row in parallel; the running right-hand side is block row T of L, x goes to dense_x (padded to T * 32)
      for (uint32_t k = T; k >= 1; --k) {
        CholBackArgs b{P->denseL.p, P->dense_x.p, n, T, k};
        hipLaunchKernelGGL(k_chol_back_step, dim3(k == T ? 1 : k), dim3(64), 0, P->stream, b);
      }
      (void)hipMemcpyAsync(P->xcg.p, P->dense_x.p, 8 * (size_t)n, hipMemcpyDeviceToDevice, P->stream);
    }
    // (exact solve: the PCG residual term of the model decrease is zero -- k_dense_assemble cleared it)
  };
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

int launch_step(gsfm_rot_problem* P) {
  StepArgs a{};
  a.n = P->n_cams; a.param_dim = P->param_dim; a.x = P->x.p; a.active = P->active.p; a.eta = P->xcg.p; a.b = P->b.p; a.rcg = P->r.p;
  a.Lam = P->Lam.p; a.Tinv = P->Tinv.p; a.x_trial = P->x_trial.p; a.q_trial = P->q_trial.p; a.partials = P->part_cam.p;
  hipLaunchKernelGGL(k_cam_step, dim3(P->nb_cam), dim3(GSFM_BLOCK), 0, P->stream, a);
  hipLaunchKernelGGL(k_sum_partials_multi, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cam.p, P->nb_cam, 5, P->scal.p + SC_STEP);
  return 0;
}

int read_scalars(gsfm_rot_problem* P, double* h) {
  return read_back(P, h, P->scal.p, SC_N * sizeof(double), "read scalars");
}

// Reverse Cuthill-McKee style relabelling (plain BFS from a minimum-degree camera of every component, reversed).  The p[col]
// and q[col] gathers of K3/K2 are bound by uncoalesced lane requests; when the neighbours of a camera sit within a few
// hundred indices of each other the lanes of a row share 128-byte lines and the gather becomes free (tools/bench_matvec.hip:
// 346 us -> 235 us at a window of 400, 284 us at 2000, no gain at 20000).  View graphs of real scenes are spatially
// coherent but their ids are arbitrary; a uniformly random graph (the C5 benchmark) has nothing to recover.  The
// relabelling is therefore adopted only if it shrinks the mean |i - j| over the edges by more than half AND brings it
// under 1024 (neighbours within about +-2000); small problems (< 2048 cameras: everything is cache-resident) are left alone.
// GSFM_REORDER=0 disables it, =1 forces adoption.  Returns true when `perm` (external -> internal) must be applied.
template <typename AdjVec>
bool reorder_for_locality(uint32_t n_cams, uint64_t n_edges, const uint32_t* ei, const uint32_t* ej, const std::vector<uint32_t>& ptr,
                          const AdjVec& adj /* neighbour | role << 31 */, std::vector<uint32_t>* perm) {
  perm->clear();
  const char* env = getenv("GSFM_REORDER");
  const int mode = env ? atoi(env) : -1;  // -1 auto, 0 off, 1 force
  if (mode == 0 || (mode < 0 && n_cams < 2048)) return false;
  double before = 0.0;
  for (uint64_t e = 0; e < n_edges; ++e) before += std::fabs((double)ei[e] - (double)ej[e]);
  before /= (double)n_edges;
  if (mode < 0 && before < 256.0) return false;  // already local (a mean index distance of 256 ~ neighbours within +-500)
  std::vector<uint32_t> stamp(n_cams, 0xffffffffu);
  if (mode < 0) {
    // cheap pre-test: in a spatially coherent graph the two-hop neighbourhood of a camera stays small; in a uniformly random
    // one it floods the graph.  32 probes, each capped at n_cams / 8 cameras.
    const uint32_t cap = n_cams / 8;
    int flooded = 0;
    for (uint32_t s = 0; s < 32; ++s) {
      const uint32_t c0 = (uint32_t)(((uint64_t)s * n_cams) / 32);
      uint32_t seen = 0;
      for (uint32_t d = ptr[c0]; d < ptr[c0 + 1] && seen < cap; ++d) {
        const uint32_t c1 = adj[d] & 0x7fffffffu;
        for (uint32_t d2 = ptr[c1]; d2 < ptr[c1 + 1] && seen < cap; ++d2) {
          const uint32_t c2 = adj[d2] & 0x7fffffffu;
          if (stamp[c2] != s) { stamp[c2] = s; ++seen; }
        }
      }
      flooded += seen >= cap;
    }
    if (flooded > 16) return false;
  }
  std::vector<uint32_t> by_degree(n_cams);
  for (uint32_t c = 0; c < n_cams; ++c) by_degree[c] = c;
  std::stable_sort(by_degree.begin(), by_degree.end(), [&](uint32_t a, uint32_t b) { return ptr[a + 1] - ptr[a] < ptr[b + 1] - ptr[b]; });
  std::vector<uint32_t> order;
  order.reserve(n_cams);
  std::vector<uint8_t> seen(n_cams, 0);
  for (uint32_t s0 : by_degree) {
    if (seen[s0]) continue;
    seen[s0] = 1;
    size_t head = order.size();
    order.push_back(s0);
    while (head < order.size()) {
      const uint32_t c = order[head++];
      for (uint32_t d = ptr[c]; d < ptr[c + 1]; ++d) { const uint32_t m = adj[d] & 0x7fffffffu; if (!seen[m]) { seen[m] = 1; order.push_back(m); } }
    }
  }
  std::vector<uint32_t> p(n_cams);
  for (uint32_t k = 0; k < n_cams; ++k) p[order[k]] = n_cams - 1 - k;
  double after = 0.0;
  for (uint64_t e = 0; e < n_edges; ++e) after += std::fabs((double)p[ei[e]] - (double)p[ej[e]]);
  after /= (double)n_edges;
  if (mode < 0 && !(after < 0.5 * before && after < 1024.0)) return false;
  perm->swap(p);
  return true;
}

// connected components of the view graph among the cameras that have at least one edge (union-find with path halving)
uint32_t count_components(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j) {
  std::vector<uint32_t> parent(n_cams);
  std::vector<uint8_t> touched(n_cams, 0);
  for (uint32_t c = 0; c < n_cams; ++c) parent[c] = c;
  auto find = [&](uint32_t v) { while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; } return v; };
  for (uint64_t e = 0; e < n_edges; ++e) {
    const uint32_t a = find(edge_i[e]), b = find(edge_j[e]);
    touched[edge_i[e]] = touched[edge_j[e]] = 1;
    if (a != b) parent[a < b ? b : a] = a < b ? a : b;
  }
  uint32_t comps = 0;
  for (uint32_t c = 0; c < n_cams; ++c) if (touched[c] && find(c) == c) ++comps;
  return comps;
}

// Per-camera host arrays (rotations, gradient, mat-vec operands) enter and leave in the caller's numbering.
const double* to_internal(gsfm_rot_problem* P, const double* ext, int width) {
  if (P->perm.empty()) return ext;
  P->h_cam.resize((size_t)P->n_cams * width);
  for (size_t k = 0; k < P->n_cams; ++k) std::memcpy(&P->h_cam[(size_t)P->perm[k] * width], ext + k * width, 8 * (size_t)width);
  return P->h_cam.data();
}
void to_external(gsfm_rot_problem* P, const double* internal, double* ext, int width) {
  for (size_t k = 0; k < P->n_cams; ++k) std::memcpy(ext + k * width, internal + (size_t)P->perm[k] * width, 8 * (size_t)width);
}

int upload_state(gsfm_rot_problem* P, const double* rot_aa) {
  const size_t N = P->n_cams;
  HIPCHK(hipMemcpyAsync(P->aa_io.p, to_internal(P, rot_aa, 3), 24 * N, hipMemcpyHostToDevice, P->stream));
  if (P->param_dim == 3) { HIPCHK(hipMemcpyAsync(P->x.p, P->aa_io.p, 24 * N, hipMemcpyDeviceToDevice, P->stream)); }
  else {  // estimator.cpp:130-136: angle-axis -> quaternion state
    hipLaunchKernelGGL(k_cam_cache, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->aa_io.p, P->n_cams, 3, (double2*)P->x.p);
  }
  launch_cache(P, P->x.p, P->q.p);
  return 0;
}
int download_state(gsfm_rot_problem* P, double* rot_aa) {
  const size_t N = P->n_cams;
  double* dst = rot_aa;
  if (!P->perm.empty()) { P->h_cam.resize(3 * N); dst = P->h_cam.data(); }
  if (P->param_dim == 3) { HIPCHK(hipMemcpyAsync(dst, P->x.p, 24 * N, hipMemcpyDeviceToHost, P->stream)); }
  else {
    hipLaunchKernelGGL(k_quat_to_aa, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->active.p, P->n_cams, P->aa_io.p);
    HIPCHK(hipMemcpyAsync(dst, P->aa_io.p, 24 * N, hipMemcpyDeviceToHost, P->stream));
  }
  if (int st = sync_check(P, "download rotations")) return st;
  if (!P->perm.empty()) to_external(P, dst, rot_aa, 3);
  return 0;
}

// ---- TrustRegionMinimizer::Minimize + LevenbergMarquardtStrategy (ceres 1.14 semantics) ----
int lm_solve(gsfm_rot_problem* P, const gsfm_rot_options& o_in, gsfm_rot_summary* sum) {
  // Several scenes batched as one disconnected graph (BASELINE C4): the PCG stopping rule is a GLOBAL relative residual, so a
  // component whose gradient is already orders of magnitude below the others' is allowed an error that is large against its own
  // right-hand side, and at vanishing damping that error lands in its weakly determined directions.  Measured on the 14-scene batch
  // with the real Madrid graph inside: 3e-5 rad on Madrid's cameras at 1e-12, 4e-9 at 1e-14 (for 9 % more PCG iterations; the
  // reference's Cholesky solves every block exactly).  Disconnected problems therefore never run looser than 1e-14.
  gsfm_rot_options o = o_in;
  if (P->n_components > 1) o.cg_relative_tolerance = std::min(o.cg_relative_tolerance, 1e-14);
  if (o.verbose && o.cg_relative_tolerance != o_in.cg_relative_tolerance)
    fprintf(stderr, "[gsfm] the view graph has %u connected components: PCG runs to a relative residual of %.0e instead of the requested %.0e\n", P->n_components, o.cg_relative_tolerance, o_in.cg_relative_tolerance);
  const double t0 = now_ms();
  std::memset(sum, 0, sizeof(*sum));
  sum->iters_to_1e6 = -1;
  sum->num_edges_used = P->cost.n;
  P->trace.clear();
  P->timer.acc[0] = P->timer.acc[1] = P->timer.acc[2] = 0;
  P->graph_launches = 0;
  P->n_collectives = P->n_pcg_collectives = P->n_pcg_launched = 0;
  P->lap = P->lap_capable;
  double h[SC_N];
  double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
  int num_invalid = 0, iteration = 0;
  double x_cost = 0, x_norm = 0, gmax = 0;

  auto record = [&](double cost, double dc, double sn, double rd, int cg) {
    const double row[GSFM_ROT_TRACE_COLS] = {(double)iteration, cost, dc, gmax, sn, rd, radius, (double)cg};
    P->trace.insert(P->trace.end(), row, row + GSFM_ROT_TRACE_COLS);
    if (o.verbose) fprintf(stderr, "[gsfm] it %3d cost %.12e dcost %.3e |g| %.3e |dx| %.3e rho %.3e radius %.3e cg %d\n",
                           iteration, cost, dc, gmax, sn, rd, radius, cg);
  };
  auto finish = [&](int term) {
    sum->termination = term; sum->num_iterations = iteration; sum->final_cost = x_cost; sum->final_gradient_max_norm = gmax;
    sum->final_radius = radius; sum->t_total_ms = now_ms() - t0;
    sum->num_graph_launches = P->graph_launches;
    sum->num_collectives = P->n_collectives; sum->num_pcg_collectives = P->n_pcg_collectives; sum->num_pcg_launched = P->n_pcg_launched;
    sum->t_linearize_ms = P->timer.acc[T_LIN]; sum->t_sweep_ms = P->timer.acc[T_SWEEP]; sum->t_cg_ms = P->timer.acc[T_CG];
    if (!std::isfinite(x_cost)) sum->nonfinite = 1;
    return 0;
  };

  // Init + IterationZero
  hipLaunchKernelGGL(k_cam_norm, dim3(P->nb_cam), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->active.p, P->n_cams, P->param_dim, P->part_cam.p);
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(GSFM_BLOCK), 0, P->stream, P->part_cam.p, P->nb_cam, P->scal.p + SC_XNORM2);
  if (int st = launch_cost(P, P->q.p, SC_COST)) return st;
  if (int st = launch_lin(P, P->q.p)) return st;
  sum->num_residual_sweeps++; sum->num_linearizations++;
  launch_prep(P, o, radius, true);
  bool prep_valid = true;
  if (int st = read_scalars(P, h)) return st;
  x_cost = h[SC_COST]; gmax = h[SC_GMAX]; x_norm = std::sqrt(h[SC_XNORM2]);
  sum->initial_cost = x_cost;
  record(x_cost, 0, 0, 0, 0);
  if (!std::isfinite(x_cost)) return finish(GSFM_TERM_FAILURE);
  if (gmax <= o.gradient_tolerance) return finish(GSFM_TERM_GRADIENT_TOLERANCE);
  bool last_successful = false, pcg_struggles = false;
  while (true) {
    if (iteration >= o.max_num_iterations) return finish(GSFM_TERM_NO_CONVERGENCE);
    if (last_successful && gmax <= o.gradient_tolerance) return finish(GSFM_TERM_GRADIENT_TOLERANCE);
    if (radius <= o.min_trust_region_radius) return finish(GSFM_TERM_FAILURE);
    ++iteration;
    last_successful = false;
    if (!prep_valid) launch_prep(P, o, radius, false);
    prep_valid = false;
    int cg = 0; double cg_rel = 0;
    bool dense_used = false;
    // dense_cholesky_max_cams > 0: exact Cholesky steps for graphs up to that size; < 0: up to |value| cameras, but only
    // once a PCG solve of this run has needed more than 150 iterations (2.5 ms of factorisation beats that many mat-vecs)
    const int64_t dense_cap = o.dense_cholesky_max_cams > 0 ? o.dense_cholesky_max_cams : -(int64_t)o.dense_cholesky_max_cams;
    if (!P->sharded && dense_cap > 0 && (int64_t)P->n_cams <= dense_cap && (o.dense_cholesky_max_cams > 0 || pcg_struggles)) {
      if (int st = run_dense(P, &dense_used)) return st;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
      if (!dense_used) {
        if (int st = coarse_build(P, pcg_struggles)) return st;
        if (int st = ((P->coarse_n == 0 && single_reduction_possible(P) && use_single_reduction(P, o)) ? run_pcg2(P, o, &cg, &cg_rel) : run_pcg(P, o, &cg, &cg_rel))) return st;
      }
      launch_step(P);
      if (int st = launch_cost(P, P->q_trial.p, SC_TRIAL)) return st;
      if (int st = read_scalars(P, h)) return st;
      if (!dense_used) break;
      int info = 0;
      std::memcpy(&info, &h[SC_DENSE_INFO], sizeof(int));
      if (info == 0) { sum->num_dense_solves++; break; }
      dense_used = false;   // not positive definite to working precision: the step just evaluated is meaningless, PCG solves it again
    }
    if (cg > 150) pcg_struggles = true;
    if (o.verbose && !dense_used && cg >= o.max_cg_iterations && cg_rel > o.cg_relative_tolerance)
      fprintf(stderr, "[gsfm] it %3d: PCG stopped at its cap of %d iterations with a relative residual of %.1e (tolerance %.1e): this step is inexact\n", iteration, o.max_cg_iterations, cg_rel, o.cg_relative_tolerance);
    sum->num_cg_iterations += cg;
    sum->num_residual_sweeps++;
    // model_cost_change = -eta.g - 1/2 eta^T B eta with B eta = -g - r_cg - Lambda eta
    const double eta_g = h[SC_STEP], eta_r = h[SC_STEP + 1], eta_L = h[SC_STEP + 2];
    const double model_cost_change = -0.5 * eta_g + 0.5 * eta_r + 0.5 * eta_L;
    const bool valid = std::isfinite(model_cost_change) && model_cost_change > 0.0;
    if (!valid) {  // HandleInvalidStep
      if (++num_invalid >= 5) return finish(GSFM_TERM_FAILURE);
      radius /= decrease_factor; decrease_factor *= 2.0;
      sum->num_unsuccessful_steps++;
      record(x_cost, 0, 0, 0, cg);
      continue;
    }
    num_invalid = 0;
    double cand_cost = h[SC_TRIAL];
    if (!std::isfinite(cand_cost)) { cand_cost = std::numeric_limits<double>::max(); sum->nonfinite = 1; }
    const double step_norm = std::sqrt(h[SC_STEP + 3]);
    const double cost_change = x_cost - cand_cost;
    const double rel_dec = cost_change / model_cost_change;
    if (sum->iters_to_1e6 < 0 && std::fabs(cost_change) <= 1e-6 * x_cost) sum->iters_to_1e6 = iteration;
    if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) { record(x_cost, cost_change, step_norm, rel_dec, cg); return finish(GSFM_TERM_PARAMETER_TOLERANCE); }
    if (std::fabs(cost_change) <= o.function_tolerance * x_cost) { record(x_cost, cost_change, step_norm, rel_dec, cg); return finish(GSFM_TERM_FUNCTION_TOLERANCE); }
    if (rel_dec > o.min_relative_decrease) {  // HandleSuccessfulStep
      std::swap(P->x.p, P->x_trial.p);
      // (a copy, not a pointer swap: the captured PCG / Cholesky graphs hold the address of the quaternions they rotate with)
      HIPCHK(hipMemcpyAsync(P->q.p, P->q_trial.p, 32 * (size_t)P->n_cams, hipMemcpyDeviceToDevice, P->stream));
      x_norm = std::sqrt(h[SC_STEP + 4]);
      x_cost = cand_cost;  // Ceres re-evaluates at the accepted point: same value
      if (int st = launch_lin(P, P->q.p)) return st;
      sum->num_residual_sweeps++; sum->num_linearizations++;
      radius = radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel_dec - 1.0, 3));
      radius = std::fmin(o.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      launch_prep(P, o, radius, false);
      prep_valid = true;
      if (int st = read_scalars(P, h)) return st;
      gmax = h[SC_GMAX];
      sum->num_successful_steps++;
      last_successful = true;
    } else {  // HandleUnsuccessfulStep
      radius /= decrease_factor; decrease_factor *= 2.0;
      sum->num_unsuccessful_steps++;
    }
    record(x_cost, cost_change, step_norm, rel_dec, cg);
  }
}

gsfm_rot_options default_options() { gsfm_rot_options o; gsfm_rot_options_default(&o); return o; }

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <typename EidVec>
int upload_planes(gsfm_rot_problem* P, EdgePlanes& pl, const EidVec& eid, const double* d_rel_aa) {
  pl.n = eid.size();
  if (pl.eid.upload(eid) != hipSuccess || pl.qr0.alloc(pl.n) != hipSuccess || pl.qr1.alloc(pl.n) != hipSuccess)
    return fail(GSFM_ERR_HIP, "uploading edge planes failed (out of memory?)");
  if (pl.n) hipLaunchKernelGGL(k_build_qrel, dim3(grid_for(pl.n)), dim3(GSFM_BLOCK), 0, P->stream, d_rel_aa, pl.eid.p, pl.n, pl.qr0.p, pl.qr1.p);
  if (P->wmode == W_MATRIX) {
    if (pl.w0.alloc(pl.n) != hipSuccess || pl.w1.alloc(pl.n) != hipSuccess || pl.w2.alloc(pl.n) != hipSuccess) return fail(GSFM_ERR_HIP, "alloc whitening planes");
  } else if (P->wmode == W_SCALAR) {
    if (pl.ws.alloc(pl.n) != hipSuccess) return fail(GSFM_ERR_HIP, "alloc weight plane");
  }
  return 0;
}
// Column-sorted layout of the directed entries (colsort_kernels.hpp): positions grouped by row block, sorted by column inside a block,
// cut into sub-chunks of GSFM_COL_SUB (each with its row-sorted slot permutation and per-row slot offsets), the sub-chunks of a block
// dealt to `nch` workgroups.  Host, once per problem, blocks in parallel.  In: the row-major CSR (rp, col with the role bit, deid = edge of
// every entry).  Out: col / deid REPLACED by their position-ordered forms (padding: GSFM_COL_PAD / edge 0), the layout arrays on the device.
int build_colsort(gsfm_rot_problem* P, const std::vector<uint32_t>& rp, hvec<uint32_t>& col, hvec<uint32_t>& deid, int n_threads) {
  constexpr uint32_t RB = GSFM_COL_RB, SUB = GSFM_COL_SUB;
  auto& C = P->cs;
  const uint32_t n_rows = P->n_rows, nblk = (n_rows + RB - 1) / RB;
  if (nblk == 0 || P->n_cams >= (1u << 22) - 1u) return 0;
  uint32_t cbits = 1;
  while (((1u << cbits) - 1u) <= P->n_cams) ++cbits;   // cameras 0 .. n_cams - 1 and the all-ones padding value
  const uint32_t cmax = cbits <= 19 ? (1u << (23 - cbits)) - 1u : 0u, kpad = (1u << cbits) - 1u;
  std::vector<size_t> sub_off((size_t)nblk + 1, 0);
  for (uint32_t b = 0; b < nblk; ++b) {
    const size_t ne = rp[std::min(n_rows, (b + 1) * RB)] - rp[b * RB];
    sub_off[b + 1] = sub_off[b] + (ne + SUB - 1) / SUB;
  }
  const size_t n_sub = sub_off[nblk], n_pos = n_sub * SUB;
  // Workgroups per block (each writes one partial sum per row, which the finish kernels add): about 22 sub-chunks (11 k entries) per
  // workgroup, but at least ~400 workgroups in all.  Measured on K3c + finish, same box each (profiles/r03_k3c_tuning.txt, r03_rank_share.txt):
  // C5 on one GPU (196 blocks of ~200 sub-chunks) 8 / 9 / 10 per block = 208 / 202 / 203 us; one rank of 4 (49 blocks) 8 / 13 / 17 / 32 =
  // 51 / 57 / 61 / 64 us; one rank of 8 (25 blocks) 8 / 16 / 24 / 32 = 39.5 / 33.3 / 38 / 39 us.  GSFM_COL_WGS=n asks for n workgroups in all.
  {
    const double per_block = (double)n_sub / nblk;
    uint32_t nch = std::max<uint32_t>((uint32_t)std::lround(per_block / 22.0), (400 + nblk - 1) / nblk);
    if (const char* e = getenv("GSFM_COL_WGS")) { const int v = atoi(e); if (v > 0) nch = ((uint32_t)v + nblk - 1) / nblk; }
    C.nch = std::min<uint32_t>(32, std::max<uint32_t>(1, nch));
  }
  if (n_pos == 0 || n_pos >= 0x7fffffffull) return 0;   // (positions are 32-bit in the kernels: stay on the row-major form)
  hvec<uint32_t> h_col(n_pos), h_eid(n_pos), h_kcol(n_pos);   // (every position is written below)
  hvec<uint2> h_meta(n_pos);
  hvec<uint16_t> h_kcnt(n_pos);
  std::vector<ColWg> h_wg((size_t)nblk * C.nch);
  parallel_run(std::max(1, std::min<int>(n_threads, (int)nblk)), [&](int t, int T) {
    std::vector<std::pair<uint64_t, uint32_t>> ent;   // (camera << 16 | local row, d): a repeated camera pair is ordered by d
    std::vector<uint32_t> cnt(RB + 1), fill(RB), chist;
    for (uint32_t b = (uint32_t)t; b < nblk; b += (uint32_t)T) {
      const uint32_t r0 = b * RB, r1 = std::min(n_rows, r0 + RB);
      const size_t ne = rp[r1] - rp[r0], ns = sub_off[b + 1] - sub_off[b];
      if ((size_t)P->n_cams <= 4 * ne + 4096) {
        // counting sort by camera: the rows are walked in order and a row's entries are in edge order, so equal cameras keep (row, d) order --
        // the same sequence as sorting the (camera, row, d) triples (199 -> ... ms of the 100k / 10M problem's creation)
        chist.assign((size_t)P->n_cams + 1, 0u);
        for (uint32_t d = rp[r0]; d < rp[r1]; ++d) chist[(col[d] & 0x7fffffffu) + 1]++;
        for (uint32_t c = 0; c < P->n_cams; ++c) chist[c + 1] += chist[c];
        ent.resize(ne);
        for (uint32_t r = r0; r < r1; ++r) for (uint32_t d = rp[r]; d < rp[r + 1]; ++d) {
          const uint32_t c = col[d] & 0x7fffffffu;
          ent[chist[c]++] = std::make_pair(((uint64_t)c << 16) | (r - r0), d);
        }
      } else {   // (a block far sparser than the camera range: forced layouts of small tests)
        ent.clear();
        for (uint32_t r = r0; r < r1; ++r) for (uint32_t d = rp[r]; d < rp[r + 1]; ++d) ent.emplace_back(((uint64_t)(col[d] & 0x7fffffffu) << 16) | (r - r0), d);
        std::sort(ent.begin(), ent.end());
      }
      for (uint32_t c = 0; c < C.nch; ++c) {
        const size_t lo = ns * c / C.nch, hi = ns * (c + 1) / C.nch;
        h_wg[(size_t)b * C.nch + c] = ColWg{(uint32_t)(sub_off[b] + lo), (uint32_t)(hi - lo), r0, 0};
      }
      for (size_t s = 0; s < ns; ++s) {
        const size_t lo = s * SUB, hi = std::min(ne, lo + SUB), base = (sub_off[b] + s) * SUB;
        std::fill(cnt.begin(), cnt.end(), 0u);
        for (size_t e = lo; e < hi; ++e) cnt[(ent[e].first & 0xffff) + 1]++;
        for (uint32_t r = 0; r < RB; ++r) cnt[r + 1] += cnt[r];
        std::copy(cnt.begin(), cnt.end() - 1, fill.begin());
        uint32_t pad_slot = (uint32_t)(hi - lo);
        for (size_t e = lo; e < lo + SUB; ++e) {
          const size_t o = base + (e - lo);
          const uint32_t p = (uint32_t)(e - lo), rows_here = (cnt[p + 1] - cnt[p]) << 10;   // position p also carries the slot count of ROW p
          if (e < hi) {
            const uint32_t rl = (uint32_t)(ent[e].first & 0xffff), d = ent[e].second;
            h_col[o] = col[d]; h_eid[o] = deid[d]; h_meta[o] = make_uint2(col[d], fill[rl]++ | rows_here | (rl << 20));
          } else { h_col[o] = GSFM_COL_PAD; h_eid[o] = 0; h_meta[o] = make_uint2(GSFM_COL_PAD, pad_slot++ | rows_here); }   // zero block, a slot no row reads
          const uint32_t rc = rows_here >> 10;
          h_kcol[o] = (h_meta[o].x == GSFM_COL_PAD ? kpad : (h_meta[o].x & 0x7fffffffu)) | ((h_meta[o].y & 0x1ffu) << cbits) | (std::min(rc, cmax) << (cbits + 9));
          h_kcnt[o] = (uint16_t)rc;
        }
      }
    }
  });
  C.n_wg = (uint32_t)h_wg.size(); C.n_pos = n_pos; C.cbits = cbits; C.cmax = cmax;
  if (C.wg.upload(h_wg) != hipSuccess || C.meta.upload(h_meta) != hipSuccess || C.kcol.upload(h_kcol) != hipSuccess || C.kcnt.upload(h_kcnt) != hipSuccess ||
      C.part.alloc((size_t)9 * C.n_wg * RB) != hipSuccess) {
    (void)hipGetLastError();
    C = gsfm_rot_problem::ColSort();   // out of memory: the row-major form needs none of this
    return 0;
  }
  col.swap(h_col); deid.swap(h_eid);
  C.active = true;
  return 0;
}

void run_whiten(gsfm_rot_problem* P, EdgePlanes& pl, const double* d_cov6, const double* d_inl) {
  if (P->wmode == W_NONE || pl.n == 0) return;
  WhitenArgs a{};
  a.cov6 = d_cov6; a.inl = d_inl; a.eid = pl.eid.p; a.n = pl.n; a.error_type = P->error_type;
  a.w0 = pl.w0.p; a.w1 = pl.w1.p; a.w2 = pl.w2.p; a.ws = pl.ws.p;
  hipLaunchKernelGGL(k_whiten, dim3(grid_for(pl.n)), dim3(GSFM_BLOCK), 0, P->stream, a);
}

}  // namespace

// =============================================================================================
extern "C" {

int gsfm_rot_abi_version(void) { return GSFM_ROT_ABI_VERSION; }
const char* gsfm_last_error(void) { return g_err.c_str(); }

void gsfm_rot_options_default(gsfm_rot_options* o) {
  std::memset(o, 0, sizeof(*o));
  o->max_num_iterations = 200; o->num_threads = 1;
  o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
  o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
  o->jacobi_scaling = 1; o->max_cg_iterations = 1000; o->cg_relative_tolerance = 1e-12; o->cg_check_interval = 8; o->verbose = 0; o->pcg_single_reduction = -1; o->cg_stall_iterations = 0; o->dense_cholesky_max_cams = 512; o->pcg_hip_graph = 1;
}

int32_t gsfm_rot_residual_dim(int32_t t) { return t == GSFM_ROT_QUATERNION_NORM ? 4 : t == GSFM_ROT_ROTATION_MAT_FNORM ? 9 : 3; }

static gsfm_status problem_create_impl(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i_in, const uint32_t* edge_j_in,
                                       const double* rel_aa, int32_t error_type, const double* cov6, const double* inlier_weight,
                                       const gsfm_rot_shard* shard, gsfm_rot_problem** out, gsfm_rot_problem** live) {
  if (!out) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  // (one rank of a sharded problem may hold no edge at all -- a slice of isolated cameras -- and still takes part in every collective)
  const bool multi_rank = shard && (shard->world_size > 1 || (shard->world_size == 1 && getenv("GSFM_FORCE_SHARD")));
  // Two failures cannot be agreed about with the other ranks and return at once: a descriptor without callbacks (there is nothing to call)
  // and a process without a HIP device (the callbacks take device pointers).  Everything else below goes through bail().
  if (multi_rank && (!shard->all_gather || !shard->all_reduce_sum)) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "bad shard descriptor: missing collective callbacks");
  if (const char* why = no_device_reason("the rotation solver")) return (gsfm_status)fail(GSFM_ERR_NO_DEVICE, why);

  gsfm_rot_problem* P = new gsfm_rot_problem;
  *live = P;   // (for the exception path of the public wrapper)
  // Sharded: a rank-local failure (bad argument, bad edge, allocation, upload, loss set-up) must not leave the other ranks blocked in a
  // collective.  Every rank passes through exactly ONE agreement all-reduce -- on the failure path from bail(), on the success path after
  // ALL of its local work -- carrying (failed?, votes against the two-level preconditioner); all ranks give up together if any of them
  // failed.  What follows the agreement are collectives only (active mask, component labels): they fail on every rank or on none.
  bool agreed = false;
  DevBuf<double> agree_buf;
  double coarse_votes_against = 0.0;
  auto agree = [&](double my_flag, double my_vote) -> int {   // number of ranks that failed, or -1 if the agreement itself could not be run
    agreed = true;
    if (!P->sharded) return 0;
    double h[2] = {my_flag, my_vote};
    if (agree_buf.alloc(2) != hipSuccess || hipMemcpy(agree_buf.p, h, 16, hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (all_reduce(P, agree_buf.p, 2) != 0) return -1;
    if (hipStreamSynchronize(P->stream) != hipSuccess || hipMemcpy(h, agree_buf.p, 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    coarse_votes_against = h[1];
    return (int)(h[0] + 0.5);
  };
  auto bail = [&](int st) {
    if (P->sharded && !agreed) { const std::string keep = g_err; (void)agree(1.0, 1.0); g_err = keep; }
    *live = nullptr;
    gsfm_rot_problem_destroy(P);
    return (gsfm_status)st;
  };
  const bool lap_on = getenv("GSFM_CREATE_TIMING") != nullptr;   // phase times of this function on stderr
  double lap_t = now_ms();
  auto lap = [&](const char* what) { if (lap_on) { const double t = now_ms(); fprintf(stderr, "gsfm create: %-28s %8.1f ms\n", what, t - lap_t); lap_t = t; } };
  (void)hipGetDevice(&P->device);
  if (hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking) != hipSuccess) { P->stream = nullptr; (void)hipGetLastError(); }   // (checked below, after the shard set-up)
  else P->own_stream = true;
  P->timer.stream = P->stream; P->timer.init();
  // GSFM_FORCE_SHARD=1 keeps the collective code path alive for a single rank (tests on a 1-GPU box)
  if (multi_rank) { P->sharded = true; P->shard = *shard; }   // from here on a failure reaches the other ranks through bail()
  if (!P->own_stream) return bail(fail(GSFM_ERR_HIP, "hipStreamCreate failed"));
  if (multi_rank && (shard->slice_width == 0 || shard->rank < 0 || shard->rank >= shard->world_size || (uint64_t)shard->slice_width * shard->world_size < n_cams))
    return bail(fail(GSFM_ERR_INVALID_ARG, "bad shard descriptor"));
  if (n_cams == 0 || (n_edges == 0 && !multi_rank)) return bail(fail(GSFM_ERR_EMPTY, "no cameras or no edges"));
  if (n_edges > 0 && (!edge_i_in || !edge_j_in || !rel_aa)) return bail(fail(GSFM_ERR_INVALID_ARG, "NULL edge arrays"));
  if (error_type < 0 || error_type > 8) return bail(fail(GSFM_ERR_INVALID_ARG, "unknown rotation error type"));
  if (n_cams >= 0x7fffffffu || n_edges >= 0x7fffffffull) return bail(fail(GSFM_ERR_INVALID_ARG, "problem too large for 31-bit indices"));
  const bool need_cov = error_type == GSFM_ROT_ANGLE_AXIS_COVARIANCE || error_type == GSFM_ROT_ANGLE_AXIS_COV_INLIERS ||
                        error_type == GSFM_ROT_ANGLE_AXIS_COVTRACE || error_type == GSFM_ROT_ANGLE_AXIS_COVNORM;
  const bool need_inl = error_type == GSFM_ROT_ANGLE_AXIS_INLIERS || error_type == GSFM_ROT_ANGLE_AXIS_COV_INLIERS;
  if (need_cov && !cov6) return bail(fail(GSFM_ERR_INVALID_ARG, "this error type needs per-edge covariances (cov6)"));
  if (need_inl && !inlier_weight) return bail(fail(GSFM_ERR_INVALID_ARG, "this error type needs per-edge inlier weights"));
  P->n_cams = n_cams; P->n_edges_in = n_edges; P->error_type = error_type;
  P->functor = error_type == GSFM_ROT_QUATERNION_COSINE ? F_QCOS : error_type == GSFM_ROT_QUATERNION_NORM ? F_QNORM
               : error_type == GSFM_ROT_ROTATION_MAT_FNORM ? F_RFNORM : F_AA;
  P->res_dim = gsfm_rot_residual_dim(error_type);
  P->param_dim = P->functor == F_AA ? 3 : 4;
  P->wmode = (error_type == GSFM_ROT_ANGLE_AXIS_COVARIANCE || error_type == GSFM_ROT_ANGLE_AXIS_COV_INLIERS) ? W_MATRIX
             : (error_type == GSFM_ROT_ANGLE_AXIS_INLIERS || error_type == GSFM_ROT_ANGLE_AXIS_COVTRACE || error_type == GSFM_ROT_ANGLE_AXIS_COVNORM) ? W_SCALAR
             : W_NONE;
  if (multi_rank) {
    P->own_begin = std::min<uint64_t>((uint64_t)shard->rank * shard->slice_width, n_cams);
    P->own_end = std::min<uint64_t>((uint64_t)(shard->rank + 1) * shard->slice_width, n_cams);
    P->n_pad = shard->slice_width * shard->world_size;
  } else { P->own_begin = 0; P->own_end = n_cams; P->n_pad = n_cams; P->shard.slice_width = n_cams; P->shard.world_size = 1; }
  P->n_rows = P->own_end - P->own_begin;
  if (hipHostMalloc(&P->pin, 256, hipHostMallocDefault) != hipSuccess) { P->pin = nullptr; (void)hipGetLastError(); }   // (read_back then copies to pageable memory)

  // ---- host-side structure: directed entries by row (counting sort), cost-owned edges ----
  const uint32_t *edge_i = edge_i_in, *edge_j = edge_j_in;
  std::vector<uint32_t> ei_perm, ej_perm;
  const uint32_t ob = P->own_begin, oe = P->own_end;
  auto owned = [&](uint32_t c) { return c >= ob && c < oe; };
  std::vector<uint32_t> rp, cost_eid;
  hvec<uint32_t> col, deid;   // (sized once, filled completely by the threads below)
  const int n_host_threads = host_threads();
  auto build_rows = [&]() -> int {
    rp.assign((size_t)P->n_rows + 1, 0);
    cost_eid.clear();
    cost_eid.reserve(P->sharded ? n_edges / 2 + 16 : n_edges);
    if (!P->sharded && n_edges >= 200000 && n_host_threads > 1) {   // one GPU: every camera and every edge is owned; count on all threads
      std::vector<int> bad((size_t)n_host_threads, 0);
      parallel_run(n_host_threads, [&](int t, int T) {
        const uint64_t lo = n_edges * t / T, hi = n_edges * (t + 1) / T;
        for (uint64_t e = lo; e < hi; ++e) if (edge_i[e] >= n_cams || edge_j[e] >= n_cams || edge_i[e] == edge_j[e]) { bad[t] = 1; break; }
      });
      for (int b : bad) if (b) return fail(GSFM_ERR_INVALID_ARG, "edge with an out-of-range or repeated camera index");
      parallel_count(n_host_threads, 2 * n_edges, n_cams, [&](size_t u) { return (u & 1) ? edge_j[u >> 1] : edge_i[u >> 1]; }, rp.data() + 1);
      cost_eid.resize(n_edges);
      for (uint64_t e = 0; e < n_edges; ++e) cost_eid[e] = (uint32_t)e;
    } else
    for (uint64_t e = 0; e < n_edges; ++e) {
      const uint32_t i = edge_i[e], j = edge_j[e];
      if (i >= n_cams || j >= n_cams || i == j) return fail(GSFM_ERR_INVALID_ARG, "edge with an out-of-range or repeated camera index");
      if (owned(i)) rp[i - ob + 1]++;
      if (owned(j)) rp[j - ob + 1]++;
      // each edge is cost-owned by exactly one rank: the owner of `first` if (i + j) is even, else of `second`
      const uint32_t c = (((i + j) & 1u) == 0u) ? i : j;
      if (owned(c)) cost_eid.push_back((uint32_t)e);
      else if (!owned(i) && !owned(j)) return fail(GSFM_ERR_INVALID_ARG, "sharded problem: edge touches no owned camera");
    }
    for (size_t r = 0; r < P->n_rows; ++r) rp[r + 1] += rp[r];
    col.resize(rp[P->n_rows]); deid.resize(rp[P->n_rows]);
    // Fill: the random writes into col / deid (8 B per directed entry) are what costs.  Every thread streams over all edges and fills only
    // the rows of its own contiguous range (ranges balanced by entry count), so each row still receives its entries in edge order:
    // the result is identical to the serial loop for any thread count.
    std::vector<uint32_t> fill(rp.begin(), rp.end() - 1);
    const int T = n_edges >= 200000 ? n_host_threads : 1;
    std::vector<uint32_t> cut((size_t)T + 1, 0);
    for (int t = 1; t < T; ++t) cut[t] = (uint32_t)(std::lower_bound(rp.begin(), rp.end(), (uint32_t)((uint64_t)rp[P->n_rows] * t / T)) - rp.begin());
    cut[T] = P->n_rows;
    for (int t = 1; t <= T; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
    parallel_run(T, [&](int t, int) {
      const uint32_t lo = ob + cut[t], hi = ob + cut[t + 1];
      if (lo >= hi) return;
      for (uint64_t e = 0; e < n_edges; ++e) {
        const uint32_t i = edge_i[e], j = edge_j[e];
        if (i >= lo && i < hi) { const uint32_t d = fill[i - ob]++; col[d] = j; deid[d] = (uint32_t)e; }
        if (j >= lo && j < hi) { const uint32_t d = fill[j - ob]++; col[d] = i | 0x80000000u; deid[d] = (uint32_t)e; }
      }
    });
    return 0;
  };
  if (int st = build_rows()) return bail(st);
  lap("directed rows (CSR)");
  // ---- optional locality relabelling of the cameras (unsharded: the rows are the full adjacency; see reorder_for_locality) ----
  if (!P->sharded && reorder_for_locality(n_cams, n_edges, edge_i, edge_j, rp, col, &P->perm)) {
    ei_perm.resize(n_edges); ej_perm.resize(n_edges);
    for (uint64_t e = 0; e < n_edges; ++e) { ei_perm[e] = P->perm[edge_i[e]]; ej_perm[e] = P->perm[edge_j[e]]; }
    edge_i = ei_perm.data(); edge_j = ej_perm.data();
    if (int st = build_rows()) return bail(st);
    // order every row by neighbour so that adjacent lanes gather adjacent cameras
    std::vector<std::pair<uint32_t, uint32_t>> row;
    for (size_t r = 0; r < P->n_rows; ++r) {
      row.clear();
      for (uint32_t d = rp[r]; d < rp[r + 1]; ++d) row.emplace_back(col[d], deid[d]);
      std::sort(row.begin(), row.end(), [](const std::pair<uint32_t, uint32_t>& a, const std::pair<uint32_t, uint32_t>& b) {
        const uint32_t ca = a.first & 0x7fffffffu, cb = b.first & 0x7fffffffu;
        return ca != cb ? ca < cb : a.second < b.second;
      });
      for (uint32_t d = rp[r]; d < rp[r + 1]; ++d) { col[d] = row[d - rp[r]].first; deid[d] = row[d - rp[r]].second; }
    }
  }
  lap("locality relabelling");
  {  // two-level preconditioner: aggregates = contiguous chunks of the camera order, which only mean something if that order is the
     // locality order (adopted above) -- GSFM_PCG_COARSE=n forces n aggregates, =0 switches it off
    const char* env = getenv("GSFM_PCG_COARSE");
    int want = env && *env ? atoi(env) : -1;
    if (want < 0 && n_cams >= 4096) {   // (sharded: n_cams is the padded index space of the partition's locality order, the edges this rank's share)
      // spatially coherent in the numbering the rows now have (relabelled above, or coherent as given)?  Mean index distance over a
      // sample of the edges: n/3 for a uniformly random graph, the neighbourhood radius for a coherent one
      double sum = 0.0; uint64_t cnt = 0;
      for (uint64_t e = 0; e < n_edges; e += 61) { sum += std::fabs((double)edge_i[e] - (double)edge_j[e]); ++cnt; }
      if (cnt == 0) { cnt = 1; sum = 0.0; }   // a rank without edges has no objection
      // What block-Jacobi cannot cope with is the DIAMETER of the graph, ~ cameras / neighbourhood radius.  Measured on coherent graphs: from a
      // ratio of ~100 the coarse space cuts the iterations 5-15x (12k cameras / radius 100: 120; 100k / 250: 400); between 32 and 100 it
      // depends on the degree (6000 cameras / 100, degree 40: 1.4x faster; 5000 / 100, degree 240: no fewer iterations, slower), so there it
      // is switched on only after a PCG solve has struggled; below, never.
      const double ratio = (double)n_cams / std::max(1.0, sum / (double)cnt);
      // one aggregate per ~256 cameras, 16 to 64 of them: more aggregates need fewer iterations but a larger dense inverse per LM step
      // (measured: 6000 cameras 16 > 64 aggregates, 100k cameras 64 > 16 and > 128)
      // (from 400k cameras a PCG iteration costs more than the 5 ms the host needs for the 384-unknown inverse: 128 aggregates there)
      want = ratio >= 32.0 ? (int)std::min<uint32_t>(n_cams >= 400000 ? 128 : 64, std::max<uint32_t>(16, n_cams / 256)) : 0;
      P->coarse_adaptive = ratio < 100.0;
    }
    if (want < 0) want = 0;
    want = std::min(want, 128);
    if (want < 2 || n_cams < 4u * (uint32_t)want) want = 0;
    if (want) {
      P->coarse_chunk = (n_cams + want - 1) / want;
      P->coarse_want = (n_cams + P->coarse_chunk - 1) / P->coarse_chunk;   // no empty aggregate
      const size_t nc = 3 * (size_t)P->coarse_want;
      if (P->coarseA.alloc(nc * nc) != hipSuccess || P->coarseAinv.alloc(nc * nc) != hipSuccess || P->coarse_rc.alloc(nc, true) != hipSuccess ||
          P->coarse_xc.alloc(nc + 1, true) != hipSuccess || P->coarse_scale.alloc(2, true) != hipSuccess || P->coarse_part.alloc(6 * (size_t)grid_for(n_cams), true) != hipSuccess) {
        P->coarseA.release(); P->coarse_want = 0; (void)hipGetLastError();   // (a sharded rank then votes against below: all ranks stay on block-Jacobi)
      }
    }
  }
  // connected components of the view graph: counted here on one GPU; a rank of a sharded problem sees only its own edges, so the
  // partitioner passes the verdict in the shard descriptor (GSFM_SHARD_DISCONNECTED)
  if (!P->sharded) P->n_components = std::max<uint32_t>(1, count_components(n_cams, n_edges, edge_i, edge_j));
  else P->n_components = (P->shard.flags & GSFM_SHARD_DISCONNECTED) ? 2 : 1;
  lap("connected components");
  const size_t nd = rp[P->n_rows];
  {
    const double mean_deg = P->n_rows ? (double)nd / P->n_rows : 0.0;
    P->G = mean_deg >= 96 ? 64 : mean_deg >= 48 ? 32 : mean_deg >= 24 ? 16 : mean_deg >= 12 ? 8 : 4;
    if (const char* g = getenv("GSFM_ROW_LANES")) {  // tuning override: lanes per camera row (power of two <= 64)
      const int v = atoi(g);
      if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64) P->G = (uint32_t)v;
    }
  }
  // cost edges ordered by the tile (camera block of `first`, camera block of `second`), and by `first` inside a
  // tile: two stable counting sorts, O(E + N + #tiles).  k_cost stages both blocks of a tile in LDS.
  std::vector<CostTile> tiles;
  {
    const size_t Ec = cost_eid.size();
    std::vector<uint32_t> tmp(Ec), cnt((size_t)n_cams + 1, 0);
    parallel_count(n_host_threads, Ec, n_cams, [&](size_t u) { return edge_i[cost_eid[u]]; }, cnt.data() + 1);
    for (size_t c = 0; c < n_cams; ++c) cnt[c + 1] += cnt[c];
    {  // stable scatter by `first`, threads own contiguous key ranges (balanced by count): same result as the serial loop
      const int T = Ec >= 200000 ? n_host_threads : 1;
      std::vector<uint32_t> kc((size_t)T + 1, 0);
      for (int t = 1; t < T; ++t) kc[t] = (uint32_t)(std::lower_bound(cnt.begin(), cnt.end(), (uint32_t)((uint64_t)Ec * t / T)) - cnt.begin());
      kc[T] = n_cams;
      for (int t = 1; t <= T; ++t) kc[t] = std::max(kc[t], kc[t - 1]);
      parallel_run(T, [&](int t, int) {
        const uint32_t lo = kc[t], hi = kc[t + 1];
        if (lo >= hi) return;
        for (size_t u = 0; u < Ec; ++u) { const uint32_t k = edge_i[cost_eid[u]]; if (k >= lo && k < hi) tmp[cnt[k]++] = cost_eid[u]; }
      });
    }
    const uint64_t nblk = ((uint64_t)n_cams + GSFM_CAMBLOCK - 1) / GSFM_CAMBLOCK;
    auto tile_of = [&](uint32_t e) { return (uint64_t)(edge_i[e] / GSFM_CAMBLOCK) * nblk + edge_j[e] / GSFM_CAMBLOCK; };
    // The bucket table has nblk^2 entries: beyond 4096 camera blocks (8.4M cameras) the edges simply stay ordered by
    // `first` (such a sweep is far too thin for LDS tiles anyway).
    const bool bucketed = nblk <= 4096;
    std::vector<size_t> tstart(bucketed ? nblk * nblk + 1 : 1, 0);
    size_t populated = 0;
    if (bucketed) {
      {
        std::vector<uint32_t> tc(nblk * nblk + 1, 0);
        parallel_count(n_host_threads, Ec, nblk * nblk, [&](size_t u) { return tile_of(tmp[u]); }, tc.data() + 1);
        for (uint64_t b = 0; b < nblk * nblk; ++b) tstart[b + 1] = tstart[b] + tc[b + 1];
      }
      std::vector<size_t> fillt(tstart.begin(), tstart.end() - 1);
      {  // stable scatter by tile, threads own contiguous tile ranges
        const int T = Ec >= 200000 ? n_host_threads : 1;
        const uint64_t nt = nblk * nblk;
        std::vector<uint64_t> kc((size_t)T + 1, 0);
        for (int t = 1; t < T; ++t) kc[t] = (uint64_t)(std::lower_bound(tstart.begin(), tstart.end(), (size_t)((uint64_t)Ec * t / T)) - tstart.begin());
        kc[T] = nt;
        for (int t = 1; t <= T; ++t) kc[t] = std::min<uint64_t>(nt, std::max(kc[t], kc[t - 1]));
        parallel_run(T, [&](int t, int) {
          const uint64_t lo = kc[t], hi = kc[t + 1];
          if (lo >= hi) return;
          for (size_t u = 0; u < Ec; ++u) { const uint64_t k = tile_of(tmp[u]); if (k >= lo && k < hi) cost_eid[fillt[k]++] = tmp[u]; }
        });
      }
      for (uint64_t b = 0; b < nblk * nblk; ++b) populated += tstart[b + 1] > tstart[b];
    } else {
      cost_eid = tmp;
    }
    // Thin tiles cannot amortise the 128 KiB LDS fill (88 B streamed per edge): below ~4096 edges per populated tile the
    // sweep gathers the quaternions directly instead (k_cost_direct).  GSFM_K1_DIRECT=0/1 overrides (A/B measurements).
    P->cost_direct = !bucketed || (populated > 0 && Ec / populated < 4096);
    if (const char* v = getenv("GSFM_K1_DIRECT")) P->cost_direct = !bucketed || atoi(v) != 0;
    if (P->cost_direct) {
      const size_t chunk = std::min<size_t>(8192, std::max<size_t>(GSFM_BLOCK, (Ec + 2047) / 2048));
      for (size_t lo = 0; lo < Ec; lo += chunk) tiles.push_back(CostTile{0, 0, (uint32_t)lo, (uint32_t)std::min(Ec, lo + chunk)});
    } else {
      // one workgroup per <= max_tile edges of a tile: ~2 workgroups per CU for big sweeps, >= 1 pass of 1024 lanes for small ones
      const size_t max_tile = std::min<size_t>(16384, std::max<size_t>(GSFM_TILE_THREADS, (Ec + 511) / 512));
      for (uint64_t b = 0; b < nblk * nblk; ++b) {
        size_t lo = tstart[b];
        const size_t hi = tstart[b + 1];
        while (lo < hi) {
          const size_t ce = std::min(hi, lo + max_tile);
          tiles.push_back(CostTile{(uint32_t)(b / nblk), (uint32_t)(b % nblk), (uint32_t)lo, (uint32_t)ce});
          lo = ce;
        }
      }
    }
    if (tiles.empty()) tiles.push_back(CostTile{0, 0, 0, 0});
  }
  std::vector<uint2> cidx(cost_eid.size());
  const uint32_t idx_mod = P->cost_direct ? 0xffffffffu : (uint32_t)GSFM_CAMBLOCK;   // global or block-local camera indices
  parallel_run(cost_eid.size() >= 200000 ? n_host_threads : 1, [&](int t, int T) {
    const size_t lo = cost_eid.size() * t / T, hi = cost_eid.size() * (t + 1) / T;
    for (size_t u = lo; u < hi; ++u)
      cidx[u] = P->cost_direct ? make_uint2(edge_i[cost_eid[u]], edge_j[cost_eid[u]]) : make_uint2(edge_i[cost_eid[u]] % idx_mod, edge_j[cost_eid[u]] % idx_mod);
  });
  P->h_cost_eid = cost_eid;

  lap("cost tiles");
  {  // K2c / K3c, the column-sorted layout of the directed entries: for large graphs whose rows offer the gathers no locality -- i.e. where
     // neither the relabelling nor the two-level preconditioner (both for spatially coherent graphs) applies.  GSFM_K3_COLSORT=0/1 overrides.
    const char* env = getenv("GSFM_K3_COLSORT");
    const int mode = env && *env ? atoi(env) : -1;
    const bool lap_ok = (P->functor == F_AA || P->functor == F_QCOS) && !(getenv("GSFM_LAPLACIAN") && atoi(getenv("GSFM_LAPLACIAN")) == 0);
    // What makes it pay is line sharing in the gathers: about one entry per camera and row block, i.e. rows of 512 * mean degree >= ~n_cams / 2
    // entries (C5: 102k entries per block for 100k cameras, on one GPU and on every rank of a sharded run alike); sparser blocks gain nothing.
    const double per_block = P->n_rows ? (double)GSFM_COL_RB * (double)nd / (double)P->n_rows : 0.0;
    if (lap_ok && nd > 0 && (mode > 0 || (mode < 0 && nd >= (size_t)1000000 && per_block >= 0.5 * (double)n_cams && P->coarse_want == 0 && P->perm.empty()))) {
      if (int st = build_colsort(P, rp, col, deid, n_host_threads)) return bail(st);
      if (P->cs.active) { P->coarse_want = 0; P->coarse_adaptive = false; }
    }
  }
  lap("column-sorted layout");
  const size_t nd_planes = P->cs.active ? P->cs.n_pos : nd;   // per-entry planes: one per position (padded sub-chunks) in the column-sorted layout
  // ---- uploads ----
  {
    DevBuf<double> d_rel;   // the measurements go up once; both sets of planes are gathered from them on the device
    if (d_rel.alloc(3 * n_edges) != hipSuccess || (n_edges > 0 && hipMemcpy(d_rel.p, rel_aa, 24 * n_edges, hipMemcpyHostToDevice) != hipSuccess))
      return bail(fail(GSFM_ERR_HIP, "uploading the relative rotations failed"));
    if (int st = upload_planes(P, P->cost, cost_eid, d_rel.p)) return bail(st);
    if (int st = upload_planes(P, P->dir, deid, d_rel.p)) return bail(st);
    if (hipStreamSynchronize(P->stream) != hipSuccess) return bail(fail(GSFM_ERR_HIP, "building the measurement planes failed"));
  }
  if (P->cost_tiles.upload(tiles) != hipSuccess) return bail(fail(GSFM_ERR_HIP, "uploading cost tiles failed"));
  P->nb_cost = (int)tiles.size();
  if (P->cost_idx.upload(cidx) != hipSuccess || P->row_ptr.upload(rp) != hipSuccess || P->col.upload(col) != hipSuccess)
    return bail(fail(GSFM_ERR_HIP, "uploading graph structure failed"));
  // planes h3, h4 (the last three of the nine values of a general block) are allocated on first use: the Laplacian form needs six
  if (P->h0.alloc(nd_planes) != hipSuccess || P->h1.alloc(nd_planes) != hipSuccess || P->h2.alloc(nd_planes) != hipSuccess)
    return bail(fail(GSFM_ERR_HIP, "allocating normal-equation blocks failed"));
  lap("edge planes -> device");

  {  // K0 whitening
    DevBuf<double> d_cov, d_inl;
    if (P->wmode != W_NONE) {
      // straight from the caller's arrays (no staging copy: cov6 is 48 B per edge)
      if (cov6 && need_cov && (d_cov.alloc(6 * n_edges) != hipSuccess || (n_edges && hipMemcpy(d_cov.p, cov6, 48 * n_edges, hipMemcpyHostToDevice) != hipSuccess)))
        return bail(fail(GSFM_ERR_HIP, "upload cov6"));
      if (inlier_weight && need_inl && (d_inl.alloc(n_edges) != hipSuccess || (n_edges && hipMemcpy(d_inl.p, inlier_weight, 8 * n_edges, hipMemcpyHostToDevice) != hipSuccess)))
        return bail(fail(GSFM_ERR_HIP, "upload inlier weights"));
      run_whiten(P, P->cost, d_cov.p, d_inl.p);
      run_whiten(P, P->dir, d_cov.p, d_inl.p);
      if (hipStreamSynchronize(P->stream) != hipSuccess) return bail(fail(GSFM_ERR_HIP, "whitening kernel failed"));
    }
  }
  lap("whitening");
  // ---- camera buffers ----
  const size_t N = n_cams, NP = P->n_pad;
  P->nb_cam = grid_for(N);
  if (P->nb_cam > GSFM_MAX_PARTIALS * 64) return bail(fail(GSFM_ERR_INVALID_ARG, "too many cameras"));
  bool ok = true;
  ok &= P->x.alloc(4 * N, true) == hipSuccess; ok &= P->x_trial.alloc(4 * N, true) == hipSuccess; ok &= P->aa_io.alloc(3 * N, true) == hipSuccess;
  ok &= P->active.alloc(NP, true) == hipSuccess; ok &= P->scale.alloc(3 * N, true) == hipSuccess; ok &= P->gD.alloc(9 * NP, true) == hipSuccess;
  ok &= P->Mblk.alloc(6 * N) == hipSuccess; ok &= P->Minv.alloc(6 * N) == hipSuccess; ok &= P->Lam.alloc(6 * N) == hipSuccess;
  ok &= P->Tinv.alloc(9 * N) == hipSuccess; ok &= P->b.alloc(3 * N) == hipSuccess; ok &= P->D6.alloc(6 * N) == hipSuccess;
  ok &= P->q.alloc(2 * N) == hipSuccess; ok &= P->q_trial.alloc(2 * N) == hipSuccess;
  ok &= P->xcg.alloc(3 * N) == hipSuccess; ok &= P->r.alloc(3 * N) == hipSuccess; ok &= P->z.alloc(3 * N) == hipSuccess;
  ok &= P->p.alloc(3 * NP, true) == hipSuccess; ok &= P->Ap.alloc(3 * NP, true) == hipSuccess; ok &= P->u_rot.alloc(3 * NP, true) == hipSuccess;
  {
    const char* env = getenv("GSFM_LAPLACIAN");   // =0: keep the general 9-value blocks (A/B measurements)
    P->lap_capable = (P->functor == F_AA || P->functor == F_QCOS) && !(env && atoi(env) == 0);
    P->lap = P->lap_capable;
  }
  ok &= P->part_a.alloc(P->nb_cam) == hipSuccess; ok &= P->part_b.alloc(P->nb_cam) == hipSuccess;
  ok &= P->part_cam.alloc((size_t)5 * P->nb_cam) == hipSuccess; ok &= P->part_cost.alloc((size_t)2 * P->nb_cost) == hipSuccess;
  ok &= P->scal.alloc(SC_N, true) == hipSuccess; ok &= P->cgsc.alloc(1, true) == hipSuccess;
  {  // fused mat-vec of the single-reduction PCG: one row group (256 / G rows) per workgroup unless that leaves too many partials
    const size_t rows_per_group = GSFM_BLOCK / P->G, groups = (P->n_rows + rows_per_group - 1) / rows_per_group;
    size_t max_partials = GSFM_MV_MAX_PARTIALS;
    if (P->sharded) {   // the delta partials travel in the tail of the all-gather slot: the same, rank-independent bound on every rank
      P->w_tail = 8u * (uint32_t)grid_for(P->shard.slice_width);
      max_partials = std::min<size_t>(max_partials, P->w_tail);
    }
    P->mv_reps = (int)std::max<size_t>(1, (groups + max_partials - 1) / max_partials);
    P->nb_mv = (int)std::max<size_t>(1, (groups + P->mv_reps - 1) / P->mv_reps);
  }
  ok &= P->s_dir.alloc(3 * N, true) == hipSuccess; ok &= P->part_g2.alloc((size_t)2 * P->nb_cam, true) == hipSuccess;
  ok &= P->part_d2.alloc(std::max(P->nb_mv, P->nb_cam), true) == hipSuccess; ok &= P->cg2sc.alloc(1, true) == hipSuccess;
  // (here, not at the first solve: an allocation that fails on one rank only must be part of the create-time agreement)
  if (P->sharded) ok &= P->w_gather.alloc(((size_t)3 * P->shard.slice_width + P->w_tail) * P->shard.world_size, true) == hipSuccess;
  if (!ok) return bail(fail(GSFM_ERR_HIP, "allocating camera buffers failed"));
  {  // cameras touched by at least one edge (Ceres only knows parameter blocks that appear in a residual block)
    std::vector<double> act(NP, 0.0);
    for (uint32_t r = 0; r < P->n_rows; ++r) act[ob + r] = (rp[r + 1] > rp[r]) ? 1.0 : 0.0;
    if (hipMemcpy(P->active.p, act.data(), 8 * NP, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(GSFM_ERR_HIP, "upload active mask"));
  }
  if (int st = prepare_loss(P, nullptr, 0)) return bail(st);
  // connected components among this rank's own edges, as one label per camera (the smallest camera index of its component; untouched
  // cameras label themselves): merged across the ranks below
  std::vector<uint32_t> comp_label;
  DevBuf<double> d_labels;
  if (P->sharded) {
    comp_label.resize(NP);
    for (uint32_t c = 0; c < NP; ++c) comp_label[c] = c;
    auto find = [&](uint32_t v) { while (comp_label[v] != v) { comp_label[v] = comp_label[comp_label[v]]; v = comp_label[v]; } return v; };
    for (uint64_t e = 0; e < n_edges; ++e) { const uint32_t a = find(edge_i[e]), b = find(edge_j[e]); if (a != b) comp_label[a < b ? b : a] = a < b ? a : b; }
    for (uint32_t c = 0; c < NP; ++c) comp_label[c] = find(c);
    if (d_labels.alloc((size_t)P->shard.world_size * NP) != hipSuccess) return bail(fail(GSFM_ERR_HIP, "allocating the component labels failed"));
    std::vector<double> lab(NP);
    for (uint32_t c = 0; c < NP; ++c) lab[c] = (double)comp_label[c];
    if (hipMemcpy(d_labels.p + (size_t)P->shard.rank * NP, lab.data(), 8 * NP, hipMemcpyHostToDevice) != hipSuccess) return bail(fail(GSFM_ERR_HIP, "upload component labels"));
  }
  lap("camera buffers");
  // ---- the agreement (sharded), then collectives only ----
  if (P->sharded) {
    // the two-level preconditioner is used only if every rank chose it (each judged the coherence of its own edges; the wait-and-see mode is single-GPU only)
    const int failed = agree(0.0, (P->coarse_want && !P->coarse_adaptive) ? 0.0 : 1.0);
    if (failed != 0) return bail(fail(GSFM_ERR_COMM, failed > 0 ? "problem creation failed on " + std::to_string(failed) + " other rank(s)" : std::string("the create-time agreement all-reduce failed")));
    if (coarse_votes_against > 0.5) P->coarse_want = 0;
    if (int st = all_gather(P, P->active.p, P->shard.slice_width)) return bail(st);
    if (int st = all_gather(P, d_labels.p, NP)) return bail(st);
    std::vector<double> all((size_t)P->shard.world_size * NP);
    if (hipStreamSynchronize(P->stream) != hipSuccess || hipMemcpy(all.data(), d_labels.p, 8 * all.size(), hipMemcpyDeviceToHost) != hipSuccess)
      return bail(fail(GSFM_ERR_HIP, "active mask / component labels all-gather failed"));
    // A rank of a sharded problem sees only its own edges, so whether the GLOBAL view graph is connected -- which decides the PCG tolerance,
    // see lm_solve -- is worked out here from every rank's local components (union of "c and its local label are connected" over all
    // ranks), identically on every rank.  (Round 2 relied on a flag the partitioner had to set; a raw C-ABI user who forgot it got a looser
    // solve than on one GPU.  The flag is still honoured.)
    std::vector<uint32_t> parent(NP);
    for (uint32_t c = 0; c < NP; ++c) parent[c] = c;
    auto find = [&](uint32_t v) { while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; } return v; };
    for (int r = 0; r < P->shard.world_size; ++r)
      for (uint32_t c = 0; c < NP; ++c) {
        const uint32_t l = (uint32_t)all[(size_t)r * NP + c];
        if (l != c && l < NP) { const uint32_t a = find(c), b = find(l); if (a != b) parent[a < b ? b : a] = a < b ? a : b; }
      }
    std::vector<double> act(NP);
    if (hipMemcpy(act.data(), P->active.p, 8 * NP, hipMemcpyDeviceToHost) != hipSuccess) return bail(fail(GSFM_ERR_HIP, "download active mask"));
    uint32_t comps = 0;
    for (uint32_t c = 0; c < NP; ++c) if (act[c] != 0.0 && find(c) == c) ++comps;
    P->n_components = std::max<uint32_t>(std::max<uint32_t>(1, comps), (P->shard.flags & GSFM_SHARD_DISCONNECTED) ? 2u : 1u);
  }
  *live = nullptr;
  *out = P;
  return GSFM_OK;
}

gsfm_status gsfm_rot_problem_create(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j, const double* rel_aa, int32_t error_type,
                                    const double* cov6, const double* inlier_weight, const gsfm_rot_shard* shard, gsfm_rot_problem** out) {
  // The host-side structure build allocates O(E) vectors and starts threads: an exception (std::bad_alloc, std::system_error) must not
  // cross the C boundary.  (On a sharded problem the peers of a rank that fails THIS way are not told: they wait in the agreement.)
  gsfm_rot_problem* live = nullptr;
  try {
    return problem_create_impl(n_cams, n_edges, edge_i, edge_j, rel_aa, error_type, cov6, inlier_weight, shard, out, &live);
  } catch (const std::exception& e) {
    if (live) gsfm_rot_problem_destroy(live);
    if (out) *out = nullptr;
    return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, std::string("problem creation ran out of host resources: ") + e.what());
  }
}

void gsfm_rot_problem_destroy(gsfm_rot_problem* P) {
  if (!P) return;
  DeviceGuard g(P->device);
  P->timer.destroy();
  P->pcg_graph.reset(); P->pcg2_graph.reset();
  if (P->dense_graph) (void)hipGraphExecDestroy(P->dense_graph);
  if (P->own_stream && P->stream) (void)hipStreamDestroy(P->stream);
  if (P->pin) (void)hipHostFree(P->pin);
  delete P;
}

gsfm_status gsfm_rot_set_stream(gsfm_rot_problem* P, void* s) {
  if (!P) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL problem");
  DeviceGuard g(P->device);
  if (P->own_stream && P->stream) { (void)hipStreamSynchronize(P->stream); (void)hipStreamDestroy(P->stream); }
  P->pcg_graph.reset(); P->pcg_graph.unusable = false; P->pcg2_graph.reset(); P->pcg2_graph.unusable = false;
  if (P->dense_graph) { (void)hipGraphExecDestroy(P->dense_graph); P->dense_graph = nullptr; }
  if (s) { P->stream = (hipStream_t)s; P->own_stream = false; }
  else { if (hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "hipStreamCreate failed"); P->own_stream = true; }
  P->timer.stream = P->stream;
  return GSFM_OK;
}

gsfm_status gsfm_rot_set_loss(gsfm_rot_problem* P, const gsfm_loss_node* prog, int32_t n) {
  if (!P || (n > 0 && !prog)) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  P->cb = nullptr;
  return (gsfm_status)prepare_loss(P, prog, n);
}

gsfm_status gsfm_rot_set_loss_callback(gsfm_rot_problem* P, gsfm_loss_callback fn, void* user) {
  if (!P || !fn) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  if (!P->rho_ext.p) {
    if (P->rho_ext.alloc(3 * P->n_edges_in) != hipSuccess || P->s_ext.alloc(P->n_edges_in) != hipSuccess)
      return (gsfm_status)fail(GSFM_ERR_HIP, "allocating callback-loss buffers failed");
  }
  P->cb = fn; P->cb_user = user;
  return GSFM_OK;
}

gsfm_status gsfm_rot_set_edge_weights(gsfm_rot_problem* P, const double* w) {
  if (!P || !w) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (P->functor != F_AA || P->wmode == W_MATRIX) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "edge weights only apply to the scalar-weight angle-axis types");
  DeviceGuard g(P->device);
  if (P->wmode == W_NONE) {  // ANGLE_AXIS: promote to a scalar-weight problem
    if (P->cost.ws.alloc(P->cost.n) != hipSuccess || P->dir.ws.alloc(P->dir.n) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc weight planes");
    P->wmode = W_SCALAR;
  }
  if (!P->w_orig.p && P->w_orig.alloc(P->n_edges_in) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc weights");
  if (hipMemcpyAsync(P->w_orig.p, w, 8 * P->n_edges_in, hipMemcpyHostToDevice, P->stream) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "upload weights");
  if (P->cost.n) hipLaunchKernelGGL(k_gather_weights, dim3(grid_for(P->cost.n)), dim3(GSFM_BLOCK), 0, P->stream, P->w_orig.p, P->cost.eid.p, P->cost.n, P->cost.ws.p);
  if (P->dir.n) hipLaunchKernelGGL(k_gather_weights, dim3(grid_for(P->dir.n)), dim3(GSFM_BLOCK), 0, P->stream, P->w_orig.p, P->dir.eid.p, P->dir.n, P->dir.ws.p);
  return (gsfm_status)sync_check(P, "set_edge_weights");
}

gsfm_status gsfm_rot_solve(gsfm_rot_problem* P, double* rot, const gsfm_rot_options* opt, gsfm_rot_summary* summary) {
  if (!P || !rot) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  const gsfm_rot_options o = opt ? *opt : default_options();
  gsfm_rot_summary local; if (!summary) summary = &local;
  const double t0 = now_ms();
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  if (int st = lm_solve(P, o, summary)) return (gsfm_status)st;
  if (int st = download_state(P, rot)) return (gsfm_status)st;
  summary->t_total_ms = now_ms() - t0;
  return GSFM_OK;
}

// EstimateRotationsWithSigmaConsensus (estimator.cpp:314-457)
gsfm_status gsfm_rot_solve_sigma_consensus(gsfm_rot_problem* P, double* rot, int32_t iters_num, double sigma_max,
                                           const gsfm_rot_options* opt, gsfm_rot_summary* summary) {
  if (!P || !rot) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (P->error_type != GSFM_ROT_ANGLE_AXIS) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "sigma consensus needs an ANGLE_AXIS problem");
  DeviceGuard g(P->device);
  const gsfm_rot_options o = opt ? *opt : default_options();
  gsfm_rot_summary local, total; if (!summary) summary = &local;
  std::memset(&total, 0, sizeof(total));
  const double t0 = now_ms();
  const MagsacConst c = magsac_const(3);
  const double squared_sigma_max_2 = sigma_max * sigma_max * 2.0;
  const double dof_minus_one_per_two = (c.nu - 1.0) / 2.0;
  const double one_over_sigma = c.C * std::pow(2.0, dof_minus_one_per_two) / sigma_max;
  const double weight_zero = one_over_sigma * (std::tgamma(dof_minus_one_per_two) - c.gk);
  // The scalar-weight planes live on the device for the whole loop, each in its kernel's own order.  There is no weight pass: the first
  // cost sweep and the first linearisation of every inner solve start from exactly the rotations the reference computes the weights at
  // (:378-416), so they compute, store and use them (kernels.hpp, SigmaDev).  The first comparison is against zero weights, like the
  // reference's zero-initialised last_weights (:352-353).
  if (P->wmode == W_NONE) {
    if (P->cost.ws.alloc(P->cost.n) != hipSuccess || P->dir.ws.alloc(P->dir.n) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc weight planes");
    P->wmode = W_SCALAR;
  }
  if (P->cost.n) HIPCHK_S(hipMemsetAsync(P->cost.ws.p, 0, 8 * P->cost.n, P->stream));
  if (!P->sigma_table.p && P->sigma_table.upload(magsac_table(3)) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc sigma consensus buffers");
  if (!P->sigma_sum.p && P->sigma_sum.alloc(2, true) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc sigma consensus buffers");
  P->sigma.table = P->sigma_table.p; P->sigma.table_len = c.n; P->sigma.on = 0; P->sigma.ssm2 = squared_sigma_max_2;
  P->sigma.one_over_sigma = one_over_sigma; P->sigma.gk = c.gk; P->sigma.weight_zero = weight_zero;
  double global_edges = (double)P->n_edges_in;
  if (P->sharded) {
    // every edge is counted in mean |w - w_old| by exactly one rank: its cost owner
    const double mine = (double)P->cost.n;
    HIPCHK_S(hipMemcpyAsync(P->sigma_sum.p + 1, &mine, 8, hipMemcpyHostToDevice, P->stream));
    if (int st = all_reduce(P, P->sigma_sum.p + 1, 1)) return (gsfm_status)st;
    HIPCHK_S(hipMemcpyAsync(&global_edges, P->sigma_sum.p + 1, 8, hipMemcpyDeviceToHost, P->stream));
    if (int st = sync_check(P, "sigma consensus: edge count")) return (gsfm_status)st;
  }
  int outer = 0;
  for (int it = 0; it < iters_num; ++it) {
    ++outer;
    P->sigma_pending_cost = P->sigma_pending_lin = true;   // consumed by the solve's first K1 / K2
    const gsfm_status sst = gsfm_rot_solve(P, rot, &o, summary);
    P->sigma_pending_cost = P->sigma_pending_lin = false;
    if (sst) return sst;
    // sum |w - w_old| was left on the device by that first sweep: read it now, after the solve (the decision it feeds comes after the
    // solve in the reference too, :448)
    if (int st = all_reduce(P, P->sigma_sum.p, 1)) return (gsfm_status)st;
    double avg = 0.0;
    if (hipMemcpyAsync(&avg, P->sigma_sum.p, 8, hipMemcpyDeviceToHost, P->stream) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "read weight change");
    if (int st = sync_check(P, "sigma consensus weights")) return (gsfm_status)st;
    avg /= global_edges;
    if (it == 0) total = *summary;
    else {
      total.num_iterations += summary->num_iterations; total.num_successful_steps += summary->num_successful_steps;
      total.num_unsuccessful_steps += summary->num_unsuccessful_steps; total.num_residual_sweeps += summary->num_residual_sweeps;
      total.num_linearizations += summary->num_linearizations; total.num_cg_iterations += summary->num_cg_iterations;
      total.final_cost = summary->final_cost; total.termination = summary->termination;
      total.final_gradient_max_norm = summary->final_gradient_max_norm; total.final_radius = summary->final_radius;
      total.num_dense_solves += summary->num_dense_solves; total.num_graph_launches += summary->num_graph_launches;
      total.num_collectives += summary->num_collectives; total.num_pcg_collectives += summary->num_pcg_collectives; total.num_pcg_launched += summary->num_pcg_launched;
      total.t_linearize_ms += summary->t_linearize_ms; total.t_sweep_ms += summary->t_sweep_ms; total.t_cg_ms += summary->t_cg_ms;
    }
    total.last_weight_change = avg;
    if (avg <= 1e-7) break;  // :448
  }
  total.outer_iterations = outer; total.t_total_ms = now_ms() - t0;
  *summary = total;
  return GSFM_OK;
}

gsfm_status gsfm_rot_residuals(gsfm_rot_problem* P, const double* rot, double* s_out, double* rho_out, double* r_out, double* cost) {
  if (!P || !rot) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  const size_t E = P->n_edges_in, Ec = P->cost.n, R = (size_t)P->res_dim;
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  // The sweep writes in the problem's own edge order (coalesced); the caller's order is restored here, at the C-ABI edge, on the host:
  // out[edge_order[u]] = device[u].  Edges this rank does not count in the cost (sharded problems) stay zero.
  DevBuf<double> dr; DevBuf<double2> dsr, dr12;
  if (((s_out || rho_out) && dsr.alloc(Ec) != hipSuccess) || (rho_out && dr12.alloc(Ec) != hipSuccess) || (r_out && dr.alloc(R * Ec) != hipSuccess))
    return (gsfm_status)fail(GSFM_ERR_HIP, "alloc outputs");
  CostOutputs out;
  out.srho = dsr.p; out.rho12 = dr12.p; out.r = dr.p;
  if (int st = launch_cost(P, P->q.p, SC_COST, out)) return (gsfm_status)st;
  double h[SC_N];
  if (int st = read_scalars(P, h)) return (gsfm_status)st;
  if (cost) *cost = h[SC_COST];
  const std::vector<uint32_t>& ord = P->h_cost_eid;
  std::vector<double> stage;
  if (s_out || rho_out) {   // device planes: (s, rho) and (rho', rho'') per edge
    stage.resize(4 * Ec);
    if (Ec && (hipMemcpy(stage.data(), dsr.p, 16 * Ec, hipMemcpyDeviceToHost) != hipSuccess || (rho_out && hipMemcpy(stage.data() + 2 * Ec, dr12.p, 16 * Ec, hipMemcpyDeviceToHost) != hipSuccess)))
      return (gsfm_status)fail(GSFM_ERR_HIP, "copy s / rho");
    if (s_out) { std::memset(s_out, 0, 8 * E); for (size_t u = 0; u < Ec; ++u) s_out[ord[u]] = stage[2 * u]; }
    if (rho_out) {
      std::memset(rho_out, 0, 24 * E);
      for (size_t u = 0; u < Ec; ++u) { double* o = rho_out + 3 * (size_t)ord[u]; o[0] = stage[2 * u + 1]; o[1] = stage[2 * Ec + 2 * u]; o[2] = stage[2 * Ec + 2 * u + 1]; }
    }
  }
  if (r_out) {
    stage.resize(R * Ec);
    if (Ec && hipMemcpy(stage.data(), dr.p, 8 * R * Ec, hipMemcpyDeviceToHost) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "copy r");
    std::memset(r_out, 0, 8 * R * E);
    for (size_t u = 0; u < Ec; ++u) for (size_t k = 0; k < R; ++k) r_out[R * (size_t)ord[u] + k] = stage[k * Ec + u];
  }
  return GSFM_OK;
}

int64_t gsfm_rot_edge_order(gsfm_rot_problem* P, uint32_t* order_out, uint64_t cap) {
  if (!P) { fail(GSFM_ERR_INVALID_ARG, "NULL problem"); return -1; }
  const size_t n = P->h_cost_eid.size();
  if (order_out) std::memcpy(order_out, P->h_cost_eid.data(), 4 * std::min<size_t>(n, cap));
  return (int64_t)n;
}

gsfm_status gsfm_rot_linearize(gsfm_rot_problem* P, const double* rot, double* gradient, double* diag_blocks, double* cost) {
  if (!P || !rot) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  DeviceGuard g(P->device);
  const size_t N = P->n_cams;
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  if (int st = launch_cost(P, P->q.p, SC_COST)) return (gsfm_status)st;
  if (int st = launch_lin(P, P->q.p)) return (gsfm_status)st;
  DevBuf<double> dg, dblk;
  if (dg.alloc(3 * N) != hipSuccess || dblk.alloc(9 * N) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc outputs");
  hipLaunchKernelGGL(k_cam_export, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->gD.p, P->n_cams, P->param_dim, dg.p, dblk.p, P->D6.p);
  double h[SC_N];
  if (int st = read_scalars(P, h)) return (gsfm_status)st;
  if (cost) *cost = h[SC_COST];
  std::vector<double> stage;
  if (!P->perm.empty()) stage.resize(9 * N);
  if (gradient) {
    if (hipMemcpy(P->perm.empty() ? gradient : stage.data(), dg.p, 24 * N, hipMemcpyDeviceToHost) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "copy gradient");
    if (!P->perm.empty()) to_external(P, stage.data(), gradient, 3);
  }
  if (diag_blocks) {
    if (hipMemcpy(P->perm.empty() ? diag_blocks : stage.data(), dblk.p, 72 * N, hipMemcpyDeviceToHost) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "copy blocks");
    if (!P->perm.empty()) to_external(P, stage.data(), diag_blocks, 9);
  }
  return GSFM_OK;
}

gsfm_status gsfm_rot_normal_matvec(gsfm_rot_problem* P, const double* v, double* y) {
  if (!P || !v || !y) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (!P->have_lin) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "call gsfm_rot_linearize first");
  DeviceGuard g(P->device);
  const size_t N = P->n_cams;
  // y = T^T B_eta (T v): xcg <- v, p <- T v, Ap <- B p, xcg <- T^T Ap
  if (hipMemcpyAsync(P->xcg.p, to_internal(P, v, 3), 24 * N, hipMemcpyHostToDevice, P->stream) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "upload v");
  hipLaunchKernelGGL(k_cam_apply_T, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->n_cams, P->param_dim, 0, P->xcg.p, P->p.p);
  if (P->lin_is_lap) hipLaunchKernelGGL(k_cam_rotT, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, (const double*)P->p.p, P->q_lin, P->n_cams, P->u_rot.p);
  if (int st = launch_matvec(P, P->D6.p, P->p.p, P->Ap.p, nullptr)) return (gsfm_status)st;
  hipLaunchKernelGGL(k_cam_apply_T, dim3(grid_for(N)), dim3(GSFM_BLOCK), 0, P->stream, P->x.p, P->n_cams, P->param_dim, 1, P->Ap.p, P->xcg.p);
  double* dst = y;
  if (!P->perm.empty()) { P->h_cam.resize(3 * N); dst = P->h_cam.data(); }
  if (hipMemcpyAsync(dst, P->xcg.p, 24 * N, hipMemcpyDeviceToHost, P->stream) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "download y");
  if (int st = sync_check(P, "normal_matvec")) return (gsfm_status)st;
  if (!P->perm.empty()) to_external(P, dst, y, 3);
  return GSFM_OK;
}

gsfm_status gsfm_rot_loss_eval(gsfm_rot_problem* P, const double* s, uint64_t n, double* rho3_out, double* value_out, double* rho1_fast_out) {
  if (!P || (n > 0 && !s)) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (P->cb) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "loss_eval needs a native loss program (a host-callback loss runs on the host)");
  if (n == 0) return GSFM_OK;
  DeviceGuard g(P->device);
  DevBuf<double> ds, d3, dv, d1;
  const int lm = loss_mode(P);
  if (rho1_fast_out && lm == LM_PROGRAM) { for (uint64_t k = 0; k < n; ++k) rho1_fast_out[k] = std::numeric_limits<double>::quiet_NaN(); rho1_fast_out = nullptr; }   // K2 has no fast path for this program
  if (ds.alloc(n) != hipSuccess || (rho3_out && d3.alloc(3 * n) != hipSuccess) || (value_out && dv.alloc(n) != hipSuccess) || (rho1_fast_out && d1.alloc(n) != hipSuccess)) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc");
  HIPCHK_S(hipMemcpyAsync(ds.p, s, 8 * n, hipMemcpyHostToDevice, P->stream));
  const dim3 grid(grid_for(n)), blk(GSFM_BLOCK);
  switch (lm) {   // the specialisation K1 / K2 are dispatched on for this program
    case LM_SIMPLE: hipLaunchKernelGGL(k_loss_eval<LM_SIMPLE>, grid, blk, 0, P->stream, (const DevLoss*)P->d_loss.p, (const double*)ds.p, (size_t)n, d3.p, dv.p, d1.p); break;
    case LM_MAGSAC: hipLaunchKernelGGL(k_loss_eval<LM_MAGSAC>, grid, blk, 0, P->stream, (const DevLoss*)P->d_loss.p, (const double*)ds.p, (size_t)n, d3.p, dv.p, d1.p); break;
    default: hipLaunchKernelGGL(k_loss_eval<LM_PROGRAM>, grid, blk, 0, P->stream, (const DevLoss*)P->d_loss.p, (const double*)ds.p, (size_t)n, d3.p, dv.p, d1.p); break;
  }
  if (rho1_fast_out) HIPCHK_S(hipMemcpyAsync(rho1_fast_out, d1.p, 8 * n, hipMemcpyDeviceToHost, P->stream));
  if (rho3_out) HIPCHK_S(hipMemcpyAsync(rho3_out, d3.p, 24 * n, hipMemcpyDeviceToHost, P->stream));
  if (value_out) HIPCHK_S(hipMemcpyAsync(value_out, dv.p, 8 * n, hipMemcpyDeviceToHost, P->stream));
  return (gsfm_status)sync_check(P, "loss_eval");
}

gsfm_status gsfm_rot_edge_sq_norms(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j, const double* rel_aa,
                                   const double* cov6, const double* rot_aa, double max_sq_norm, double* s_out, uint8_t* keep_out,
                                   uint64_t* n_kept, double* kernel_ms) {
  if (n_cams == 0 || n_edges == 0) { if (n_kept) *n_kept = 0; return GSFM_OK; }
  if (!edge_i || !edge_j || !rel_aa || !rot_aa || !s_out) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  for (uint64_t e = 0; e < n_edges; ++e) if (edge_i[e] >= n_cams || edge_j[e] >= n_cams) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "edge with an out-of-range camera index");
  // (no host fallback: like every entry point of this library the sweep runs on the device or fails loudly)
  if (const char* why = no_device_reason("the edge sweep")) return (gsfm_status)fail(GSFM_ERR_NO_DEVICE, why);
  // One device slab, one private stream, stream-ordered copies: no hipDeviceSynchronize (it would stall every problem's stream of the
  // process) and nothing to leak on an error path (the guard below owns stream, events and slab).
  struct Guard {
    hipStream_t s = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr; void* slab = nullptr;
    ~Guard() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); if (slab) (void)hipFree(slab); if (s) (void)hipStreamDestroy(s); }
  } G;
  auto up = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t o_i = 0, o_j = o_i + up(4 * n_edges), o_rel = o_j + up(4 * n_edges), o_cov = o_rel + up(24 * n_edges), o_rot = o_cov + (cov6 ? up(48 * n_edges) : 0),
               o_s = o_rot + up(24 * (size_t)n_cams), o_q = o_s + up(8 * n_edges), o_keep = o_q + up(32 * (size_t)n_cams), o_cnt = o_keep + (keep_out ? up(n_edges) : 0), total = o_cnt + 256;
  HIPCHK_S(hipStreamCreateWithFlags(&G.s, hipStreamNonBlocking));
  HIPCHK_S(hipEventCreate(&G.e0)); HIPCHK_S(hipEventCreate(&G.e1));
  if (hipMalloc(&G.slab, total) != hipSuccess) { G.slab = nullptr; return (gsfm_status)fail(GSFM_ERR_HIP, "allocating the edge sweep buffers failed"); }
  char* base = (char*)G.slab;
  HIPCHK_S(hipMemcpyAsync(base + o_i, edge_i, 4 * n_edges, hipMemcpyHostToDevice, G.s)); HIPCHK_S(hipMemcpyAsync(base + o_j, edge_j, 4 * n_edges, hipMemcpyHostToDevice, G.s));
  HIPCHK_S(hipMemcpyAsync(base + o_rel, rel_aa, 24 * n_edges, hipMemcpyHostToDevice, G.s)); HIPCHK_S(hipMemcpyAsync(base + o_rot, rot_aa, 24 * (size_t)n_cams, hipMemcpyHostToDevice, G.s));
  if (cov6) HIPCHK_S(hipMemcpyAsync(base + o_cov, cov6, 48 * n_edges, hipMemcpyHostToDevice, G.s));
  HIPCHK_S(hipMemsetAsync(base + o_cnt, 0, 8, G.s));
  hipLaunchKernelGGL(k_cam_cache, dim3(grid_for(n_cams)), dim3(GSFM_BLOCK), 0, G.s, (const double*)(base + o_rot), n_cams, 3, (double2*)(base + o_q));
  EdgeSweepArgs a{};
  a.n = n_edges; a.ei = (const uint32_t*)(base + o_i); a.ej = (const uint32_t*)(base + o_j); a.rel_aa = (const double*)(base + o_rel); a.cov6 = cov6 ? (const double*)(base + o_cov) : nullptr;
  a.q = (const double2*)(base + o_q); a.max_sq = max_sq_norm; a.s_out = (double*)(base + o_s); a.keep = keep_out ? (uint8_t*)(base + o_keep) : nullptr; a.n_kept = (unsigned long long*)(base + o_cnt);
  HIPCHK_S(hipEventRecord(G.e0, G.s));
  hipLaunchKernelGGL(k_edge_sweep, dim3(grid_for(n_edges)), dim3(GSFM_BLOCK), 0, G.s, a);
  HIPCHK_S(hipEventRecord(G.e1, G.s));
  HIPCHK_S(hipMemcpyAsync(s_out, base + o_s, 8 * n_edges, hipMemcpyDeviceToHost, G.s));
  if (keep_out) HIPCHK_S(hipMemcpyAsync(keep_out, base + o_keep, n_edges, hipMemcpyDeviceToHost, G.s));
  unsigned long long cnt = 0;
  HIPCHK_S(hipMemcpyAsync(&cnt, base + o_cnt, 8, hipMemcpyDeviceToHost, G.s));
  HIPCHK_S(hipStreamSynchronize(G.s));
  HIPCHK_S(hipGetLastError());
  float ms = 0; (void)hipEventElapsedTime(&ms, G.e0, G.e1);
  if (kernel_ms) *kernel_ms = ms;
  if (n_kept) *n_kept = keep_out ? (uint64_t)cnt : n_edges;
  return GSFM_OK;
}

int64_t gsfm_rot_count_components(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j) {
  if ((n_edges > 0 && (!edge_i || !edge_j)) || n_cams >= 0x7fffffffu) { fail(GSFM_ERR_INVALID_ARG, "bad argument"); return -1; }
  for (uint64_t e = 0; e < n_edges; ++e) if (edge_i[e] >= n_cams || edge_j[e] >= n_cams) { fail(GSFM_ERR_INVALID_ARG, "edge with an out-of-range camera index"); return -1; }
  return (int64_t)count_components(n_cams, n_edges, edge_i, edge_j);
}

int32_t gsfm_rot_locality_order(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j, uint32_t* perm_out) {
  if (!perm_out || (n_edges > 0 && (!edge_i || !edge_j)) || n_cams >= 0x7fffffffu || n_edges >= 0x7fffffffull) { fail(GSFM_ERR_INVALID_ARG, "bad argument"); return -1; }
  std::vector<uint32_t> ptr((size_t)n_cams + 1, 0), adj(2 * n_edges);
  for (uint64_t e = 0; e < n_edges; ++e) {
    if (edge_i[e] >= n_cams || edge_j[e] >= n_cams || edge_i[e] == edge_j[e]) { fail(GSFM_ERR_INVALID_ARG, "edge with an out-of-range or repeated camera index"); return -1; }
    ptr[edge_i[e] + 1]++; ptr[edge_j[e] + 1]++;
  }
  for (size_t c = 0; c < n_cams; ++c) ptr[c + 1] += ptr[c];
  {
    std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1);
    for (uint64_t e = 0; e < n_edges; ++e) { adj[fill[edge_i[e]]++] = edge_j[e]; adj[fill[edge_j[e]]++] = edge_i[e] | 0x80000000u; }
  }
  std::vector<uint32_t> perm;
  const bool adopted = n_edges > 0 && reorder_for_locality(n_cams, n_edges, edge_i, edge_j, ptr, adj, &perm);
  for (uint32_t c = 0; c < n_cams; ++c) perm_out[c] = adopted ? perm[c] : c;
  return adopted ? 1 : 0;
}

int32_t gsfm_rot_get_trace(gsfm_rot_problem* P, double* out, int32_t cap_rows) {
  if (!P) return 0;
  const int rows = (int)(P->trace.size() / GSFM_ROT_TRACE_COLS);
  const int m = std::min(rows, cap_rows);
  if (out && m > 0) std::memcpy(out, P->trace.data(), sizeof(double) * GSFM_ROT_TRACE_COLS * m);
  return rows;
}

gsfm_status gsfm_rot_time_sweep(gsfm_rot_problem* P, const double* rot, int32_t reps, double* mean_ms) {
  if (!P || !rot || !mean_ms || reps <= 0) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "bad argument");
  if (P->cb) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "time_sweep needs a native loss");
  DeviceGuard g(P->device);
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  const CostArgs a = cost_args(P, P->q.p);
  for (int k = 0; k < 3; ++k) if (dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost)) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "dispatch");
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "event create");
  (void)hipEventRecord(e0, P->stream);
  for (int k = 0; k < reps; ++k) dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost);
  (void)hipEventRecord(e1, P->stream);
  int st = sync_check(P, "time_sweep");
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *mean_ms = ms / reps;
  return (gsfm_status)st;
}

gsfm_status gsfm_rot_time_sweep_variants(gsfm_rot_problem* P, const double* rot, int32_t reps, double* out_ms8) {
  if (!P || !rot || !out_ms8 || reps <= 0) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "bad argument");
  if (P->cb) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "time_sweep_variants needs a native loss");
  DeviceGuard g(P->device);
  const size_t Ec = P->cost.n;
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  DevBuf<double> ds; DevBuf<double2> dsr, dr12;
  if (ds.alloc(Ec, true) != hipSuccess || dsr.alloc(Ec, true) != hipSuccess || dr12.alloc(Ec, true) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "alloc outputs");
  const CostArgs base = cost_args(P, P->q.p);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "event create");
  auto timed = [&](auto&& launch, double* out) -> int {   // two rounds, the faster one counts (a round now and then is hit by something else on the box)
    double best = 0.0;
    for (int round = 0; round < 2; ++round) {
      for (int k = -2; k < reps; ++k) { if (k == 0) (void)hipEventRecord(e0, P->stream); launch(); }
      (void)hipEventRecord(e1, P->stream);
      if (int st = sync_check(P, "time_sweep_variants")) return st;
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      if (round == 0 || ms / reps < best) best = ms / reps;
    }
    *out = best; return 0;
  };
  int st = 0;
  for (int k = 0; k < 8; ++k) out_ms8[k] = 0.0;
  { CostArgs a = base; st = timed([&] { dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost); }, &out_ms8[0]); }                                   // trial cost: rho value only
  if (!st) { CostArgs a = base; a.srho_out = dsr.p; a.rho12_out = dr12.p; st = timed([&] { dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost); }, &out_ms8[1]); }  // s + rho triple
  if (!st) { CostArgs a = base; a.s_out = ds.p; a.s_only = 1; st = timed([&] { dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost); }, &out_ms8[2]); }   // s only (callback pass 1)
  if (!st) { CostArgs a = base; a.rho1_out = ds.p; st = timed([&] { dispatch<CostArgs, CostLauncher>(P, a, P->nb_cost); }, &out_ms8[3]); }             // the reweight sweep of SURVEY 8(d): rho' out
  if (!st && P->functor == F_AA && P->wmode == W_SCALAR && P->cost.ws.p && P->dir.ws.p) {
    // sigma consensus: the first cost sweep / linearisation of an inner solve with the weight computation fused in (estimator.cpp:400-416),
    // against the same kernels without it.  The sweeps overwrite the problem's own weight planes: save and restore them.
    const MagsacConst c = magsac_const(3);
    if (!P->sigma_table.p && P->sigma_table.upload(magsac_table(3)) != hipSuccess) st = fail(GSFM_ERR_HIP, "alloc");
    if (!st && !P->sigma_sum.p && P->sigma_sum.alloc(2, true) != hipSuccess) st = fail(GSFM_ERR_HIP, "alloc");
    DevBuf<double> keep_c, keep_d;
    if (!st && (keep_c.alloc(P->cost.n) != hipSuccess || keep_d.alloc(P->dir.n) != hipSuccess)) st = fail(GSFM_ERR_HIP, "alloc");
    if (!st) {
      const double sigma_max = 0.02, one_over_sigma = c.C * std::pow(2.0, (c.nu - 1.0) / 2.0) / sigma_max;
      SigmaDev sg{};
      sg.table = P->sigma_table.p; sg.table_len = c.n; sg.on = 1; sg.ssm2 = 2.0 * sigma_max * sigma_max; sg.one_over_sigma = one_over_sigma;
      sg.gk = c.gk; sg.weight_zero = one_over_sigma * (std::tgamma((c.nu - 1.0) / 2.0) - c.gk);
      (void)hipMemcpyAsync(keep_c.p, P->cost.ws.p, 8 * P->cost.n, hipMemcpyDeviceToDevice, P->stream);
      (void)hipMemcpyAsync(keep_d.p, P->dir.ws.p, 8 * P->dir.n, hipMemcpyDeviceToDevice, P->stream);
      const SigmaDev keep_sigma = P->sigma;
      P->sigma = sg;
      st = timed([&] { P->sigma_pending_cost = true; (void)launch_cost(P, P->q.p, SC_COST); }, &out_ms8[4]);    // K1 with the weights fused in (+ the two scalar reductions)
      if (!st) st = timed([&] { (void)launch_cost(P, P->q.p, SC_COST); }, &out_ms8[5]);                          // the same sweep without
      if (!st) st = timed([&] { P->sigma_pending_lin = true; (void)launch_lin(P, P->q.p); }, &out_ms8[6]);       // K2 with the weights fused in
      if (!st) st = timed([&] { (void)launch_lin(P, P->q.p); }, &out_ms8[7]);                                    // K2 without
      P->sigma = keep_sigma; P->sigma_pending_cost = P->sigma_pending_lin = false;
      (void)hipMemcpyAsync(P->cost.ws.p, keep_c.p, 8 * P->cost.n, hipMemcpyDeviceToDevice, P->stream);
      (void)hipMemcpyAsync(P->dir.ws.p, keep_d.p, 8 * P->dir.n, hipMemcpyDeviceToDevice, P->stream);
      if (int s2 = sync_check(P, "time_sweep_variants restore")) st = st ? st : s2;
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return (gsfm_status)st;
}

gsfm_status gsfm_rot_time_kernels(gsfm_rot_problem* P, const double* rot, int32_t reps, double* out_ms4) {
  if (!P || !rot || !out_ms4 || reps <= 0) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "bad argument");
  if (P->cb) return (gsfm_status)fail(GSFM_ERR_UNSUPPORTED, "time_kernels needs a native loss");
  DeviceGuard g(P->device);
  const gsfm_rot_options o = default_options();
  if (int st = upload_state(P, rot)) return (gsfm_status)st;
  if (int st = launch_lin(P, P->q.p)) return (gsfm_status)st;
  launch_prep(P, o, o.initial_trust_region_radius, true);
  if (int st = sync_check(P, "time_kernels setup")) return (gsfm_status)st;
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (gsfm_status)fail(GSFM_ERR_HIP, "event create");
  for (int which = 0; which < 4; ++which) {
    out_ms4[which] = 0.0;
    if (which == 3) continue;   // (reserved)
    for (int round = 0; round < 2; ++round) {   // two rounds of `reps` launches each, the faster round's mean counts
      for (int k = -2; k < reps; ++k) {  // two warm-up launches
        if (k == 0) (void)hipEventRecord(e0, P->stream);
        if (which == 0) { if (int st = launch_cost(P, P->q.p, SC_COST)) return (gsfm_status)st; }
        else if (which == 1) { if (int st = launch_lin(P, P->q.p)) return (gsfm_status)st; }
        else if (which == 2) { if (int st = launch_matvec(P, P->Mblk.p, P->b.p, P->Ap.p, nullptr)) return (gsfm_status)st; }
      }
      (void)hipEventRecord(e1, P->stream);
      if (int st = sync_check(P, "time_kernels")) return (gsfm_status)st;
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (round == 0 || ms / reps < out_ms4[which]) out_ms4[which] = ms / reps;
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return GSFM_OK;
}

gsfm_status gsfm_rot_matvec_bytes(gsfm_rot_problem* P, double* layout_bytes, double* lin_bytes, int32_t* form) {
  if (!P) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL problem");
  const double N = (double)P->n_rows;
  const bool lap = P->lap_capable;
  if (lap && P->cs.active) {
    // mat-vec, per position: its 4- or 6-byte record (column | slot | row count) + body-frame block 48; partial sums written and
    // read once; the gathered vector, the diagonal blocks, p, q in, y out once per camera.  Linearisation, per position: record 8 +
    // q_rel 32 + whitening 48 in, block 48 out; nine partial sums per row and workgroup
    if (layout_bytes) *layout_bytes = (P->cs.cmax ? 52.0 : 54.0) * (double)P->cs.n_pos + 2.0 * 24.0 * P->cs.n_wg * GSFM_COL_RB + (24.0 + 48.0 + 24.0 + 32.0 + 24.0) * N;
    if (lin_bytes) *lin_bytes = (8.0 + 32.0 + (P->wmode == W_MATRIX ? 48.0 : P->wmode == W_SCALAR ? 8.0 : 0.0) + 48.0) * (double)P->cs.n_pos + 2.0 * 72.0 * P->cs.n_wg * GSFM_COL_RB + (32.0 + 72.0) * N;
    if (form) *form = 2;
  } else {
    if (layout_bytes) *layout_bytes = (double)P->dir.n * (lap ? 52.0 : 76.0) + 2.0 * 24.0 * N;
    const double w = P->wmode == W_MATRIX ? 48.0 : P->wmode == W_SCALAR ? 8.0 : 0.0;
    if (lin_bytes) *lin_bytes = (double)P->dir.n * (4.0 + 32.0 + w + (lap ? 48.0 : 72.0)) + (32.0 + 72.0) * N;
    if (form) *form = lap ? 1 : 0;
  }
  return GSFM_OK;
}

gsfm_status gsfm_rot_sweep_bytes(gsfm_rot_problem* P, double* algorithmic, double* layout) {
  if (!P) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL problem");
  // indices 8 B + measurement 24 B (32 B for the quaternion types) + whitening 48/8/0 B + weight out 8 B
  const double w = P->wmode == W_MATRIX ? 48.0 : P->wmode == W_SCALAR ? 8.0 : 0.0;
  if (algorithmic) *algorithmic = 8.0 + (P->functor == F_AA ? 24.0 : 32.0) + w + 8.0;
  // as laid out: uint2 idx + 32 B quaternion + whitening planes; the weight is consumed in-kernel (no per-edge store)
  if (layout) *layout = 8.0 + 32.0 + w;
  return GSFM_OK;
}

gsfm_status gsfm_cov_estimate(uint64_t n_edges, const uint64_t* match_ptr, const double* matches, const double* intrinsics,
                              const double* rot_in, const double* trans_in, int32_t max_iterations, double* cov9_out,
                              double* rot_out, double* trans_out, int32_t* status_out, int32_t* iters_out, double* kernel_ms) {
  if (!match_ptr || !matches || !intrinsics || !rot_in || !trans_in || !cov9_out || !rot_out || !trans_out || !status_out)
    return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "NULL argument");
  if (n_edges == 0) return GSFM_OK;
  if (const char* why = no_device_reason("the covariance estimator")) return (gsfm_status)fail(GSFM_ERR_NO_DEVICE, why);
  const uint64_t n_matches = match_ptr[n_edges];
  for (uint64_t e = 0; e < n_edges; ++e) if (match_ptr[e + 1] < match_ptr[e]) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "match_ptr must be non-decreasing");
  DevBuf<uint64_t> d_ptr; DevBuf<double4> d_m; DevBuf<double> d_K, d_r, d_t, d_cov, d_ro, d_to; DevBuf<int> d_st, d_it;
  bool ok = d_ptr.alloc(n_edges + 1) == hipSuccess && d_m.alloc(std::max<uint64_t>(n_matches, 1)) == hipSuccess && d_K.alloc(6 * n_edges) == hipSuccess &&
            d_r.alloc(3 * n_edges) == hipSuccess && d_t.alloc(3 * n_edges) == hipSuccess && d_cov.alloc(9 * n_edges) == hipSuccess &&
            d_ro.alloc(3 * n_edges) == hipSuccess && d_to.alloc(3 * n_edges) == hipSuccess && d_st.alloc(n_edges) == hipSuccess && d_it.alloc(n_edges) == hipSuccess;
  if (!ok) return (gsfm_status)fail(GSFM_ERR_HIP, "allocating covariance buffers failed");
  HIPCHK_S(hipMemcpy(d_ptr.p, match_ptr, 8 * (n_edges + 1), hipMemcpyHostToDevice));
  if (n_matches) HIPCHK_S(hipMemcpy(d_m.p, matches, 32 * n_matches, hipMemcpyHostToDevice));
  HIPCHK_S(hipMemcpy(d_K.p, intrinsics, 48 * n_edges, hipMemcpyHostToDevice));
  HIPCHK_S(hipMemcpy(d_r.p, rot_in, 24 * n_edges, hipMemcpyHostToDevice));
  HIPCHK_S(hipMemcpy(d_t.p, trans_in, 24 * n_edges, hipMemcpyHostToDevice));
  CovArgs a{};
  a.n_edges = n_edges; a.match_ptr = d_ptr.p; a.matches = d_m.p; a.intr = d_K.p; a.rot_in = d_r.p; a.trans_in = d_t.p;
  a.max_iterations = max_iterations; a.cov9 = d_cov.p; a.rot_out = d_ro.p; a.trans_out = d_to.p; a.status = d_st.p; a.iters = d_it.p;
  hipEvent_t e0, e1;
  HIPCHK_S(hipEventCreate(&e0)); HIPCHK_S(hipEventCreate(&e1));
  const int grid = (int)((n_edges + (GSFM_BLOCK / 64) - 1) / (GSFM_BLOCK / 64));
  HIPCHK_S(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_cov_estimate, dim3(grid), dim3(GSFM_BLOCK), 0, 0, a);
  HIPCHK_S(hipEventRecord(e1, 0));
  HIPCHK_S(hipDeviceSynchronize());
  HIPCHK_S(hipGetLastError());
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (kernel_ms) *kernel_ms = ms;
  HIPCHK_S(hipMemcpy(cov9_out, d_cov.p, 72 * n_edges, hipMemcpyDeviceToHost));
  HIPCHK_S(hipMemcpy(rot_out, d_ro.p, 24 * n_edges, hipMemcpyDeviceToHost));
  HIPCHK_S(hipMemcpy(trans_out, d_to.p, 24 * n_edges, hipMemcpyDeviceToHost));
  HIPCHK_S(hipMemcpy(status_out, d_st.p, 4 * n_edges, hipMemcpyDeviceToHost));
  if (iters_out) HIPCHK_S(hipMemcpy(iters_out, d_it.p, 4 * n_edges, hipMemcpyDeviceToHost));
  return GSFM_OK;
}

int32_t gsfm_magsac_table(int32_t nu, double* out, int32_t cap) {
  if (nu != 3 && nu != 4 && nu != 9) return -1;
  const std::vector<double>& t = magsac_table(nu);
  const int m = std::min((int)t.size(), cap);
  if (out && m > 0) std::memcpy(out, t.data(), 8 * (size_t)m);
  return (int)t.size();
}
gsfm_status gsfm_magsac_constants(int32_t nu, double* C, double* q, double* gk) {
  if (nu != 3 && nu != 4 && nu != 9) return (gsfm_status)fail(GSFM_ERR_INVALID_ARG, "nu must be 3, 4 or 9");
  const MagsacConst c = magsac_const(nu);
  if (C) *C = c.C; if (q) *q = c.q; if (gk) *gk = c.gk;
  return GSFM_OK;
}

}  // extern "C"
