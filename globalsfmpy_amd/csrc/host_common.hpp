// Host side of libgsfm_rot.so, part 1 of 6 (one translation unit: gsfm_rot.hip includes the parts in order): error plumbing, host threads,
// MAGSAC constants, device buffers, the event timer and the problem object every other part works on.
#pragma once

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <exception>
#include <stdexcept>
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
#include "comp_kernels.hpp"
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
// An exception in any thread (std::bad_alloc of a per-thread table, say) is carried to the caller after ALL threads have been joined: thrown
// from a std::thread body it would call std::terminate, and thrown on the caller with joinable threads alive the vector's destructor would.
template <typename F> void parallel_run(int T, F body) {
  if (T <= 1) { body(0, 1); return; }
  std::vector<std::exception_ptr> err((size_t)T);
  std::vector<std::thread> th;
  struct Joiner { std::vector<std::thread>& v; ~Joiner() { for (auto& x : v) if (x.joinable()) x.join(); } } joiner{th};
  auto guarded = [&body, &err, T](int t) { try { body(t, T); } catch (...) { err[(size_t)t] = std::current_exception(); } };
  try {
    for (int t = 1; t < T; ++t) th.emplace_back(guarded, t);
  } catch (...) { err[0] = std::current_exception(); }   // (std::system_error: no more threads -- the ones started are joined below, their ranges stay undone)
  if (!err[0]) guarded(0);
  for (auto& x : th) x.join();
  if ((int)th.size() != T - 1 && !err[0]) err[0] = std::make_exception_ptr(std::runtime_error("could not start the host threads of the structure build"));
  for (auto& e : err) if (e) std::rethrow_exception(e);
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
  bool mute = false;   // no events while set (the device-controlled exact-step pipeline: solver_lm.hpp)
  bool mute_all = false;   // the problem is too small to afford them (apply_rule)
  double acc[3] = {0, 0, 0};
  hipStream_t stream = nullptr;
  // Every begin / end is an event record on the solver's stream (~3 us each, ten to twelve per LM iteration).  Problems whose kernels take
  // hundreds of microseconds do not notice; in the latency regime they are 11 % of a solve (10k cameras / 200k edges: 3.69 -> 3.29 ms;
  // Madrid 34.3 -> 32.6 ms).  Rule (apply_rule, at problem creation): timers from 2 M directed entries on; below, the summary's t_*_ms
  // stay zero.  GSFM_PHASE_TIMERS=1 / 0 forces them on / off.
  void init() { ok = true; for (int k = 0; k < 2 * NPAIR; ++k) if (hipEventCreate(&ev[k]) != hipSuccess) ok = false; }
  void apply_rule(size_t directed_entries) {
    const char* e = getenv("GSFM_PHASE_TIMERS");
    const bool want = e && *e ? atoi(e) != 0 : directed_entries >= (size_t)2000000;
    if (!want) mute_all = true;
  }
  void destroy() { if (ok) for (int k = 0; k < 2 * NPAIR; ++k) (void)hipEventDestroy(ev[k]); ok = false; }
  int begin(int category) {
    if (!ok || mute || mute_all) return -1;
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

  bool q3 = false;            // the measurement planes hold three quaternion components, 24 B (kernels.hpp qrel_three: W_MATRIX problems of >= 1 M edges)
  EdgePlanes cost;            // cost-owned edges
  double cost_n_global = 0.0; // edges counted in the cost over ALL ranks (= cost.n unsharded; sharded: summed in the create-time agreement, identical on every rank)
  DevBuf<uint2> cost_idx;
  DevBuf<CostTile> cost_tiles;
  // Laplacian form of the normal matrix (kernels.hpp, lin_rows): chosen per linearisation; u_rot = R^T p for the mat-vec
  const double2* q_lin = nullptr;   // quaternions the current blocks were linearised at
  bool lap = false, lap_capable = false, lin_is_lap = false;   // lin_is_lap: what the stored blocks currently are
  DevBuf<double> u_rot;
  // locality relabelling adopted at create (empty = identity): internal id = perm[external id]
  std::vector<uint32_t> perm;
  std::vector<double> h_cam;   // staging for permuted per-camera transfers
  DevBuf<uint32_t> d_perm;     // perm on the device (gsfm_rot_solve_resident; uploaded on first use)
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
  const double* b_rhs = nullptr;   // non-null: the right-hand side PCG starts from instead of b (solver_components.hpp: b with the factorised components zeroed)
  DevBuf<double> w_gather;   // sharded single-reduction PCG: per rank [slice of A u | delta partials of its rows] (run_pcg2)
  uint32_t w_tail = 0;
  DevBuf<Cg2Scalars> cg2sc;
  // two-level preconditioner (kernels.hpp, k_coarse_*): aggregates wanted (0 = off, decided at create) / in use for the current LM step
  uint32_t coarse_want = 0, coarse_n = 0, coarse_chunk = 0;
  bool coarse_adaptive = false;     // use it only once a block-Jacobi PCG solve of the run has needed more than 150 iterations
  DevBuf<double> coarseA, coarseAinv, coarse_rc, coarse_xc, coarse_scale, coarse_part;
  std::vector<double> h_coarse, h_coarse_inv;
  void* pin = nullptr;              // 512 B of pinned host memory: [0, 256) staging for the small read-backs of the solve loop (read_back), [256, 264) the deferred gradient norm (lm_solve)
  DevBuf<double> denseA, denseL, dense_x;
  // Disconnected view graph (unsharded): the small components are factorised exactly, side by side (solver_components.hpp)
  struct Components {
    std::vector<uint32_t> comp_of;     // per camera (internal numbering): component index, 0xffffffff for a camera without edges
    std::vector<uint32_t> size;        // cameras per component
    int built_cap = -1;                // the size limit the batch below was built for (-1: not built)
    bool fresh_solve = true;           // set at the top of every solve: the first component step of the solve resets `frozen` / `stepmax` (not "LM iteration 1": a step can reach this path later, e.g. behind a failed dense factorisation)
    uint32_t n_items = 0, Tmax = 0, n_dense_cams = 0, n_pcg_comps = 0;
    size_t a_words = 0;                // doubles of the A tiles (the head of the slab: one memset clears them)
    bool all_dense = false;            // every camera with an edge lies in a factorised component: no PCG at all
    DevBuf<int32_t> cam_item;
    DevBuf<uint32_t> cam_loc;
    DevBuf<CholBatchItem> items;
    DevBuf<double> slab, b_pcg;
    DevBuf<int> info, active;          // per item: factorisation status; has anything to solve in this LM step (k_comp_activity)
    DevBuf<uint32_t> item_ptr, item_cams;   // the cameras of every item, item by item
    DevBuf<unsigned long long> stepmax, stepprev;   // per item: bits of the largest camera update (rad) of its last exact step, and of the one measured before it
    DevBuf<int> frozen;                      // per item: put to rest for the remainder of the solve (k_comp_activity)
    double graph_freeze = -1.0;
    hipGraphExec_t graph = nullptr;
    bool graph_lap = false;
    void drop_graph() { if (graph) (void)hipGraphExecDestroy(graph); graph = nullptr; }
    // the factorisations run on a stream of their own, beside the PCG solve of the large components (fork / join by events)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int side_state = 0;   // 0 not tried, 1 usable, -1 unavailable
    void drop_side() {
      if (ev_fork) (void)hipEventDestroy(ev_fork);
      if (ev_join) (void)hipEventDestroy(ev_join);
      if (side) (void)hipStreamDestroy(side);
      side = nullptr; ev_fork = ev_join = nullptr; side_state = 0;
    }
  } comps;
  uint64_t loss_epoch = 0;                // bumped whenever the loss (and with it the choice of kernels) changes
  hipGraphExec_t dense_graph = nullptr;   // zero + assemble + blocked Cholesky + solve, captured once
  bool dense_graph_lap = false;           // form of the blocks the captured assemble kernel expects
  int nb_mv = 1, mv_reps = 1;
  bool packed = false;        // sharded: no rank holds an edge that leaves its slice (whole components per rank): every rank runs its own PCG, no collective in the loop
  bool pcg_local = false;     //   ... set while such a PCG runs: the mat-vec does not gather
  DevBuf<double> b_own;       //   ... its right-hand side: b with the other ranks' cameras zeroed
  bool component_rest = true; // gsfm_rot_options::component_rest of the solve in hand (lm_solve): converged components of a disconnected problem are put to rest
  uint32_t n_components = 1;  // connected components of the view graph (1 when sharded: a rank sees only its own edges)
  DevBuf<double> part_a, part_b, part_cost, part_cam, part_gauge, scal;
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
    // K3c's 2-byte delta-coded record (colsort_kernels.hpp, ColLayoutDev::k16): built where its escapes are rare
    bool k16_active = false;
    DevBuf<uint16_t> k16;
    DevBuf<uint32_t> kbase, kdel;
    ColLayoutDev dev() const { return ColLayoutDev{wg.p, meta.p, kcol.p, kcnt.p, cbits, cmax, n_wg, nch, k16_active ? k16.p : nullptr, kbase.p, kdel.p}; }
  } cs;

  // sigma consensus (gsfm_rot_solve_sigma_consensus): the weights are computed inside the first cost sweep / linearisation of a solve
  SigmaDev sigma{};
  bool sigma_pending_cost = false, sigma_pending_lin = false;
  DevBuf<double> sigma_table, sigma_sum;   // nu = 3 table; [0] = sum |w - w_old| over this rank's cost edges

  bool have_lin = false;
  double* rec_host = nullptr;            // LM control on the device: ring of four per-iteration records in mapped host memory ((CT_N + 1) doubles each; the host polls the stamp)
  double* rec_dev = nullptr;             //   ... and its device address
  double* mail_host = nullptr;           // PCG status mailbox (k_pcg_mail): GSFM_MAIL_WORDS + 1 doubles of mapped host memory, its device address, the device-resident
  double* mail_dev = nullptr;            //   count of posts and the count the host expects next; mail_state: 0 not tried, 1 usable, -1 unavailable
  DevBuf<double> mail_count;
  double mail_expected = 0.0;
  int mail_state = 0;
  bool loss_staircase = false;  // the loss program contains a MAGSAC leaf: the cost is a staircase in every edge's s
  bool loss_cuts_off = false;   // the loss program contains a leaf whose rho' is identically zero beyond a cut-off (Tukey)
  bool fast_lin_ok = true;   // the loss's rho'' is <= 0 for every s (decided from leaf kind AND parameter signs in prepare_loss): K2's alpha = 0 fast path applies
  int graph_launches = 0;
  int n_collectives = 0, n_pcg_collectives = 0, n_pcg_launched = 0;   // issued (or replayed from a graph) since the solve started
  std::vector<double> trace;
  EventTimer timer;
};
