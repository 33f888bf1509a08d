// Host side, part 6: gsfm_rot_problem_create -- replaces the edge loop that fills the ceres::Problem (estimator.cpp:47-65, 110-166, 228-295):
// locality relabelling, block-CSR / column-sorted layouts of the directed entries, cost tiles, uploads, whitening, sharded create-time agreement.
#pragma once
#include "host_common.hpp"

namespace {

// Reverse Cuthill-McKee style relabelling (plain BFS from a minimum-degree camera of every component, reversed).  The p[col]
// and q[col] gathers of K3/K2 are bound by uncoalesced lane requests; when the neighbours of a camera sit within a few
// hundred indices of each other the lanes of a row share 128-byte lines and the gather becomes free (tools/archive/bench_matvec.hip:
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


gsfm_rot_options default_options() { gsfm_rot_options o; gsfm_rot_options_default(&o); return o; }

struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) { if (hipGetDevice(&prev) == hipSuccess && prev != dev) (void)hipSetDevice(dev); else prev = -1; }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

template <typename EidVec>
int upload_planes(gsfm_rot_problem* P, EdgePlanes& pl, const EidVec& eid, const double* d_rel_aa) {
  pl.n = eid.size();
  // (qr1: the third stored component, one double per position, on the W_MATRIX problems -- kernels.hpp, qrel_three; the full quaternion's (z, w) pairs otherwise)
  const bool three = P->q3;
  if (pl.eid.upload(eid) != hipSuccess || pl.qr0.alloc(pl.n) != hipSuccess || pl.qr1.alloc(three ? (pl.n + 1) / 2 : pl.n) != hipSuccess)
    return fail(GSFM_ERR_HIP, "uploading edge planes failed (out of memory?)");
  if (pl.n) hipLaunchKernelGGL(k_build_qrel, dim3(grid_for(pl.n)), dim3(GSFM_BLOCK), 0, P->stream, d_rel_aa, pl.eid.p, pl.n, pl.qr0.p, pl.qr1.p, three ? 1 : 0);
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
  if (nblk == 0 || P->n_cams >= (1u << (31 - GSFM_COL_SLOT_BITS)) - 1u) return 0;   // (camera | slot | count in one 32-bit word: 2^22 cameras at 512 rows per block)
  uint32_t cbits = 1;
  while (((1u << cbits) - 1u) <= P->n_cams) ++cbits;   // cameras 0 .. n_cams - 1 and the all-ones padding value
  const uint32_t cmax = cbits + GSFM_COL_SLOT_BITS <= 28 ? (1u << (32 - GSFM_COL_SLOT_BITS - cbits)) - 1u : 0u, kpad = (1u << cbits) - 1u;   // (RB = 512: counts up to 2^(23 - cbits) - 1 in the word, as before)
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
    // A layout that is ONE round of workgroups (fewer than the 1280 tasks from which the sizes are graded: a rank's share of a sharded problem, a
    // mid-size graph) runs as long as its most loaded CU: about two workgroups per CU -- 500 tasks -- measured best for K3c AND K2c on the
    // shares of the benchmark graph (profiles/r05_rank_wgs.txt: 2 ranks 882 -> 490 tasks K3c 106.6 -> 86.2 us; 8 ranks 400 -> 500 tasks K2c
    // 114.1 -> 104.6 us, K3c 34.6 -> 33.9; 525 or 750 tasks lose 15 %: a third workgroup on some CUs).
    if ((size_t)nblk * nch < 1280) nch = std::max<uint32_t>(1u, (uint32_t)std::lround(500.0 / nblk));
    if (const char* e = getenv("GSFM_COL_WGS")) { const int v = atoi(e); if (v > 0) nch = ((uint32_t)v + nblk - 1) / nblk; }
    C.nch = std::min<uint32_t>(32, std::max<uint32_t>(1, nch));
  }
  if (n_pos == 0 || n_pos >= 0x7fffffffull) return 0;   // (positions are 32-bit in the kernels: stay on the row-major form)
  hvec<uint32_t> h_col(n_pos), h_eid(n_pos), h_kcol(n_pos);   // (every position is written below)
  hvec<uint2> h_meta(n_pos);
  hvec<uint16_t> h_kcnt(n_pos);
  // K3c's 2-byte record (ColLayoutDev::k16): GSFM_K3C_K16=0 keeps the 4-byte one (A/B), =1 forces it whatever its escapes cost
  const char* k16_env = getenv("GSFM_K3C_K16");
  const int k16_mode = k16_env && *k16_env ? atoi(k16_env) : -1;
  hvec<uint16_t> h_k16(k16_mode != 0 ? n_pos : 0);
  hvec<uint32_t> h_kbase(k16_mode != 0 ? n_sub * (SUB / 64) : 0), h_kdel(k16_mode != 0 ? n_pos : 0);
  std::atomic<uint64_t> k16_escapes{0};
  std::vector<ColWg> h_wg((size_t)nblk * C.nch);
  // (graded only where the tasks outnumber the chip's resident workgroups several times over -- K2c holds 512, K3c 1024: with a single round,
  // as on one rank's share of a sharded problem (400 tasks), the kernel takes as long as its LARGEST task, and grading made K3c 33 -> 40 us there)
  const bool graded = C.nch > 1 && (size_t)nblk * C.nch >= (size_t)1280;
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
      // The tasks of a block: its sub-chunks cut into nch ranges of DECREASING size (1.5 x the mean down to 0.5 x), launched chunk-major --
      // the large tasks of all blocks first, the small ones last.  Equal tasks fill the chip in whole rounds (K2c: 2 x 256 resident
      // workgroups, 1764 equal tasks = 3.45 rounds, the last one half empty; measured as a saw-tooth in the task count: 616 us at 1960
      // tasks, 645 at 2156, 612 at 2548: profiles/r04b_wgs_sweep.txt); with graded sizes the tail is as long as the SMALLEST task.
      for (uint32_t c = 0; c < C.nch; ++c) {
        size_t lo, hi;
        if (graded && ns >= 4 * (size_t)C.nch) {
          // cumulative weight of chunks 0 .. c-1 with w_c = 1.5 - c / (nch - 1), total nch
          auto cum = [&](uint32_t k) { return 1.5 * k - 0.5 * (double)k * (k - 1) / (double)(C.nch - 1); };
          lo = (size_t)std::llround((double)ns * cum(c) / (double)C.nch); hi = (size_t)std::llround((double)ns * cum(c + 1) / (double)C.nch);
          if (c + 1 == C.nch) hi = ns;
        } else { lo = ns * c / C.nch; hi = ns * (c + 1) / C.nch; }
        h_wg[(size_t)c * nblk + b] = ColWg{(uint32_t)(sub_off[b] + lo), (uint32_t)(hi - lo), r0, b * C.nch + c};
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
          const uint32_t p = (uint32_t)(e - lo), rc = cnt[p + 1] - cnt[p];   // position p also carries the slot count of ROW p
          if (e < hi) {
            const uint32_t rl = (uint32_t)(ent[e].first & 0xffff), d = ent[e].second;
            h_col[o] = col[d]; h_eid[o] = deid[d]; h_meta[o] = make_uint2(col[d], col_pack(fill[rl]++, rc, rl));
          } else { h_col[o] = GSFM_COL_PAD; h_eid[o] = 0; h_meta[o] = make_uint2(GSFM_COL_PAD, col_pack(pad_slot++, rc, 0)); }   // zero block, a slot no row reads
          h_kcol[o] = (h_meta[o].x == GSFM_COL_PAD ? kpad : (h_meta[o].x & 0x7fffffffu)) | (col_slot(h_meta[o].y) << cbits) | (std::min(rc, cmax) << (cbits + GSFM_COL_SLOT_BITS));
          h_kcnt[o] = (uint16_t)rc;
        }
        if (k16_mode != 0) {   // the same sub-chunk once more: cameras as steps inside each wavefront's 64 positions
          uint32_t prev = 0, esc = 0;
          for (uint32_t p = 0; p < SUB; ++p) {
            const size_t o = base + p;
            const uint32_t cam = h_meta[o].x == GSFM_COL_PAD ? prev : (h_meta[o].x & 0x7fffffffu);   // (position 0 of a sub-chunk is never padding)
            uint32_t step = 0;
            if ((p & 63u) == 0) h_kbase[(sub_off[b] + s) * (SUB / 64) + (p >> 6)] = cam; else step = cam - prev;
            prev = cam;
            const uint32_t rc = h_kcnt[o];
            h_kdel[o] = step;
            esc += (step >= GSFM_K16_DEL_ESC) + (rc >= GSFM_K16_CNT_ESC);
            h_k16[o] = (uint16_t)(col_slot(h_meta[o].y) | (std::min(rc, GSFM_K16_CNT_ESC) << GSFM_COL_SLOT_BITS) | (std::min(step, GSFM_K16_DEL_ESC) << 12));
          }
          k16_escapes.fetch_add(esc, std::memory_order_relaxed);
        }
      }
    }
  });
  C.n_wg = (uint32_t)h_wg.size(); C.n_pos = n_pos; C.cbits = cbits; C.cmax = cmax;
  // The 2-byte record pays where an escape (a step of 15 cameras or more, a row with 7 or more entries in one sub-chunk: each its own 32-byte
  // sector) is rare: below one position in a hundred -- the benchmark graph has 1e-4 --; an allocation that fails leaves the 4-byte record in use.
  if (k16_mode != 0 && (k16_mode > 0 || (double)k16_escapes.load() <= 0.01 * (double)n_pos)) {
    if (C.k16.upload(h_k16) == hipSuccess && C.kbase.upload(h_kbase) == hipSuccess && C.kdel.upload(h_kdel) == hipSuccess) C.k16_active = true;
    else { (void)hipGetLastError(); C.k16.release(); C.kbase.release(); C.kdel.release(); }
  }
  if (C.wg.upload(h_wg) != hipSuccess || C.meta.upload(h_meta) != hipSuccess || (!C.k16_active && C.kcol.upload(h_kcol) != hipSuccess) || C.kcnt.upload(h_kcnt) != hipSuccess ||
      C.part.alloc((size_t)9 * C.n_wg * RB) != hipSuccess) {
    (void)hipGetLastError();
    C = gsfm_rot_problem::ColSort();   // out of memory: the row-major form needs none of this
    return 0;
  }
  if (getenv("GSFM_CREATE_TIMING")) fprintf(stderr, "gsfm create: column-sorted layout, %zu positions, 2-byte record %s (%llu escapes)\n", n_pos, C.k16_active ? "on" : "off", (unsigned long long)k16_escapes.load());
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
  // The agreement also carries the number of edges this rank counts in the cost: their sum -- the problem's edge count, the same exact
  // double on every rank -- is what every LM decision that depends on "how many edges" is taken from (solver_lm.hpp: the staircase band of
  // the forcing schedule).  A rank-local count there would let one rank restart while the others enter a collective (round-5 advisor).
  auto agree = [&](double my_flag, double my_vote) -> int {   // number of ranks that failed, or -1 if the agreement itself could not be run
    agreed = true;
    if (!P->sharded) return 0;
    double h[3] = {my_flag, my_vote, (double)P->cost.n};
    if (agree_buf.alloc(3) != hipSuccess || hipMemcpy(agree_buf.p, h, 24, hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (all_reduce(P, agree_buf.p, 3) != 0) return -1;
    if (hipStreamSynchronize(P->stream) != hipSuccess || hipMemcpy(h, agree_buf.p, 24, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    coarse_votes_against = h[1];
    P->cost_n_global = h[2];
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
  if (hipHostMalloc(&P->pin, 512, hipHostMallocDefault) != hipSuccess) { P->pin = nullptr; (void)hipGetLastError(); }   // (read_back then copies to pageable memory)

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
  if (!P->sharded) {
    P->n_components = std::max<uint32_t>(1, count_components(n_cams, n_edges, edge_i, edge_j));
    if (P->n_components > 1) component_labels(n_cams, n_edges, edge_i, edge_j, &P->comps.comp_of, &P->comps.size);   // (internal numbering: solver_components.hpp)
  }
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
  P->timer.apply_rule(nd);
  {  // the measurement planes as three quaternion components (kernels.hpp, qrel_three): covariance-whitened problems whose sweeps are bound by the
     // stream -- at least a million edges held by this process; GSFM_QREL3=0/1 overrides (tests force it on small graphs)
    const char* e = getenv("GSFM_QREL3");
    P->q3 = GSFM_QREL3 != 0 && P->wmode == W_MATRIX && (e && *e ? atoi(e) != 0 : n_edges >= (uint64_t)1000000);
  }
  {  // K2c / K3c, the column-sorted layout of the directed entries: for large graphs whose rows offer the gathers no locality -- i.e. where
     // neither the relabelling nor the two-level preconditioner (both for spatially coherent graphs) applies.  GSFM_K3_COLSORT=0/1 overrides.
    const char* env = getenv("GSFM_K3_COLSORT");
    const int mode = env && *env ? atoi(env) : -1;
    const bool lap_ok = (P->functor == F_AA || P->functor == F_QCOS) && !(getenv("GSFM_LAPLACIAN") && atoi(getenv("GSFM_LAPLACIAN")) == 0);
    // ROUND 6: no density condition any more.  Rounds 3-5 also asked for 512 rows x mean degree >= cameras / 2 ("sparser blocks gain nothing": the
    // gathers of a block then share no lines) -- a rule set when the layout was first built and never re-measured.  Measured on the current kernels
    // (tools/r06_density_probe.py, profiles/r06_density_rule.txt; same generator, default choice against the forced layout, whole solves):
    //   400k cameras / 40 M edges   86.8 -> 52.1 ms   (mat-vec 19.2 -> 9.2 ns per 1000 directed entries; C5: 9.35)
    //   800k / 80 M                198.2 -> 121.7 ms   (22.5 -> 11.7)        1.5 M / 150 M   423.8 -> 245.1 ms   (24.3 -> 11.4)
    //   1 M / 10 M (degree 20)     376 -> 244 ms       500k / 12.5 M (degree 50)   51.1 -> 32.8 ms     2 M / 10 M (degree 10)   12.6 -> 9.5 s
    // with the final cost equal to the last bit every time.  (Why the sorted order wins even where a wavefront's 64 gathers touch 64 different
    // lines was not chased with counters: those lines are neighbours in memory, a row-major row's are scattered over the whole vector.)
    // What remains of the rule: at least 1 M directed entries (below, the launch-latency regime, the row-major
    // kernels with their 2-kernel PCG iteration are the faster ones) and no locality found by the relabelling / no coarse space (coherent graphs).
    if (lap_ok && nd > 0 && (mode > 0 || (mode < 0 && nd >= (size_t)1000000 && P->coarse_want == 0 && P->perm.empty()))) {
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
  ok &= P->xcg.alloc(3 * NP, true) == hipSuccess; ok &= P->r.alloc(3 * NP, true) == hipSuccess; ok &= P->z.alloc(3 * N) == hipSuccess;   // (xcg, r: padded like p / Ap -- a packed sharded problem all-gathers them)
  ok &= P->p.alloc(3 * NP, true) == hipSuccess; ok &= P->Ap.alloc(3 * NP, true) == hipSuccess; ok &= P->u_rot.alloc(3 * NP, true) == hipSuccess;
  {
    const char* env = getenv("GSFM_LAPLACIAN");   // =0: keep the general 9-value blocks (A/B measurements)
    P->lap_capable = (P->functor == F_AA || P->functor == F_QCOS) && !(env && atoi(env) == 0);
    P->lap = P->lap_capable;
  }
  ok &= P->part_a.alloc(P->nb_cam) == hipSuccess; ok &= P->part_b.alloc(P->nb_cam) == hipSuccess;
  ok &= P->part_cam.alloc((size_t)6 * P->nb_cam) == hipSuccess; ok &= P->part_gauge.alloc((size_t)9 * P->nb_cam) == hipSuccess; ok &= P->part_cost.alloc((size_t)2 * P->nb_cost) == hipSuccess;
  ok &= P->scal.alloc(SC_ALL, true) == hipSuccess; ok &= P->cgsc.alloc(1, true) == hipSuccess;
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
    // PACKED (round 5; SURVEY 8(e): "whole components can be packed per GPU => zero cross-GPU coupling except the scalar cost"): no rank holds an
    // edge that leaves its own slice -- every camera a rank's edges touch, and the label they gave it, lies in that rank's slice.  Read off the
    // gathered labels, so every rank arrives at the same verdict without another collective.  The normal matrix is then block diagonal ACROSS
    // the ranks and each rank solves its own block with its own PCG (solver_pcg.hpp): no collective inside the PCG loop at all.
    bool packed = P->n_components > 1;
    for (int r = 0; r < P->shard.world_size && packed; ++r) {
      const uint64_t lo = std::min<uint64_t>((uint64_t)r * P->shard.slice_width, NP), hi = std::min<uint64_t>((uint64_t)(r + 1) * P->shard.slice_width, NP);
      for (uint32_t c = 0; c < NP; ++c) {
        const uint32_t l = (uint32_t)all[(size_t)r * NP + c];
        if (l != c && !(c >= lo && c < hi && l >= lo && l < hi)) { packed = false; break; }
      }
    }
    P->packed = packed;
    if (packed) {   // this rank's own components, for the per-component step (solver_components.hpp): labels of its own cameras that carry an edge
      auto& C = P->comps;
      C.comp_of.assign(n_cams, 0xffffffffu); C.size.clear();
      std::vector<uint32_t> id_of_root(NP, 0xffffffffu);
      for (uint32_t c = P->own_begin; c < P->own_end; ++c) {
        if (act[c] == 0.0) continue;
        const uint32_t r = comp_label[c];
        if (id_of_root[r] == 0xffffffffu) { id_of_root[r] = (uint32_t)C.size.size(); C.size.push_back(0); }
        C.comp_of[c] = id_of_root[r];
        C.size[id_of_root[r]]++;
      }
    }
    if (packed) P->coarse_want = 0;   // (its coarse matrix is an all-reduce per LM step and its use a decision taken from the iteration counts, which now differ from rank to rank)
  }
  *live = nullptr;
  *out = P;
  return GSFM_OK;
}

}  // namespace
