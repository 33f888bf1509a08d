// Host side, part 4b: the LM step of a DISCONNECTED view graph (BASELINE C4: the 14 1DSfM scenes as one problem).  The normal matrix is block
// diagonal, one block per connected component, and the reference's Cholesky factorises it block by block -- every component solved exactly,
// whatever the others' right-hand sides look like (src/GSfM_nonlinear_rotation_estimator.cpp:299-305).  Until round 4 the device ran ONE PCG
// over all components with a global stopping rule, which forced a tolerance of 1e-14 (a component that has converged while the batch iterates
// on is otherwise left with an error that is large against its own right-hand side) and made the one ill-conditioned scene set the iteration
// count of all: 3 167 PCG iterations over 46 LM steps on C4.  Now:
//   * components of at most dense_cholesky_max_cams cameras (default 512: 1 536 unknowns) are factorised EXACTLY, all of them side by side in
//     one chain of launches as long as the largest one's (dense_kernels.hpp, k_chol_*_batch; comp_kernels.hpp) -- the reference's step, to
//     rounding;
//   * the larger components stay with PCG, started on the right-hand side with the factorised components' entries zeroed: those cameras'
//     iterates stay exactly zero, they contribute nothing to any of its scalars, and the solve converges at the rate of the components it is
//     actually solving;
//   * a factorisation that meets a non-positive pivot hands the whole step back to the plain PCG (lm_solve).
// Unsharded problems, and PACKED sharded ones (whole components per rank, problem_create.hpp): there every rank runs this on its own components and the
// step is gathered once per LM iteration (packed_exchange); a sharded problem with cut components keeps one PCG over its replicated vectors.
#pragma once
#include "host_common.hpp"

namespace {

// component index per camera (internal numbering), in order of first appearance; cameras without an edge get 0xffffffff
void component_labels(uint32_t n_cams, uint64_t n_edges, const uint32_t* edge_i, const uint32_t* edge_j, std::vector<uint32_t>* comp_of, std::vector<uint32_t>* size) {
  std::vector<uint32_t> parent(n_cams);
  std::vector<uint8_t> touched(n_cams, 0);
  for (uint32_t c = 0; c < n_cams; ++c) parent[c] = c;
  auto find = [&](uint32_t v) { while (parent[v] != v) { parent[v] = parent[parent[v]]; v = parent[v]; } return v; };
  for (uint64_t e = 0; e < n_edges; ++e) {
    const uint32_t a = find(edge_i[e]), b = find(edge_j[e]);
    touched[edge_i[e]] = touched[edge_j[e]] = 1;
    if (a != b) parent[a < b ? b : a] = a < b ? a : b;
  }
  comp_of->assign(n_cams, 0xffffffffu);
  size->clear();
  std::vector<uint32_t> id_of_root(n_cams, 0xffffffffu);
  for (uint32_t c = 0; c < n_cams; ++c) {
    if (!touched[c]) continue;
    const uint32_t r = find(c);
    if (id_of_root[r] == 0xffffffffu) { id_of_root[r] = (uint32_t)size->size(); size->push_back(0); }
    (*comp_of)[c] = id_of_root[r];
    (*size)[id_of_root[r]]++;
  }
}

// (Re)build the batch for components of at most `cap` cameras.  Returns false if there is nothing to factorise (or no memory for it).
bool comps_build(gsfm_rot_problem* P, int cap) {
  auto& C = P->comps;
  if (C.built_cap == cap) return C.n_items > 0;
  C.built_cap = cap; C.n_items = 0; C.drop_graph();
  if (C.comp_of.size() != P->n_cams || C.size.empty() || (!P->sharded && C.size.size() < 2)) return false;   // (a packed rank may hold ONE of the problem's components)
  const uint32_t NC = (uint32_t)C.size.size();
  std::vector<int32_t> item_of_comp(NC, -1);
  std::vector<CholBatchItem> items;
  std::vector<size_t> offA, offL, offX;
  size_t words = 0;
  uint32_t Tmax = 0, n_dense = 0, pcg_comps = 0;
  // the A tiles of all items first (one memset clears them), then L, then x
  for (uint32_t c = 0; c < NC; ++c) {
    if ((int64_t)C.size[c] > (int64_t)cap) { ++pcg_comps; continue; }
    const uint32_t n = 3 * C.size[c], T = (n + GSFM_CB - 1) / GSFM_CB;
    item_of_comp[c] = (int32_t)items.size();
    CholBatchItem it{};
    it.T = T; it.n = n;
    items.push_back(it);
    offA.push_back(words); words += chol_num_tiles(T) * GSFM_TILE_ELEMS;
    Tmax = std::max(Tmax, T); n_dense += C.size[c];
  }
  if (items.empty()) return false;
  const size_t a_words = words;
  for (auto& it : items) { offL.push_back(words); words += chol_num_tiles(it.T) * GSFM_TILE_ELEMS; }
  for (auto& it : items) { offX.push_back(words); words += (size_t)it.T * GSFM_CB; }
  std::vector<int32_t> cam_item(P->n_cams, -1);
  std::vector<uint32_t> cam_loc(P->n_cams, 0), next(NC, 0);
  for (uint32_t k = 0; k < P->n_cams; ++k) {
    const uint32_t c = C.comp_of[k];
    if (c == 0xffffffffu) continue;
    cam_loc[k] = next[c]++;
    cam_item[k] = item_of_comp[c];
  }
  std::vector<uint32_t> item_ptr(items.size() + 1, 0), item_cams;
  for (uint32_t k = 0; k < P->n_cams; ++k) if (cam_item[k] >= 0) item_ptr[cam_item[k] + 1]++;
  for (size_t i = 0; i < items.size(); ++i) item_ptr[i + 1] += item_ptr[i];
  item_cams.resize(item_ptr.back());
  { std::vector<uint32_t> fill(item_ptr.begin(), item_ptr.end() - 1); for (uint32_t k = 0; k < P->n_cams; ++k) if (cam_item[k] >= 0) item_cams[fill[cam_item[k]]++] = k; }
  if (C.slab.alloc(words, true) != hipSuccess || C.info.alloc(items.size(), true) != hipSuccess || C.active.alloc(items.size(), true) != hipSuccess ||
      C.stepmax.alloc(items.size(), true) != hipSuccess || C.stepprev.alloc(items.size(), true) != hipSuccess || C.frozen.alloc(items.size(), true) != hipSuccess ||
      C.item_ptr.upload(item_ptr) != hipSuccess || C.item_cams.upload(item_cams) != hipSuccess || C.b_pcg.alloc(3 * (size_t)P->n_cams, true) != hipSuccess) {
    (void)hipGetLastError(); C.slab.release(); return false;
  }
  for (size_t i = 0; i < items.size(); ++i) { items[i].A = C.slab.p + offA[i]; items[i].L = C.slab.p + offL[i]; items[i].x = C.slab.p + offX[i]; items[i].info = C.info.p + i; items[i].active = C.active.p + i; }
  if (C.items.upload(items) != hipSuccess || C.cam_item.upload(cam_item) != hipSuccess || C.cam_loc.upload(cam_loc) != hipSuccess) { (void)hipGetLastError(); return false; }
  { std::vector<unsigned long long> inf(items.size(), 0x7ff0000000000000ull); if (hipMemcpy(C.stepmax.p, inf.data(), 8 * items.size(), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(C.stepprev.p, inf.data(), 8 * items.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return false; } }   // nothing measured yet
  C.n_items = (uint32_t)items.size(); C.Tmax = Tmax; C.n_dense_cams = n_dense; C.all_dense = pcg_comps == 0;
  C.a_words = a_words; C.n_pcg_comps = pcg_comps;
  return true;
}

// enqueue: clear + assemble + factorise + substitute, all factorised components side by side
void comps_enqueue_dense(gsfm_rot_problem* P, hipStream_t st) {
  auto& C = P->comps;
  (void)hipMemsetAsync(C.slab.p, 0, 8 * C.a_words, st);
  hipLaunchKernelGGL(k_comp_activity, dim3(C.n_items), dim3(GSFM_BLOCK), 0, st, (const uint32_t*)C.item_ptr.p, (const uint32_t*)C.item_cams.p, (const double*)P->b.p,
                     (const double*)P->Minv.p, (const double*)(P->scal.p + SC_ZBOUND), pcg_abs_floor2(P), C.active.p, C.stepmax.p, C.stepprev.p, C.frozen.p, comp_freeze_below(P), (const double*)(P->scal.p + SC_FREEZE_OK));
  DenseArgs a{};
  a.n_rows = P->n_rows; a.row_ptr = P->row_ptr.p; a.col = P->col.p; a.h0 = P->h0.p; a.h1 = P->h1.p; a.h2 = P->h2.p; a.h3 = P->h3.p; a.h4 = P->h4.p;
  a.Mblk = P->Mblk.p; a.b = P->b.p; a.A = nullptr; a.n = 0; a.T = 0; a.q = P->q_lin; a.lap = P->lin_is_lap; a.info_slot = P->scal.p + SC_DENSE_INFO; a.rcg = P->r.p;
  const CompMap cm{C.cam_item.p, C.cam_loc.p, P->own_begin};
  hipLaunchKernelGGL(k_comp_assemble, dim3((uint32_t)C.item_cams.n), dim3(GSFM_BLOCK), 0, st, a, cm, (const CholBatchItem*)C.items.p, (const uint32_t*)C.item_cams.p);
  // one launch per TWO block columns: their panels beside the update with the two columns before them (dense_kernels.hpp, k_chol_look2_batch; round 6: the
  // fused step's bits at one elimination per block row and column -- the six scenes of C4 side by side 985 -> 660 us per factorisation + solve);
  // GSFM_CHOL_FUSED=1: the fused step
  const char* fused_env = getenv("GSFM_CHOL_FUSED");
  const bool fused = fused_env && fused_env[0] == '1';
  if (!fused) {
    hipLaunchKernelGGL(k_chol_look2_batch<0>, dim3(chol_look2_grid(C.Tmax, 0, false), C.n_items), dim3(256), 0, st, (const CholBatchItem*)C.items.p, 0u);
    for (uint32_t c0 = 2; c0 < C.Tmax; c0 += 2)
      hipLaunchKernelGGL(k_chol_look2_batch<2>, dim3(chol_look2_grid(C.Tmax, c0, true), C.n_items), dim3(256), 0, st, (const CholBatchItem*)C.items.p, c0);
  } else for (uint32_t k = 0; k < C.Tmax; ++k) {
    const uint32_t m = C.Tmax - k, nt = chol_step_tiles_per_wg(m);
    const dim3 grid(chol_step_grid(m, nt), C.n_items);
    if (nt == 3) hipLaunchKernelGGL(k_chol_step_batch<3>, grid, dim3(256), 0, st, (const CholBatchItem*)C.items.p, k);
    else if (nt == 2) hipLaunchKernelGGL(k_chol_step_batch<2>, grid, dim3(256), 0, st, (const CholBatchItem*)C.items.p, k);
    else hipLaunchKernelGGL(k_chol_step_batch<1>, grid, dim3(256), 0, st, (const CholBatchItem*)C.items.p, k);
  }
  constexpr uint32_t GR = 8;
  for (uint32_t g = 0; g * GR < C.Tmax; ++g) {
    hipLaunchKernelGGL(k_chol_back_group_batch<GR>, dim3(1, C.n_items), dim3(64 * GR), 0, st, (const CholBatchItem*)C.items.p, g);
    const uint32_t k1 = C.Tmax - g * GR, k0 = k1 > GR ? k1 - GR : 0;
    if (k0) hipLaunchKernelGGL(k_chol_back_update_batch<GR>, dim3(k0, C.n_items), dim3(32 * GR), 0, st, (const CholBatchItem*)C.items.p, g);
  }
}

// The step of a disconnected problem: *used = false if the path does not apply (the caller then runs its generic one).  On return the step
// vector and the PCG residual are complete and the status word of the scalar block says whether every factorisation went through.
// freeze_ok: this step's damping is not what would make a component's step small (lm_solve: the trust radius is at or above its initial value)
int run_component_step(gsfm_rot_problem* P, const gsfm_rot_options& o, double tol_requested, bool pcg_struggles, bool freeze_ok, bool* used, int* cg, double* cg_rel) {
  *used = false; *cg = 0; *cg_rel = 0.0;
  if ((P->sharded && !P->packed) || P->n_components <= 1 || o.dense_cholesky_max_cams <= 0 || P->cs.active) return 0;
  if (!comps_build(P, o.dense_cholesky_max_cams)) return 0;
  auto& C = P->comps;
  if (C.fresh_solve) {   // the first component step of a solve, whichever LM iteration reaches it (lm_solve sets the flag at its top): nobody is at rest, no step has been measured (+inf)
    C.fresh_solve = false;
    std::vector<unsigned long long> inf(C.n_items, 0x7ff0000000000000ull);
    HIPCHK(hipMemsetAsync(C.frozen.p, 0, sizeof(int) * C.n_items, P->stream));
    HIPCHK(hipMemcpyAsync(C.stepmax.p, inf.data(), 8 * C.n_items, hipMemcpyHostToDevice, P->stream));
    HIPCHK(hipMemcpyAsync(C.stepprev.p, inf.data(), 8 * C.n_items, hipMemcpyHostToDevice, P->stream));
    HIPCHK(hipStreamSynchronize(P->stream));   // (the staging vector dies with this scope)
  }
  hipLaunchKernelGGL(k_set_double, dim3(1), dim3(1), 0, P->stream, P->scal.p + SC_FREEZE_OK, freeze_ok ? 1.0 : 0.0);
  // The factorisations and the PCG solve of the large components touch disjoint outputs and read the same inputs: with something left for PCG
  // the chain of the factorisations runs on a stream of its own beside it (fork behind the damping's kernels, join in front of the scatter).
  if (C.side_state == 0) {
    C.side_state = -1;
    if (hipStreamCreateWithFlags(&C.side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&C.ev_fork, hipEventDisableTiming) == hipSuccess &&
        hipEventCreateWithFlags(&C.ev_join, hipEventDisableTiming) == hipSuccess) C.side_state = 1;
    else { (void)hipGetLastError(); C.drop_side(); C.side_state = -1; }
  }
  const bool beside = !C.all_dense && C.side_state == 1;
  const hipStream_t ds = beside ? C.side : P->stream;
  const int tk = P->timer.begin(T_CG);
  if (C.graph && (C.graph_lap != P->lin_is_lap || C.graph_freeze != comp_freeze_below(P))) C.drop_graph();
  if (beside) { HIPCHK(hipEventRecord(C.ev_fork, P->stream)); HIPCHK(hipStreamWaitEvent(ds, C.ev_fork, 0)); }
  if (!C.graph && o.pcg_hip_graph && !P->pcg_graph.unusable) {
    C.graph_lap = P->lin_is_lap; C.graph_freeze = comp_freeze_below(P);
    hipGraph_t captured = nullptr;
    if (hipStreamBeginCapture(ds, hipStreamCaptureModeThreadLocal) == hipSuccess) {
      comps_enqueue_dense(P, ds);
      if (hipStreamEndCapture(ds, &captured) != hipSuccess || !captured || hipGraphInstantiate(&C.graph, captured, nullptr, nullptr, 0) != hipSuccess) C.graph = nullptr;
      if (captured) (void)hipGraphDestroy(captured);
    }
    if (!C.graph) (void)hipGetLastError();
  }
  if (C.graph) { HIPCHK(hipGraphLaunch(C.graph, ds)); P->graph_launches++; }
  else comps_enqueue_dense(P, ds);
  if (beside) HIPCHK(hipEventRecord(C.ev_join, ds));
  P->timer.end(tk);
  const CompMap cm{C.cam_item.p, C.cam_loc.p, P->own_begin};
  if (!C.all_dense) {
    hipLaunchKernelGGL(k_comp_mask_rhs, dim3(grid_for(P->n_cams)), dim3(GSFM_BLOCK), 0, P->stream, (const double*)P->b.p, cm, P->n_cams, C.b_pcg.p);
    P->b_rhs = C.b_pcg.p;
    if (int st = coarse_build(P, pcg_struggles)) { P->b_rhs = nullptr; return st; }
    // one large component left: the global stopping rule is its own, the requested tolerance stands; several: the round-3 safeguard (1e-14)
    const double tol = C.n_pcg_comps > 1 ? std::min(tol_requested, 1e-14) : tol_requested;
    const bool pcg2 = P->coarse_n == 0 && use_single_reduction(P, o);
    const int st = pcg2 ? run_pcg2(P, o, tol, 0.0, -1, cg, cg_rel) : run_pcg(P, o, tol, 0.0, -1, cg, cg_rel);
    P->b_rhs = nullptr;
    if (st) return st;
    if (*cg_rel <= tol) *cg_rel = std::min(*cg_rel, o.cg_relative_tolerance);   // (held against the caller's tolerance afterwards)
  }
  if (beside) HIPCHK(hipStreamWaitEvent(P->stream, C.ev_join, 0));
  hipLaunchKernelGGL(k_comp_scatter, dim3(grid_for(P->n_cams)), dim3(GSFM_BLOCK), 0, P->stream, cm, (const CholBatchItem*)C.items.p, C.n_items, P->n_cams, C.all_dense ? 1 : 0,
                     P->xcg.p, P->r.p, P->scal.p + SC_DENSE_INFO, (const double*)P->Tinv.p, C.stepmax.p, P->packed ? P->scal.p + SC_COMPBAD : nullptr);
  if (o.verbose) {   // which components were live in this step (a read-back: verbose runs only)
    std::vector<int> act(C.n_items), fr(C.n_items);
    std::vector<unsigned long long> sm(C.n_items);
    HIPCHK(hipMemcpyAsync(act.data(), C.active.p, sizeof(int) * C.n_items, hipMemcpyDeviceToHost, P->stream));
    HIPCHK(hipMemcpyAsync(fr.data(), C.frozen.p, sizeof(int) * C.n_items, hipMemcpyDeviceToHost, P->stream));
    HIPCHK(hipMemcpyAsync(sm.data(), C.stepmax.p, 8 * C.n_items, hipMemcpyDeviceToHost, P->stream));
    HIPCHK(hipStreamSynchronize(P->stream));
    std::string line = "[gsfm] components:";
    for (uint32_t i = 0; i < C.n_items; ++i) { double v; std::memcpy(&v, &sm[i], 8); char b[64]; snprintf(b, sizeof b, " %s%.1e", fr[i] ? "rest:" : act[i] ? "" : "idle:", v); line += b; }
    fprintf(stderr, "%s  (largest camera update of each factorised component, rad; PCG %d iterations)\n", line.c_str(), *cg);
  }
  *used = true;
  return 0;
}

}  // namespace

namespace {
// Packed sharded problem, once per evaluated step: every rank's block of the step and of the PCG residual to everybody (the LM scalars are
// computed replicated from them), and the number of ranks whose component factorisation broke down (they all solve that step again by PCG).
int packed_exchange(gsfm_rot_problem* P) {
  if (int st = all_gather(P, P->xcg.p, (size_t)P->shard.slice_width * 3)) return st;
  if (int st = all_gather(P, P->r.p, (size_t)P->shard.slice_width * 3)) return st;
  return all_reduce(P, P->scal.p + SC_COMPBAD, 1);
}
}  // namespace
