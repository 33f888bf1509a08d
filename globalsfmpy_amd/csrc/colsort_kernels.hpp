// K2c / K3c: linearisation and normal-equation mat-vec on the COLUMN-SORTED layout of the directed entries, for large graphs WITHOUT
// locality (the C5 benchmark graph is a uniformly random expander).
//
// Why: counters on the row-major K3 (profiles/r03_k3_ceiling.md) show that every u[col] gather is its own L1 -> L2 request (32.1 M
// requests per product for 20 M entries + 8 M stream lines, no line shared between lanes), and that the vector L1 sustains only ~100
// requests in flight at a ~555-cycle loaded L2 latency: one 128-byte line per ~5.5 cycles per CU, 295 us for those 32 M lines whatever
// else the kernel does.  The only way below that is fewer lines.
//
// How: rows are grouped into blocks of RB = 512 and the entries of a block are sorted by COLUMN; this order replaces the row-major one
// for all per-entry planes of the problem (measurements, whitening, blocks).  A wavefront then touches 64 neighbouring cameras -- at the C5
// shape 100k cameras / 102k entries per block, ~5 lanes per 128-byte line -- and the gather's L2 requests drop about fivefold; the
// mat-vec becomes a stream of its own 52 B per entry (48 B block + a 4-byte record; 60 when the layout was first built).  Per-row sums stay deterministic and atomic-free: a sub-chunk of SUB entries is
// evaluated entry-parallel, each entry's contribution is written to its slot of a ROW-sorted LDS staging area (2-byte permutation index
// per entry), and after a barrier the lane that owns row r adds the slots of row r, in slot order, to the sums it keeps in registers
// (the number of slots of row r in a sub-chunk rides in spare bits of the record at POSITION r of that sub-chunk -- as many rows as
// positions -- and a workgroup-wide prefix sum turns the counts into slot ranges: no separate index stream).  The sub-chunks of a block
// are dealt to NCH workgroups; a finishing kernel adds the NCH partials of a row in fixed order.  K2c stores the edge blocks in the BODY
// frame, B = R_k^T G R_k (it has R_k at hand), so K3c applies R_k once per row, in its finish, instead of once per entry.
//   measured at C5: prototype on random data (tools/bench_matvec6.hip) row-major 322 us -> 193 + 3 us; in the product 296 -> 182-201 us
//   (profiles/r03_k3c_tuning.txt has every step in between).
#pragma once
#include "kernels.hpp"

namespace gsfm {

#define GSFM_COL_RB 512    // rows per block = positions per sub-chunk = lanes of a K3c workgroup (1024 measured slower: profiles/r04b_rb1024_ab.txt; the variant is gone)
#define GSFM_COL_SUB GSFM_COL_RB
#define GSFM_COL_SLOT_BITS 9   // bits of a slot / a row inside its block; a row's count in a sub-chunk needs one more (0 .. SUB)
static_assert(GSFM_COL_RB == (1 << GSFM_COL_SLOT_BITS), "row blocks of 512 rows");
#ifndef GSFM_COL_EPL
#define GSFM_COL_EPL 1     // K3c: sub-chunks in flight per iteration (entries per lane); -DGSFM_COL_EPL=n: 1 measured best in the product (2: +4..9 %, 3: +8 %)
#endif
#define GSFM_COL_PAD 0xffffffffu   // column value of a padding position (zero block, reads camera 0)

// One workgroup's task: a range of sub-chunks of one row block; `row0` = first row of the block (local to the owned rows); `part` = where its
// partial row sums go (block * nch + chunk: what the finishing kernels index by).  The array is in LAUNCH order, which is not the order of
// `part`: the tasks of a block have unequal sizes and the large ones of all blocks are launched first (problem_create.hpp, build_colsort).
struct ColWg { uint32_t first_sub, n_sub, row0, part; };
struct ColLayoutDev {
  const ColWg* wg;
  const uint2* meta;        // per position p of a sub-chunk: .x = neighbour camera | role << 31 (GSFM_COL_PAD = padding);
                            //   .y = slot of the entry in the row-sorted staging area (bits 0-9) | number of entries ROW p of the block has in this
                            //   sub-chunk (bits 10-19) | row of the entry inside its block (bits 20-28)  -- one 8-byte load per entry
  // The mat-vec's own, narrower copy of what it needs of the record: kcol = camera (all ones in `cbits` bits = padding) | slot << cbits |
  // row count << (cbits + 9), the count saturating at `cmax` = 2^(23 - cbits) - 1, in which case -- and always when cmax == 0: from 2^19
  // cameras on the word has no room for counts (the layout is not built for 2^22 cameras or more) -- the true count is read from kcnt:
  // 4 bytes per position for graphs below 2^19 cameras, 6 above, instead of 8.
  const uint32_t* kcol;
  const uint16_t* kcnt;
  uint32_t cbits, cmax;
  uint32_t n_wg, nch;
  // The 2-byte, delta-coded form of the same record (round 5; built where it pays: problem_create.hpp, build_colsort): the positions of a block are
  // sorted by camera, so the camera of a position is the camera of its wavefront's first lane (kbase: one word per 64 positions) plus the
  // sum of the steps up to it -- a DPP prefix sum over the wavefront.  k16 = slot (bits 0-8) | row count (bits 9-11; 7: read kcnt) | step from
  // the previous lane's camera (bits 12-15; 15: read kdel; lane 0: 0).  Padding positions repeat the last camera (their block is zero and
  // no row reads their slot).  50 B per position instead of 52; the sums are the same numbers added in the same order.  Null: not built.
  const uint16_t* k16;
  const uint32_t* kbase;
  const uint32_t* kdel;
};
#define GSFM_K16_CNT_ESC 7u
#define GSFM_K16_DEL_ESC 15u
// the record's second word: slot (SLOT_BITS) | count of row p (SLOT_BITS + 1) | row of the entry inside its block (SLOT_BITS)
__host__ __device__ constexpr uint32_t col_pack(uint32_t slot, uint32_t rowcount, uint32_t rowl) { return slot | (rowcount << GSFM_COL_SLOT_BITS) | (rowl << (2 * GSFM_COL_SLOT_BITS + 1)); }
__host__ __device__ __forceinline__ uint32_t col_slot(uint32_t y) { return y & ((1u << GSFM_COL_SLOT_BITS) - 1u); }
__host__ __device__ __forceinline__ uint32_t col_rowcount(uint32_t y) { return (y >> GSFM_COL_SLOT_BITS) & ((2u << GSFM_COL_SLOT_BITS) - 1u); }
__host__ __device__ __forceinline__ uint32_t col_rowl(uint32_t y) { return y >> (2 * GSFM_COL_SLOT_BITS + 1); }
// inclusive prefix sum over the 64 lanes of a wavefront: DPP row shifts inside the rows of 16 lanes, then the two row broadcasts (six VALU
// instructions, no LDS traffic)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
  int v = (int)x;
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
  return (uint32_t)v;
}

__device__ __forceinline__ uint2 col_load_meta(const uint2* p) {
  uint2 v;
  v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y);
  return v;
}

// ---- K3c ----------------------------------------------------------------------------------------------------------------
struct ColMatvecArgs {
  ColLayoutDev L;
  const double2 *b0, *b1, *b2;   // body-frame blocks (b00 b01) (b02 b11) (b12 b22); zero at padding positions
  const double* u;          // u_m = R_m^T p_m, 3 per camera
  double* part;             // 3 planes of [n_wg * RB]
  const int* done;          // PCG convergence flag (may be null)
};
// `stop_after_request`: evaluated once the first sub-chunk's streams are in flight (the single-reduction PCG decides about convergence
// there, behind the loads instead of in front of them); true = leave without touching anything.
template <bool K16, typename Stop>
__device__ __forceinline__ void mv_col_body(const ColMatvecArgs& a, Stop stop_after_request) {
  constexpr int RB = GSFM_COL_RB, EPL = GSFM_COL_EPL;
  // plane-major: the slot-contiguous reads of one row are conflict-free; two buffers, so one barrier per iteration suffices (a buffer is
  // written again two iterations later, after the barrier every lane passes once it has finished reading it)
  __shared__ double slots[2][EPL][3][RB];
  __shared__ uint32_t wtot[2][EPL][RB / 64];   // per wavefront: number of slots of its 64 rows
  const ColWg w = a.L.wg[blockIdx.x];
  const uint32_t r = threadIdx.x, wave = r >> 6;
  double y0 = 0.0, y1 = 0.0, y2 = 0.0;
  uint32_t m[EPL], pos[EPL], cbase[EPL]; double2 A[EPL], B[EPL], C[EPL];
  const uint32_t cmask = (1u << a.L.cbits) - 1u, cshift = a.L.cbits + GSFM_COL_SLOT_BITS, cmax = a.L.cmax;
  auto request = [&](uint32_t s) {   // the streams of sub-chunks s .. s + EPL - 1 of this workgroup (past its end: the last one again, discarded)
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const uint32_t sc = w.first_sub + min(s + (uint32_t)k, w.n_sub - 1);
      const size_t e = (size_t)sc * RB + r;
      pos[k] = (uint32_t)e;
      if constexpr (K16) { m[k] = __builtin_nontemporal_load(a.L.k16 + e); cbase[k] = a.L.kbase[(size_t)sc * (RB / 64) + __builtin_amdgcn_readfirstlane(wave)]; }
      else { m[k] = __builtin_nontemporal_load(a.L.kcol + e); cbase[k] = 0u; }
      A[k] = nt_load2(a.b0 + e); B[k] = nt_load2(a.b1 + e); C[k] = nt_load2(a.b2 + e);
    }
  };
  if (w.n_sub) request(0);
  if (stop_after_request()) return;
  int buf = 0;
  for (uint32_t s = 0; s < w.n_sub; s += EPL, buf ^= 1) {
    uint32_t cnt[EPL], inc[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      uint32_t cam, pm, c;
      if constexpr (K16) {
        uint32_t dl = m[k] >> 12;
        if (dl == GSFM_K16_DEL_ESC) dl = a.L.kdel[pos[k]];
        cam = cbase[k] + wave_incl_scan(dl);
        pm = m[k] & ((1u << GSFM_COL_SLOT_BITS) - 1u);
        c = (m[k] >> GSFM_COL_SLOT_BITS) & 7u;
        if (c == GSFM_K16_CNT_ESC) c = a.L.kcnt[pos[k]];
      } else {
        cam = (m[k] & cmask) == cmask ? 0u : (m[k] & cmask); pm = (m[k] >> a.L.cbits) & ((1u << GSFM_COL_SLOT_BITS) - 1u);
        c = m[k] >> cshift;
        if (c == cmax) c = a.L.kcnt[pos[k]];   // (saturated: a row with very many entries in this sub-chunk, or no room for counts in the word)
      }
      const double* um = a.u + 3 * (size_t)cam;
      const double u0 = um[0], u1 = um[1], u2 = um[2];
      slots[buf][k][0][pm] = A[k].x * u0 + A[k].y * u1 + B[k].x * u2;
      slots[buf][k][1][pm] = A[k].y * u0 + B[k].y * u1 + C[k].x * u2;
      slots[buf][k][2][pm] = B[k].x * u0 + C[k].x * u1 + C[k].y * u2;
      // slot range of row r: prefix sum of the rows' counts, inside the wavefront here, across wavefronts after the barrier
      cnt[k] = s + (uint32_t)k < w.n_sub ? c : 0u;
      inc[k] = wave_incl_scan(cnt[k]);
      if ((r & 63u) == 63u) wtot[buf][k][wave] = inc[k];
    }
    if (s + EPL < w.n_sub) request(s + EPL);   // the next iteration's streams are in flight across the barrier and the row phase
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      uint32_t s1 = inc[k];
      for (uint32_t v = 0; v < wave; ++v) s1 += wtot[buf][k][v];
      for (uint32_t t = s1 - cnt[k]; t < s1; ++t) { y0 += slots[buf][k][0][t]; y1 += slots[buf][k][1][t]; y2 += slots[buf][k][2][t]; }
    }
  }
  const size_t o = (size_t)w.part * RB + r, plane = (size_t)a.L.n_wg * RB;
  a.part[o] = y0; a.part[plane + o] = y1; a.part[2 * plane + o] = y2;
}
// (a 256-lane form of the same product -- more, smaller barrier groups per CU -- measured equal or slower, profiles/r04b_k3c_threads_ab.txt, and was removed in round 5)
#define GSFM_K3C_THREADS GSFM_COL_RB
template <bool K16>
__global__ void __launch_bounds__(GSFM_K3C_THREADS) k_mv_col(ColMatvecArgs a) {
  if (a.done && *a.done) return;
  mv_col_body<K16>(a, [] { return false; });
}
// The same product as the mat-vec of the single-reduction PCG (run_pcg2): the convergence decision of k_matvec_cg at its entry -- every
// workgroup re-sums the gamma partials in the order and with the reduction tree of the 256-lane kernels, so all of them, and the vector
// kernel that follows, see bit-identical scalars -- then the rows.  The delta partials come from k_mv_col_finish (dot_part).
struct ColMatvecCgArgs { ColMatvecArgs mv; Cg2Args cg; };
template <bool K16>
__global__ void __launch_bounds__(GSFM_K3C_THREADS) k_mv_col_cg(ColMatvecCgArgs aa) {
  __shared__ double lds[4];
  const Cg2Args& c = aa.cg;
  mv_col_body<K16>(aa.mv, [&]() -> bool {
    const int done = c.sc->done, iters = c.sc->iters;
    const double gamma0 = c.sc->gamma0, tol = c.sc->tol;
    const bool estop = cg_energy_stop(c.sc->einc, c.sc->esum, c.sc->etol2, iters);
    double gpart = 0.0;
    if (threadIdx.x < GSFM_BLOCK) for (int k = threadIdx.x; k < c.nb_cam; k += GSFM_BLOCK) gpart += c.part_g[(size_t)c.par * c.nb_cam + k];
    gpart = wave_sum(gpart);
    if ((threadIdx.x & 63u) == 0 && threadIdx.x < GSFM_BLOCK) lds[threadIdx.x >> 6] = gpart;
    __syncthreads();
    double gamma = 0.0;
#pragma unroll
    for (int k = 0; k < GSFM_BLOCK / 64; ++k) gamma += lds[k];
    if (done) return true;
    bool conv;
    if (c.first) {
      const double rz_abs = c.sc->rz_abs;   // (as k_matvec_cg)
      conv = !(gamma > rz_abs);
      if (blockIdx.x == 0 && threadIdx.x == 0) { c.sc->gamma0 = gamma; c.sc->tol = cg_tol_with_floor(tol, rz_abs, gamma); if (conv) { c.sc->done = 1; c.sc->last_rel = 0.0; } }
    } else {
      const double rel = sqrt(gamma / gamma0);
      conv = !(rel > tol) || iters >= c.max_iters || estop;
      if (blockIdx.x == 0 && threadIdx.x == 0) { c.sc->last_rel = rel; if (conv) c.sc->done = 1; }
    }
    return conv;
  });
}
// y_k = M_k p_k - R_k sum_{c < NCH} part[block(k) * NCH + c][k mod RB]
struct ColFinishArgs {
  uint32_t n_rows, row_base, nch, n_wg;
  const double* part; const double* Mblk; const double* p; const double2* q; double* y; const int* done;
  double* dot_part;   // PCG on one GPU: this block's share of p . (A p), so that k_cg_dot need not be launched (null: not wanted)
};
__global__ void __launch_bounds__(GSFM_BLOCK) k_mv_col_finish(ColFinishArgs a) {
  if (a.done && *a.done) return;
  __shared__ double lds[8];
  const uint32_t row = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  double pAp = 0.0;
  if (row < a.n_rows) {
  const uint32_t blk = row / GSFM_COL_RB, r = row % GSFM_COL_RB;
  const size_t plane = (size_t)a.n_wg * GSFM_COL_RB;
  double t0 = 0.0, t1 = 0.0, t2 = 0.0;
  // (the partials of eight chunks are requested together, then added in chunk order -- the same sums as a plain loop, bit for bit; as a plain loop
  // the 3 x nch loads of a lane went out one dependent round trip after the other: 8.2-9.6 us for the 12.5 k rows of one rank of 8 under
  // rocprofv3, a quarter of its mat-vec, profiles/r06_rank8_kernel_stats.txt)
  constexpr uint32_t FB = 8;
  for (uint32_t c0 = 0; c0 < a.nch; c0 += FB) {
    double v0[FB], v1[FB], v2[FB];
#pragma unroll
    for (uint32_t j = 0; j < FB; ++j) {
      const size_t o = ((size_t)blk * a.nch + min(c0 + j, a.nch - 1)) * GSFM_COL_RB + r;
      v0[j] = a.part[o]; v1[j] = a.part[plane + o]; v2[j] = a.part[2 * plane + o];
    }
#pragma unroll
    for (uint32_t j = 0; j < FB; ++j) if (c0 + j < a.nch) { t0 += v0[j]; t1 += v1[j]; t2 += v2[j]; }
  }
  const size_t k = a.row_base + row;
  double R[9], mp[3];
  qmat(load_q(a.q, (uint32_t)k), R);
  sym3_mulvec(a.Mblk + 6 * k, a.p + 3 * k, mp);
  const double y0 = mp[0] - (R[0] * t0 + R[1] * t1 + R[2] * t2), y1 = mp[1] - (R[3] * t0 + R[4] * t1 + R[5] * t2), y2 = mp[2] - (R[6] * t0 + R[7] * t1 + R[8] * t2);
  a.y[3 * k] = y0; a.y[3 * k + 1] = y1; a.y[3 * k + 2] = y2;
  pAp = a.p[3 * k] * y0 + a.p[3 * k + 1] * y1 + a.p[3 * k + 2] * y2;   // (the same product, block size and reduction tree as k_cg_dot: same bits)
  }
  if (a.dot_part) {
    const double t = block_sum_bcast(pAp, lds);
    if (threadIdx.x == 0) a.dot_part[blockIdx.x] = t;
  }
}

// s = |r_e|^2 of every entry of the layout, written per local edge (the column-sorted twin of k_row_s: host-callback losses on a sharded
// problem need s for every edge a rank holds).  One workgroup per ColWg, one position per lane and trip.
struct ColRowSArgs {
  ColLayoutDev L;
  uint32_t row_base, n_rows;
  const uint32_t* eid;
  const double2 *qr0, *qr1, *w0, *w1, *w2;
  const double* ws;
  const double2* q;
  double* s_out;
};
template <int F, int WM>
__global__ void __launch_bounds__(GSFM_BLOCK) k_col_s(ColRowSArgs a) {
  constexpr int R = ResDim<F>::R;
  const ColWg w = a.L.wg[blockIdx.x];
  for (uint32_t d = w.first_sub * GSFM_COL_SUB + threadIdx.x; d < (w.first_sub + w.n_sub) * GSFM_COL_SUB; d += GSFM_BLOCK) {
    const uint2 mt = col_load_meta(a.L.meta + d);
    if (mt.x == GSFM_COL_PAD) continue;
    const Quat qk = load_q(a.q, a.row_base + w.row0 + col_rowl(mt.y)), qm = load_q(a.q, mt.x & 0x7fffffffu);
    double2 r0, r1;
    qrel_load<WM>(a.qr0, a.qr1, d, r0, r1);
    const Quat qr = qrel_quat<WM>(r0, r1);
    const EdgeW W = load_w<WM>(a.w0, a.w1, a.w2, a.ws, d);
    double r[R];
    if (mt.x >> 31) edge_residual<F, WM>(qm, qk, qr, W, r);
    else edge_residual<F, WM>(qk, qm, qr, W, r);
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < R; ++c) s += r[c] * r[c];
    a.s_out[a.eid[d]] = s;
  }
}

// Dense assembly for the exact Cholesky step (k_dense_assemble's twin on this layout): the blocks are stored in the body frame,
// H_km = -R_k B R_m^T; the lower triangle takes the entry whose row camera has the larger index.  One workgroup per ColWg for the
// entries, a grid-stride loop over the cameras for the diagonal blocks, the right-hand side and the padding.  A is zero-filled before.
__global__ void __launch_bounds__(GSFM_BLOCK) k_dense_assemble_col(DenseArgs a, ColLayoutDev L) {
  for (uint32_t row = blockIdx.x * GSFM_BLOCK + threadIdx.x; row < a.n_rows; row += gridDim.x * GSFM_BLOCK) {
    for (int c = 0; c < 3; ++c) a.rcg[3 * (size_t)row + c] = 0.0;
    if (row == 0) *a.info_slot = 0.0;
    const double* M = a.Mblk + 6 * (size_t)row;
    const double m[9] = {M[0], M[1], M[2], M[1], M[3], M[4], M[2], M[4], M[5]};
    for (int r = 0; r < 3; ++r) for (int c = 0; c <= r; ++c) *dense_elem(a.A, 3 * row + r, 3 * row + c) = m[3 * r + c];
    for (int c = 0; c < 3; ++c) a.A[(((size_t)a.T * (a.T + 1) / 2) + (3 * row + c) / 32) * 1024 + (3 * row + c) % 32] = a.b[3 * (size_t)row + c];
    if (row == 0) for (uint32_t g = a.n; g < a.T * 32; ++g) *dense_elem(a.A, g, g) = 1.0;
  }
  const ColWg w = L.wg[blockIdx.x];
  for (uint32_t d = w.first_sub * GSFM_COL_SUB + threadIdx.x; d < (w.first_sub + w.n_sub) * GSFM_COL_SUB; d += GSFM_BLOCK) {
    const uint2 mt = col_load_meta(L.meta + d);
    if (mt.x == GSFM_COL_PAD) continue;
    const uint32_t row = w.row0 + col_rowl(mt.y), m = mt.x & 0x7fffffffu;
    if (m >= row) continue;
    const double2 A0 = a.h0[d], B0 = a.h1[d], C0 = a.h2[d];
    const double Bm[9] = {A0.x, A0.y, B0.x, A0.y, B0.y, C0.x, B0.x, C0.x, C0.y};
    double Rk[9], Rm[9], T[9], H[9];
    qmat(load_q(a.q, row), Rk);
    qmat(load_q(a.q, m), Rm);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) T[3 * r + c] = Bm[3 * r] * Rm[3 * c] + Bm[3 * r + 1] * Rm[3 * c + 1] + Bm[3 * r + 2] * Rm[3 * c + 2];   // B R_m^T
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[3 * r + c] = -(Rk[3 * r] * T[c] + Rk[3 * r + 1] * T[3 + c] + Rk[3 * r + 2] * T[6 + c]);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) atomicAdd(dense_elem(a.A, 3 * row + r, 3 * m + c), H[3 * r + c]);
  }
}

// ---- K2c ----------------------------------------------------------------------------------------------------------------
// The linearisation in the same order: one entry per lane and trip, 256 lanes work through a sub-chunk of GSFM_COL_SUB positions in two
// trips; the nine per-entry contributions to (g, D) of the row camera go through the LDS slots, lanes t and t + 256 own rows t and t + 256
// of the block (37 KB of slots + 16 KB of row quaternions per workgroup: three workgroups, i.e. three waves per SIMD, per CU).  The row cameras' quaternions are staged in LDS once per workgroup (a lane's row is arbitrary inside the block); the
// neighbour's quaternion is gathered -- line-sharing, like u in K3c.  a.h0..h2 receive the BODY-frame block at the entry's position.
#ifndef GSFM_COLLIN_THREADS
#define GSFM_COLLIN_THREADS 256
#endif
// which instantiations of K2c hand BODY-frame row sums to the finishing kernel (the host asks the same question: solver_launch.hpp)
__host__ __device__ constexpr bool col_lin_body_frame(int functor, bool fast) { return fast && functor == F_AA; }
struct ColLinArgs {
  LinArgs lin;              // streams (qr, w, col, eid: all in position order), q, loss, rho_ext, sigma, h0..h2 (out); row_base
  ColLayoutDev L;
  double* part;             // 9 planes of [n_wg * RB]
};
template <int F, int WM, int LM, bool FAST>
__device__ __forceinline__ void lin_col_body(const ColLinArgs& a) {
  constexpr int RB = GSFM_COL_RB, SUB = GSFM_COL_SUB, T = GSFM_COLLIN_THREADS, RPL = RB / T;
  constexpr bool BODY = col_lin_body_frame(F, FAST);
  __shared__ double slots[9][SUB];
  __shared__ double2 qrow[2][RB];
  __shared__ uint32_t wtot[RPL][T / 64];
  const ColWg w = a.L.wg[blockIdx.x];
  const uint32_t t = threadIdx.x;
  const LossView<LM> lv = loss_view<LM>(a.lin.loss);   // (before the first store: scalar loads, see loss_dev.hpp)
  for (uint32_t r = t; r < RB; r += T) {
    const uint32_t k = min(a.lin.row_base + w.row0 + r, a.lin.row_base + a.lin.n_rows - 1);   // (a ragged last block re-reads its last row)
    qrow[0][r] = a.lin.q[2 * (size_t)k]; qrow[1][r] = a.lin.q[2 * (size_t)k + 1];
  }
  double acc[RPL][9];
#pragma unroll
  for (int j = 0; j < RPL; ++j)
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[j][c] = 0.0;
  __syncthreads();
  // Software pipeline over the sub-chunks (round 4): the records and streams of sub-chunk s + 1 are requested BEFORE sub-chunk s is
  // evaluated and its neighbour quaternions before the row phase, so the HBM round trips run behind the ~1 100 VALU instructions of the
  // evaluation, the LDS row phase and the two barriers instead of in front of them.  The registers for it (2 x 24 per lane) are the ones
  // the scalar-register transcendentals freed (devmath.hpp; 228 -> 169 without the pipeline).  (The round-3 form -- both trips of ONE sub-chunk
  // requested together, nothing in flight during the evaluation -- measured 614 against 567 us, profiles/r04b_k2c_pipeline_ab.txt, and is gone.)
  constexpr int K = SUB / T;
  static_assert(K == RPL, "one record per owned row");
  uint2 mt[K];
  LinStreams S[K];
  Quat qm[K];
  auto request = [&](uint32_t s, uint2* m, LinStreams* St) {   // (past the end: the last sub-chunk again, discarded)
    const uint32_t sc = w.first_sub + min(s, w.n_sub - 1);
#pragma unroll
    for (int k = 0; k < K; ++k) m[k] = col_load_meta(a.L.meta + (sc * SUB + k * T + t));
#pragma unroll
    for (int k = 0; k < K; ++k) St[k] = lin_load_streams<WM>(a.lin, sc * SUB + k * T + t);
  };
  auto gather = [&](const uint2* m, Quat* q) {   // (padding positions read camera 0; nothing of it is used)
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = load_q(a.lin.q, m[k].x == GSFM_COL_PAD ? 0u : (m[k].x & 0x7fffffffu));
  };
  if (w.n_sub) { request(0, mt, S); gather(mt, qm); }
  for (uint32_t s = 0; s < w.n_sub; ++s) {
    const uint32_t sc = w.first_sub + s;
    uint2 mtn[K];
    LinStreams Sn[K];
    Quat qmn[K];
    request(s + 1, mtn, Sn);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t d = sc * SUB + k * T + t;
      const uint32_t cr = mt[k].x, pm = col_slot(mt[k].y);
      double g3[3] = {0, 0, 0}, G6[6] = {0, 0, 0, 0, 0, 0}, B6[6] = {0, 0, 0, 0, 0, 0};
      if constexpr (BODY) {   // body-frame evaluation (kernels.hpp, lin_entry_body_aa): the slots carry (gb, B), the finishing kernel rotates the row sums
        if (cr != GSFM_COL_PAD) {
          const uint32_t rl = col_rowl(mt[k].y);
          const double2 k0 = qrow[0][rl], k1 = qrow[1][rl];
          lin_entry_body_aa<WM, LM>(a.lin, lv, d, cr, Quat{k0.x, k0.y, k1.x, k1.y}, qm[k], S[k], g3, B6);
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) G6[c] = B6[c];
      } else if (cr != GSFM_COL_PAD) {
        const uint32_t rl = col_rowl(mt[k].y);
        const double2 k0 = qrow[0][rl], k1 = qrow[1][rl];
        const Quat qk{k0.x, k0.y, k1.x, k1.y};
        lin_entry_eval<F, WM, LM, FAST>(a.lin, lv, d, cr, qk, qm[k], S[k], g3, G6);
        // body frame: B = R_k^T G R_k  (the BODY instantiations produce B directly)
        double R[9], Tm[9];
        qmat(qk, R);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          Tm[c] = G6[0] * R[c] + G6[1] * R[3 + c] + G6[2] * R[6 + c];
          Tm[3 + c] = G6[1] * R[c] + G6[3] * R[3 + c] + G6[4] * R[6 + c];
          Tm[6 + c] = G6[2] * R[c] + G6[4] * R[3 + c] + G6[5] * R[6 + c];
        }
        B6[0] = R[0] * Tm[0] + R[3] * Tm[3] + R[6] * Tm[6]; B6[1] = R[0] * Tm[1] + R[3] * Tm[4] + R[6] * Tm[7]; B6[2] = R[0] * Tm[2] + R[3] * Tm[5] + R[6] * Tm[8];
        B6[3] = R[1] * Tm[1] + R[4] * Tm[4] + R[7] * Tm[7]; B6[4] = R[1] * Tm[2] + R[4] * Tm[5] + R[7] * Tm[8]; B6[5] = R[2] * Tm[2] + R[5] * Tm[5] + R[8] * Tm[8];
      }
      nt_store2(a.lin.h0 + d, B6[0], B6[1]);
      nt_store2(a.lin.h1 + d, B6[2], B6[3]);
      nt_store2(a.lin.h2 + d, B6[4], B6[5]);
#pragma unroll
      for (int c = 0; c < 3; ++c) slots[c][pm] = g3[c];
#pragma unroll
      for (int c = 0; c < 6; ++c) slots[3 + c][pm] = G6[c];
    }
    // slot ranges of rows t + j * T: the rows' counts ride in the records of positions t + j * T, i.e. in mt[j] (K == RPL)
    uint32_t cnt[RPL], inc[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      cnt[j] = col_rowcount(mt[j].y);
      inc[j] = wave_incl_scan(cnt[j]);
      if ((t & 63u) == 63u) wtot[j][t >> 6] = inc[j];
    }
    gather(mtn, qmn);   // the next sub-chunk's neighbour quaternions travel during the row phase
    __syncthreads();
    uint32_t carry = 0, s0[RPL], nmax = 0;
#pragma unroll
    for (int j = 0; j < RPL; ++j) {
      uint32_t s1 = carry + inc[j];
      for (uint32_t v = 0; v < T / 64; ++v) { const uint32_t wt = wtot[j][v]; if (v < (t >> 6)) s1 += wt; carry += wt; }
      s0[j] = s1 - cnt[j];
      nmax = max(nmax, cnt[j]);
    }
    // the rows of a lane side by side (their LDS reads are independent: one latency per step instead of one per row and step); every row
    // still adds its slots in slot order
    for (uint32_t u = 0; u < nmax; ++u) {
#pragma unroll
      for (int j = 0; j < RPL; ++j) {
        if (u < cnt[j]) {
#pragma unroll
          for (int c = 0; c < 9; ++c) acc[j][c] += slots[c][s0[j] + u];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) { mt[k] = mtn[k]; S[k] = Sn[k]; qm[k] = qmn[k]; }
  }
  const size_t plane = (size_t)a.L.n_wg * RB;
#pragma unroll
  for (int j = 0; j < RPL; ++j) {
    const size_t o = (size_t)w.part * RB + t + j * T;
#pragma unroll
    for (int c = 0; c < 9; ++c) a.part[(size_t)c * plane + o] = acc[j][c];
  }
}
// Registers at the compiler's choice (~200-250: two waves per SIMD, two workgroups per CU): asking for three waves spills the eighteen
// row accumulators (76 B of scratch per lane) and costs 1010 us against 785 at C5.
template <int F, int WM, int LM, bool FAST>
#ifndef GSFM_K2C_ATTR
#define GSFM_K2C_ATTR
#endif
__global__ void __launch_bounds__(GSFM_COLLIN_THREADS) GSFM_K2C_ATTR k_lin_col(ColLinArgs a) { lin_col_body<F, WM, LM, FAST>(a); }

// gD[k] = sum_{c < NCH} part[block(k) * NCH + c][k mod RB]  (nine values per camera)
// q != null: the partial sums are in the rows' body frames (sum gb, sum B): g = R_k sum gb, D = R_k (sum B) R_k^T
__global__ void __launch_bounds__(GSFM_BLOCK) k_lin_col_finish(uint32_t n_rows, uint32_t row_base, uint32_t nch, uint32_t n_wg, const double* __restrict__ part, double* __restrict__ gD, const double2* __restrict__ q) {
  const uint32_t row = blockIdx.x * GSFM_BLOCK + threadIdx.x;
  if (row >= n_rows) return;
  const uint32_t blk = row / GSFM_COL_RB, r = row % GSFM_COL_RB;
  const size_t plane = (size_t)n_wg * GSFM_COL_RB;
  double v[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  constexpr uint32_t FB = 4;   // (the nine partials of four chunks in flight together, added in chunk order: the same sums as a plain loop -- see k_mv_col_finish)
  for (uint32_t c0 = 0; c0 < nch; c0 += FB) {
    double w[FB][9];
#pragma unroll
    for (uint32_t j = 0; j < FB; ++j) {
      const size_t o = ((size_t)blk * nch + min(c0 + j, nch - 1)) * GSFM_COL_RB + r;
#pragma unroll
      for (int x = 0; x < 9; ++x) w[j][x] = part[(size_t)x * plane + o];
    }
#pragma unroll
    for (uint32_t j = 0; j < FB; ++j) if (c0 + j < nch) {
#pragma unroll
      for (int x = 0; x < 9; ++x) v[x] += w[j][x];
    }
  }
  double* out = gD + 9 * (size_t)(row_base + row);
  if (q) {
    double R[9], T[9];
    qmat(load_q(q, row_base + row), R);
    const double g0 = v[0], g1 = v[1], g2 = v[2];
    v[0] = R[0] * g0 + R[1] * g1 + R[2] * g2; v[1] = R[3] * g0 + R[4] * g1 + R[5] * g2; v[2] = R[6] * g0 + R[7] * g1 + R[8] * g2;
    const double S[9] = {v[3], v[4], v[5], v[4], v[6], v[7], v[5], v[7], v[8]};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) T[3 * r + c] = R[3 * r] * S[c] + R[3 * r + 1] * S[3 + c] + R[3 * r + 2] * S[6 + c];   // R S
    v[3] = T[0] * R[0] + T[1] * R[1] + T[2] * R[2]; v[4] = T[0] * R[3] + T[1] * R[4] + T[2] * R[5]; v[5] = T[0] * R[6] + T[1] * R[7] + T[2] * R[8];
    v[6] = T[3] * R[3] + T[4] * R[4] + T[5] * R[5]; v[7] = T[3] * R[6] + T[4] * R[7] + T[5] * R[8]; v[8] = T[6] * R[6] + T[7] * R[7] + T[8] * R[8];
  }
#pragma unroll
  for (int x = 0; x < 9; ++x) out[x] = v[x];
}

}  // namespace gsfm
