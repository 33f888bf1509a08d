// Dev micro-benchmark (not part of the product; round 6, review item 3): the column-sorted mat-vec BEYOND the density rule -- graphs whose 512-row
// blocks hold far fewer entries than there are cameras (1.5 M cameras / 150 M edges on one GPU; every rank of an 8-GPU weak-scaling run at 800 k
// cameras), where the product switches K2c / K3c off (problem_create.hpp: 512 rows x mean degree >= cameras / 2) and the row-major kernels run at
// 0.47 x the C5 rate per edge.  The remedy DESIGN section 9 has named since round 3: row blocks of RBB = 2048-4096 rows, so that a block holds about as
// many entries as there are cameras and the u[col] gathers share lines again; a row no longer has a lane of its own, so the row sums live in LDS
// (3 x RBB doubles per workgroup) and the HEAD of each run of equal-row slots adds its run to them (one barrier per sub-chunk, deterministic: a row
// has one run per sub-chunk, the sub-chunks of a workgroup are sequential).  Per position: col 4 B + slot 2 B + (row | head flag) 2 B + block 48 B.
// Measured against the row-major form and the product's RB = 512 column-sorted form on the same graph.
//   hipcc --offload-arch=gfx950 -O3 -o bench_matvec8 bench_matvec8.hip ;  ./bench_matvec8 [N = 400000] [DEG = 200]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double2 nt2(const double2* p) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }

// ---- baseline: the product's row-major form (G lanes per row) ----
struct ArgsRow { unsigned n_rows, G; const unsigned* row_ptr; const unsigned* col; const double2 *h0, *h1, *h2; const double* u; const double* M; const double* p; const double2* q; double* y; };
__global__ void __launch_bounds__(256) k_mv_row(ArgsRow a) {
  const unsigned G = a.G, t = blockIdx.x * 256 + threadIdx.x, row = t / G, lane = t % G;
  const bool live = row < a.n_rows;
  double y0 = 0, y1 = 0, y2 = 0;
  if (live) {
    const double2 qa = a.q[2 * (size_t)row], qb = a.q[2 * (size_t)row + 1];
    const double R0 = qa.x, R1 = qa.y, R2 = qb.x, R3 = qb.y, R4 = qa.x * qb.y, R5 = qa.y * qb.x, R6 = qa.x + qb.x, R7 = qa.y - qb.y, R8 = qb.x * qb.y;
    const unsigned end = a.row_ptr[row + 1];
    for (unsigned d = a.row_ptr[row] + lane; d < end; d += G) {
      const unsigned m = __builtin_nontemporal_load(a.col + d) & 0x7fffffffu;
      const double2 A = nt2(a.h0 + d), B = nt2(a.h1 + d), C = nt2(a.h2 + d);
      const double* um = a.u + 3 * (size_t)m;
      const double u0 = um[0], u1 = um[1], u2 = um[2];
      const double w0 = R0 * u0 + R1 * u1 + R2 * u2, w1 = R3 * u0 + R4 * u1 + R5 * u2, w2 = R6 * u0 + R7 * u1 + R8 * u2;
      y0 += A.x * w0 + A.y * w1 + B.x * w2; y1 += A.y * w0 + B.y * w1 + C.x * w2; y2 += B.x * w0 + C.x * w1 + C.y * w2;
    }
  }
  for (unsigned off = G >> 1; off > 0; off >>= 1) { y0 += __shfl_down(y0, off, G); y1 += __shfl_down(y1, off, G); y2 += __shfl_down(y2, off, G); }
  if (live && lane == 0) {
    const double* M = a.M + 6 * (size_t)row; const double* pk = a.p + 3 * (size_t)row;
    a.y[3 * (size_t)row] = M[0] * pk[0] + M[1] * pk[1] + M[2] * pk[2] - y0; a.y[3 * (size_t)row + 1] = M[1] * pk[0] + M[3] * pk[1] + M[4] * pk[2] - y1;
    a.y[3 * (size_t)row + 2] = M[2] * pk[0] + M[4] * pk[1] + M[5] * pk[2] - y2;
  }
}

// ---- column-sorted row blocks ----
struct WgDesc { unsigned first_sub, n_sub, block, pad; };
struct ArgsCol {
  const WgDesc* wg; const unsigned* col; const uint16_t* perm; const uint16_t* seg;   // seg: (RB + 1) per sub-chunk
  const double2 *b0, *b1, *b2; const double* u; const double4* u4; double* part;       // part: 3 planes of [n_wg * RB]
  unsigned n_wg;
};
// PADU: gather from a 32-byte padded vector.  PF: request the next sub-chunk's streams before this one's row phase.
template <int RB, int EPL, bool PADU, bool PF>
__global__ void __launch_bounds__(RB) k_mv_col(ArgsCol a) {
  constexpr int SUB = RB * EPL;
  __shared__ double slots[2][3][SUB];   // plane-major: conflict-free for slot-contiguous reads
  const WgDesc w = a.wg[blockIdx.x];
  const unsigned r = threadIdx.x;
  double y0 = 0, y1 = 0, y2 = 0;
  unsigned m[EPL]; double2 A[EPL], B[EPL], C[EPL]; uint16_t pm[EPL];
  auto request = [&](unsigned sc) {
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const size_t e = (size_t)sc * SUB + (size_t)k * RB + r;
      m[k] = __builtin_nontemporal_load(a.col + e);
      A[k] = nt2(a.b0 + e); B[k] = nt2(a.b1 + e); C[k] = nt2(a.b2 + e);
      pm[k] = __builtin_nontemporal_load(a.perm + e);
    }
  };
  if (PF) request(w.first_sub);
  for (unsigned s = 0; s < w.n_sub; ++s) {
    const unsigned sc = w.first_sub + s;
    if (!PF) request(sc);
    double c0[EPL], c1[EPL], c2[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      double u0, u1, u2;
      if (PADU) { const double4 v = a.u4[m[k]]; u0 = v.x; u1 = v.y; u2 = v.z; }
      else { const double* um = a.u + 3 * (size_t)m[k]; u0 = um[0]; u1 = um[1]; u2 = um[2]; }
      c0[k] = A[k].x * u0 + A[k].y * u1 + B[k].x * u2; c1[k] = A[k].y * u0 + B[k].y * u1 + C[k].x * u2; c2[k] = B[k].x * u0 + C[k].x * u1 + C[k].y * u2;
    }
    const int buf = s & 1;
#pragma unroll
    for (int k = 0; k < EPL; ++k) { slots[buf][0][pm[k]] = c0[k]; slots[buf][1][pm[k]] = c1[k]; slots[buf][2][pm[k]] = c2[k]; }
    const uint16_t* sg = a.seg + (size_t)sc * (RB + 1);
    const unsigned s0 = sg[r], s1 = sg[r + 1];
    if (PF && s + 1 < w.n_sub) request(sc + 1);
    __syncthreads();
    for (unsigned t = s0; t < s1; ++t) { y0 += slots[buf][0][t]; y1 += slots[buf][1][t]; y2 += slots[buf][2][t]; }
  }
  const size_t o = (size_t)blockIdx.x * RB + r, plane = (size_t)a.n_wg * RB;
  a.part[o] = y0; a.part[plane + o] = y1; a.part[2 * plane + o] = y2;
}

// ---- column-sorted BIG row blocks: row sums in LDS ----
struct ArgsBig {
  const WgDesc* wg; const unsigned* col; const uint16_t* perm; const uint16_t* srow;   // srow: per SLOT of a sub-chunk: row inside the block | 0x8000 = head of its run
  const double2 *b0, *b1, *b2; const double* u; double* part;                            // part: 3 planes of [n_wg * RBB]
  unsigned n_wg;
};
template <int RBB, int T>
__global__ void __launch_bounds__(T) k_mv_big(ArgsBig a) {
  constexpr int SUB = T;
  __shared__ double acc[3][RBB];
  __shared__ double slots[2][3][SUB];
  __shared__ uint16_t srl[2][SUB + 1];
  const WgDesc w = a.wg[blockIdx.x];
  const unsigned r = threadIdx.x;
  for (unsigned k = r; k < 3 * RBB; k += T) (&acc[0][0])[k] = 0.0;
  unsigned m; double2 A, B, C; uint16_t pm, sr;
  auto request = [&](unsigned sc) {
    const size_t e = (size_t)sc * SUB + r;
    m = __builtin_nontemporal_load(a.col + e);
    A = nt2(a.b0 + e); B = nt2(a.b1 + e); C = nt2(a.b2 + e);
    pm = __builtin_nontemporal_load(a.perm + e); sr = __builtin_nontemporal_load(a.srow + e);
  };
  if (w.n_sub) request(w.first_sub);
  if (r == 0) { srl[0][SUB] = 0x8000; srl[1][SUB] = 0x8000; }
  __syncthreads();
  for (unsigned s = 0; s < w.n_sub; ++s) {
    const unsigned sc = w.first_sub + s;
    const int buf = s & 1;
    const double* um = a.u + 3 * (size_t)m;
    const double u0 = um[0], u1 = um[1], u2 = um[2];
    slots[buf][0][pm] = A.x * u0 + A.y * u1 + B.x * u2; slots[buf][1][pm] = A.y * u0 + B.y * u1 + C.x * u2; slots[buf][2][pm] = B.x * u0 + C.x * u1 + C.y * u2;
    srl[buf][r] = sr;
    const unsigned mine = sr;
    if (s + 1 < w.n_sub) request(sc + 1);
    __syncthreads();
    if (mine & 0x8000u) {
      const unsigned row = mine & 0x7fffu;
      if (row < RBB) {   // (the padding run carries row 0x7fff)
        double t0 = slots[buf][0][r], t1 = slots[buf][1][r], t2 = slots[buf][2][r];
        for (unsigned v = r + 1; !(srl[buf][v] & 0x8000u); ++v) { t0 += slots[buf][0][v]; t1 += slots[buf][1][v]; t2 += slots[buf][2][v]; }
        acc[0][row] += t0; acc[1][row] += t1; acc[2][row] += t2;
      }
    }
  }
  __syncthreads();
  const size_t plane = (size_t)a.n_wg * RBB;
  for (unsigned k = r; k < RBB; k += T) { const size_t o = (size_t)blockIdx.x * RBB + k; a.part[o] = acc[0][k]; a.part[plane + o] = acc[1][k]; a.part[2 * plane + o] = acc[2][k]; }
}
// y_k = M_k p_k - R_k sum_chunks part
struct ArgsFin { unsigned n_rows, RB, NCH, n_wg; const double* part; const double* M; const double* p; const double2* q; double* y; };
__global__ void __launch_bounds__(256) k_mv_finish(ArgsFin a) {
  const unsigned k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.n_rows) return;
  const unsigned blk = k / a.RB, r = k % a.RB;
  const size_t plane = (size_t)a.n_wg * a.RB;
  double t0 = 0, t1 = 0, t2 = 0;
  for (unsigned c = 0; c < a.NCH; ++c) { const size_t o = ((size_t)blk * a.NCH + c) * a.RB + r; t0 += a.part[o]; t1 += a.part[plane + o]; t2 += a.part[2 * plane + o]; }
  const double2 qa = a.q[2 * (size_t)k], qb = a.q[2 * (size_t)k + 1];
  const double R0 = qa.x, R1 = qa.y, R2 = qb.x, R3 = qb.y, R4 = qa.x * qb.y, R5 = qa.y * qb.x, R6 = qa.x + qb.x, R7 = qa.y - qb.y, R8 = qb.x * qb.y;
  const double* M = a.M + 6 * (size_t)k; const double* pk = a.p + 3 * (size_t)k;
  a.y[3 * (size_t)k] = M[0] * pk[0] + M[1] * pk[1] + M[2] * pk[2] - (R0 * t0 + R1 * t1 + R2 * t2);
  a.y[3 * (size_t)k + 1] = M[1] * pk[0] + M[3] * pk[1] + M[4] * pk[2] - (R3 * t0 + R4 * t1 + R5 * t2);
  a.y[3 * (size_t)k + 2] = M[2] * pk[0] + M[4] * pk[1] + M[5] * pk[2] - (R6 * t0 + R7 * t1 + R8 * t2);
}

template <typename F> float timeit(F f, int reps = 10) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  f(); f();
  CHK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) f();
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3f;
}

struct ColLayout {
  std::vector<WgDesc> wg; std::vector<unsigned> col, src; std::vector<uint16_t> perm, seg; unsigned NCH; size_t n_ent;
};
// rp / col: row-major CSR.  Returns the column-sorted block layout; src[e] = row-major entry stored at position e (0xffffffff = padding).
ColLayout build_layout(unsigned N, const std::vector<unsigned>& rp, const std::vector<unsigned>& col, unsigned RB, unsigned EPL, unsigned NCH) {
  ColLayout L; L.NCH = NCH;
  const unsigned SUB = RB * EPL, nblk = (N + RB - 1) / RB;
  std::vector<std::pair<uint64_t, unsigned>> ent;   // (col << 16 | local row, d)
  for (unsigned b = 0; b < nblk; ++b) {
    const unsigned r0 = b * RB, r1 = std::min(N, r0 + RB);
    ent.clear();
    for (unsigned r = r0; r < r1; ++r) for (unsigned d = rp[r]; d < rp[r + 1]; ++d) ent.emplace_back(((uint64_t)col[d] << 16) | (r - r0), d);
    std::sort(ent.begin(), ent.end());
    const size_t ne = ent.size(), n_sub_total = (ne + SUB - 1) / SUB;
    for (unsigned c = 0; c < NCH; ++c) {   // sub-chunks dealt to the NCH workgroups of the block in contiguous runs
      const size_t s_lo = n_sub_total * c / NCH, s_hi = n_sub_total * (c + 1) / NCH;
      WgDesc w{(unsigned)(L.col.size() / SUB), (unsigned)(s_hi - s_lo), b, 0};
      for (size_t s = s_lo; s < s_hi; ++s) {
        const size_t lo = s * SUB, hi = std::min(ne, lo + SUB);
        // slot order inside the sub-chunk: by local row, then by position (stable counting sort)
        std::vector<unsigned> cnt(RB + 1, 0);
        for (size_t e = lo; e < hi; ++e) cnt[(ent[e].first & 0xffff) + 1]++;
        for (unsigned r = 0; r < RB; ++r) cnt[r + 1] += cnt[r];
        for (unsigned r = 0; r <= RB; ++r) L.seg.push_back((uint16_t)cnt[r]);
        std::vector<unsigned> fill(cnt.begin(), cnt.end() - 1);
        unsigned pad_slot = (unsigned)(hi - lo);
        for (size_t e = lo; e < lo + SUB; ++e) {
          if (e < hi) { L.col.push_back((unsigned)(ent[e].first >> 16)); L.src.push_back(ent[e].second); L.perm.push_back((uint16_t)fill[ent[e].first & 0xffff]++); }
          else { L.col.push_back(0); L.src.push_back(0xffffffffu); L.perm.push_back((uint16_t)pad_slot++); }   // padding: zero block, a slot no row reads
        }
      }
      L.wg.push_back(w);
    }
  }
  L.n_ent = L.col.size();
  return L;
}


struct BigLayout { std::vector<WgDesc> wg; std::vector<unsigned> col, src; std::vector<uint16_t> perm, srow; unsigned NCH; size_t n_ent; };
BigLayout build_big(unsigned N, const std::vector<unsigned>& rp, const std::vector<unsigned>& col, unsigned RBB, unsigned SUB, unsigned NCH) {
  BigLayout L; L.NCH = NCH;
  const unsigned nblk = (N + RBB - 1) / RBB;
  std::vector<std::pair<uint64_t, unsigned>> ent;
  for (unsigned b = 0; b < nblk; ++b) {
    const unsigned r0 = b * RBB, r1 = std::min(N, r0 + RBB);
    ent.clear();
    for (unsigned r = r0; r < r1; ++r) for (unsigned d = rp[r]; d < rp[r + 1]; ++d) ent.emplace_back(((uint64_t)col[d] << 16) | (r - r0), d);
    std::sort(ent.begin(), ent.end());
    const size_t ne = ent.size(), n_sub_total = (ne + SUB - 1) / SUB;
    for (unsigned c = 0; c < NCH; ++c) {
      const size_t s_lo = n_sub_total * c / NCH, s_hi = n_sub_total * (c + 1) / NCH;
      WgDesc w{(unsigned)(L.col.size() / SUB), (unsigned)(s_hi - s_lo), b, 0};
      for (size_t s = s_lo; s < s_hi; ++s) {
        const size_t lo = s * SUB, hi = std::min(ne, lo + SUB);
        // slot order: by local row, then by position
        std::vector<std::pair<unsigned, unsigned>> byrow;   // (local row, position in the sub-chunk)
        for (size_t e = lo; e < hi; ++e) byrow.emplace_back((unsigned)(ent[e].first & 0xffff), (unsigned)(e - lo));
        std::stable_sort(byrow.begin(), byrow.end(), [](const std::pair<unsigned, unsigned>& x, const std::pair<unsigned, unsigned>& y) { return x.first < y.first; });
        std::vector<uint16_t> slot_of(SUB), sr(SUB);
        for (unsigned t = 0; t < byrow.size(); ++t) { slot_of[byrow[t].second] = (uint16_t)t; sr[t] = (uint16_t)(byrow[t].first | ((t == 0 || byrow[t].first != byrow[t - 1].first) ? 0x8000u : 0u)); }
        for (unsigned t = (unsigned)byrow.size(); t < SUB; ++t) sr[t] = (uint16_t)(0x7fffu | (t == byrow.size() ? 0x8000u : 0u));   // padding: one run nobody adds
        for (size_t e = lo; e < lo + SUB; ++e) {
          if (e < hi) { L.col.push_back((unsigned)(ent[e].first >> 16)); L.src.push_back(ent[e].second); L.perm.push_back(slot_of[e - lo]); }
          else { L.col.push_back(0); L.src.push_back(0xffffffffu); L.perm.push_back((uint16_t)(e - lo)); }
        }
        for (unsigned t = 0; t < SUB; ++t) L.srow.push_back(sr[t]);
      }
      L.wg.push_back(w);
    }
  }
  L.n_ent = L.col.size();
  return L;
}

int main(int argc, char** argv) {
  const unsigned N = argc > 1 ? atoi(argv[1]) : 400000, DEG = argc > 2 ? atoi(argv[2]) : 200; const int jitter = argc > 3 ? atoi(argv[3]) : 14;
  std::mt19937 rng(1);
  std::vector<unsigned> rp(N + 1, 0);
  std::normal_distribution<double> nd01(0, 1);
  for (unsigned r = 0; r < N; ++r) { int dg = (int)(DEG + jitter * nd01(rng) + 0.5); if (dg < 1) dg = 1; rp[r + 1] = rp[r] + dg; }
  const size_t nd = rp[N];
  std::vector<unsigned> col(nd);
  for (size_t d = 0; d < nd; ++d) col[d] = rng() % N;
  std::uniform_real_distribution<double> U(-1, 1);
  std::vector<double> hb(6 * nd), hu(3 * (size_t)N), hM(6 * (size_t)N), hp(3 * (size_t)N), hq(4 * (size_t)N);
  for (auto& v : hb) v = U(rng);
  for (auto& v : hu) v = U(rng);
  for (auto& v : hM) v = U(rng);
  for (auto& v : hp) v = U(rng);
  for (auto& v : hq) v = U(rng);
  // device: row-major
  unsigned *d_rp, *d_col; double2 *h0, *h1, *h2, *q; double *u, *M, *p, *y, *y2; double4* u4;
  CHK(hipMalloc(&d_rp, 4 * (N + 1))); CHK(hipMalloc(&d_col, 4 * nd));
  CHK(hipMalloc(&h0, 16 * nd)); CHK(hipMalloc(&h1, 16 * nd)); CHK(hipMalloc(&h2, 16 * nd)); CHK(hipMalloc(&q, 32 * (size_t)N));
  CHK(hipMalloc(&u, 24 * (size_t)N)); CHK(hipMalloc(&u4, 32 * (size_t)N)); CHK(hipMalloc(&M, 48 * (size_t)N)); CHK(hipMalloc(&p, 24 * (size_t)N)); CHK(hipMalloc(&y, 24 * (size_t)N)); CHK(hipMalloc(&y2, 24 * (size_t)N));
  {
    std::vector<double2> a0(nd), a1(nd), a2(nd);
    for (size_t d = 0; d < nd; ++d) { a0[d] = make_double2(hb[6 * d], hb[6 * d + 1]); a1[d] = make_double2(hb[6 * d + 2], hb[6 * d + 3]); a2[d] = make_double2(hb[6 * d + 4], hb[6 * d + 5]); }
    CHK(hipMemcpy(h0, a0.data(), 16 * nd, hipMemcpyHostToDevice)); CHK(hipMemcpy(h1, a1.data(), 16 * nd, hipMemcpyHostToDevice)); CHK(hipMemcpy(h2, a2.data(), 16 * nd, hipMemcpyHostToDevice));
    std::vector<double> up(4 * (size_t)N, 0.0);
    for (size_t k = 0; k < N; ++k) { up[4 * k] = hu[3 * k]; up[4 * k + 1] = hu[3 * k + 1]; up[4 * k + 2] = hu[3 * k + 2]; }
    CHK(hipMemcpy(u4, up.data(), 32 * (size_t)N, hipMemcpyHostToDevice));
  }
  CHK(hipMemcpy(d_rp, rp.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CHK(hipMemcpy(d_col, col.data(), 4 * nd, hipMemcpyHostToDevice));
  CHK(hipMemcpy(u, hu.data(), 24 * (size_t)N, hipMemcpyHostToDevice)); CHK(hipMemcpy(M, hM.data(), 48 * (size_t)N, hipMemcpyHostToDevice));
  CHK(hipMemcpy(p, hp.data(), 24 * (size_t)N, hipMemcpyHostToDevice)); CHK(hipMemcpy(q, hq.data(), 32 * (size_t)N, hipMemcpyHostToDevice));
  const double bytes = 52.0 * nd + 48.0 * N;
  printf("rows %u, directed entries %zu (mean degree %.1f, sd %d), %.3f GB per product in the row-major layout\n", N, nd, (double)nd / N, jitter, bytes * 1e-9);
  for (unsigned G : {64u, 32u}) {
    ArgsRow a{N, G, d_rp, d_col, h0, h1, h2, u, M, p, q, y};
    const int grid = (int)(((size_t)N * G + 255) / 256);
    const float t = timeit([&] { hipLaunchKernelGGL(k_mv_row, dim3(grid), dim3(256), 0, 0, a); });
    printf("row-major G=%2u                                   %8.1f us  %6.2f TB/s\n", G, t, bytes / t * 1e-6);
  }
  std::vector<double> yref(3 * (size_t)N);
  { ArgsRow a{N, 64, d_rp, d_col, h0, h1, h2, u, M, p, q, y}; hipLaunchKernelGGL(k_mv_row, dim3((int)(((size_t)N * 64 + 255) / 256)), dim3(256), 0, 0, a); CHK(hipDeviceSynchronize()); }
  // NOTE the row-major stand-in applies "R" per entry to u, the column form applies it to the row sum: same value up to rounding
  CHK(hipMemcpy(yref.data(), y, 24 * (size_t)N, hipMemcpyDeviceToHost));

  auto run = [&](unsigned RB, unsigned EPL, unsigned NCH, auto kern, const char* name) {
    const ColLayout L = build_layout(N, rp, col, RB, EPL, NCH);
    const size_t ne = L.n_ent, nwg = L.wg.size();
    WgDesc* d_wg; unsigned* c2; uint16_t *pm, *sg; double2 *b0, *b1, *b2; double* part;
    CHK(hipMalloc(&d_wg, sizeof(WgDesc) * nwg)); CHK(hipMalloc(&c2, 4 * ne)); CHK(hipMalloc(&pm, 2 * ne)); CHK(hipMalloc(&sg, 2 * L.seg.size()));
    CHK(hipMalloc(&b0, 16 * ne)); CHK(hipMalloc(&b1, 16 * ne)); CHK(hipMalloc(&b2, 16 * ne)); CHK(hipMalloc(&part, 24 * nwg * RB));
    std::vector<double2> a0(ne), a1(ne), a2(ne);
    for (size_t e = 0; e < ne; ++e) {
      const unsigned d = L.src[e];
      if (d == 0xffffffffu) { a0[e] = a1[e] = a2[e] = make_double2(0, 0); continue; }
      a0[e] = make_double2(hb[6 * (size_t)d], hb[6 * (size_t)d + 1]); a1[e] = make_double2(hb[6 * (size_t)d + 2], hb[6 * (size_t)d + 3]); a2[e] = make_double2(hb[6 * (size_t)d + 4], hb[6 * (size_t)d + 5]);
    }
    CHK(hipMemcpy(d_wg, L.wg.data(), sizeof(WgDesc) * nwg, hipMemcpyHostToDevice)); CHK(hipMemcpy(c2, L.col.data(), 4 * ne, hipMemcpyHostToDevice));
    CHK(hipMemcpy(pm, L.perm.data(), 2 * ne, hipMemcpyHostToDevice)); CHK(hipMemcpy(sg, L.seg.data(), 2 * L.seg.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(b0, a0.data(), 16 * ne, hipMemcpyHostToDevice)); CHK(hipMemcpy(b1, a1.data(), 16 * ne, hipMemcpyHostToDevice)); CHK(hipMemcpy(b2, a2.data(), 16 * ne, hipMemcpyHostToDevice));
    ArgsCol a{d_wg, c2, pm, sg, b0, b1, b2, u, u4, part, (unsigned)nwg};
    ArgsFin f{N, RB, NCH, (unsigned)nwg, part, M, p, q, y2};
    const float t = timeit([&] { kern(a, (int)nwg, RB); hipLaunchKernelGGL(k_mv_finish, dim3((N + 255) / 256), dim3(256), 0, 0, f); });
    const float tm = timeit([&] { kern(a, (int)nwg, RB); });
    std::vector<double> yy(3 * (size_t)N);
    CHK(hipMemcpy(yy.data(), y2, 24 * (size_t)N, hipMemcpyDeviceToHost));
    // reference for the column form in double on the host (the row-major kernel rotates before the block, this one after: compare with a
    // host evaluation of THIS form instead)
    double maxerr = 0, maxref = 0;
    for (unsigned k = 0; k < N; k += 97) {
      double t0 = 0, t1 = 0, t2 = 0;
      for (unsigned d = rp[k]; d < rp[k + 1]; ++d) {
        const double* b = &hb[6 * (size_t)d]; const double* um = &hu[3 * (size_t)col[d]];
        t0 += b[0] * um[0] + b[1] * um[1] + b[2] * um[2]; t1 += b[1] * um[0] + b[3] * um[1] + b[4] * um[2]; t2 += b[2] * um[0] + b[4] * um[1] + b[5] * um[2];
      }
      const double* qq = &hq[4 * (size_t)k];
      const double R[9] = {qq[0], qq[1], qq[2], qq[3], qq[0] * qq[3], qq[1] * qq[2], qq[0] + qq[2], qq[1] - qq[3], qq[2] * qq[3]};
      const double* Mk = &hM[6 * (size_t)k]; const double* pk = &hp[3 * (size_t)k];
      const double r0 = Mk[0] * pk[0] + Mk[1] * pk[1] + Mk[2] * pk[2] - (R[0] * t0 + R[1] * t1 + R[2] * t2);
      const double r1 = Mk[1] * pk[0] + Mk[3] * pk[1] + Mk[4] * pk[2] - (R[3] * t0 + R[4] * t1 + R[5] * t2);
      const double r2 = Mk[2] * pk[0] + Mk[4] * pk[1] + Mk[5] * pk[2] - (R[6] * t0 + R[7] * t1 + R[8] * t2);
      maxerr = std::max({maxerr, std::fabs(r0 - yy[3 * (size_t)k]), std::fabs(r1 - yy[3 * (size_t)k + 1]), std::fabs(r2 - yy[3 * (size_t)k + 2])});
      maxref = std::max({maxref, std::fabs(r0), std::fabs(r1), std::fabs(r2)});
    }
    const double cb = ne * (4.0 + 48.0 + 2.0) + 2.0 * L.seg.size() + 48.0 * nwg * RB + 72.0 * N;
    printf("%-46s %8.1f us (+finish %5.1f)  %6.2f TB/s on its own %.3f GB, %zu WGs, pad %.1f %%, max err %.1e (|y| %.1e); per directed entry %.2f ns (C5's K3c: 8.9)\n", name, tm, t - tm, cb / t * 1e-6, cb * 1e-9, nwg,
           100.0 * ((double)ne / nd - 1.0), maxerr, maxref, 1e3 * t / nd);
    for (void* ptr : {(void*)d_wg, (void*)c2, (void*)pm, (void*)sg, (void*)b0, (void*)b1, (void*)b2, (void*)part}) CHK(hipFree(ptr));
  };
#define KERN(RB, EPL, PADU, PF) [](ArgsCol a, int grid, unsigned rb) { hipLaunchKernelGGL((k_mv_col<RB, EPL, PADU, PF>), dim3(grid), dim3(rb), 0, 0, a); }
  const unsigned nch512 = std::max(1u, (unsigned)(1764.0 * N / 100000.0 / ((N + 511) / 512) + 0.5));
  run(512, 1, 9, KERN(512, 1, false, true), "col-sorted RB=512 EPL=1 NCH=9 prefetch (the product's shape)");
  auto run_big = [&](unsigned RBB, unsigned T, unsigned NCH, auto kern, const char* name) {
    const BigLayout L = build_big(N, rp, col, RBB, T, NCH);
    const size_t ne = L.n_ent, nwg = L.wg.size();
    WgDesc* d_wg; unsigned* c2; uint16_t *pm, *sr; double2 *b0, *b1, *b2; double* part;
    CHK(hipMalloc(&d_wg, sizeof(WgDesc) * nwg)); CHK(hipMalloc(&c2, 4 * ne)); CHK(hipMalloc(&pm, 2 * ne)); CHK(hipMalloc(&sr, 2 * ne));
    CHK(hipMalloc(&b0, 16 * ne)); CHK(hipMalloc(&b1, 16 * ne)); CHK(hipMalloc(&b2, 16 * ne)); CHK(hipMalloc(&part, 24 * nwg * RBB));
    std::vector<double2> a0(ne), a1(ne), a2(ne);
    for (size_t e = 0; e < ne; ++e) {
      const unsigned d = L.src[e];
      if (d == 0xffffffffu) { a0[e] = a1[e] = a2[e] = make_double2(0, 0); continue; }
      a0[e] = make_double2(hb[6 * (size_t)d], hb[6 * (size_t)d + 1]); a1[e] = make_double2(hb[6 * (size_t)d + 2], hb[6 * (size_t)d + 3]); a2[e] = make_double2(hb[6 * (size_t)d + 4], hb[6 * (size_t)d + 5]);
    }
    CHK(hipMemcpy(d_wg, L.wg.data(), sizeof(WgDesc) * nwg, hipMemcpyHostToDevice)); CHK(hipMemcpy(c2, L.col.data(), 4 * ne, hipMemcpyHostToDevice));
    CHK(hipMemcpy(pm, L.perm.data(), 2 * ne, hipMemcpyHostToDevice)); CHK(hipMemcpy(sr, L.srow.data(), 2 * ne, hipMemcpyHostToDevice));
    CHK(hipMemcpy(b0, a0.data(), 16 * ne, hipMemcpyHostToDevice)); CHK(hipMemcpy(b1, a1.data(), 16 * ne, hipMemcpyHostToDevice)); CHK(hipMemcpy(b2, a2.data(), 16 * ne, hipMemcpyHostToDevice));
    ArgsBig a{d_wg, c2, pm, sr, b0, b1, b2, u, part, (unsigned)nwg};
    ArgsFin f{N, RBB, NCH, (unsigned)nwg, part, M, p, q, y2};
    const float t = timeit([&] { kern(a, (int)nwg); hipLaunchKernelGGL(k_mv_finish, dim3((N + 255) / 256), dim3(256), 0, 0, f); });
    const float tm = timeit([&] { kern(a, (int)nwg); });
    std::vector<double> yy(3 * (size_t)N);
    CHK(hipMemcpy(yy.data(), y2, 24 * (size_t)N, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (unsigned k = 0; k < N; k += 97) {
      double t0 = 0, t1 = 0, t2 = 0;
      for (unsigned d = rp[k]; d < rp[k + 1]; ++d) {
        const double* b = &hb[6 * (size_t)d]; const double* um = &hu[3 * (size_t)col[d]];
        t0 += b[0] * um[0] + b[1] * um[1] + b[2] * um[2]; t1 += b[1] * um[0] + b[3] * um[1] + b[4] * um[2]; t2 += b[2] * um[0] + b[4] * um[1] + b[5] * um[2];
      }
      const double* qq = &hq[4 * (size_t)k];
      const double R[9] = {qq[0], qq[1], qq[2], qq[3], qq[0] * qq[3], qq[1] * qq[2], qq[0] + qq[2], qq[1] - qq[3], qq[2] * qq[3]};
      const double* Mk = &hM[6 * (size_t)k]; const double* pk = &hp[3 * (size_t)k];
      const double r0 = Mk[0] * pk[0] + Mk[1] * pk[1] + Mk[2] * pk[2] - (R[0] * t0 + R[1] * t1 + R[2] * t2);
      const double r1 = Mk[1] * pk[0] + Mk[3] * pk[1] + Mk[4] * pk[2] - (R[3] * t0 + R[4] * t1 + R[5] * t2);
      const double r2 = Mk[2] * pk[0] + Mk[4] * pk[1] + Mk[5] * pk[2] - (R[6] * t0 + R[7] * t1 + R[8] * t2);
      maxerr = std::max({maxerr, std::fabs(r0 - yy[3 * (size_t)k]), std::fabs(r1 - yy[3 * (size_t)k + 1]), std::fabs(r2 - yy[3 * (size_t)k + 2])});
      maxref = std::max({maxref, std::fabs(r0), std::fabs(r1), std::fabs(r2)});
    }
    const double cb = ne * (4.0 + 48.0 + 2.0 + 2.0) + 48.0 * nwg * RBB + 72.0 * N;
    printf("%-46s %8.1f us (+finish %5.1f)  %6.2f TB/s on its own %.3f GB, %zu WGs, pad %.1f %%, max err %.1e (|y| %.1e); per directed entry %.2f ns (C5's K3c: 8.9)\n", name, tm, t - tm, cb / t * 1e-6, cb * 1e-9, nwg,
           100.0 * ((double)ne / nd - 1.0), maxerr, maxref, 1e3 * t / nd);
    for (void* ptr : {(void*)d_wg, (void*)c2, (void*)pm, (void*)sr, (void*)b0, (void*)b1, (void*)b2, (void*)part}) CHK(hipFree(ptr));
  };
#define KBIG(RBB, T) [](ArgsBig a, int grid) { hipLaunchKernelGGL((k_mv_big<RBB, T>), dim3(grid), dim3(T), 0, 0, a); }
  const unsigned nb4 = (N + 4095) / 4096, nb2 = (N + 2047) / 2048;
  run_big(4096, 512, std::max(1u, 768 / nb4), KBIG(4096, 512), "big blocks RBB=4096 T=512, ~768 WGs");
  run_big(4096, 512, std::max(1u, 1536 / nb4), KBIG(4096, 512), "big blocks RBB=4096 T=512, ~1536 WGs");
  run_big(4096, 1024, std::max(1u, 768 / nb4), KBIG(4096, 1024), "big blocks RBB=4096 T=1024, ~768 WGs");
  run_big(2048, 512, std::max(1u, 1536 / nb2), KBIG(2048, 512), "big blocks RBB=2048 T=512, ~1536 WGs");
  run_big(2048, 512, std::max(1u, 3072 / nb2), KBIG(2048, 512), "big blocks RBB=2048 T=512, ~3072 WGs");
  CHK(hipDeviceSynchronize());
  return 0;
}
