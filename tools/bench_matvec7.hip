// Dev micro-benchmark (not part of the product): the Laplacian-form mat-vec with EACH EDGE'S BLOCK STREAMED ONCE (round-3 review, item 5).
//
// In the body frame the normal matrix is a graph Laplacian with one symmetric 3 x 3 weight B_e per UNDIRECTED edge e = (i, j): the
// product needs t_i += B_e u_j and t_j += B_e u_i.  The column-sorted layout of the product (K3c, bench_matvec6) stores the block under
// both rows: 2 x 52 B per edge.  Here an edge is stored once, 48 + 8 B:
//   * rows are cut into STRIPS of RB cameras; edge (i < j) belongs to strip(i); inside a strip the edges are sorted by column j;
//   * a workgroup owns a run of sub-chunks (SUB edges, one per lane) of ONE strip and keeps u_I and the strip's row sums t_I in LDS
//     (2 x 24 B x RB: 96 KB at RB = 2048);
//   * forward, t_i += B u_j: u_j is a line-sharing gather (sorted columns), the contribution goes to its slot of a ROW-sorted LDS staging
//     area; after the barrier the lane at the head of each row's run of slots adds the run, in slot order, into t_I (one owner per row and
//     sub-chunk: plain read-modify-write, deterministic, no atomics);
//   * reverse, t_j += B u_i: u_i comes from LDS, the contribution is staged in POSITION order; edges of one column are contiguous, so the
//     head of each column's run adds it in position order and stores one 24-byte partial per (strip, column) -- runs never straddle a
//     sub-chunk (padding: ~0.2 %);
//   * a finishing kernel adds, per camera, the strip partials of its column side and the workgroup partials of its row side in fixed order,
//     applies R_k once and adds M_k p_k.
// Bytes at the C5 shape (100k cameras, 10M edges): stream 560 MB + column partials 24 B x N x strips / 2 written and read (2 x 59 MB at
// RB = 2048) + row partials 2 x 48 KB per workgroup  ~  0.72 GB against 1.10 GB for K3c.
//   hipcc --offload-arch=gfx950 -O3 -o bench_matvec7 bench_matvec7.hip ;  ./bench_matvec7 [N] [E]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ double2 nt2(const double2* p) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
#define PADJ 0xffffffffu

struct SymWg { unsigned first_sub, n_sub, strip, pad; };
struct ArgsSym {
  const SymWg* wg; const uint2* rec;          // rec.x = column j (PADJ = padding); rec.y = local row | forward slot << 12
  const double2 *b0, *b1, *b2;                // (b00 b01) (b02 b11) (b12 b22)
  const double* u;                            // 3 per camera
  double* partJ;                              // [3][n_strips][N]
  double* partI;                              // [3][n_wg][RB]
  unsigned N, n_wg;
};

template <int RB, int SUB>
__global__ void __launch_bounds__(SUB) k_mv_sym(ArgsSym a) {
  extern __shared__ double lds[];
  double* uI = lds;                    // [3][RB]
  double* tI = uI + 3 * RB;            // [3][RB]
  double* fst = tI + 3 * RB;           // [3][SUB]  forward contributions, row-sorted slots
  double* rst = fst + 3 * SUB;         // [3][SUB]  reverse contributions, position order
  unsigned* jst = (unsigned*)(rst + 3 * SUB);   // [SUB] column per position
  unsigned short* frow = (unsigned short*)(jst + SUB);   // [SUB] local row per slot (0xffff: padding)
  const SymWg w = a.wg[blockIdx.x];
  const unsigned t = threadIdx.x, row0 = w.strip * RB;
  for (unsigned r = t; r < RB; r += SUB) {
    const unsigned k = min(row0 + r, a.N - 1);
    uI[r] = a.u[3 * (size_t)k]; uI[RB + r] = a.u[3 * (size_t)k + 1]; uI[2 * RB + r] = a.u[3 * (size_t)k + 2];
    tI[r] = 0.0; tI[RB + r] = 0.0; tI[2 * RB + r] = 0.0;
  }
  uint2 rc; double2 A, B, C;
  auto request = [&](unsigned s) {
    const size_t e = (size_t)(w.first_sub + s) * SUB + t;
    rc.x = __builtin_nontemporal_load(&a.rec[e].x); rc.y = __builtin_nontemporal_load(&a.rec[e].y);
    A = nt2(a.b0 + e); B = nt2(a.b1 + e); C = nt2(a.b2 + e);
  };
  if (w.n_sub) request(0);
  __syncthreads();
  const size_t planeJ = (size_t)gridDim.y * 0 + (size_t)a.N;   // (partJ plane stride per strip = N)
  for (unsigned s = 0; s < w.n_sub; ++s) {
    const unsigned j = rc.x, il = rc.y & 0xfffu, fs = rc.y >> 12;
    const bool pad = j == PADJ;
    const double* uj = a.u + 3 * (size_t)(pad ? 0u : j);
    const double j0 = uj[0], j1 = uj[1], j2 = uj[2];
    const double i0 = uI[il], i1 = uI[RB + il], i2 = uI[2 * RB + il];
    fst[fs] = A.x * j0 + A.y * j1 + B.x * j2; fst[SUB + fs] = A.y * j0 + B.y * j1 + C.x * j2; fst[2 * SUB + fs] = B.x * j0 + C.x * j1 + C.y * j2;
    frow[fs] = pad ? (unsigned short)0xffff : (unsigned short)il;
    rst[t] = A.x * i0 + A.y * i1 + B.x * i2; rst[SUB + t] = A.y * i0 + B.y * i1 + C.x * i2; rst[2 * SUB + t] = B.x * i0 + C.x * i1 + C.y * i2;
    jst[t] = j;
    if (s + 1 < w.n_sub) request(s + 1);
    __syncthreads();
    {  // forward: head of each row's run of slots
      const unsigned row = frow[t], prev = t ? frow[t - 1] : 0xfffeu;
      if (row != 0xffffu && row != prev) {
        double s0 = fst[t], s1 = fst[SUB + t], s2 = fst[2 * SUB + t];
        for (unsigned k = t + 1; k < SUB && frow[k] == row; ++k) { s0 += fst[k]; s1 += fst[SUB + k]; s2 += fst[2 * SUB + k]; }
        tI[row] += s0; tI[RB + row] += s1; tI[2 * RB + row] += s2;
      }
    }
    {  // reverse: head of each column's run of positions
      const unsigned jj = jst[t], prev = t ? jst[t - 1] : (PADJ - 1u);
      if (jj != PADJ && jj != prev) {
        double s0 = rst[t], s1 = rst[SUB + t], s2 = rst[2 * SUB + t];
        for (unsigned k = t + 1; k < SUB && jst[k] == jj; ++k) { s0 += rst[k]; s1 += rst[SUB + k]; s2 += rst[2 * SUB + k]; }
        double* o = a.partJ + (size_t)w.strip * planeJ + jj;
        const size_t cplane = (size_t)a.N * ((a.N + RB - 1) / RB);
        o[0] = s0; o[cplane] = s1; o[2 * cplane] = s2;
      }
    }
    __syncthreads();
  }
  const size_t plane = (size_t)a.n_wg * RB;
  for (unsigned r = t; r < RB; r += SUB) {
    const size_t o = (size_t)blockIdx.x * RB + r;
    a.partI[o] = tI[r]; a.partI[plane + o] = tI[RB + r]; a.partI[2 * plane + o] = tI[2 * RB + r];
  }
}

// y_k = M_k p_k - R_k (sum_{s <= strip(k)} partJ[s][k] + sum_{wg of strip(k)} partI[wg][k mod RB])
struct ArgsFin { unsigned N, RB, n_wg, n_strips; const unsigned* wg_first; const double* partJ; const double* partI; const double* M; const double* p; const double2* q; double* y; };
__global__ void __launch_bounds__(256) k_sym_finish(ArgsFin a) {
  const unsigned k = blockIdx.x * 256 + threadIdx.x;
  if (k >= a.N) return;
  const unsigned st = k / a.RB, r = k % a.RB;
  const size_t cplane = (size_t)a.N * a.n_strips, plane = (size_t)a.n_wg * a.RB;
  double t0 = 0, t1 = 0, t2 = 0;
  for (unsigned s = 0; s <= st; ++s) { const size_t o = (size_t)s * a.N + k; t0 += a.partJ[o]; t1 += a.partJ[cplane + o]; t2 += a.partJ[2 * cplane + o]; }
  for (unsigned w = a.wg_first[st]; w < a.wg_first[st + 1]; ++w) { const size_t o = (size_t)w * a.RB + r; t0 += a.partI[o]; t1 += a.partI[plane + o]; t2 += a.partI[2 * plane + o]; }
  const double2 qa = a.q[2 * (size_t)k], qb = a.q[2 * (size_t)k + 1];
  const double R0 = qa.x, R1 = qa.y, R2 = qb.x, R3 = qb.y, R4 = qa.x * qb.y, R5 = qa.y * qb.x, R6 = qa.x + qb.x, R7 = qa.y - qb.y, R8 = qb.x * qb.y;
  const double* M = a.M + 6 * (size_t)k; const double* pk = a.p + 3 * (size_t)k;
  a.y[3 * (size_t)k] = M[0] * pk[0] + M[1] * pk[1] + M[2] * pk[2] - (R0 * t0 + R1 * t1 + R2 * t2);
  a.y[3 * (size_t)k + 1] = M[1] * pk[0] + M[3] * pk[1] + M[4] * pk[2] - (R3 * t0 + R4 * t1 + R5 * t2);
  a.y[3 * (size_t)k + 2] = M[2] * pk[0] + M[4] * pk[1] + M[5] * pk[2] - (R6 * t0 + R7 * t1 + R8 * t2);
}

// what K2 would have to do to feed this layout from its own (directed, column-sorted) order: the block of every edge stored at a position
// given by an index plane -- three scattered 16-byte stores per edge -- against the coalesced stores it does now
__global__ void __launch_bounds__(256) k_scatter_store(const unsigned* pos, double2* b0, double2* b1, double2* b2, size_t n, int scattered) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const size_t d = scattered ? pos[e] : e;
  const double v = (double)e;
  __builtin_nontemporal_store(v, &b0[d].x); __builtin_nontemporal_store(v + 1, &b0[d].y);
  __builtin_nontemporal_store(v + 2, &b1[d].x); __builtin_nontemporal_store(v + 3, &b1[d].y);
  __builtin_nontemporal_store(v + 4, &b2[d].x); __builtin_nontemporal_store(v + 5, &b2[d].y);
}

template <typename F> float timeit(F f, int reps = 10) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  f(); f();
  CHK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) f();
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3f;
}

struct Edge { unsigned i, j; size_t id; };

template <int RB, int SUB>
void run(unsigned N, const std::vector<Edge>& edges, const std::vector<double>& hb, const std::vector<double>& hu, const std::vector<double>& hM, const std::vector<double>& hp,
         const std::vector<double>& hq, unsigned wg_target, const char* name) {
  const size_t E = edges.size();
  const unsigned n_strips = (N + RB - 1) / RB;
  std::vector<Edge> ed(edges);
  std::sort(ed.begin(), ed.end(), [&](const Edge& a, const Edge& b) { const unsigned sa = a.i / RB, sb = b.i / RB; return sa != sb ? sa < sb : (a.j != b.j ? a.j < b.j : a.i < b.i); });
  std::vector<uint2> rec; std::vector<size_t> src;   // per position
  std::vector<unsigned> strip_sub0(n_strips + 1, 0);
  size_t pos = 0;
  for (unsigned s = 0; s < n_strips; ++s) {
    strip_sub0[s] = (unsigned)(rec.size() / SUB);
    size_t lo = pos; while (pos < E && ed[pos].i / RB == s) ++pos;
    size_t e = lo;
    while (e < pos) {   // fill one sub-chunk with whole column runs
      const size_t base = rec.size();
      size_t fill = 0;
      while (e < pos) {
        size_t r = e; while (r < pos && ed[r].j == ed[e].j) ++r;
        if (r - e > (size_t)SUB) { printf("a column run longer than a sub-chunk: unsupported in this prototype\n"); exit(1); }
        if (fill + (r - e) > (size_t)SUB) break;
        for (; e < r; ++e, ++fill) { rec.push_back(make_uint2(ed[e].j, ed[e].i % RB)); src.push_back(ed[e].id); }
      }
      for (; fill < (size_t)SUB; ++fill) { rec.push_back(make_uint2(PADJ, 0)); src.push_back((size_t)-1); }
      // forward slots: stable sort of the positions by local row, padding last
      std::vector<unsigned> order(SUB);
      for (unsigned k = 0; k < (unsigned)SUB; ++k) order[k] = k;
      std::stable_sort(order.begin(), order.end(), [&](unsigned x, unsigned y) {
        const unsigned rx = rec[base + x].x == PADJ ? 0xffffu : rec[base + x].y, ry = rec[base + y].x == PADJ ? 0xffffu : rec[base + y].y; return rx < ry; });
      for (unsigned slot = 0; slot < (unsigned)SUB; ++slot) rec[base + order[slot]].y |= slot << 12;
    }
  }
  strip_sub0[n_strips] = (unsigned)(rec.size() / SUB);
  const size_t n_sub = rec.size() / SUB, n_pos = rec.size();
  // workgroups: each strip gets a share of wg_target proportional to its sub-chunks (at least one if it has any)
  std::vector<SymWg> wgs; std::vector<unsigned> wg_first(n_strips + 1, 0);
  for (unsigned s = 0; s < n_strips; ++s) {
    wg_first[s] = (unsigned)wgs.size();
    const unsigned ns = strip_sub0[s + 1] - strip_sub0[s];
    if (!ns) continue;
    const unsigned nw = std::max(1u, std::min(ns, (unsigned)std::llround((double)wg_target * ns / n_sub)));
    for (unsigned c = 0; c < nw; ++c) { const unsigned lo = (unsigned)((uint64_t)ns * c / nw), hi = (unsigned)((uint64_t)ns * (c + 1) / nw); wgs.push_back(SymWg{strip_sub0[s] + lo, hi - lo, s, 0}); }
  }
  wg_first[n_strips] = (unsigned)wgs.size();
  const unsigned n_wg = (unsigned)wgs.size();
  // device
  SymWg* d_wg; uint2* d_rec; double2 *b0, *b1, *b2, *q; double *u, *M, *p, *y, *partJ, *partI; unsigned* d_wgf;
  CHK(hipMalloc(&d_wg, sizeof(SymWg) * n_wg)); CHK(hipMalloc(&d_rec, 8 * n_pos)); CHK(hipMalloc(&b0, 16 * n_pos)); CHK(hipMalloc(&b1, 16 * n_pos)); CHK(hipMalloc(&b2, 16 * n_pos));
  CHK(hipMalloc(&q, 32 * (size_t)N)); CHK(hipMalloc(&u, 24 * (size_t)N)); CHK(hipMalloc(&M, 48 * (size_t)N)); CHK(hipMalloc(&p, 24 * (size_t)N)); CHK(hipMalloc(&y, 24 * (size_t)N));
  CHK(hipMalloc(&partJ, 24 * (size_t)N * n_strips)); CHK(hipMalloc(&partI, 24 * (size_t)n_wg * RB)); CHK(hipMalloc(&d_wgf, 4 * (n_strips + 1)));
  CHK(hipMemset(partJ, 0, 24 * (size_t)N * n_strips));
  {
    std::vector<double2> a0(n_pos), a1(n_pos), a2(n_pos);
    for (size_t e = 0; e < n_pos; ++e) {
      if (src[e] == (size_t)-1) { a0[e] = a1[e] = a2[e] = make_double2(0, 0); continue; }
      const double* b = &hb[6 * src[e]];
      a0[e] = make_double2(b[0], b[1]); a1[e] = make_double2(b[2], b[3]); a2[e] = make_double2(b[4], b[5]);
    }
    CHK(hipMemcpy(b0, a0.data(), 16 * n_pos, hipMemcpyHostToDevice)); CHK(hipMemcpy(b1, a1.data(), 16 * n_pos, hipMemcpyHostToDevice)); CHK(hipMemcpy(b2, a2.data(), 16 * n_pos, hipMemcpyHostToDevice));
  }
  CHK(hipMemcpy(d_wg, wgs.data(), sizeof(SymWg) * n_wg, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_rec, rec.data(), 8 * n_pos, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_wgf, wg_first.data(), 4 * (n_strips + 1), hipMemcpyHostToDevice));
  CHK(hipMemcpy(u, hu.data(), 24 * (size_t)N, hipMemcpyHostToDevice)); CHK(hipMemcpy(M, hM.data(), 48 * (size_t)N, hipMemcpyHostToDevice));
  CHK(hipMemcpy(p, hp.data(), 24 * (size_t)N, hipMemcpyHostToDevice)); CHK(hipMemcpy(q, hq.data(), 32 * (size_t)N, hipMemcpyHostToDevice));
  const size_t lds_bytes = 8 * (6 * (size_t)RB + 6 * (size_t)SUB) + 4 * (size_t)SUB + 2 * (size_t)SUB;
  CHK(hipFuncSetAttribute((const void*)k_mv_sym<RB, SUB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  ArgsSym a{d_wg, d_rec, b0, b1, b2, u, partJ, partI, N, n_wg};
  ArgsFin f{N, (unsigned)RB, n_wg, n_strips, d_wgf, partJ, partI, M, p, q, y};
  auto mv = [&] { hipLaunchKernelGGL((k_mv_sym<RB, SUB>), dim3(n_wg), dim3(SUB), lds_bytes, 0, a); };
  auto fin = [&] { hipLaunchKernelGGL(k_sym_finish, dim3((N + 255) / 256), dim3(256), 0, 0, f); };
  mv(); CHK(hipDeviceSynchronize()); CHK(hipGetLastError());
  const float tm = timeit(mv), tt = timeit([&] { mv(); fin(); });
  std::vector<double> yy(3 * (size_t)N);
  CHK(hipMemcpy(yy.data(), y, 24 * (size_t)N, hipMemcpyDeviceToHost));
  // host reference
  std::vector<double> tsum(3 * (size_t)N, 0.0);
  for (const Edge& e : edges) {
    const double* b = &hb[6 * e.id]; const double* ui = &hu[3 * (size_t)e.i]; const double* uj = &hu[3 * (size_t)e.j];
    tsum[3 * (size_t)e.i] += b[0] * uj[0] + b[1] * uj[1] + b[2] * uj[2]; tsum[3 * (size_t)e.i + 1] += b[1] * uj[0] + b[3] * uj[1] + b[4] * uj[2]; tsum[3 * (size_t)e.i + 2] += b[2] * uj[0] + b[4] * uj[1] + b[5] * uj[2];
    tsum[3 * (size_t)e.j] += b[0] * ui[0] + b[1] * ui[1] + b[2] * ui[2]; tsum[3 * (size_t)e.j + 1] += b[1] * ui[0] + b[3] * ui[1] + b[4] * ui[2]; tsum[3 * (size_t)e.j + 2] += b[2] * ui[0] + b[4] * ui[1] + b[5] * ui[2];
  }
  double maxerr = 0, maxref = 0;
  for (unsigned k = 0; k < N; ++k) {
    const double* qq = &hq[4 * (size_t)k]; const double* t = &tsum[3 * (size_t)k];
    const double R[9] = {qq[0], qq[1], qq[2], qq[3], qq[0] * qq[3], qq[1] * qq[2], qq[0] + qq[2], qq[1] - qq[3], qq[2] * qq[3]};
    const double* Mk = &hM[6 * (size_t)k]; const double* pk = &hp[3 * (size_t)k];
    const double r0 = Mk[0] * pk[0] + Mk[1] * pk[1] + Mk[2] * pk[2] - (R[0] * t[0] + R[1] * t[1] + R[2] * t[2]);
    const double r1 = Mk[1] * pk[0] + Mk[3] * pk[1] + Mk[4] * pk[2] - (R[3] * t[0] + R[4] * t[1] + R[5] * t[2]);
    const double r2 = Mk[2] * pk[0] + Mk[4] * pk[1] + Mk[5] * pk[2] - (R[6] * t[0] + R[7] * t[1] + R[8] * t[2]);
    maxerr = std::max({maxerr, std::fabs(r0 - yy[3 * (size_t)k]), std::fabs(r1 - yy[3 * (size_t)k + 1]), std::fabs(r2 - yy[3 * (size_t)k + 2])});
    maxref = std::max({maxref, std::fabs(r0), std::fabs(r1), std::fabs(r2)});
  }
  // bit-reproducibility: a second product must give the same bits
  mv(); fin();
  std::vector<double> y2(3 * (size_t)N);
  CHK(hipMemcpy(y2.data(), y, 24 * (size_t)N, hipMemcpyDeviceToHost));
  const bool same = std::equal(yy.begin(), yy.end(), y2.begin());
  // bytes: stream + column partials (written once per (strip, column with edges): ~ N (n_strips + 1) / 2, read by the finish) + row partials + vectors
  const double colp = 24.0 * N * (n_strips + 1) / 2.0;
  const double bytes = 56.0 * n_pos + 2.0 * colp + 2.0 * 24.0 * n_wg * RB + 24.0 * n_wg * RB /* u_I loads */ + 152.0 * N;
  printf("%-40s %8.1f us (+finish %5.1f) = %6.2f TB/s on its own %.3f GB | 80 B/edge survey bytes: %.2f of 8 TB/s | %u WGs, %u strips, pad %.2f %%, LDS %zu KB, max err %.1e (|y| %.1e), bitwise repeatable: %s\n",
         name, tm, tt - tm, bytes / tt * 1e-6, bytes * 1e-9, (80.0 * E + 48.0 * N) / tt * 1e-6 / 8.0, n_wg, n_strips, 100.0 * ((double)n_pos / E - 1.0), lds_bytes / 1024, maxerr, maxref,
         same ? "yes" : "NO");
  for (void* ptr : {(void*)d_wg, (void*)d_rec, (void*)b0, (void*)b1, (void*)b2, (void*)q, (void*)u, (void*)M, (void*)p, (void*)y, (void*)partJ, (void*)partI, (void*)d_wgf}) CHK(hipFree(ptr));
}

int main(int argc, char** argv) {
  const unsigned N = argc > 1 ? atoi(argv[1]) : 100000; const size_t E = argc > 2 ? atoll(argv[2]) : 10000000;
  std::mt19937_64 rng(1);
  std::vector<Edge> edges(E);
  for (size_t e = 0; e < E; ++e) { unsigned i = rng() % N, j = rng() % N; while (j == i) j = rng() % N; edges[e] = Edge{std::min(i, j), std::max(i, j), e}; }
  std::uniform_real_distribution<double> U(-1, 1);
  std::vector<double> hb(6 * E), hu(3 * (size_t)N), hM(6 * (size_t)N), hp(3 * (size_t)N), hq(4 * (size_t)N);
  for (auto& v : hb) v = U(rng);
  for (auto& v : hu) v = U(rng);
  for (auto& v : hM) v = U(rng);
  for (auto& v : hp) v = U(rng);
  for (auto& v : hq) v = U(rng);
  printf("%u cameras, %zu undirected edges (mean degree %.1f); K3c streams 2 x 52 B per edge = %.3f GB per product, measured 178-193 us at this shape\n", N, E, 2.0 * E / N, (104.0 * E + 152.0 * N) * 1e-9);
  run<2048, 1024>(N, edges, hb, hu, hM, hp, hq, 256, "sym RB=2048 SUB=1024 ~256 WGs");
  run<2048, 1024>(N, edges, hb, hu, hM, hp, hq, 512, "sym RB=2048 SUB=1024 ~512 WGs");
  run<2048, 1024>(N, edges, hb, hu, hM, hp, hq, 1024, "sym RB=2048 SUB=1024 ~1024 WGs");
  run<2048, 512>(N, edges, hb, hu, hM, hp, hq, 512, "sym RB=2048 SUB=512 ~512 WGs");
  run<1024, 1024>(N, edges, hb, hu, hM, hp, hq, 512, "sym RB=1024 SUB=1024 ~512 WGs (2 per CU)");
  run<1024, 512>(N, edges, hb, hu, hM, hp, hq, 1024, "sym RB=1024 SUB=512 ~1024 WGs");
  // the store K2 would need: block of every edge to an arbitrary position
  {
    std::vector<unsigned> perm(E);
    for (size_t e = 0; e < E; ++e) perm[e] = (unsigned)e;
    std::shuffle(perm.begin(), perm.end(), rng);
    unsigned* d_pos; double2 *b0, *b1, *b2;
    CHK(hipMalloc(&d_pos, 4 * E)); CHK(hipMalloc(&b0, 16 * E)); CHK(hipMalloc(&b1, 16 * E)); CHK(hipMalloc(&b2, 16 * E));
    CHK(hipMemcpy(d_pos, perm.data(), 4 * E, hipMemcpyHostToDevice));
    for (int sc = 0; sc < 2; ++sc) {
      const float t = timeit([&] { hipLaunchKernelGGL(k_scatter_store, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, 0, d_pos, b0, b1, b2, E, sc); });
      printf("48-byte block store per edge, %-9s: %8.1f us (%.2f TB/s of payload)\n", sc ? "scattered" : "coalesced", t, 48.0 * E / t * 1e-6);
    }
  }
  CHK(hipDeviceSynchronize());
  return 0;
}
