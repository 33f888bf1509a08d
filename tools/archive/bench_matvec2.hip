// Dev micro-benchmark: LDS-staged, column-blocked block-CSR mat-vec prototype.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define CB 4096
#define TT 1024

struct Tile { unsigned block, step_begin, step_end, pad; };
struct Args {
  const Tile* tiles; unsigned n_cams;
  const unsigned* step_ptr;    // [n_steps+1] entry offsets
  const unsigned* step_slot;   // [n_steps] first slot id of the step
  const unsigned* erow; const unsigned short* ecol;
  const double2 *h0, *h1, *h2, *h3; const double* h4;
  const double* p; double* yseg;  // 3 per slot
};

__device__ __forceinline__ double2 ntl2(const double2* p) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }

__global__ void __launch_bounds__(TT) k_mv_tile(Args a) {
  __shared__ double2 pxy[CB];
  __shared__ double pz[CB];
  const Tile t = a.tiles[blockIdx.x];
  const unsigned base = t.block * CB, cnt = min((unsigned)CB, a.n_cams - base);
  for (unsigned c = threadIdx.x; c < cnt; c += TT) { const double* pp = a.p + 3 * (size_t)(base + c); pxy[c] = make_double2(pp[0], pp[1]); pz[c] = pp[2]; }
  __syncthreads();
  const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned st = t.step_begin + wave; st < t.step_end; st += TT / 64) {
    const unsigned b = a.step_ptr[st], len = a.step_ptr[st + 1] - b;
    double y0 = 0, y1 = 0, y2 = 0; unsigned row = 0xffffffffu;
    if (lane < len) {
      const unsigned d = b + lane;
      row = __builtin_nontemporal_load(a.erow + d);
      const unsigned c = __builtin_nontemporal_load(a.ecol + d);
      const double2 A = ntl2(a.h0 + d), B = ntl2(a.h1 + d), C = ntl2(a.h2 + d), D = ntl2(a.h3 + d);
      const double E = __builtin_nontemporal_load(a.h4 + d);
      const double2 pa = pxy[c]; const double p2 = pz[c];
      y0 = A.x * pa.x + A.y * pa.y + B.x * p2; y1 = B.y * pa.x + C.x * pa.y + C.y * p2; y2 = D.x * pa.x + D.y * pa.y + E * p2;
    }
    // segmented reduction over runs of equal row ids
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned r2 = __shfl_down(row, off, 64);
      const double v0 = __shfl_down(y0, off, 64), v1 = __shfl_down(y1, off, 64), v2 = __shfl_down(y2, off, 64);
      if (lane + off < 64 && r2 == row) { y0 += v0; y1 += v1; y2 += v2; }
    }
    const unsigned prev = __shfl_up(row, 1, 64);
    const bool head = lane < len && (lane == 0 || prev != row);
    const unsigned long long hm = __ballot(head);
    if (head) {
      const unsigned slot = a.step_slot[st] + __popcll(hm & ((1ull << lane) - 1ull));
      double* o = a.yseg + 3 * (size_t)slot;
      o[0] = y0; o[1] = y1; o[2] = y2;
    }
  }
}
__global__ void __launch_bounds__(256) k_mv_final(unsigned n, const unsigned* cam_ptr, const unsigned* cam_slots, const double* yseg, double* y) {
  const unsigned k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  double s0 = 0, s1 = 0, s2 = 0;
  for (unsigned t = cam_ptr[k]; t < cam_ptr[k + 1]; ++t) { const double* v = yseg + 3 * (size_t)cam_slots[t]; s0 += v[0]; s1 += v[1]; s2 += v[2]; }
  y[3 * (size_t)k] = s0; y[3 * (size_t)k + 1] = s1; y[3 * (size_t)k + 2] = s2;
}

template <typename F> float timeit(F f, int reps = 10) {
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  f(); f();
  CHK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) f();
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const unsigned N = argc > 1 ? atoi(argv[1]) : 100000, DEG = argc > 2 ? atoi(argv[2]) : 200;
  const size_t nd = (size_t)N * DEG;
  const unsigned nblk = (N + CB - 1) / CB;
  std::mt19937 rng(1);
  // entries (row, col) -> sort by (col block, row)
  std::vector<unsigned> row(nd), col(nd);
  for (unsigned r = 0; r < N; ++r) for (unsigned k = 0; k < DEG; ++k) { row[(size_t)r * DEG + k] = r; col[(size_t)r * DEG + k] = rng() % N; }
  std::vector<unsigned> bstart(nblk + 1, 0), perm(nd);
  for (size_t d = 0; d < nd; ++d) bstart[col[d] / CB + 1]++;
  for (unsigned b = 0; b < nblk; ++b) bstart[b + 1] += bstart[b];
  { std::vector<unsigned> f(bstart.begin(), bstart.end() - 1); for (size_t d = 0; d < nd; ++d) perm[f[col[d] / CB]++] = (unsigned)d; }  // stable: rows stay sorted
  std::vector<unsigned> erow(nd); std::vector<unsigned short> ecol(nd);
  for (size_t t = 0; t < nd; ++t) { erow[t] = row[perm[t]]; ecol[t] = (unsigned short)(col[perm[t]] % CB); }
  // steps: whole row segments, <= 64 entries
  std::vector<unsigned> step_ptr, step_slot; std::vector<Tile> tiles;
  std::vector<std::pair<unsigned, unsigned>> slot_cam;  // (camera, slot)
  unsigned slot = 0;
  const size_t per_wg = std::max<size_t>(16384, nd / 512);
  for (unsigned b = 0; b < nblk; ++b) {
    size_t d = bstart[b]; const size_t hi = bstart[b + 1];
    const unsigned first_step = (unsigned)step_ptr.size();
    std::vector<unsigned> tile_breaks; size_t since = 0;
    while (d < hi) {
      step_ptr.push_back((unsigned)d); step_slot.push_back(slot);
      size_t e = d; unsigned used = 0;
      while (e < hi) {
        size_t f = e; while (f < hi && erow[f] == erow[e]) ++f;
        unsigned seg = (unsigned)(f - e);
        if (used + seg > 64) { if (used == 0) { seg = 64; f = e + 64; } else break; }
        slot_cam.push_back({erow[e], slot}); ++slot; used += seg; e = f;
        if (used == 64) break;
      }
      since += e - d; d = e;
      if (since >= per_wg) { tile_breaks.push_back((unsigned)step_ptr.size()); since = 0; }
    }
    unsigned sb = first_step;
    for (unsigned br : tile_breaks) { tiles.push_back(Tile{b, sb, br, 0}); sb = br; }
    if (sb < step_ptr.size()) tiles.push_back(Tile{b, sb, (unsigned)step_ptr.size(), 0});
  }
  step_ptr.push_back((unsigned)nd);
  const unsigned n_steps = (unsigned)step_slot.size(), n_slots = slot;
  std::sort(slot_cam.begin(), slot_cam.end());
  std::vector<unsigned> cam_ptr(N + 1, 0), cam_slots(n_slots);
  for (auto& sc : slot_cam) cam_ptr[sc.first + 1]++;
  for (unsigned k = 0; k < N; ++k) cam_ptr[k + 1] += cam_ptr[k];
  for (size_t t = 0; t < slot_cam.size(); ++t) cam_slots[t] = slot_cam[t].second;
  printf("entries %zu steps %u (%.1f entries/step) slots %u tiles %zu\n", nd, n_steps, (double)nd / n_steps, n_slots, tiles.size());

  unsigned *d_sp, *d_ss, *d_erow, *d_cp, *d_cs; unsigned short* d_ecol; Tile* d_tiles;
  double2 *h0, *h1, *h2, *h3; double *h4, *p, *yseg, *y;
  CHK(hipMalloc(&d_sp, 4 * (n_steps + 1))); CHK(hipMalloc(&d_ss, 4 * n_steps)); CHK(hipMalloc(&d_erow, 4 * nd)); CHK(hipMalloc(&d_ecol, 2 * nd));
  CHK(hipMalloc(&d_cp, 4 * (N + 1))); CHK(hipMalloc(&d_cs, 4 * (size_t)n_slots)); CHK(hipMalloc(&d_tiles, sizeof(Tile) * tiles.size()));
  CHK(hipMalloc(&h0, 16 * nd)); CHK(hipMalloc(&h1, 16 * nd)); CHK(hipMalloc(&h2, 16 * nd)); CHK(hipMalloc(&h3, 16 * nd)); CHK(hipMalloc(&h4, 8 * nd));
  CHK(hipMalloc(&p, 24 * (size_t)N)); CHK(hipMalloc(&yseg, 24 * (size_t)n_slots)); CHK(hipMalloc(&y, 24 * (size_t)N));
  CHK(hipMemcpy(d_sp, step_ptr.data(), 4 * (n_steps + 1), hipMemcpyHostToDevice)); CHK(hipMemcpy(d_ss, step_slot.data(), 4 * n_steps, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_erow, erow.data(), 4 * nd, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_ecol, ecol.data(), 2 * nd, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_cp, cam_ptr.data(), 4 * (N + 1), hipMemcpyHostToDevice)); CHK(hipMemcpy(d_cs, cam_slots.data(), 4 * (size_t)n_slots, hipMemcpyHostToDevice));
  CHK(hipMemcpy(d_tiles, tiles.data(), sizeof(Tile) * tiles.size(), hipMemcpyHostToDevice));
  { std::vector<double> hp(3 * (size_t)N); for (auto& v : hp) v = (rng() % 1000) * 1e-3; CHK(hipMemcpy(p, hp.data(), 24 * (size_t)N, hipMemcpyHostToDevice));
    std::vector<double> ones(2 * nd, 1.0); CHK(hipMemcpy(h0, ones.data(), 16 * nd, hipMemcpyHostToDevice)); }
  CHK(hipMemset(h1, 0, 16 * nd)); CHK(hipMemset(h2, 0, 16 * nd)); CHK(hipMemset(h3, 0, 16 * nd)); CHK(hipMemset(h4, 0, 8 * nd));
  Args a{d_tiles, N, d_sp, d_ss, d_erow, d_ecol, h0, h1, h2, h3, h4, p, yseg};
  const float t1 = timeit([&] { hipLaunchKernelGGL(k_mv_tile, dim3((unsigned)tiles.size()), dim3(TT), 0, 0, a); });
  const float t2 = timeit([&] { hipLaunchKernelGGL(k_mv_final, dim3((N + 255) / 256), dim3(256), 0, 0, N, d_cp, d_cs, yseg, y); });
  const float t12 = timeit([&] { hipLaunchKernelGGL(k_mv_tile, dim3((unsigned)tiles.size()), dim3(TT), 0, 0, a);
                                  hipLaunchKernelGGL(k_mv_final, dim3((N + 255) / 256), dim3(256), 0, 0, N, d_cp, d_cs, yseg, y); });
  printf("tile kernel %.1f us (%.2f TB/s of 78 B/entry)  final %.1f us  both %.1f us\n", t1, 78.0 * nd / t1 * 1e-6, t2, t12);
  // check: y0 of row r = sum over entries of (p0 + p1) since H00=H01=1, rest 0
  std::vector<double> hy(3 * (size_t)N), hp(3 * (size_t)N);
  CHK(hipMemcpy(hy.data(), y, 24 * (size_t)N, hipMemcpyDeviceToHost)); CHK(hipMemcpy(hp.data(), p, 24 * (size_t)N, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (unsigned r = 0; r < N; r += 997) { double s = 0; for (unsigned k = 0; k < DEG; ++k) { const unsigned c = col[(size_t)r * DEG + k]; s += hp[3 * (size_t)c] + hp[3 * (size_t)c + 1]; } maxerr = std::max(maxerr, std::abs(s - hy[3 * (size_t)r])); }
  printf("check max err %.3e\n", maxerr);
  return 0;
}
