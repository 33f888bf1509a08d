// Cost of a grid-wide barrier inside one persistent kernel on MI355X (8 XCDs, non-coherent L2s): the number that decides whether a
// persistent PCG kernel can beat two dependent launches per iteration in the latency regime.
// Each round: every workgroup writes a slice of a vector, barrier, reads a slice written by a workgroup on another XCD and checks it.
// Spins are bounded: a barrier that is not passed within LIMIT polls raises an error flag and the kernel exits.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bench_gridsync tools/archive/bench_gridsync.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int LIMIT = 2000000;

// state layout (unsigned words, every counter on its own 4 KiB page so that contended atomics land in different channels):
//   [0]            top counter        [1024]  release flag        [2048 + 1024 g]  counter of group g
template <int VARIANT>
__device__ __forceinline__ bool grid_barrier(unsigned* st, unsigned round, int* err) {   // round = 1, 2, 3, ...
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned G = gridDim.x;
    if (VARIANT == 0) {          // flat: one counter, everybody polls it
      __hip_atomic_fetch_add(st, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int spins = 0;
      while (__hip_atomic_load(st, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < G * round) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > LIMIT) { *err = 1; break; }
      }
    } else {
      bool last;
      if (VARIANT == 1) {        // flat counter, separate release flag written by the last arrival
        last = __hip_atomic_fetch_add(st, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == G * round - 1;
      } else {                   // two levels: groups of GROUP consecutive workgroups, then the group leaders
        constexpr unsigned GROUP = VARIANT;
        const unsigned g = blockIdx.x / GROUP, ng = (G + GROUP - 1) / GROUP;
        const unsigned gsize = (g == ng - 1) ? G - g * GROUP : GROUP;
        last = false;
        if (__hip_atomic_fetch_add(st + 2048 + 1024 * g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gsize * round - 1)
          last = __hip_atomic_fetch_add(st, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == ng * round - 1;
      }
      if (last) __hip_atomic_store(st + 1024, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else {
        int spins = 0;
        while (__hip_atomic_load(st + 1024, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > LIMIT) { *err = 1; break; }
        }
      }
    }
  }
  __syncthreads();
  return *err == 0;
}

template <int PAYLOAD, int VARIANT>   // doubles written / read per thread per round
__global__ void __launch_bounds__(256) k_rounds(unsigned* cnt, double* vec, int rounds, int* err, unsigned long long* bad) {
  const unsigned G = gridDim.x;
  unsigned long long nbad = 0;
  for (int it = 0; it < rounds; ++it) {
    for (int p = 0; p < PAYLOAD; ++p) vec[((size_t)blockIdx.x * PAYLOAD + p) * 256 + threadIdx.x] = (double)(it * 7 + p);
    if (!grid_barrier<VARIANT>(cnt, 2 * it + 1, err)) return;
    const unsigned other = (blockIdx.x + 3) % G;   // consecutive workgroup ids sit on different XCDs
    for (int p = 0; p < PAYLOAD; ++p) {
      const double v = vec[((size_t)other * PAYLOAD + p) * 256 + threadIdx.x];
      if (v != (double)(it * 7 + p)) ++nbad;
    }
    if (!grid_barrier<VARIANT>(cnt, 2 * it + 2, err)) return;
  }
  if (nbad) atomicAdd(bad, nbad);
}

template <int PAYLOAD, int VARIANT>
void run(unsigned* st, double* vec, int* err, unsigned long long* bad, hipEvent_t e0, hipEvent_t e1) {
  const int rounds = 500;
  for (int G : {32, 64, 128, 256}) {
    float best = 1e30f; int h_err = 0; unsigned long long h_bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(st, 0, 4 * (2048 + 1024 * 64))); CK(hipMemset(err, 0, 4)); CK(hipMemset(bad, 0, 8));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_rounds<PAYLOAD, VARIANT>), dim3(G), dim3(256), 0, 0, st, vec, rounds, err, bad);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      CK(hipMemcpy(&h_err, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost));
      if (h_err) break;
    }
    printf("variant %2d  payload %d doubles/thread  G=%4d workgroups: %.2f us per barrier  timeout=%d  stale reads=%llu\n", VARIANT, PAYLOAD, G,
           1e3 * best / (2 * rounds), h_err, h_bad);
  }
}

int main() {
  unsigned* st; int* err; unsigned long long* bad; double* vec;
  CK(hipMalloc(&st, 4 * (2048 + 1024 * 64))); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 8)); CK(hipMalloc(&vec, (size_t)1024 * 8 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  run<0, 0>(st, vec, err, bad, e0, e1); run<0, 1>(st, vec, err, bad, e0, e1); run<0, 8>(st, vec, err, bad, e0, e1); run<0, 16>(st, vec, err, bad, e0, e1);
  run<1, 0>(st, vec, err, bad, e0, e1); run<1, 1>(st, vec, err, bad, e0, e1); run<1, 8>(st, vec, err, bad, e0, e1); run<1, 16>(st, vec, err, bad, e0, e1);
  return 0;
}
