// Peer-store collectives for gsfm_rot_shard (include/gsfm_rot.h): the per-iteration all-gather of a sharded PCG solve without a
// collective-library launch in the loop ("measure RCCL ring vs tree vs custom P2P": ~4 us of wire time for a
// direct exchange of 0.3 MB slices over the seven xGMI links in parallel, against ~27 us + 14 hop latencies for a ring).
//
// Every rank owns a MAILBOX in device memory -- [2 buffers][world slots][cap doubles], then world flag words -- allocated uncached /
// fine-grained and mapped into every peer through hipIpcMemHandle.  One collective = two kernels on the solver's stream:
//   k_peer_push   copies the rank's slice into slot[rank] of EVERY rank's mailbox (plain stores through the mapped pointers: over xGMI on a
//                 multi-GPU node); the last workgroup to finish publishes the call's sequence number in flag[rank] of every mailbox
//                 (system-scope release after a system-scope fence of every storing thread);
//   k_peer_pull   waits (system-scope acquire, bounded) until all `world` flags of its OWN mailbox carry the sequence number, then copies
//                 the slots into the caller's buffer -- or, for the sum all-reduce, adds them in rank order (the same order on every rank:
//                 bit-identical sums everywhere).
// The sequence number is a device-resident counter advanced by the last workgroup of k_peer_pull, so both kernels take no per-call arguments and a chunk of PCG
// iterations containing them replays as a hipGraph.  Buffers alternate with the call's parity: a rank can be at most one call ahead of a
// peer (it needs that peer's flag to finish its own call), so the slot it overwrites was read two calls ago.
// A call larger than the mailbox, or a communicator that could not map every peer, goes to the fallback callbacks (RCCL, or the
// torch.distributed ones): the choice is the caller's (sharding.make_comm(..., exchange="peer")).
// A wait that runs into its bound (a peer that never delivers) is an ERROR of the solve it happens in and the end of the peer path for this
// communicator: the flag is looked at on entry of every call -- the first call that sees it returns non-zero (the solver then returns
// GSFM_ERR_COMM; a rank whose waits all succeeded meets the same bound one call later, because the failed rank has stopped storing) -- and every
// later call goes to the fallback callbacks, call by call.  gsfm_peer_error_take() lets the host layer fail a solve whose LAST calls timed
// out (inside a replayed hipGraph no callback runs on the host).
// Built with hipcc --offload-arch=gfx950 into libgsfm_peer.so.  Tested without a multi-GPU node: N processes sharing one GPU exchange
// IPC handles the same way (tests/test_gpu_sharded.py: bitwise equality with the host-staged path at 2 / 3 / 8 ranks).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace {

constexpr int MAX_WORLD = 64;
constexpr int PEER_BLOCK = 256;

struct PeerDev {
  int rank, world;
  unsigned long long cap;            // doubles per slot
  double* box[MAX_WORLD];            // mailbox base of every rank (own entry = local pointer)
  unsigned long long* flags[MAX_WORLD];   // flag words of every rank's mailbox: flags[p][r] = last call rank r has delivered to rank p
  unsigned long long* seq;           // local: calls completed
  unsigned int* done_blocks;         // local: [0] workgroups of the current push that have finished storing, [1] of the current pull that have finished reading
  int* error;                        // local, host-visible: 1 = a wait ran into its bound
};

__device__ __forceinline__ double* slot_ptr(const PeerDev& d, int owner, unsigned long long call, int slot) {
  return d.box[owner] + ((call & 1ull) * (unsigned long long)d.world + (unsigned long long)slot) * d.cap;
}

__global__ void __launch_bounds__(PEER_BLOCK) k_peer_push(PeerDev d, const double* src, unsigned long long count) {
  const unsigned long long call = *d.seq + 1ull;   // (advanced by k_peer_pull, which runs after this kernel on the same stream)
  for (int p = 0; p < d.world; ++p) {
    double* dst = slot_ptr(d, p, call, d.rank);
    for (unsigned long long i = (unsigned long long)blockIdx.x * PEER_BLOCK + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * PEER_BLOCK)
      __builtin_nontemporal_store(src[i], dst + i);
  }
  __threadfence_system();            // this thread's stores are visible system-wide before the block is counted
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int n = __hip_atomic_fetch_add(d.done_blocks, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (n == gridDim.x - 1) {        // last workgroup: everything is out; publish the call in every mailbox
      __hip_atomic_store(d.done_blocks, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      for (int p = 0; p < d.world; ++p) __hip_atomic_store(d.flags[p] + d.rank, call, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// mode 0: all-gather -- dst + p * count <- slot p (own slot excluded: the caller's slice is already in place); mode 1: dst[i] <- sum_p slot p [i]
__global__ void __launch_bounds__(PEER_BLOCK) k_peer_pull(PeerDev d, double* dst, unsigned long long count, int mode) {
  __shared__ int ok;
  const unsigned long long call = *d.seq + 1ull;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if (threadIdx.x < (unsigned)d.world) {
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(d.flags[d.rank] + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < call) {
      // (a communicator that has timed out once does not wait again: the calls still in flight -- a replayed PCG chunk holds eight of them --
      // leave at once instead of spending 5 s each; the host fails the solve at its next look)
      if (*(volatile int*)d.error != 0 || wall_clock64() - t0 > 500000000ull) { ok = 0; break; }   // ~5 s of the 100 MHz wall clock: a peer that never arrives must not hang the GPU
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (!ok) {   // give up on this call (flagged for the host), but keep the call counter moving so that later calls line up again
    if (threadIdx.x == 0) {
      *d.error = 1;
      const unsigned int n = __hip_atomic_fetch_add(d.done_blocks + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (n == gridDim.x - 1) { __hip_atomic_store(d.done_blocks + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *d.seq = call; }
    }
    return;
  }
  const unsigned long long stride = (unsigned long long)gridDim.x * PEER_BLOCK, i0 = (unsigned long long)blockIdx.x * PEER_BLOCK + threadIdx.x;
  if (mode == 0) {
    for (int p = 0; p < d.world; ++p) {
      if (p == d.rank) continue;
      const double* s = slot_ptr(d, d.rank, call, p);
      double* o = dst + (unsigned long long)p * count;
      for (unsigned long long i = i0; i < count; i += stride) o[i] = __builtin_nontemporal_load(s + i);
    }
  } else {
    for (unsigned long long i = i0; i < count; i += stride) {
      double acc = 0.0;
      for (int p = 0; p < d.world; ++p) acc += __builtin_nontemporal_load(slot_ptr(d, d.rank, call, p) + i);
      dst[i] = acc;
    }
  }
  // the last workgroup to finish advances the call counter (every workgroup has read it by then: it reads it before anything else)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int n = __hip_atomic_fetch_add(d.done_blocks + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (n == gridDim.x - 1) { __hip_atomic_store(d.done_blocks + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *d.seq = call; }
  }
}

typedef int (*coll_fn)(void*, double*, size_t, void*);

struct Peer {
  PeerDev dev{};
  void* local = nullptr;             // own mailbox allocation
  void* mapped[MAX_WORLD] = {};      // peers' allocations as opened here
  size_t bytes = 0, flag_off = 0;
  unsigned long long* d_seq = nullptr;
  unsigned int* d_done = nullptr;
  int* h_error = nullptr;            // pinned
  bool connected = false;
  bool reported = false;             // the time-out has been returned to the solver (or taken by the host layer) once
  void* fb_ctx = nullptr; coll_fn fb_gather = nullptr, fb_reduce = nullptr;
  long n_peer = 0, n_fallback = 0;
};
thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return 1; }

}  // namespace

extern "C" {

const char* gsfm_peer_last_error(void) { return g_err.c_str(); }
int gsfm_peer_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

// Allocates this rank's mailbox on the current HIP device and writes its IPC handle (gsfm_peer_handle_bytes() bytes) to handle_out.
void* gsfm_peer_create(int rank, int world, size_t cap_doubles, char* handle_out) {
  if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world || cap_doubles == 0) { fail("gsfm_peer_create: bad arguments"); return nullptr; }
  Peer* P = new Peer;
  P->dev.rank = rank; P->dev.world = world; P->dev.cap = cap_doubles;
  P->flag_off = 2 * (size_t)world * cap_doubles * sizeof(double);
  P->bytes = P->flag_off + (size_t)world * sizeof(unsigned long long);
  // fine-grained (system-coherent, not cached across kernels) device memory: peers store into it and the local kernels poll it
  // (no coarse-grained hipMalloc fallback: in such memory a system-scope poll may never see a peer's store -- every call would time out; the
  // caller falls back to the collective communicator instead, sharding.make_comm)
  hipError_t e = hipExtMallocWithFlags(&P->local, P->bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&P->local, P->bytes, hipDeviceMallocFinegrained); }
  if (e != hipSuccess) { (void)hipGetLastError(); fail(std::string("fine-grained mailbox allocation: ") + hipGetErrorString(e)); delete P; return nullptr; }
  bool ok = hipMemset(P->local, 0, P->bytes) == hipSuccess;
  ok = ok && hipMalloc((void**)&P->d_seq, sizeof(unsigned long long)) == hipSuccess && hipMemset(P->d_seq, 0, sizeof(unsigned long long)) == hipSuccess;
  ok = ok && hipMalloc((void**)&P->d_done, 2 * sizeof(unsigned int)) == hipSuccess && hipMemset(P->d_done, 0, 2 * sizeof(unsigned int)) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&P->h_error, sizeof(int), hipHostMallocMapped) == hipSuccess;
  hipIpcMemHandle_t h;
  ok = ok && hipIpcGetMemHandle(&h, P->local) == hipSuccess;
  if (!ok) { fail(std::string("mailbox set-up: ") + hipGetErrorString(hipGetLastError())); if (P->local) (void)hipFree(P->local); delete P; return nullptr; }
  *P->h_error = 0;
  std::memcpy(handle_out, &h, sizeof(h));
  (void)hipDeviceSynchronize();
  P->dev.seq = P->d_seq; P->dev.done_blocks = P->d_done; P->dev.error = P->h_error;
  return P;
}

// all_handles: world x gsfm_peer_handle_bytes() bytes, rank order (own entry ignored).  Every rank must have created its mailbox before.
int gsfm_peer_connect(void* ctx, const char* all_handles) {
  Peer* P = static_cast<Peer*>(ctx);
  const int w = P->dev.world;
  for (int p = 0; p < w; ++p) {
    void* base = P->local;
    if (p != P->dev.rank) {
      hipIpcMemHandle_t h;
      std::memcpy(&h, all_handles + (size_t)p * sizeof(h), sizeof(h));
      const hipError_t e = hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { (void)hipGetLastError(); return fail(std::string("hipIpcOpenMemHandle(rank ") + std::to_string(p) + "): " + hipGetErrorString(e)); }
      P->mapped[p] = base;
    }
    P->dev.box[p] = static_cast<double*>(base);
    P->dev.flags[p] = reinterpret_cast<unsigned long long*>(static_cast<char*>(base) + P->flag_off);
  }
  P->connected = true;
  return 0;
}

void gsfm_peer_set_fallback(void* ctx, void* fb_ctx, void* all_gather_fn, void* all_reduce_fn) {
  Peer* P = static_cast<Peer*>(ctx);
  P->fb_ctx = fb_ctx; P->fb_gather = reinterpret_cast<coll_fn>(all_gather_fn); P->fb_reduce = reinterpret_cast<coll_fn>(all_reduce_fn);
}

static int peer_collective(Peer* P, double* buf, size_t count, void* stream, int mode) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool broken = *static_cast<volatile int*>(P->h_error) != 0;
  if (broken && !P->reported) {
    P->reported = true;
    return fail("peer exchange: a wait for a peer's slice ran into its 5 s bound in an earlier call of this solve -- its result is invalid; this communicator uses its fallback collectives from now on");
  }
  if (!P->connected || count > P->dev.cap || broken) {
    coll_fn f = mode ? P->fb_reduce : P->fb_gather;
    if (!f) return fail("peer exchange: the call does not fit the mailbox and there is no fallback");
    P->n_fallback++;
    return f(P->fb_ctx, buf, count, stream);
  }
  if (count == 0) return 0;
  P->n_peer++;
  const double* src = mode ? buf : buf + (size_t)P->dev.rank * count;
  const unsigned grid = (unsigned)((count + PEER_BLOCK - 1) / PEER_BLOCK) < 64u ? (unsigned)((count + PEER_BLOCK - 1) / PEER_BLOCK) : 64u;
  hipLaunchKernelGGL(k_peer_push, dim3(grid), dim3(PEER_BLOCK), 0, s, P->dev, src, (unsigned long long)count);
  hipLaunchKernelGGL(k_peer_pull, dim3(grid), dim3(PEER_BLOCK), 0, s, P->dev, buf, (unsigned long long)count, mode);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(std::string("peer exchange launch: ") + hipGetErrorString(e));
  return 0;
}
int gsfm_peer_all_gather(void* ctx, double* buf, size_t count, void* stream) { return peer_collective(static_cast<Peer*>(ctx), buf, count, stream, 0); }
int gsfm_peer_all_reduce_sum(void* ctx, double* buf, size_t count, void* stream) { return peer_collective(static_cast<Peer*>(ctx), buf, count, stream, 1); }

// 1 if a wait ever ran into its bound (the data of that call is then garbage: the caller must treat the solve as failed)
int gsfm_peer_error(void* ctx) { return *static_cast<Peer*>(ctx)->h_error; }
// 1 exactly once per communicator: a time-out has happened and no collective call has returned it to the solver yet (call behind a
// synchronisation of the solver's stream, i.e. after a solve)
int gsfm_peer_error_take(void* ctx) {
  Peer* P = static_cast<Peer*>(ctx);
  if (*static_cast<volatile int*>(P->h_error) == 0 || P->reported) return 0;
  P->reported = true;
  return 1;
}
// test hook: behave as if a wait had timed out
void gsfm_peer_inject_error(void* ctx) { *static_cast<Peer*>(ctx)->h_error = 1; }
long gsfm_peer_calls(void* ctx, int fallback) { Peer* P = static_cast<Peer*>(ctx); return fallback ? P->n_fallback : P->n_peer; }

void gsfm_peer_destroy(void* ctx) {
  Peer* P = static_cast<Peer*>(ctx);
  if (!P) return;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < P->dev.world; ++p) if (P->mapped[p]) (void)hipIpcCloseMemHandle(P->mapped[p]);
  if (P->local) (void)hipFree(P->local);
  if (P->d_seq) (void)hipFree(P->d_seq);
  if (P->d_done) (void)hipFree(P->d_done);
  if (P->h_error) (void)hipHostFree(P->h_error);
  delete P;
}

}  // extern "C"
