// Probe: what does a DEPENDENCY between two layers cost on this chip, by the way it is expressed?
//   launches   R dependent launches replayed from a HIP graph (what the step graph does today: ~10 us of fixed cost per
//              implicit-GEMM launch)
//   fence      ONE persistent launch (one block per CU), layers separated by a grid barrier: an arrival counter + bounded
//              spin, plain stores / loads made visible by agent-scope fences (__threadfence: L2 write-back + invalidate)
//   sc1        the same, but the tile traffic itself is device-coherent (`sc1` stores / loads, as the in-launch split-K finish
//              of csrc/igemm.hip) and the barrier is only the relaxed agent-scope counter
// Every round a block WRITES a tile (0 / 16 / 64 / 256 KiB) and, behind the dependency, READS the tile another block -- on
// another XCD -- wrote in the same round, and checks every word (a stale line would show).  Reported: us per round.
// The plan of DESIGN 8 (one persistent launch per transformer block at the 16 x 16 / 8 x 8 levels) stands or falls with
// the gap between the first line and the other two.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int THREADS = 256;
constexpr long SPIN_LIMIT = 1L << 22;      // a barrier that is never completed ends the kernel with err = 1 instead of hanging

__device__ __forceinline__ void store16(bool sc1, u32x4* p, const u32x4& v) {
  if (sc1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else *p = v;
}
__device__ __forceinline__ u32x4 load16(bool sc1, const u32x4* p) {
  u32x4 v;
  if (sc1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  else v = *p;
  return v;
}

// one round of a block: write my tile of this round, [dependency], read + check the partner's tile of this round
__device__ __forceinline__ void write_tile(bool sc1, u32x4* tiles, int words16, int round, int block) {
  u32x4* mine = tiles + ((long)(round & 1) * gridDim.x + block) * words16;
  const unsigned tag = (unsigned)round * 1000003u + (unsigned)block * 7919u;
  for (int i = threadIdx.x; i < words16; i += THREADS) store16(sc1, mine + i, u32x4{tag, tag + (unsigned)i, tag ^ 0x5a5a5a5au, (unsigned)i});
}
__device__ __forceinline__ int read_tile(bool sc1, const u32x4* tiles, int words16, int round, int partner) {
  const u32x4* theirs = tiles + ((long)(round & 1) * gridDim.x + partner) * words16;
  const unsigned tag = (unsigned)round * 1000003u + (unsigned)partner * 7919u;
  int bad = 0;
  for (int i = threadIdx.x; i < words16; i += THREADS) {
    const u32x4 v = load16(sc1, theirs + i);
    bad += (v[0] != tag) | (v[1] != tag + (unsigned)i) | (v[2] != (tag ^ 0x5a5a5a5au)) | (v[3] != (unsigned)i);
  }
  return bad;
}

__global__ __launch_bounds__(THREADS) void layer_kernel(u32x4* tiles, int words16, int round, int* bad_out) {
  // the launch form: the PREVIOUS launch wrote round `round - 1`'s tiles... here one launch = write round r, and the next
  // launch reads it: so a launch reads round - 1 first (its dependency is the kernel boundary), then writes round
  const int block = blockIdx.x, partner = (block + 37) % gridDim.x;
  int bad = 0;
  if (round > 0) bad = read_tile(false, tiles, words16, round - 1, partner);
  write_tile(false, tiles, words16, round, block);
  if (bad) atomicAdd(bad_out, bad);
}

template <bool SC1>
__global__ __launch_bounds__(THREADS) void persistent_kernel(u32x4* tiles, int words16, int rounds, unsigned* counter, int* bad_out, int* err) {
  const int block = blockIdx.x, partner = (block + 37) % gridDim.x;
  int bad = 0;
  for (int r = 0; r < rounds; ++r) {
    write_tile(SC1, tiles, words16, r, block);
    // ---- grid barrier ----
    if (SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my sc1 stores are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!SC1) __threadfence();                                         // release: plain stores written back
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * gridDim.x;
      long spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > SPIN_LIMIT) { *err = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      if (!SC1) __threadfence();                                         // acquire: stale lines dropped
    }
    __syncthreads();
    bad += read_tile(SC1, tiles, words16, r, partner);
  }
  if (bad) atomicAdd(bad_out, bad);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = cus > 0 ? cus : 256, rounds = 200;
  const int sizes_kib[] = {0, 16, 64, 256};
  const long max_words16 = 256L * 1024 / 16;
  u32x4* tiles; unsigned* counter; int *bad, *err;
  CK(hipMalloc(&tiles, 2L * blocks * max_words16 * 16 + 64));
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&err, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("%d blocks of %d threads (one per CU), %d rounds; each round: write a tile, dependency, read + check the tile of block + 37\n", blocks, THREADS, rounds);
  for (int kib : sizes_kib) {
    const int words16 = kib * 1024 / 16;
    // ---- launches, replayed from a graph ----
    {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipMemsetAsync(bad, 0, 4, st));
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
      for (int r = 0; r <= rounds; ++r) hipLaunchKernelGGL(layer_kernel, dim3(blocks), dim3(THREADS), 0, st, tiles, words16, r, bad);
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      int hbad = 0; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
      printf("tile %3d KiB  launches (graph)      %7.2f us per round   mismatches %d\n", kib, best * 1e3f / (rounds + 1), hbad);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // ---- one persistent launch, grid barriers ----
    for (int sc1 = 0; sc1 < 2; ++sc1) {
      float best = 1e30f; int hbad = 0, herr = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(bad, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
        CK(hipEventRecord(e0, st));
        if (sc1) hipLaunchKernelGGL((persistent_kernel<true>), dim3(blocks), dim3(THREADS), 0, st, tiles, words16, rounds, counter, bad, err);
        else     hipLaunchKernelGGL((persistent_kernel<false>), dim3(blocks), dim3(THREADS), 0, st, tiles, words16, rounds, counter, bad, err);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        if (herr) break;
      }
      printf("tile %3d KiB  persistent, %-9s  %7.2f us per round   mismatches %d%s\n", kib, sc1 ? "sc1" : "fence", best * 1e3f / rounds, hbad,
             herr ? "   BARRIER TIMED OUT (blocks not co-resident?)" : "");
    }
  }
  return 0;
}
