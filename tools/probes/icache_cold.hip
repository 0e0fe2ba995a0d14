// Probe: what does a wave pay for executing straight-line code it has never fetched -- the situation of every epilogue
// (fully unrolled, run once per block) and of the first k-steps of every launch in a chain of DIFFERENT kernels (each
// launch starts with a cold instruction cache; a pair of CUs shares 64 KB of it, the ring kernels are 43-62 KB of code).
// One wave per block, one block per CU; the body is N KiB of independent instructions executed `reps` times in a
// runtime loop: reps = 1 is the cold pass, the difference between reps = 2 and reps = 1 the warm pass of the same code.
//   snop   s_nop 0            4 bytes, issues every cycle or so: fetch-bound if anything is
//   valu   v_add_f32 (VOP3)   8 bytes, 4+ cycles of issue each: what an unrolled epilogue looks like to the fetcher
//   hipcc --offload-arch=gfx950 -O2 tools/probes/icache_cold.hip -o /tmp/icache_cold && /tmp/icache_cold
#include <hip/hip_runtime.h>
#include <stdio.h>

#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define R256(x) R16(R16(x))
#define R1024(x) R4(R256(x))

// KIB kibibytes of body per pass
template <int KIB, bool VALU>
__global__ __launch_bounds__(64) void body_kernel(unsigned long long* out, int reps, float* sink) {
  float a = (float)threadIdx.x, b = 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (VALU) {
      // 128 VOP3 instructions of 8 bytes = 1 KiB per R16(R4(2 instr))
#pragma unroll
      for (int k = 0; k < KIB; ++k) {
        R16(R4(asm volatile("v_add_f32_e64 %0, %0, %1\n\tv_add_f32_e64 %1, %1, 1.0" : "+v"(a), "+v"(b));))
      }
    } else {
      // 256 s_nop of 4 bytes = 1 KiB
#pragma unroll
      for (int k = 0; k < KIB; ++k) {
        R256(asm volatile("s_nop 0");)
      }
    }
    asm volatile("" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (a == 123456.0f) *sink = a + b;
}

// keeps the chip busy (and its clocks up: an idle GPU drops them within milliseconds) in front of every measured launch
__global__ __launch_bounds__(1024) void busy_kernel(float* sink, int iters) {
  float a = (float)threadIdx.x, b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) a = a * b + 0.5f;
  }
  if (a == 123456.0f) *sink = a;
}

template <int KIB, bool VALU>
static double run(unsigned long long* d, unsigned long long* h, float* sink, int blocks, int reps) {
  // a DIFFERENT kernel in between evicts nothing by itself, but a launch starts with an invalidated instruction cache anyway
  hipLaunchKernelGGL(busy_kernel, dim3(blocks * 2), dim3(1024), 0, 0, sink, 20000);      // ~10 ms of full-rate FMA right before
  hipLaunchKernelGGL((body_kernel<KIB, VALU>), dim3(blocks), dim3(64), 0, 0, d, reps, sink);
  hipDeviceSynchronize();
  hipMemcpy(h, d, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks; ++i) s += (double)h[i];
  return s / blocks;
}

template <int KIB, bool VALU>
static void report(unsigned long long* d, unsigned long long* h, float* sink, int blocks) {
  double c1 = 1e30, c2 = 1e30, c3 = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    const double a = run<KIB, VALU>(d, h, sink, blocks, 1), b = run<KIB, VALU>(d, h, sink, blocks, 2), c = run<KIB, VALU>(d, h, sink, blocks, 3);
    if (a < c1) c1 = a;
    if (b < c2) c2 = b;
    if (c < c3) c3 = c;
  }
  // s_memtime is the SHADER clock: cycles
  const double cold = c1, warm = c3 - c2;
  printf("%-5s %3d KiB of code: cold pass %8.0f cycles, warm pass %8.0f cycles -> %5.0f cycles of fetch stalls (%4.1f per KiB)\n", VALU ? "valu" : "snop", KIB, cold,
         warm, cold - warm, (cold - warm) / KIB);
}

int main() {
  int dev = 0, cus = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const int blocks = cus > 0 ? cus : 256;
  unsigned long long* d; float* sink;
  hipMalloc(&d, blocks * sizeof(unsigned long long)); hipMalloc(&sink, 4);
  unsigned long long* h = (unsigned long long*)malloc(blocks * sizeof(unsigned long long));
  hipLaunchKernelGGL(busy_kernel, dim3(blocks * 2), dim3(1024), 0, 0, sink, 600000);       // ~0.3 s: clocks up before anything is measured
  hipDeviceSynchronize();
  printf("%d blocks of one wave (one per CU); body executed 1 / 2 / 3 times in a runtime loop; times from s_memtime (shader cycles), mean over blocks\n", blocks);
  report<4, false>(d, h, sink, blocks);
  report<16, false>(d, h, sink, blocks);
  report<48, false>(d, h, sink, blocks);
  report<4, true>(d, h, sink, blocks);
  report<16, true>(d, h, sink, blocks);
  report<48, true>(d, h, sink, blocks);
  return 0;
}
