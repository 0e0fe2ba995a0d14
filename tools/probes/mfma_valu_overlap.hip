// Probe: do vector instructions issue in the shadow of a wave's own MFMAs on gfx950?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/probes/mfma_valu_overlap.hip && /tmp/mvo
// One block per CU, W waves per SIMD (block = 256 W threads); per loop step a wave issues 14 x v_mfma_f32_32x32x16_bf16
// and / or N vector instructions (v_fma_f32 or v_exp_f32), either in one burst after the MFMAs or dealt evenly into the
// MFMA gaps.  Prints shader cycles (s_memtime) per step and the wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE, int NV, bool EXP>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int steps) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.01f + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    constexpr int PER = (NV + 13) / 14;
#pragma unroll
    for (int m = 0; m < 14; ++m) {
      if (MODE & 1) {
        if (m < 7) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        else acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      }
      if ((MODE & 2) && !(MODE & 4)) {                       // vector work dealt into the gaps
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int k = (m * PER + j) & 15;
          if (EXP) v[k] = __builtin_amdgcn_exp2f(v[k]); else v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((MODE & 2) && (MODE & 4)) {                          // vector work in one burst behind the MFMAs
#pragma unroll
      for (int j = 0; j < PER * 14; ++j) {
        const int k = j & 15;
        if (EXP) v[k] = __builtin_amdgcn_exp2f(v[k]); else v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int NV, bool EXP>
void run(const char* tag, int waves_per_simd) {
  const int steps = 2000, blocks = 256, threads = 256 * waves_per_simd;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, NV, EXP><<<blocks, threads>>>(out, cyc, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE, NV, EXP><<<blocks, threads>>>(out, cyc, steps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double c = 0; for (auto x : h) c += (double)x; c /= blocks;
  printf("%-58s waves/SIMD %d: %7.0f cycles/step  %7.1f ns/step  (%.2f GHz)\n", tag, waves_per_simd, c / steps, ms * 1e6 / steps,
         c / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<1, 0, false>("14 MFMA 32x32x16", w);
    run<2, 84, false>("84 v_fma_f32", w);
    run<3, 84, false>("14 MFMA + 84 v_fma_f32, 6 per gap", w);
    run<7, 84, false>("14 MFMA + 84 v_fma_f32, one burst", w);
    run<2, 42, true>("42 v_exp_f32", w);
    run<3, 42, true>("14 MFMA + 42 v_exp_f32, 3 per gap", w);
    run<3, 84, true>("14 MFMA + 84 v_exp_f32, 6 per gap", w);
    run<3, 140, false>("14 MFMA + 140 v_fma_f32, 10 per gap", w);
  }
  return 0;
}
