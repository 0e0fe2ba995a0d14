// Probe: how much vector-instruction issue fits in the shadow of a wave's own MFMAs on gfx950, instruction by instruction
// (inline asm, so hipcc neither packs the f32 operations into v_pk_* nor moves them).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfill tools/probes/mfma_fill.hip && /tmp/mfill
// One block per CU, W waves per SIMD.  A loop step issues NM MFMAs; gap g carries the fillers the pattern names.
// Prints shader cycles per step (s_memtime) and wall ns per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define MF32(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MF16(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define ADD(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define FMA(x, y, z) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define MAX3(x, y, z) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define OR3(x, y, z) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))
#define CVT(d, x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y))
#define PKMAX3(x, y, z) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(x) : "v"(y), "v"(z))

// MODE: 0 MFMA32 only | 1 +6 v_add / gap | 2 +6 v_fma / gap | 3 softmax mix (per 14 gaps: 32 exp, 16 cvt_pk, 8 or3)
//       4 fillers of mode 3 alone | 5 mode 3 with max3 (16) instead of or3 (8) | 6 fillers of mode 1 alone
//       7 28 x MFMA16 only | 8 28 x MFMA16 + the mode-3 mix | 9 mode 3 + 16 v_fma (the un-folded subtraction) + 16 max3 (today's mix)
//       10 11 x MFMA32 + mode-3 mix | 11 mode 3 with 5 exp in a row per gap pattern (burstier)
template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* cyc, int steps) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
  f32x16 acc0, acc1;
  f32x4 c0, c1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  for (int i = 0; i < 4; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  float v[32];
  unsigned w[16];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 0.01f - i;
  for (int i = 0; i < 16; ++i) w[i] = i;
  unsigned orr = 0;
  float mx = 0.f;
  const float k1 = 1.0001f, k2 = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    constexpr bool M32 = MODE <= 3 || MODE == 5 || MODE == 9 || MODE == 10 || MODE == 11;
    constexpr bool M16 = MODE == 7 || MODE == 8;
    constexpr bool MIX = MODE == 3 || MODE == 4 || MODE == 5 || MODE == 8 || MODE == 9 || MODE == 10 || MODE == 11;
    constexpr int NG = MODE == 10 ? 11 : 14;
    int e = 0, cv = 0, o3 = 0, f = 0, m3 = 0;       // compile-time counters after unrolling
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (M32) { if (g & 1) MF32(acc1); else MF32(acc0); }
      if (M16) { MF16(c0); }
      if (MODE == 1 || MODE == 6) {
#pragma unroll
        for (int j = 0; j < 6; ++j) ADD(v[(g * 6 + j) & 31], k1);
      }
      if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 6; ++j) FMA(v[(g * 6 + j) & 31], k1, k2);
      }
      if (MIX) {
        // deal 32 exp / 16 cvt / 8 or3 (or 16 max3) evenly over NG gaps
        const int e_end = (32 * (g + 1)) / NG, c_end = (16 * (g + 1)) / NG, o_end = (8 * (g + 1)) / NG;
        const int m_end = (16 * (g + 1)) / NG;
        if (MODE == 9) {
          for (; f < m_end; ++f) { FMA(v[(2 * f) & 31], k1, k2); FMA(v[(2 * f + 1) & 31], k1, k2); }
        }
        for (; e < e_end; ++e) EXP(v[e & 31]);
        for (; cv < c_end; ++cv) CVT(w[cv & 15], v[(2 * cv) & 31], v[(2 * cv + 1) & 31]);
        if (MODE == 5 || MODE == 9) {
          for (; m3 < m_end; ++m3) MAX3(mx, v[(2 * m3) & 31], v[(2 * m3 + 1) & 31]);
        } else {
          for (; o3 < o_end; ++o3) OR3(orr, w[(2 * o3) & 15], w[(2 * o3 + 1) & 15]);
        }
      }
      if (M16) { MF16(c1); }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = mx + (float)orr;
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + (float)w[i];
  for (int i = 0; i < 32; ++i) r += v[i];
  for (int i = 0; i < 4; ++i) r += c0[i] + c1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Role split inside a 512-thread block (two waves per SIMD): waves 0-3 issue NM MFMAs per step, waves 4-7 the softmax mix
// (32 exp + 16 cvt_pk + 8 or3); ROLE bit 0 = the MFMA half runs, bit 1 = the vector half runs, bit 2 = a barrier per step.
template <int ROLE, int NM>
__global__ __launch_bounds__(512) void split(float* out, unsigned long long* cyc, int steps) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float v[32];
  unsigned w[16];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 0.01f - i;
  for (int i = 0; i < 16; ++i) w[i] = i;
  unsigned orr = 0;
  const bool mf = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) < 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int s = 0; s < steps; ++s) {
    if (mf) {
      if (ROLE & 1) {
#pragma unroll
        for (int g = 0; g < NM; ++g) { if (g & 1) MF32(acc1); else MF32(acc0); }
      }
    } else if (ROLE & 2) {
#pragma unroll
      for (int e = 0; e < 32; ++e) EXP(v[e]);
#pragma unroll
      for (int c = 0; c < 16; ++c) CVT(w[c], v[2 * c], v[2 * c + 1]);
#pragma unroll
      for (int o3 = 0; o3 < 8; ++o3) OR3(orr, w[2 * o3], w[2 * o3 + 1]);
    }
    if (ROLE & 4) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float r = (float)orr;
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + (float)w[i];
  for (int i = 0; i < 32; ++i) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// Emulation of attention_pipe_kernel's step at two waves per SIMD: 14 MFMAs, each followed by 2-3 exp + cvt_pk (+ or3);
// EMU bit 0 = a barrier per step, bit 1 = an LDS fragment read behind every MFMA (8 x 2 ds_read_b64_tr_b16, 6 x ds_read_b128)
// that nobody consumes (s_waitcnt lgkmcnt(0) before the barrier), bit 2 = the MFMA's A operand IS last step's read (counted
// waits), bit 3 = two ds_write_b128 at the top of the step.
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
template <int EMU>
__global__ __launch_bounds__(512) void emu(float* out, unsigned long long* cyc, int steps) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(0.5f + i); }
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
  __syncthreads();
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  float v[32];
  unsigned w[16];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 0.01f - i;
  for (int i = 0; i < 16; ++i) w[i] = i;
  unsigned orr = 0;
  u32x4_t fr[14];
  for (int i = 0; i < 14; ++i) fr[i] = u32x4_t{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  const unsigned lane = threadIdx.x & 63;
  const unsigned base = (threadIdx.x >> 6) * 4096 + lane * 16;
  for (int s = 0; s < steps; ++s) {
    if (EMU & 8) {
      asm volatile("ds_write_b128 %0, %1" :: "v"(base + 32768), "v"(fr[0]) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:1024" :: "v"(base + 32768), "v"(fr[1]) : "memory");
    }
#pragma unroll
    for (int g = 0; g < 14; ++g) {
      if (EMU & 4) {
        // operand = the fragment read behind this MFMA one step ago: at most 13 younger reads may still be in flight
        asm volatile("s_waitcnt lgkmcnt(13)" ::: "memory");
        if (g & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(fr[g]), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(fr[g]), "v"(b));
      } else {
        if (g & 1) MF32(acc1); else MF32(acc0);
      }
      EXP(v[(2 * g) & 31]); EXP(v[(2 * g + 1) & 31]);
      if (g % 3 == 0) { EXP(v[(g + 7) & 31]); }
      CVT(w[g & 15], v[(2 * g) & 31], v[(2 * g + 1) & 31]);
      if (g & 1) OR3(orr, w[g & 15], w[(g + 1) & 15]);
      if (EMU & 2) {
        if (g < 8) {
          u32x2_t lo, hi;
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(base + g * 64) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(hi) : "v"(base + g * 64) : "memory");
          fr[g] = u32x4_t{lo[0], lo[1], hi[0], hi[1]};
        } else {
          asm volatile("ds_read_b128 %0, %1" : "=v"(fr[g]) : "v"(base + g * 16) : "memory");
        }
      }
    }
    if (EMU & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (EMU & 1) __builtin_amdgcn_s_barrier();
  }
  float r = (float)orr;
  for (int i = 0; i < 16; ++i) r += acc0[i] + acc1[i] + (float)w[i];
  for (int i = 0; i < 32; ++i) r += v[i];
  for (int i = 0; i < 14; ++i) r += (float)fr[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int EMU>
void run_emu(const char* tag) {
  const int steps = 4000, blocks = 256, threads = 512;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  emu<EMU><<<blocks, threads>>>(out, cyc, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  emu<EMU><<<blocks, threads>>>(out, cyc, steps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-72s %7.1f ns/step (wall, two wave tiles per SIMD)\n", tag, ms * 1e6 / steps);
  hipFree(out); hipFree(cyc);
}

template <int ROLE, int NM>
void run_split(const char* tag) {
  const int steps = 4000, blocks = 256, threads = 512;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  split<ROLE, NM><<<blocks, threads>>>(out, cyc, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  split<ROLE, NM><<<blocks, threads>>>(out, cyc, steps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-64s %7.1f ns/step (wall)\n", tag, ms * 1e6 / steps);
  hipFree(out); hipFree(cyc);
}

template <int MODE>
void run(const char* tag, int waves_per_simd) {
  const int steps = 4000, blocks = 256, threads = 256 * waves_per_simd;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<blocks, threads>>>(out, cyc, steps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE><<<blocks, threads>>>(out, cyc, steps);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double c = 0; for (auto x : h) c += (double)x; c /= blocks;
  printf("%-64s waves/SIMD %d: %7.0f cycles/step  %7.1f ns/step  (%.2f GHz)\n", tag, waves_per_simd, c / steps,
         ms * 1e6 / steps, c / (ms * 1e6));
  hipFree(out); hipFree(cyc);
}

int main() {
  run_emu<0>("emu: 14 x (MFMA + 2-3 exp + cvt + or3), two waves per SIMD");
  run_emu<1>("emu: + a barrier per step");
  run_emu<2>("emu: + an LDS fragment read behind every MFMA (unused)");
  run_emu<3>("emu: + reads + barrier");
  run_emu<7>("emu: + reads feed next step's MFMAs + barrier");
  run_emu<15>("emu: + two ds_write_b128 at the top");
  run_split<1, 14>("split: waves 0-3 14 MFMA, waves 4-7 idle");
  run_split<2, 14>("split: waves 0-3 idle, waves 4-7 softmax mix");
  run_split<3, 14>("split: waves 0-3 14 MFMA | waves 4-7 softmax mix");
  run_split<7, 14>("split: the same + a barrier per step");
  run_split<1, 8>("split: waves 0-3 8 MFMA, waves 4-7 idle");
  run_split<3, 8>("split: waves 0-3 8 MFMA | waves 4-7 softmax mix");
  run_split<7, 8>("split: 8 MFMA | mix + a barrier per step");
  for (int w = 1; w <= 2; ++w) {
    run<0>("14 MFMA 32x32x16", w);
    run<6>("84 v_add_f32 alone", w);
    run<1>("14 MFMA + 6 v_add_f32 per gap", w);
    run<2>("14 MFMA + 6 v_fma_f32 per gap", w);
    run<4>("softmax mix alone: 32 exp + 16 cvt_pk + 8 or3", w);
    run<3>("14 MFMA + 32 exp + 16 cvt_pk + 8 or3", w);
    run<5>("14 MFMA + 32 exp + 16 cvt_pk + 16 max3", w);
    run<9>("14 MFMA + 32 fma + 32 exp + 16 cvt_pk + 16 max3 (today)", w);
    run<10>("11 MFMA + 32 exp + 16 cvt_pk + 8 or3", w);
    run<7>("28 MFMA 16x16x32", w);
    run<8>("28 MFMA 16x16x32 + 32 exp + 16 cvt_pk + 8 or3", w);
  }
  return 0;
}
