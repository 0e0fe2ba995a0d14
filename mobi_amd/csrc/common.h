// Shared device helpers for the MObI gfx950 engine (CDNA4, wave64).
// Activations are channels-last ("NHWC": [n][h][w][c], c contiguous) in a 16-bit
// storage type T (f16 or bf16); every reduction / accumulation is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mobi_engine.h"

namespace mobi {

typedef _Float16 f16_t;
typedef __bf16 bf16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Vec8;
template <> struct Vec8<f16_t> { typedef f16x8 type; typedef f16x4 half_type; };
template <> struct Vec8<bf16_t> { typedef bf16x8 type; typedef bf16x4 half_type; };

__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

// 8 x T  <->  8 x fp32 (one 16-byte access)
template <typename T>
__device__ __forceinline__ void unpack8(const u32x4& raw, float (&f)[8]) {
  typedef typename Vec8<T>::type V;
  V v = __builtin_bit_cast(V, raw);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T>
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  typedef typename Vec8<T>::type V;
  V v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (T)f[i];
  return __builtin_bit_cast(u32x4, v);
}
template <typename T>
__device__ __forceinline__ u32x2 pack4(const float (&f)[4]) {
  typedef typename Vec8<T>::half_type V;
  V v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (T)f[i];
  return __builtin_bit_cast(u32x2, v);
}

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

// 8 consecutive floats through two 16-byte accesses (LDS stage rows, bias, per-image vectors, fp32 activations)
__device__ __forceinline__ void ld8f(const float* p, float (&f)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// erf-form GELU (F.gelu default, attention.py:45).  erf by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the 16-bit output rounding): one rcp, one exp, five FMAs.
__device__ __forceinline__ float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float r = 1.0f - p * t * e;
  return x < 0.f ? -r : r;
}
#ifdef MOBI_DBG_NOGELU   // timing only (wrong results): what the GELU arithmetic of a GEGLU epilogue costs
__device__ __forceinline__ float gelu_erf_f(float x) { return x; }
#else
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f)); }
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// MFMA wrappers, fp32 accumulate.
__device__ __forceinline__ f32x4 mfma16(const f16x8& a, const f16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// XCD-aware block remap (8 XCDs, blocks dealt round-robin): consecutive logical
// ids run on one XCD so tiles that share an operand panel share that XCD's L2.
// Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7, s = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

}  // namespace mobi

#define MOBI_CHECK_LAUNCH()                                     \
  do {                                                          \
    hipError_t e__ = hipGetLastError();                         \
    if (e__ != hipSuccess) return MOBI_ERR_LAUNCH;              \
  } while (0)
