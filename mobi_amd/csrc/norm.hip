// GroupNorm(32)+SiLU and LayerNorm for gfx950: HBM-bound streaming kernels,
// 16-byte accesses, fp32 statistics (the reference's GroupNorm32 computes in fp32,
// ldm/modules/diffusionmodules/util.py:214-216).
//
// GroupNorm has three forms (launch_gn picks; MOBI_GN_FUSED = 0 / 1 force the second / third for A/Bs):
//   registers (gn_regs_kernel): one launch, a block owns 1 / 2 / 4 adjacent groups of one image and keeps them in
//          registers; 2 B read + 2 B written per element.  Every shape of the denoising step except the 64 x 64 level's
//          640- and 960-channel inputs;
//   two launches: stats: grid (chunks, images); each block sweeps a chunk of pixels with every thread owning
//          a fixed 8-channel column, folds per-channel sums into the 32 groups through LDS in
//          a fixed order (deterministic) and writes one (sum, sumsq) pair per group;
//          apply: every block first combines its image's chunk partials in fp64 (cancellation-safe
//          E[x^2] - mean^2), then streams y = x * a[c] + b[c], optional SiLU.
//          2 B read (stats) + 2 B read + 2 B written (apply) per element; fully coalesced 16-byte accesses;
//   LDS (gn_fused_kernel): one launch, one block per (group, image), the slab in LDS (kept for A/Bs).
#include "common.h"
#include "tuning.h"

namespace mobi {

static inline int gn_chunks(int hw) {
  int c = (hw + 31) / 32;
  return c < 1 ? 1 : (c > 64 ? 64 : c);
}

struct GnArgs {
  const void* src0; const void* src1;
  int c0, c1, C, hw, chunks;
  const float* gamma; const float* beta; float eps; int silu;
  void* out; float* ws;
  int* sync;       // gn_coop_kernel: one arrival counter per image (zero before and after the launch), or NULL
  int src_f32;     // 1: src0 is f32 (one source): the VAE decoder's fp32 streams
  int out_mode;    // 0: T [..][C];  1: T [..][2 C] = hi | lo with hi = T(y), lo = T(y - hi) (a consumer with duplicated weights then
                   //    multiplies y to ~22 bits);  2: f32 [..][C];  3: T [..][3 C] = hi | lo | hi (weights [W ; W ; W - T(W)]: the weights'
                   //    rounding corrected too)
  // src0 as the unfinished split-K partial sums of its producer (mobi_split_source; gn_regs_kernel<.., SLAB = true> only)
  const float* slabs; int splits, slab_row; long long slab_stride;
  const float* s_bias; const float* s_rowvec; int s_rowvec_stride;
  const void* s_resid; long long s_res_img;
  void* finished;
};

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnArgs a) {
  // LDS: per (row slot, channel) partial sums; rows*C <= 2048+ or C <= 2560 when one slot
  __shared__ float s_sum[2560 + 512];
  __shared__ float s_sq[2560 + 512];
  const int tid = threadIdx.x;
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int V = a.C >> 3;                              // 8-channel columns per pixel
  const int rows = V <= 256 ? 256 / V : 1;             // pixels swept per iteration
  const int per = (a.hw + a.chunks - 1) / a.chunks;
  const int p_begin = chunk * per;
  const int p_end = min(a.hw, p_begin + per);
  const T* __restrict__ s0 = reinterpret_cast<const T*>(a.src0) + (long long)img * a.hw * a.c0;
  const T* __restrict__ s1 = a.src1 ? reinterpret_cast<const T*>(a.src1) + (long long)img * a.hw * a.c1 : nullptr;

  const int slot = V <= 256 ? tid / V : 0;
  const int col0 = V <= 256 ? tid - slot * V : tid;
  const bool active = V <= 256 ? slot < rows : true;
  for (int col = col0; col < V; col += 256) {
    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sum[j] = 0.f; sq[j] = 0.f; }
    if (active) {
      const int c = col * 8;
      const bool second = c >= a.c0;
      const T* __restrict__ base = second ? s1 + (c - a.c0) : s0 + c;
      const int cs = second ? a.c1 : a.c0;
      // four independent 16-byte loads in flight per thread (the loop is latency-bound otherwise: one load per
      // thread and iteration keeps only ~32 KB per CU in flight)
      int p = p_begin + slot;
      for (; p + 3 * rows < p_end; p += 4 * rows) {
        u32x4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = ld16(base + (long long)(p + u * rows) * cs);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float f[8];
          unpack8<T>(r[u], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) { sum[j] += f[j]; sq[j] += f[j] * f[j]; }
        }
      }
      for (; p < p_end; p += rows) {
        float f[8];
        unpack8<T>(ld16(base + (long long)p * cs), f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sum[j] += f[j]; sq[j] += f[j] * f[j]; }
      }
      float* ds = s_sum + slot * a.C + c;
      float* dq = s_sq + slot * a.C + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ds[j] = sum[j]; dq[j] = sq[j]; }
    }
    if (V <= 256) break;
  }
  __syncthreads();
  if (tid < 32) {
    const int cpg = a.C / 32;
    float gs = 0.f, gq = 0.f;
    for (int r = 0; r < rows; ++r)
      for (int j = 0; j < cpg; ++j) {
        gs += s_sum[r * a.C + tid * cpg + j];
        gq += s_sq[r * a.C + tid * cpg + j];
      }
    float* w = a.ws + ((long long)(img * a.chunks + chunk) * 32 + tid) * 2;
    w[0] = gs; w[1] = gq;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnArgs a) {
  // per-channel scale / shift of this image, built once per block:  y = x * sc[c] + sh[c]
  __shared__ float s_mean[32], s_rstd[32];
  __shared__ __attribute__((aligned(16))) float s_sc[2560];
  __shared__ __attribute__((aligned(16))) float s_sh[2560];
  const int tid = threadIdx.x;
  const int img = blockIdx.y;
  // chunk partials -> (mean, rstd) of the 32 groups.  All 256 threads take part: thread (sub = tid >> 5, g = tid & 31)
  // sums chunks sub, sub + 8, ... of group g in fp64 (independent loads, one round trip), LDS folds the eight
  // partial sums in a fixed order.  (32 threads walking all chunks serially cost ~8 us of exposed latency per block.)
  __shared__ double s_ps[8][32], s_pq[8][32];
  {
    const int g = tid & 31, sub = tid >> 5;
    double s = 0.0, q = 0.0;
    const float* w = a.ws + ((long long)img * a.chunks * 32 + g) * 2;
    for (int c = sub; c < a.chunks; c += 8) { s += (double)w[c * 64]; q += (double)w[c * 64 + 1]; }
    s_ps[sub][g] = s; s_pq[sub][g] = q;
  }
  __syncthreads();
  if (tid < 32) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += s_ps[k][tid]; q += s_pq[k][tid]; }
    const double n = (double)a.hw * (double)(a.C / 32);
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  const int cpg = a.C / 32;
  for (int c = tid; c < a.C; c += 256) {
    const int g = c / cpg;
    const float sc = s_rstd[g] * a.gamma[c];
    s_sc[c] = sc;
    s_sh[c] = a.beta[c] - s_mean[g] * sc;
  }
  __syncthreads();
  const int V = a.C >> 3;
  const int total = a.hw * V;                                // 16-byte items of one image (host: < 2^31)
  const T* __restrict__ s0 = reinterpret_cast<const T*>(a.src0) + (long long)img * a.hw * a.c0;
  const T* __restrict__ s1 = a.src1 ? reinterpret_cast<const T*>(a.src1) + (long long)img * a.hw * a.c1 : nullptr;
  T* __restrict__ out = reinterpret_cast<T*>(a.out) + (long long)img * a.hw * a.C;
  // item i = pixel * V + column.  A thread walks i, i + S, i + 2 S, ... (S = threads of the grid row); (pixel, column)
  // advance by (S / V, S % V) with one carry instead of a division per item, and two items are in flight.
  const int S = (int)gridDim.x * 256;
  const int dp = S / V, dv = S - dp * V;
  int i = (int)blockIdx.x * 256 + tid;
  int p = i / V, v = i - p * V;
  auto src_of = [&](int pp, int c) -> const T* {
    return c >= a.c0 ? s1 + (long long)pp * a.c1 + (c - a.c0) : s0 + (long long)pp * a.c0 + c;
  };
  auto finish = [&](const u32x4& raw, int pp, int c) {
    float f[8];
    unpack8<T>(raw, f);
    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(s_sc + c), sc1 = *reinterpret_cast<const f32x4*>(s_sc + c + 4);
    const f32x4 sh0 = *reinterpret_cast<const f32x4*>(s_sh + c), sh1 = *reinterpret_cast<const f32x4*>(s_sh + c + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[j] = f[j] * sc0[j] + sh0[j];
      f[4 + j] = f[4 + j] * sc1[j] + sh1[j];
    }
    if (a.silu) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = silu_f(f[j]);
    }
    st16(out + (long long)pp * a.C + c, pack8<T>(f));
  };
  while (i < total) {
    int p2 = p + dp, v2 = v + dv;
    if (v2 >= V) { v2 -= V; ++p2; }
    const bool two = i + S < total;
    const u32x4 r0 = ld16(src_of(p, v * 8));
    u32x4 r1 = r0;
    if (two) r1 = ld16(src_of(p2, v2 * 8));
    finish(r0, p, v * 8);
    if (two) finish(r1, p2, v2 * 8);
    i += 2 * S;
    p = p2 + dp; v = v2 + dv;
    if (v >= V) { v -= V; ++p; }
  }
}

// ---------------------------------------------------------------------------------------
// The two-launch form for fp32 sources and for the "precise" outputs (GnArgs.src_f32 / out_mode): the lidar decoder's tail of
// the fp16 parity configuration (ldm/modules/diffusionmodules/model.py).  Same statistics arithmetic as gn_stats_kernel /
// gn_apply_kernel (per-chunk fp32 partial sums, fp64 fold); not tuned -- a handful of launches per decode.
__global__ __launch_bounds__(256) void gn_stats_f32_kernel(const GnArgs a) {
  __shared__ float s_sum[2560 + 512];
  __shared__ float s_sq[2560 + 512];
  const int tid = threadIdx.x;
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int V = a.C >> 3;
  const int rows = V <= 256 ? 256 / V : 1;
  const int per = (a.hw + a.chunks - 1) / a.chunks;
  const int p_begin = chunk * per;
  const int p_end = min(a.hw, p_begin + per);
  const float* __restrict__ s0 = reinterpret_cast<const float*>(a.src0) + (long long)img * a.hw * a.C;
  const int slot = V <= 256 ? tid / V : 0;
  const int col0 = V <= 256 ? tid - slot * V : tid;
  const bool active = V <= 256 ? slot < rows : true;
  for (int col = col0; col < V; col += 256) {
    float sum[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sum[j] = 0.f; sq[j] = 0.f; }
    if (active) {
      const int c = col * 8;
      for (int p = p_begin + slot; p < p_end; p += rows) {
        float f[8];
        ld8f(s0 + (long long)p * a.C + c, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { sum[j] += f[j]; sq[j] += f[j] * f[j]; }
      }
      float* ds = s_sum + slot * a.C + c;
      float* dq = s_sq + slot * a.C + c;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ds[j] = sum[j]; dq[j] = sq[j]; }
    }
    if (V <= 256) break;
  }
  __syncthreads();
  if (tid < 32) {
    const int cpg = a.C / 32;
    float gs = 0.f, gq = 0.f;
    for (int r = 0; r < rows; ++r)
      for (int j = 0; j < cpg; ++j) {
        gs += s_sum[r * a.C + tid * cpg + j];
        gq += s_sq[r * a.C + tid * cpg + j];
      }
    float* w = a.ws + ((long long)(img * a.chunks + chunk) * 32 + tid) * 2;
    w[0] = gs; w[1] = gq;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_x_kernel(const GnArgs a) {
  __shared__ float s_mean[32], s_rstd[32];
  __shared__ __attribute__((aligned(16))) float s_sc[2560];
  __shared__ __attribute__((aligned(16))) float s_sh[2560];
  __shared__ double s_ps[8][32], s_pq[8][32];
  const int tid = threadIdx.x;
  const int img = blockIdx.y;
  {
    const int g = tid & 31, sub = tid >> 5;
    double s = 0.0, q = 0.0;
    const float* w = a.ws + ((long long)img * a.chunks * 32 + g) * 2;
    for (int c = sub; c < a.chunks; c += 8) { s += (double)w[c * 64]; q += (double)w[c * 64 + 1]; }
    s_ps[sub][g] = s; s_pq[sub][g] = q;
  }
  __syncthreads();
  if (tid < 32) {
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += s_ps[k][tid]; q += s_pq[k][tid]; }
    const double n = (double)a.hw * (double)(a.C / 32);
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  const int cpg = a.C / 32;
  for (int c = tid; c < a.C; c += 256) {
    const int g = c / cpg;
    const float sc = s_rstd[g] * a.gamma[c];
    s_sc[c] = sc;
    s_sh[c] = a.beta[c] - s_mean[g] * sc;
  }
  __syncthreads();
  const int V = a.C >> 3;
  const long long total = (long long)a.hw * V;
  const long long ibase = (long long)img * a.hw;
  for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256) {
    const int p = (int)(i / V), c = (int)(i - (long long)p * V) * 8;
    float f[8];
    if (a.src_f32) {
      ld8f(reinterpret_cast<const float*>(a.src0) + (ibase + p) * a.C + c, f);
    } else {
      const T* src = c >= a.c0 ? reinterpret_cast<const T*>(a.src1) + (ibase + p) * a.c1 + (c - a.c0)
                               : reinterpret_cast<const T*>(a.src0) + (ibase + p) * a.c0 + c;
      unpack8<T>(ld16(src), f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = f[j] * s_sc[c + j] + s_sh[c + j];
      if (a.silu) f[j] = silu_f(f[j]);
    }
    if (a.out_mode == 2) {
      float* o = reinterpret_cast<float*>(a.out) + (ibase + p) * a.C + c;
      *reinterpret_cast<f32x4*>(o) = f32x4{f[0], f[1], f[2], f[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{f[4], f[5], f[6], f[7]};
    } else if (a.out_mode == 1 || a.out_mode == 3) {
      T* o = reinterpret_cast<T*>(a.out) + (ibase + p) * ((a.out_mode == 3 ? 3 : 2) * a.C) + c;
      float hi[8], lo[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { hi[j] = (float)(T)f[j]; lo[j] = f[j] - hi[j]; }
      const u32x4 h8 = pack8<T>(hi);
      st16(o, h8);
      st16(o + a.C, pack8<T>(lo));
      if (a.out_mode == 3) st16(o + 2 * a.C, h8);
    } else {
      st16(reinterpret_cast<T*>(a.out) + (ibase + p) * a.C + c, pack8<T>(f));
    }
  }
}

// ---------------------------------------------------------------------------------------
// One-launch GroupNorm(+SiLU) for SMALL tensors: one block per (group, image) keeps the group's slab (hw x C/32 channels,
// at most GN1_MAX elements of T) in LDS: read once, mean, variance about the mean (two passes over LDS, fixed order),
// y = (x - mean) * rstd * gamma + beta, optional SiLU, written once.  The 16 x 16 / 8 x 8 levels of mobi_nusc_512 and every
// level of mobi_nusc_256 qualify: there the two-launch form above is bound by its launches (7-8 us each), not by bytes.
// Algorithmic bytes: 2 B read + 2 B written per element.
constexpr int GN1_MAX = 32768;
template <typename T>
__global__ __launch_bounds__(256) void gn_fused_kernel(const GnArgs a) {
  __shared__ unsigned slab[GN1_MAX / 2];               // channel pairs (C / 32 is even for every C % 64 == 0 ... see host)
  __shared__ float s_red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x, img = blockIdx.y;
  const int cpg = a.C / 32, hp = cpg >> 1;             // channels / channel pairs of the group
  const int n2 = a.hw * hp;                            // pairs in the slab
  const int c_first = g * cpg;
  const T* __restrict__ s0 = reinterpret_cast<const T*>(a.src0) + (long long)img * a.hw * a.c0;
  const T* __restrict__ s1 = a.src1 ? reinterpret_cast<const T*>(a.src1) + (long long)img * a.hw * a.c1 : nullptr;
  typedef T T2 __attribute__((ext_vector_type(2)));
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();                                   // (also orders this reduction behind the previous one's reads)
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  };
  float sum = 0.f;
  for (int d = tid; d < n2; d += 256) {
    const int p = d / hp, j = d - p * hp;
    const int c = c_first + 2 * j;
    const T* src = c >= a.c0 ? s1 + (long long)p * a.c1 + (c - a.c0) : s0 + (long long)p * a.c0 + c;
    const unsigned raw = *reinterpret_cast<const unsigned*>(src);
    slab[d] = raw;
    const T2 v = __builtin_bit_cast(T2, raw);
    sum += (float)v[0] + (float)v[1];
  }
  const float n = (float)a.hw * (float)cpg;
  const float mean = block_sum(sum) / n;
  float sq = 0.f;
  for (int d = tid; d < n2; d += 256) {
    const T2 v = __builtin_bit_cast(T2, slab[d]);
    const float d0 = (float)v[0] - mean, d1 = (float)v[1] - mean;
    sq += d0 * d0 + d1 * d1;
  }
  const float rstd = rsqrtf(block_sum(sq) / n + a.eps);
  T* __restrict__ out = reinterpret_cast<T*>(a.out) + (long long)img * a.hw * a.C;
  for (int d = tid; d < n2; d += 256) {
    const int p = d / hp, j = d - p * hp;
    const int c = c_first + 2 * j;
    const T2 v = __builtin_bit_cast(T2, slab[d]);
    float y0 = ((float)v[0] - mean) * rstd * a.gamma[c] + a.beta[c];
    float y1 = ((float)v[1] - mean) * rstd * a.gamma[c + 1] + a.beta[c + 1];
    if (a.silu) { y0 = silu_f(y0); y1 = silu_f(y1); }
    T2 o;
    o[0] = (T)y0;
    o[1] = (T)y1;
    *reinterpret_cast<unsigned*>(out + (long long)p * a.C + c) = __builtin_bit_cast(unsigned, o);
  }
}

// ---------------------------------------------------------------------------------------
// One-launch GroupNorm(+SiLU) with the slab in REGISTERS: a block owns GB adjacent groups of one image (GB = 1 / 2 / 4, the
// smallest count whose channels are whole pieces of PW = 8 or 4 channels: a pixel's segment is GB * C / 32 contiguous
// channels) and every thread keeps its pieces -- a FIXED column `col` of the segment, pixels slot, slot + SLOTS, ... -- in
// ITEMS x PW / 2 registers: all loads of the block are in flight at once, the tensor is read once and written once.  A
// column lies in at most two groups (C / 32 >= PW): its first `k` channels belong to group g0, the rest to g0 + 1.
// Statistics: fp32 mean, then the variance about the mean (from registers), both reduced in a fixed order (per-thread
// partials in LDS by group, one wave slice per group, then the slices).  The (group block, image) -> workgroup map hands
// each XCD whole images: the cache lines neighbouring blocks share are fetched from HBM into ONE L2.
// Algorithmic bytes: 2 B read + 2 B written per element.
struct GnRegsGeom { int gb, ppr, slots; };               // groups per block, pieces per pixel segment, pixel slots

// SLAB: the columns of src0 are not loaded but RECONSTRUCTED from the fp32 partial sums of the split-K launch that produces
// src0 -- slabs summed in ascending order from zero, + bias, + per-image vector, + residual, one rounding to T: the arithmetic
// of igemm_splitk_reduce_kernel (csrc/igemm.hip) bit for bit, which this launch replaces; the rounded pieces are what the
// statistics and the output see (the same tensor the two launches would have passed through memory), and they are written to
// `finished` when the tensor has other readers (a ResBlock's output: residual / skip connection of what follows).
template <typename T, int THREADS, int ITEMS, int PW, bool SLAB = false>
__global__ __launch_bounds__(THREADS) void gn_regs_kernel(const GnArgs a, const GnRegsGeom geo) {
  constexpr int NW = THREADS / 64, PR = PW / 2;
  __shared__ float s_part[4][THREADS];                    // per-thread partial of each of the block's groups
  __shared__ float s_wave[NW];
  __shared__ float s_stat[2][4];                          // mean, rstd of the block's groups
  __shared__ __attribute__((aligned(16))) float s_sc[4 * 80], s_sh[4 * 80];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int GB = geo.gb, PPR = geo.ppr, SLOTS = geo.slots;
  const int cpg = a.C / 32, ngb = 32 / GB;
  // workgroup -> (image, group block): XCD x (= id % 8) takes the contiguous range x * n / 8 ... of the work list
  const int nblk = (int)gridDim.x;
  int w = (int)blockIdx.x;
  if (!(nblk & 7)) w = (w & 7) * (nblk >> 3) + (w >> 3);
  const int img = w / ngb, gblk = w - img * ngb;
  const int c_first = gblk * GB * cpg;                    // first channel of the block's segment (a multiple of PW)

  const int slot = tid / PPR, col = tid - slot * PPR;
  const bool live = slot < SLOTS;
  const int cb = col * PW;                                // channel offset inside the segment
  const int g0 = cb / cpg;
  const int k = min(PW, (g0 + 1) * cpg - cb);             // channels of this column that lie in group g0
  const int c = c_first + cb;
  const bool second = c >= a.c0;
  const int cs = second ? a.c1 : a.c0;
  const T* __restrict__ src = second ? reinterpret_cast<const T*>(a.src1) + (long long)img * a.hw * a.c1 + (c - a.c0)
                                     : reinterpret_cast<const T*>(a.src0) + (long long)img * a.hw * a.c0 + c;
  typedef unsigned piece_t __attribute__((ext_vector_type(PR)));
  typedef T T2 __attribute__((ext_vector_type(2)));
  unsigned raw[ITEMS][PR];
  if (SLAB && !second) {
    float bia[PW], rvv[PW];                               // bias, per-image vector of this column (added one after the other, as the reduce kernel does)
#pragma unroll
    for (int j = 0; j < PW; j += 4) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      const f32x4 b4 = a.s_bias ? *reinterpret_cast<const f32x4*>(a.s_bias + c + j) : z;
      const f32x4 r4 = a.s_rowvec ? *reinterpret_cast<const f32x4*>(a.s_rowvec + (long long)img * a.s_rowvec_stride + c + j) : z;
#pragma unroll
      for (int e = 0; e < 4; ++e) { bia[j + e] = b4[e]; rvv[j + e] = r4[e]; }
    }
    const T* __restrict__ resid = reinterpret_cast<const T*>(a.s_resid);
    T* __restrict__ fin = reinterpret_cast<T*>(a.finished);
    const long long row0 = (long long)img * a.hw;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int p = slot + i * SLOTS;
      piece_t r = piece_t(0u);
      if (live && p < a.hw) {
        float o[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) o[j] = 0.f;
        const float* q = a.slabs + (row0 + p) * a.slab_row + c;
        int sp = 0;
        // (more loads in flight -- a batch of items x slabs requested before anything is added, 16-24 sixteen-byte loads per
        //  thread -- did NOT help: 15.5 against 14.3 us on [8, 256, 640] x 4 slabs, 23.8 against 22.2 on [16, 256, 1280] x 4: the
        //  launch is bound by the slabs' bytes (84 MB in the 13 us it takes beyond the plain GroupNorm), and the batches' registers
        //  cost occupancy; profiles/r05_defer_breakdown.txt)
        for (; sp + 4 <= a.splits; sp += 4) {             // four slabs requested at once, summed in ascending order
          f32x4 v[4][PW / 4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < PW / 4; ++j) v[u][j] = *reinterpret_cast<const f32x4*>(q + (sp + u) * a.slab_stride + 4 * j);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < PW; ++j) o[j] += v[u][j >> 2][j & 3];
        }
        for (; sp < a.splits; ++sp) {
#pragma unroll
          for (int j = 0; j < PW / 4; ++j) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(q + sp * a.slab_stride + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * j + e] += v[e];
          }
        }
        if (a.s_bias) {
#pragma unroll
          for (int j = 0; j < PW; ++j) o[j] += bia[j];
        }
        if (a.s_rowvec) {
#pragma unroll
          for (int j = 0; j < PW; ++j) o[j] += rvv[j];
        }
        if (resid) {
          // (the words go through a plain array: `__builtin_bit_cast(T2, rr[j])` straight on the elements of a `const piece_t`
          //  compiled -- hipcc of ROCm 7.2 -- to ONE 4-byte load of element 0 used for every j: 6 of 8 channels without residual)
          piece_t rr = *reinterpret_cast<const piece_t*>(resid + (long long)img * a.s_res_img + (long long)p * a.c0 + c);
          unsigned rw[PR];
#pragma unroll
          for (int j = 0; j < PR; ++j) rw[j] = rr[j];
#pragma unroll
          for (int j = 0; j < PR; ++j) {
            const T2 v = __builtin_bit_cast(T2, rw[j]);
            o[2 * j] += (float)v[0];
            o[2 * j + 1] += (float)v[1];
          }
        }
#pragma unroll
        for (int j = 0; j < PR; ++j) {
          T2 v;
          v[0] = (T)o[2 * j];
          v[1] = (T)o[2 * j + 1];
          r[j] = __builtin_bit_cast(unsigned, v);
        }
        if (fin) *reinterpret_cast<piece_t*>(fin + (row0 + p) * a.c0 + c) = r;
      }
#pragma unroll
      for (int j = 0; j < PR; ++j) raw[i][j] = r[j];
    }
  } else {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      const int p = slot + i * SLOTS;
      piece_t r = piece_t(0u);
      if (live && p < a.hw) r = *reinterpret_cast<const piece_t*>(src + (long long)p * cs);
#pragma unroll
      for (int j = 0; j < PR; ++j) raw[i][j] = r[j];
    }
  }
  auto unpack = [&](const unsigned (&r)[PR], float (&f)[PW]) {
#pragma unroll
    for (int j = 0; j < PR; ++j) {
      const T2 v = __builtin_bit_cast(T2, r[j]);
      f[2 * j] = (float)v[0];
      f[2 * j + 1] = (float)v[1];
    }
  };
  // (the packed pieces are what stays in registers between the passes: without the pins the compiler keeps the unpacked
  // floats of the first pass alive, twice the registers)
  auto pin = [&]() {
#pragma unroll
    for (int i = 0; i < ITEMS; ++i)
#pragma unroll
      for (int j = 0; j < PR; ++j) asm volatile("" : "+v"(raw[i][j]));
  };
  // sum over the block's threads of (lo -> group g0, hi -> group g0 + 1), fixed order; result for every group in s_stat[which]
  auto reduce = [&](float lo, float hi, int which, float scale, bool to_rstd) {
    __syncthreads();
    for (int g = 0; g < GB; ++g) s_part[g][tid] = g == g0 ? lo : (g == g0 + 1 ? hi : 0.f);
    __syncthreads();
    {
      const int g = wave % GB, part = wave / GB, parts = NW / GB, len = THREADS / parts;
      float v = 0.f;
      for (int t = part * len + lane; t < (part + 1) * len; t += 64) v += s_part[g][t];
      v = wave_sum(v);
      if (lane == 0) s_wave[wave] = v;
    }
    __syncthreads();
    if (tid < GB) {
      float v = 0.f;
      for (int part = 0; part < NW / GB; ++part) v += s_wave[part * GB + tid];
      v *= scale;
      s_stat[which][tid] = to_rstd ? rsqrtf(v + a.eps) : v;
    }
    __syncthreads();
  };
  const float inv_n = 1.f / ((float)a.hw * (float)cpg);
  // per-channel accumulators over the thread's pixels (one add / one fma per element), split into the two groups once
  auto split = [&](const float (&acc)[PW], float& lo, float& hi) {
    lo = 0.f; hi = 0.f;
#pragma unroll
    for (int j = 0; j < PW; ++j) { lo += j < k ? acc[j] : 0.f; hi += j < k ? 0.f : acc[j]; }
  };
  {
    float acc[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) acc[j] = 0.f;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      float f[PW];
      unpack(raw[i], f);
#pragma unroll
      for (int j = 0; j < PW; ++j) acc[j] += f[j];
    }
    float lo, hi;
    split(acc, lo, hi);
    reduce(lo, hi, 0, inv_n, false);
  }
  pin();
  {
    const float m_lo = s_stat[0][g0], m_hi = s_stat[0][min(g0 + 1, GB - 1)];
    float acc[PW], m[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) { acc[j] = 0.f; m[j] = j < k ? m_lo : m_hi; }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (live && slot + i * SLOTS < a.hw) {
        float f[PW];
        unpack(raw[i], f);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
          const float d = f[j] - m[j];
          acc[j] += d * d;
        }
      }
    }
    float lo, hi;
    split(acc, lo, hi);
    reduce(lo, hi, 1, inv_n, true);
  }
  pin();
  // y = x * sc[c] + sh[c]
  for (int cc = tid; cc < GB * cpg; cc += THREADS) {
    const int g = cc / cpg;
    const float sc = s_stat[1][g] * a.gamma[c_first + cc];
    s_sc[cc] = sc;
    s_sh[cc] = a.beta[c_first + cc] - s_stat[0][g] * sc;
  }
  __syncthreads();
  if (!live) return;
  T* __restrict__ out = reinterpret_cast<T*>(a.out) + (long long)img * a.hw * a.C + c;
  float sc[PW], sh[PW];
#pragma unroll
  for (int j = 0; j < PW; j += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(s_sc + cb + j), u = *reinterpret_cast<const f32x4*>(s_sh + cb + j);
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[j + e] = v[e]; sh[j + e] = u[e]; }
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int p = slot + i * SLOTS;
    if (p < a.hw) {
      float f[PW];
      unpack(raw[i], f);
#pragma unroll
      for (int j = 0; j < PW; ++j) {
        f[j] = f[j] * sc[j] + sh[j];
        if (a.silu) f[j] = silu_f(f[j]);
      }
      piece_t o;
#pragma unroll
      for (int j = 0; j < PR; ++j) {
        T2 v;
        v[0] = (T)f[2 * j];
        v[1] = (T)f[2 * j + 1];
        o[j] = __builtin_bit_cast(unsigned, v);
      }
      *reinterpret_cast<piece_t*>(out + (long long)p * a.C) = o;
    }
  }
}

// geometry of the register form for (C, hw): 16-byte pieces when a thread then holds at most 16 of them, else 8-byte pieces
// (half the groups per block, up to 24 per thread); false when the tensor does not qualify (odd C / 32, or too many
// pieces per thread: the 64 x 64 level's 960-channel input)
static bool gn_regs_geometry(int C, int hw, int batch, GnRegsGeom* geo, int* threads, int* items, int* pw_out) {
  const int cpg = C / 32;
  if (cpg < 8 || (cpg & 1)) return false;
  static const int steps8[] = {1, 2, 3, 4, 6, 8, 12, 16};
  static const int steps4[] = {8, 12, 16, 24};
  bool found = false;
  for (int pw = 8; pw >= 4; pw >>= 1) {
    int gb = 1;
    while (gb <= 4 && ((gb * cpg) & (pw - 1))) gb <<= 1;
    if (gb > 4) continue;
    if (pw == 8) while (gb * cpg < 32 && gb < 4) gb <<= 1;       // segments of at least 64 bytes
    const int ppr = gb * cpg / pw;
    if (gb * cpg > 4 * 80) continue;
    const long long pieces = (long long)hw * ppr;
    const int th = pieces >= 2048 ? 1024 : 256;
    if (pw == 4 && th != 1024) continue;
    const int slots = th / ppr;
    if (slots < 1) continue;
    const int it = (hw + slots - 1) / slots;
    int pick = 0;
    if (pw == 8) { for (int s : steps8) if (s >= it) { pick = s; break; } }
    else         { for (int s : steps4) if (s >= it) { pick = s; break; } }
    if (!pick) continue;
    if (th == 1024 && pick >= 12 && 32 / gb * batch > 256) continue;   // one such block per CU: a second round of blocks loses
    // 16-byte pieces unless the 8-byte form (half the groups per block) is what gives every CU a block
    if (found && !(32 / geo->gb * batch < 256 && 32 / gb * batch > 32 / geo->gb * batch)) continue;
    geo->gb = gb; geo->ppr = ppr; geo->slots = slots;
    *threads = th; *items = pick; *pw_out = pw;
    found = true;
  }
  return found;
}

// src0 from split-K slabs: the register form with at most 8 pieces per thread (split launches have few pixels: the 32 x 32
// level's 320-channel tensors are the largest -- 8 pieces of 8 bytes), both sources split at a piece boundary
static bool gn_regs_takes_split(int c0, int c1, int hw, int batch) {
  GnRegsGeom geo;
  int th, it, pw;
  if (!gn_regs_geometry(c0 + c1, hw, batch, &geo, &th, &it, &pw)) return false;
  // 8-byte pieces (C = 320: the 32 x 32 level of mobi_nusc_256) only on request: in the step graph 21.0 us against 9.8 + 6.5 for
  // GroupNorm + reduce launch on [8, 1024, 320] x 4 slabs (tools/defer_breakdown.sh) -- 128 blocks read 42 MB of slabs
  if (pw == 4 && tuning().gn_split_pw4 != 1) return false;
  return it <= 8 && c0 % pw == 0;
}

template <typename T>
static bool launch_gn_regs_split(const GnArgs& a, int batch, hipStream_t st) {
  GnRegsGeom geo;
  int th, it, pw;
  if (!gn_regs_geometry(a.C, a.hw, batch, &geo, &th, &it, &pw) || it > 8 || a.c0 % pw) return false;
  if (pw == 4 && tuning().gn_split_pw4 != 1) return false;
  const dim3 grid((unsigned)(32 / geo.gb * batch));
#define MOBI_GNS(TH_, IT_, PW_) hipLaunchKernelGGL((gn_regs_kernel<T, TH_, IT_, PW_, true>), grid, dim3(TH_), 0, st, a, geo)
  if (pw == 4) MOBI_GNS(1024, 8, 4);
  else if (th == 256) {
    switch (it) {
      case 1: MOBI_GNS(256, 1, 8); break;   case 2: MOBI_GNS(256, 2, 8); break;   case 3: MOBI_GNS(256, 3, 8); break;
      case 4: MOBI_GNS(256, 4, 8); break;   case 6: MOBI_GNS(256, 6, 8); break;   default: MOBI_GNS(256, 8, 8); break;
    }
  } else {
    switch (it) {
      case 1: MOBI_GNS(1024, 1, 8); break;   case 2: MOBI_GNS(1024, 2, 8); break;   case 3: MOBI_GNS(1024, 3, 8); break;
      case 4: MOBI_GNS(1024, 4, 8); break;   case 6: MOBI_GNS(1024, 6, 8); break;   default: MOBI_GNS(1024, 8, 8); break;
    }
  }
#undef MOBI_GNS
  return true;
}

template <typename T>
static bool launch_gn_regs(const GnArgs& a, int batch, hipStream_t st) {
  GnRegsGeom geo;
  int th, it, pw;
  if (!gn_regs_geometry(a.C, a.hw, batch, &geo, &th, &it, &pw)) return false;
  const dim3 grid((unsigned)(32 / geo.gb * batch));
#define MOBI_GNR(TH_, IT_, PW_) hipLaunchKernelGGL((gn_regs_kernel<T, TH_, IT_, PW_>), grid, dim3(TH_), 0, st, a, geo)
  if (pw == 4) {
    switch (it) {
      case 8: MOBI_GNR(1024, 8, 4); break;   case 12: MOBI_GNR(1024, 12, 4); break;
      case 16: MOBI_GNR(1024, 16, 4); break; default: MOBI_GNR(1024, 24, 4); break;
    }
  } else if (th == 256) {
    switch (it) {
      case 1: MOBI_GNR(256, 1, 8); break;   case 2: MOBI_GNR(256, 2, 8); break;   case 3: MOBI_GNR(256, 3, 8); break;
      case 4: MOBI_GNR(256, 4, 8); break;   case 6: MOBI_GNR(256, 6, 8); break;   case 8: MOBI_GNR(256, 8, 8); break;
      case 12: MOBI_GNR(256, 12, 8); break; default: MOBI_GNR(256, 16, 8); break;
    }
  } else {
    switch (it) {
      case 1: MOBI_GNR(1024, 1, 8); break;   case 2: MOBI_GNR(1024, 2, 8); break;   case 3: MOBI_GNR(1024, 3, 8); break;
      case 4: MOBI_GNR(1024, 4, 8); break;   case 6: MOBI_GNR(1024, 6, 8); break;   case 8: MOBI_GNR(1024, 8, 8); break;
      case 12: MOBI_GNR(1024, 12, 8); break; default: MOBI_GNR(1024, 16, 8); break;
    }
  }
#undef MOBI_GNR
  return true;
}

// ---------------------------------------------------------------------------------------
struct LnArgs {
  const void* src; void* out;
  int images, rows, C; long long src_img, out_img;
  const float* gamma; const float* beta; float eps;
};

// LPR lanes per token row (64 / LPR rows per wave, four waves per block); a lane holds up to MAXV 8-channel pieces of its row
// (pieces lane_in_row + LPR * i).  With C = 320 / 640 / 1280 that is 8 / 16 / 32 lanes and 5 pieces: every lane has five
// 16-byte loads in flight and a wave covers 5 KB of rows per pass (one row per wave left 24 of 64 lanes idle at C = 320 and
// a single load per lane in flight).  The sums are reduced over the row's lanes by a butterfly of log2(LPR) steps.
template <typename T, int MAXV, int LPR>
__global__ __launch_bounds__(256) void layernorm_kernel(const LnArgs a) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, lr = lane & (LPR - 1);
  const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + (lane / LPR);
  const long long total = (long long)a.images * a.rows;
  const bool live = row < total;
  const long long rr = live ? row : total - 1;               // idle lanes shadow the last row (no store): uniform shuffles
  const int img = (int)(rr / a.rows);
  const int r = (int)(rr - (long long)img * a.rows);
  const T* __restrict__ src = reinterpret_cast<const T*>(a.src) + img * a.src_img + (long long)r * a.C;
  T* __restrict__ out = reinterpret_cast<T*>(a.out) + img * a.out_img + (long long)r * a.C;
  const int V = a.C >> 3;
  float f[MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lr + LPR * i;
    if (v < V) {
      unpack8<T>(ld16(src + v * 8), f[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += f[i][j];
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float mean = s / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lr + LPR * i;
    if (v < V) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; q += d * d; }
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  const float rstd = rsqrtf(q / (float)a.C + a.eps);
  if (!live) return;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = lr + LPR * i;
    if (v < V) {
      float g[8], bt[8], o[8];
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.gamma + v * 8), g1 = *reinterpret_cast<const f32x4*>(a.gamma + v * 8 + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.beta + v * 8), b1 = *reinterpret_cast<const f32x4*>(a.beta + v * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { g[j] = g0[j]; g[4 + j] = g1[j]; bt[j] = b0[j]; bt[4 + j] = b1[j]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (f[i][j] - mean) * rstd * g[j] + bt[j];
      st16(out + v * 8, pack8<T>(o));
    }
  }
}

// ---------------------------------------------------------------------------------------
// One-launch GroupNorm(+SiLU) for tensors whose groups are too large for one block (the 64 x 64 level's 320 channels: a group
// is 80 KB per image and its 20-byte pixel segments lie 640 bytes apart -- gn_regs_kernel reads them as 8-byte pieces, 40 useful
// bytes per 128-byte line and request).  Here a block owns a CHUNK of pixels x ALL channels of one image in registers: every
// access is a fully coalesced 16-byte piece of a dense row, the tensor is read once and written once.  The chunks of an image
// meet through memory, not through a second launch:
//   every block writes the (sum, sum of squares) of its chunk for the 32 groups, ARRIVES at the image's counter and waits
//   (bounded spin, one lane) until all chunks of the image have; then it folds the image's partials in fp64 in chunk order -- the
//   two-launch form's arithmetic -- and applies scale / shift (+ SiLU) to the pieces it still holds.
// The grid is at most one block per CU (the host checks), so every block of an image is resident while its siblings spin.
// Coherence as in the in-launch split-K finish (csrc/igemm.hip): the partials are stored and loaded device-coherently (sc1), the
// counter is a device-scope atomic, nothing else changes its cache policy; a block's SECOND arrival (after its fold) lets the last
// one return the counter to zero, so one zeroed buffer serves every launch of a stream.
// Algorithmic bytes: 2 B read + 2 B written per element.
struct GnCoopGeom { int chunks, ppb, slots, V; };

__device__ __forceinline__ void st8_dev(float* p, float x, float y) {
  const u32x2 v = {__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, y)};
  asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

template <typename T, int ITEMS>
__global__ __launch_bounds__(1024) void gn_coop_kernel(const GnArgs a, const GnCoopGeom geo) {
  __shared__ float s_part[4][1024];                      // per thread: (sum, squares) of its column's first- / second-group channels
  __shared__ float s_col[4][320];                        // per column, over the block's pixel slots
  __shared__ double s_ps[32][32], s_pq[32][32];          // [chunk][group] of the image
  __shared__ float s_mean[32], s_rstd[32];
  __shared__ int s_seen;
  const int tid = threadIdx.x;
  const int V = geo.V, SLOTS = geo.slots;
  const int cpg = a.C / 32;
  const int nblk = (int)gridDim.x;
  int w = (int)blockIdx.x;
  if (!(nblk & 7)) w = (w & 7) * (nblk >> 3) + (w >> 3);  // an XCD's blocks: whole images
  const int img = w / geo.chunks, chunk = w - img * geo.chunks;
  const int slot = tid / V, col = tid - slot * V;
  const bool live = slot < SLOTS;
  const int c = col * 8;
  const int g0 = c / cpg;
  const int k = min(8, (g0 + 1) * cpg - c);              // channels of the column in group g0 (the rest in g0 + 1)
  const bool second = c >= a.c0;
  const int cs = second ? a.c1 : a.c0;
  const T* __restrict__ src = second ? reinterpret_cast<const T*>(a.src1) + (long long)img * a.hw * a.c1 + (c - a.c0)
                                     : reinterpret_cast<const T*>(a.src0) + (long long)img * a.hw * a.c0 + c;
  const int p0 = chunk * geo.ppb;
  u32x4 raw[ITEMS];
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int q = slot + i * SLOTS;
    raw[i] = u32x4{0u, 0u, 0u, 0u};
    if (live && q < geo.ppb && p0 + q < a.hw) raw[i] = ld16(src + (long long)(p0 + q) * cs);
  }
  {
    float sm[8], sq[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sm[j] = 0.f; sq[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {                    // (pieces past the chunk are zeros: they add nothing)
      float f[8];
      unpack8<T>(raw[i], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { sm[j] += f[j]; sq[j] += f[j] * f[j]; }
    }
    float lo_s = 0.f, hi_s = 0.f, lo_q = 0.f, hi_q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      lo_s += j < k ? sm[j] : 0.f; hi_s += j < k ? 0.f : sm[j];
      lo_q += j < k ? sq[j] : 0.f; hi_q += j < k ? 0.f : sq[j];
    }
    s_part[0][tid] = live ? lo_s : 0.f; s_part[1][tid] = live ? hi_s : 0.f;
    s_part[2][tid] = live ? lo_q : 0.f; s_part[3][tid] = live ? hi_q : 0.f;
  }
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) asm volatile("" : "+v"(raw[i]));      // the packed pieces are what stays in registers
  __syncthreads();
  for (int t = tid; t < V * 4; t += 1024) {              // per column, over the pixel slots in slot order
    const int cc = t >> 2, q = t & 3;
    float v = 0.f;
    for (int s_ = 0; s_ < SLOTS; ++s_) v += s_part[q][s_ * V + cc];
    s_col[q][cc] = v;
  }
  __syncthreads();
  float* wsp = a.ws + ((long long)(img * geo.chunks + chunk) * 32) * 2;
  if (tid < 32) {                                         // group tid: the columns that overlap it, in column order
    const int g = tid;
    const int c_lo = max(0, (g * cpg) / 8 - 1), c_hi = min(V - 1, ((g + 1) * cpg - 1) / 8);
    float gs = 0.f, gq = 0.f;
    for (int cc = c_lo; cc <= c_hi; ++cc) {
      const int gg = (cc * 8) / cpg;
      if (gg == g) { gs += s_col[0][cc]; gq += s_col[2][cc]; }
      else if (gg + 1 == g) { gs += s_col[1][cc]; gq += s_col[3][cc]; }
    }
    st8_dev(wsp + g * 2, gs, gq);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // acknowledged before this block arrives
  }
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_fetch_add(a.sync + img, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int seen = 0;
    for (int spin = 0; spin < (1 << 22); ++spin) {       // bounded: a lost sibling ends in a wrong image, never in a hung chip
      seen = __hip_atomic_load(a.sync + img, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (seen >= geo.chunks) break;
      __builtin_amdgcn_s_sleep(4);
    }
    s_seen = seen;
  }
  __syncthreads();
  {
    const int g = tid & 31, ch = tid >> 5;                // 32 x 32 threads: one (chunk, group) pair each
    double s = 0.0, q = 0.0;
    if (ch < geo.chunks) {
      // (one descriptor over the whole workspace -- a descriptor is wave-uniform --, the pair's byte offset per lane; sc1)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, nblk * 256, 0x00020000);
      // (two dword loads: the b64 form returned the dwords at +0 and +8 here -- seen as the next group's sum in place of this
      //  group's squares)
      const unsigned off = (unsigned)(((img * geo.chunks + ch) * 32 + g) * 8);
      s = (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 16));
      q = (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4u, 0, 16));
    }
    s_ps[ch][g] = s; s_pq[ch][g] = q;
  }
  __syncthreads();
  if (tid == 0) {                                         // second arrival: the last block past its loads returns the counter to zero
    const int old = __hip_atomic_fetch_add(a.sync + img, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 2 * geo.chunks - 1) __hip_atomic_store(a.sync + img, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < 32) {
    double s = 0.0, q = 0.0;
    for (int ch = 0; ch < geo.chunks; ++ch) { s += s_ps[ch][tid]; q += s_pq[ch][tid]; }
    const double n = (double)a.hw * (double)cpg;
    const double mean = s / n;
    double var = q / n - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    s_mean[tid] = (float)mean;
    s_rstd[tid] = (float)(1.0 / sqrt(var + (double)a.eps));
  }
  __syncthreads();
  if (!live) return;
  float sc[8], sh[8];
  {
    float gm[8], bt[8];
    ld8f(a.gamma + c, gm);
    ld8f(a.beta + c, bt);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = j < k ? g0 : g0 + 1;
      sc[j] = s_rstd[g] * gm[j];
      sh[j] = bt[j] - s_mean[g] * sc[j];
    }
  }
  T* __restrict__ out = reinterpret_cast<T*>(a.out) + (long long)img * a.hw * a.C + c;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int q = slot + i * SLOTS;
    if (q < geo.ppb && p0 + q < a.hw) {
      float f[8];
      unpack8<T>(raw[i], f);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f[j] = f[j] * sc[j] + sh[j];
        if (a.silu) f[j] = silu_f(f[j]);
      }
      st16(out + (long long)(p0 + q) * a.C, pack8<T>(f));
    }
  }
}

// chunks per image (a power of two, at most 32), pixels per block, pixel slots; false: the shape is not the kernel's
static bool gn_coop_geometry(int C, int hw, int batch, int cus, GnCoopGeom* geo, int* items) {
  const int cpg = C / 32, V = C / 8;
  if (cpg < 8 || V > 320 || V < 1) return false;
  const int slots = 1024 / V;
  int chunks = 32;
  while (chunks > 1 && ((long long)batch * chunks > cus || hw % chunks)) chunks >>= 1;
  if (chunks < 2) return false;
  const int ppb = hw / chunks;
  const int need = (ppb + slots - 1) / slots;
  static const int steps[] = {2, 4, 6, 8, 12, 16};
  int pick = 0;
  for (int s_ : steps) if (s_ >= need) { pick = s_; break; }
  if (!pick) return false;
  geo->chunks = chunks; geo->ppb = ppb; geo->slots = slots; geo->V = V;
  *items = pick;
  return true;
}

static int gn_compute_units() {
  static int cached = 0;
  if (!cached) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached = n;
  }
  return cached;
}

// where the chunked kernel replaces the register kernel by default: NOWHERE.  Measured (tools/gn_lab.py, profiles/r05_gn_lab.txt, 16
// images, graph-timed on cold inputs): [4096, 320] 33.0 against 30.2 us (registers), [1024, 640] 24.1 against 17.6, [256, 1280]
// 18.7 against 10.6 -- the fully coalesced accesses buy nothing: both kernels are ONE-SHOT (every block loads its whole share,
// reduces, then stores, all 256 blocks in the same phase), so the launch costs the read phase PLUS the write phase (~2.7 TB/s of
// the 6 a copy reaches by overlapping them), and the meeting through memory adds ~8 us.  Kept for the A/B (MOBI_GN_COOP=1).
static bool gn_coop_default(int C, int hw, int batch) {
  (void)C; (void)hw; (void)batch;
  return false;
}

template <typename T>
static bool launch_gn_coop(const GnArgs& a, int batch, hipStream_t st) {
  GnCoopGeom geo;
  int it;
  if (!a.sync || !gn_coop_geometry(a.C, a.hw, batch, gn_compute_units(), &geo, &it)) return false;
  const dim3 grid((unsigned)(batch * geo.chunks));
#define MOBI_GNC(IT_) hipLaunchKernelGGL((gn_coop_kernel<T, IT_>), grid, dim3(1024), 0, st, a, geo)
  switch (it) {
    case 2: MOBI_GNC(2); break;   case 4: MOBI_GNC(4); break;   case 6: MOBI_GNC(6); break;
    case 8: MOBI_GNC(8); break;   case 12: MOBI_GNC(12); break; default: MOBI_GNC(16); break;
  }
#undef MOBI_GNC
  return true;
}

template <typename T>
static int launch_gn(const GnArgs& a, int batch, hipStream_t st) {
  // small tensors: one launch, the group's slab in LDS (channel pairs: C / 32 even; both sources split at an even channel)
  const int cpg = a.C / 32;
  if (a.slabs) {                                        // src0 from its producer's split-K slabs: the register form or nothing
    if (!launch_gn_regs_split<T>(a, batch, st)) return MOBI_ERR_UNSUPPORTED;
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  if (a.src_f32 || a.out_mode) {                        // fp32 source / precise outputs: the two-launch form of their own
    if (a.src_f32) hipLaunchKernelGGL(gn_stats_f32_kernel, dim3(a.chunks, batch), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(a.chunks, batch), dim3(256), 0, st, a);
    MOBI_CHECK_LAUNCH();
    long long blocks = ((long long)a.hw * (a.C >> 3) + 256 * 4 - 1) / (256 * 4);
    const long long cap = (2048 + batch - 1) / batch;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((gn_apply_x_kernel<T>), dim3((unsigned)blocks, batch), dim3(256), 0, st, a);
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  const int mode = tuning().gn_fused;                   // 0: two launches; 1: the LDS form where it fits; else registers first
  // pixel chunks meeting through memory (gn_coop_kernel): MOBI_GN_COOP = 1 wherever its geometry fits, 0 never; default: the shapes
  // it is measured faster on (tools/gn_lab.py)
  {
    const int coop = tuning().gn_coop;
    const bool want = coop == 1 || (coop != 0 && gn_coop_default(a.C, a.hw, batch));
    if (want && mode != 0 && mode != 1 && launch_gn_coop<T>(a, batch, st)) {
      MOBI_CHECK_LAUNCH();
      return MOBI_OK;
    }
  }
  if (mode != 0 && mode != 1 && launch_gn_regs<T>(a, batch, st)) {
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  if (!(cpg & 1) && (long long)a.hw * cpg <= GN1_MAX && mode != 0) {
    hipLaunchKernelGGL((gn_fused_kernel<T>), dim3(32, batch), dim3(256), 0, st, a);
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  hipLaunchKernelGGL((gn_stats_kernel<T>), dim3(a.chunks, batch), dim3(256), 0, st, a);
  MOBI_CHECK_LAUNCH();
  const long long vecs = (long long)a.hw * (a.C >> 3);
  long long blocks = (vecs + 256 * 4 - 1) / (256 * 4);
  const long long cap = (2048 + batch - 1) / batch;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((gn_apply_kernel<T>), dim3((unsigned)blocks, batch), dim3(256), 0, st, a);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

template <typename T>
static int launch_ln(const LnArgs& a, hipStream_t st) {
  const long long total = (long long)a.images * a.rows;
  const int V = a.C >> 3;
#define MOBI_LN(MAXV_, LPR_)                                                                                           \
  hipLaunchKernelGGL((layernorm_kernel<T, MAXV_, LPR_>), dim3((unsigned)((total + 4 * (64 / LPR_) - 1) / (4 * (64 / LPR_)))), \
                     dim3(256), 0, st, a)
  if (V <= 8) MOBI_LN(1, 8);
  else if (V <= 40) MOBI_LN(5, 8);                            // C <= 320
  else if (V <= 80) MOBI_LN(5, 16);                           // C <= 640
  else if (V <= 160) MOBI_LN(5, 32);                          // C <= 1280
  else if (V <= 320) MOBI_LN(5, 64);                          // C <= 2560
  else return MOBI_ERR_UNSUPPORTED;
#undef MOBI_LN
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

}  // namespace mobi

extern "C" size_t mobi_groupnorm_workspace_bytes(int32_t batch, int32_t hw) {
  if (batch <= 0 || hw <= 0) return 0;
  const int chunks = mobi::gn_chunks(hw) > 32 ? mobi::gn_chunks(hw) : 32;     // (gn_coop_kernel: up to 32 chunks per image)
  return (size_t)batch * chunks * 32 * 2 * sizeof(float);
}

extern "C" int mobi_groupnorm_takes_split(int32_t c0, int32_t c1, int32_t batch, int32_t hw) {
  if (c0 <= 0 || (c0 & 31) || c1 < 0 || (c1 & 31) || batch <= 0 || hw <= 0 || c0 + c1 > 2560) return 0;
  if (mobi::tuning().gn_fused == 0 || mobi::tuning().gn_fused == 1) return 0;      // a forced other form (A/B): no register form
  return mobi::gn_regs_takes_split(c0, c1, hw, batch) ? 1 : 0;
}

extern "C" int mobi_groupnorm(const mobi_groupnorm_params* p, void* stream) {
  using namespace mobi;
  if (!p || (!p->src0 && !p->src0_split) || !p->out || !p->ws || !p->gamma || !p->beta) return MOBI_ERR_ARG;
  if (p->dtype != MOBI_F16 && p->dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (p->c0 <= 0 || (p->c0 & 31) || p->c1 < 0 || (p->c1 & 31) || (p->c1 > 0 && !p->src1)) return MOBI_ERR_UNSUPPORTED;
  if (p->batch <= 0 || p->hw <= 0 || p->batch > 65535) return MOBI_ERR_ARG;
  const int C = p->c0 + p->c1;
  if (C > 2560 || (long long)p->hw * (C >> 3) >= 0x7fffffffLL) return MOBI_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p->src0) | reinterpret_cast<uintptr_t>(p->src1) |
       reinterpret_cast<uintptr_t>(p->out)) & 15) return MOBI_ERR_ALIGN;
  GnArgs a;
  a.src0 = p->src0; a.src1 = p->c1 ? p->src1 : nullptr; a.c0 = p->c0; a.c1 = p->c1; a.C = C;
  a.hw = p->hw; a.chunks = gn_chunks(p->hw);
  a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps; a.silu = p->silu;
  a.out = p->out; a.ws = reinterpret_cast<float*>(p->ws);
  if (p->src_f32 != 0 && p->src_f32 != 1) return MOBI_ERR_ARG;
  if (p->out_mode < 0 || p->out_mode > 3) return MOBI_ERR_ARG;
  if (p->src_f32 && p->c1) return MOBI_ERR_UNSUPPORTED;
  a.src_f32 = p->src_f32; a.out_mode = p->out_mode;
  if (reinterpret_cast<uintptr_t>(p->sync) & 3) return MOBI_ERR_ALIGN;
  a.sync = reinterpret_cast<int*>(p->sync);
  a.slabs = nullptr; a.splits = 0; a.slab_row = 0; a.slab_stride = 0;
  a.s_bias = nullptr; a.s_rowvec = nullptr; a.s_rowvec_stride = 0; a.s_resid = nullptr; a.s_res_img = 0; a.finished = nullptr;
  if (const mobi_split_source* ss = p->src0_split) {
    if (!ss->slabs || ss->count < 2 || ss->count > 64 || ss->row_stride < p->c0 || (ss->row_stride & 3)) return MOBI_ERR_ARG;
    if (p->src_f32 || p->out_mode) return MOBI_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(ss->slabs) | reinterpret_cast<uintptr_t>(ss->bias) | reinterpret_cast<uintptr_t>(ss->rowvec) |
         reinterpret_cast<uintptr_t>(ss->residual) | reinterpret_cast<uintptr_t>(ss->finished)) & 15) return MOBI_ERR_ALIGN;
    if ((ss->rowvec_stride & 3) || (ss->res_img_stride & 7)) return MOBI_ERR_ALIGN;
    if (!mobi_groupnorm_takes_split(p->c0, p->c1, p->batch, p->hw)) return MOBI_ERR_UNSUPPORTED;
    a.slabs = ss->slabs; a.splits = ss->count; a.slab_row = ss->row_stride;
    a.slab_stride = (long long)p->batch * p->hw * ss->row_stride;
    a.s_bias = ss->bias; a.s_rowvec = ss->rowvec; a.s_rowvec_stride = ss->rowvec_stride ? ss->rowvec_stride : p->c0;
    a.s_resid = ss->residual; a.s_res_img = ss->res_img_stride ? ss->res_img_stride : (long long)p->hw * p->c0;
    a.finished = ss->finished;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return p->dtype == MOBI_F16 ? launch_gn<f16_t>(a, p->batch, st) : launch_gn<bf16_t>(a, p->batch, st);
}

extern "C" int mobi_layernorm(const mobi_layernorm_params* p, void* stream) {
  using namespace mobi;
  if (!p || !p->src || !p->out || !p->gamma || !p->beta) return MOBI_ERR_ARG;
  if (p->dtype != MOBI_F16 && p->dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (p->images <= 0 || p->rows_per_image <= 0 || p->channels <= 0 || (p->channels & 7)) return MOBI_ERR_UNSUPPORTED;
  if ((p->src_img_stride & 7) || (p->out_img_stride & 7)) return MOBI_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(p->src) | reinterpret_cast<uintptr_t>(p->out) | reinterpret_cast<uintptr_t>(p->gamma) |
       reinterpret_cast<uintptr_t>(p->beta)) & 15) return MOBI_ERR_ALIGN;
  LnArgs a;
  a.src = p->src; a.out = p->out; a.images = p->images; a.rows = p->rows_per_image; a.C = p->channels;
  const long long dense = (long long)p->rows_per_image * p->channels;
  a.src_img = p->src_img_stride ? p->src_img_stride : dense;
  a.out_img = p->out_img_stride ? p->out_img_stride : dense;
  a.gamma = p->gamma; a.beta = p->beta; a.eps = p->eps;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return p->dtype == MOBI_F16 ? launch_ln<f16_t>(a, st) : launch_ln<bf16_t>(a, st);
}
