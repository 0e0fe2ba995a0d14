// GroupNorm(32)+SiLU and LayerNorm for gfx950: HBM-bound streaming kernels,
// 16-byte accesses, fp32 statistics (the reference's GroupNorm32 computes in fp32,
// ldm/modules/diffusionmodules/util.py:214-216).
//
// GroupNorm runs as two launches:
//   stats: grid (chunks, images); each block sweeps a chunk of pixels with every thread owning
//          a fixed 8-channel column, folds per-channel sums into the 32 groups through LDS in
//          a fixed order (deterministic) and writes one (sum, sumsq) pair per group;
//   apply: every block first combines its image's chunk partials in fp64 (cancellation-safe
//          E[x^2] - mean^2), then streams y = x * a[c] + b[c], optional SiLU.
// Algorithmic bytes: 2 B read (stats) + 2 B read + 2 B written (apply) per element.
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

template <typename T>
static int launch_gn(const GnArgs& a, int batch, hipStream_t st) {
  // small tensors: one launch, the group's slab in LDS (channel pairs: C / 32 even; both sources split at an even channel)
  const int cpg = a.C / 32;
  if (!(cpg & 1) && (long long)a.hw * cpg <= GN1_MAX && tuning().gn_fused != 0) {
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
  return (size_t)batch * mobi::gn_chunks(hw) * 32 * 2 * sizeof(float);
}

extern "C" int mobi_groupnorm(const mobi_groupnorm_params* p, void* stream) {
  using namespace mobi;
  if (!p || !p->src0 || !p->out || !p->ws || !p->gamma || !p->beta) return MOBI_ERR_ARG;
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
