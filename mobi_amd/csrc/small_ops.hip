// Small / HBM-bound kernels around the matrix-core paths: skinny dense layers on fp32
// vectors, timestep embedding, few-token context attention, row softmax, direct
// convolutions for the thin ends of the networks, layout converters.
#include "common.h"
#include "tuning.h"

namespace mobi {

// ---------------------------------------------------------------------------------------
// skinny linear (m <= 16 rows of fp32 activations against a tall T weight matrix): the weights are the only
// real traffic (read once, 16 B per lane); x is staged in LDS per 512-wide k chunk (optionally through SiLU)
// and shared by the block's 16 output columns; each wave owns 4 columns, two at a time.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void skinny_linear_kernel(const mobi_skinny_linear_params a) {
  constexpr int KC = 512, MMAX = 16, PAIRS = 2;
  __shared__ __attribute__((aligned(16))) float xs[MMAX * KC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_base = blockIdx.x * (8 * PAIRS) + wave * (2 * PAIRS);
  const T* __restrict__ wbase = reinterpret_cast<const T*>(a.weight);
  float acc[PAIRS][2][MMAX];        // [pair][column of pair][row]  -- fully unrolled below
#pragma unroll
  for (int p = 0; p < PAIRS; ++p)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int m = 0; m < MMAX; ++m) acc[p][c][m] = 0.f;
  for (int k0 = 0; k0 < a.k; k0 += KC) {
    __syncthreads();
    for (int i = tid; i < a.m * (KC / 4); i += 256) {
      const int m = i / (KC / 4), kq = (i - m * (KC / 4)) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k0 + kq < a.k) v = *reinterpret_cast<const f32x4*>(a.x + (long long)m * a.x_row_stride + k0 + kq);
      if (a.pre_act == MOBI_ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
      }
      *reinterpret_cast<f32x4*>(xs + m * KC + kq) = v;
    }
    __syncthreads();
    const int kk = lane * 8;                       // this lane's 8 k values of the chunk
    const bool kok = k0 + kk < a.k;
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      float w0[8], w1[8];
      const int n0 = n_base + 2 * p, n1 = n0 + 1;
      u32x4 r0 = {0u, 0u, 0u, 0u}, r1 = {0u, 0u, 0u, 0u};
      if (kok && n0 < a.n) r0 = ld16(wbase + (long long)n0 * a.k + k0 + kk);
      if (kok && n1 < a.n) r1 = ld16(wbase + (long long)n1 * a.k + k0 + kk);
      unpack8<T>(r0, w0);
      unpack8<T>(r1, w1);
#pragma unroll
      for (int m = 0; m < MMAX; ++m) {
        if (m < a.m) {
          const f32x4 xa = *reinterpret_cast<const f32x4*>(xs + m * KC + kk);
          const f32x4 xb = *reinterpret_cast<const f32x4*>(xs + m * KC + kk + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[p][0][m] += w0[j] * xa[j] + w0[4 + j] * xb[j];
            acc[p][1][m] += w1[j] * xa[j] + w1[4 + j] * xb[j];
          }
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < PAIRS; ++p)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int n = n_base + 2 * p + c;
#pragma unroll
      for (int m = 0; m < MMAX; ++m) {
        if (m < a.m) {
          const float s = wave_sum(acc[p][c][m]);
          if (lane == 0 && n < a.n) {
            float o = s + (a.bias ? a.bias[n] : 0.f);
            if (a.post_act == MOBI_ACT_SILU) o = silu_f(o);
            else if (a.post_act == MOBI_ACT_GELU) o = gelu_erf_f(o);
            a.out[(long long)m * a.out_row_stride + n] = o;
          }
        }
      }
    }
}

// The same layer on the matrix cores (k % 32 == 0).  The vector-ALU kernel above spends m FMAs per weight element and keeps two
// 16-byte weight loads in flight per wave: 27-100 us per launch of the time-embedding chain (0.16 ms per denoising step).  Here
// the fp32 rows are split into three T pieces (hi + mid + lo = x to 24 bits for bf16, more for fp16: the products are exact,
// the sums fp32 -- within 1e-6 of the fp32 FMA chain) and staged in LDS as MFMA 16x16x32 A-operand images per 512-deep k chunk; a
// block owns 16 output columns, its four waves take every fourth 32-deep step, a lane loads its column's 8 weights of a step
// straight from global memory (all of the chunk's loads in flight), three MFMAs per step, partial sums folded through LDS
// in wave order.  The weights are the traffic: 45 MB for the batched emb_layers projection.
// WIDE (many columns: the batched projection): a block owns 64 columns, 16 per wave, every wave walks all steps of a chunk --
// the staging of x (SiLU, split: the same for every block) is shared by four times the columns.  Otherwise a block owns 16
// columns and its four waves take every fourth step (few columns: more blocks, shorter chains).
template <typename T, bool WIDE>
__global__ __launch_bounds__(256) void skinny_linear_mfma_kernel(const mobi_skinny_linear_params a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int KC = 512, STEPS = KC / 32, SPW = WIDE ? STEPS : STEPS / 4;     // steps per chunk / per wave and chunk
  __shared__ __attribute__((aligned(16))) unsigned char xs[3 * STEPS * 1024];     // [piece][step][row][64 B]
  __shared__ float part[WIDE ? 1 : 4][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  const int n = WIDE ? (blockIdx.x * 4 + wave) * 16 + r16 : blockIdx.x * 16 + r16;
  const bool n_ok = n < a.n;
  const T* __restrict__ wrow = reinterpret_cast<const T*>(a.weight) + (long long)(n_ok ? n : 0) * a.k + 8 * g4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < a.k; k0 += KC) {
    const int steps = min(STEPS, (a.k - k0) >> 5);
    // this wave's weight pieces of the chunk: requested before the staging barrier, consumed behind it
    u32x4 wf[SPW];
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      const int st = WIDE ? i : wave + 4 * i;
      wf[i] = u32x4{0u, 0u, 0u, 0u};
      if (st < steps && n_ok) wf[i] = ld16(wrow + k0 + 32 * st);
    }
    __syncthreads();                                                  // (the previous chunk's images are read)
    for (int i = tid; i < 16 * (KC / 8); i += 256) {                   // (row, 8 consecutive k): split and store three 16-byte pieces
      const int m = i / (KC / 8), kq = (i - m * (KC / 8)) * 8;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0.f;
      if (m < a.m && k0 + kq < a.k) {
        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(a.x + (long long)m * a.x_row_stride + k0 + kq);
        const f32x4 hi4 = *reinterpret_cast<const f32x4*>(a.x + (long long)m * a.x_row_stride + k0 + kq + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = lo4[j]; v[4 + j] = hi4[j]; }
        if (a.pre_act == MOBI_ACT_SILU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = silu_f(v[j]);
        }
      }
      float p0[8], p1[8], p2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const T h = (T)v[j];
        const float r1 = v[j] - (float)h;
        const T md = (T)r1;
        p0[j] = (float)h; p1[j] = (float)md; p2[j] = r1 - (float)md;
      }
      unsigned char* dst = xs + ((kq >> 5) * 16 + m) * 64 + ((kq >> 3) & 3) * 16;
      st16(dst, pack8<T>(p0));
      st16(dst + STEPS * 1024, pack8<T>(p1));
      st16(dst + 2 * STEPS * 1024, pack8<T>(p2));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SPW; ++i) {
      const int st = WIDE ? i : wave + 4 * i;
      if (st < steps) {                                               // (wave-uniform)
        const unsigned char* img = xs + (st * 16 + r16) * 64 + g4 * 16;
        const frag_t w8 = __builtin_bit_cast(frag_t, wf[i]);
        acc = mfma16(__builtin_bit_cast(frag_t, ld16(img)), w8, acc);
        acc = mfma16(__builtin_bit_cast(frag_t, ld16(img + STEPS * 1024)), w8, acc);
        acc = mfma16(__builtin_bit_cast(frag_t, ld16(img + 2 * STEPS * 1024)), w8, acc);
      }
    }
  }
  // lane (column r16, g4) holds rows 4 g4 .. 4 g4 + 3; without WIDE the four waves' partial sums are folded in wave order
  if constexpr (!WIDE) {
#pragma unroll
    for (int j = 0; j < 4; ++j) part[wave][lane][j] = acc[j];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (part[0][lane][j] + part[1][lane][j]) + (part[2][lane][j] + part[3][lane][j]);
  }
  if (n_ok) {
    const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = 4 * g4 + j;
      if (m < a.m) {
        float o = acc[j] + bias;
        if (a.post_act == MOBI_ACT_SILU) o = silu_f(o);
        else if (a.post_act == MOBI_ACT_GELU) o = gelu_erf_f(o);
        a.out[(long long)m * a.out_row_stride + n] = o;
      }
    }
  }
}

// LayerNorm over the last axis of a few fp32 rows (the token mapper / bbox MLP work on ONE fp32 token per image): one wave
// per row, two passes in registers (mean, then centred variance), torch's formula (x - mean) * rsqrt(var + eps) * g + b
__global__ void layernorm_rows_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                          const float* __restrict__ beta, float* __restrict__ out, int rows, int cols,
                                          int x_stride, int out_stride, float eps) {
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long long)row * x_stride;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += xr[c];
  const float mean = wave_sum(s) / (float)cols;
  float v = 0.f;
  for (int c = lane; c < cols; c += 64) { const float d = xr[c] - mean; v += d * d; }
  const float rstd = rsqrtf(wave_sum(v) / (float)cols + eps);
  float* o = out + (long long)row * out_stride;
  for (int c = lane; c < cols; c += 64) o[c] = (xr[c] - mean) * rstd * gamma[c] + beta[c];
}

// CLIP's "quick_gelu" (transformers: x * sigmoid(1.702 x)), elementwise on a T tensor, 8 values per thread
template <typename T>
__global__ void quick_gelu_kernel(const T* __restrict__ src, T* __restrict__ out, long long vecs) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (long long)gridDim.x * blockDim.x) {
    float f[8];
    unpack8<T>(ld16(src + i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f[j] / (1.0f + __expf(-1.702f * f[j]));
    st16(out + i * 8, pack8<T>(f));
  }
}

__global__ void timestep_embedding_kernel(const int64_t* t, const float* freqs, float* out, int n, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half) return;
  const int b = i / half, f = i - b * half;
  const float arg = (float)t[b] * freqs[f];
  out[(long long)b * 2 * half + f] = cosf(arg);
  out[(long long)b * 2 * half + half + f] = sinf(arg);
}

// ---------------------------------------------------------------------------------------
// attention against <= 8 context tokens: one thread per (token row, head)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ctx_attention_kernel(const mobi_ctx_attention_params a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.images * a.tq * a.heads;
  if (idx >= total) return;
  const int head = (int)(idx % a.heads);
  const long long row = idx / a.heads;
  const int img = (int)(row / a.tq);
  const int C = a.heads * a.dh;
  const T* __restrict__ q = reinterpret_cast<const T*>(a.q) + row * C + head * a.dh;
  T* __restrict__ o = reinterpret_cast<T*>(a.out) + row * C + head * a.dh;
  const float* __restrict__ kb = a.k + (long long)img * a.tk * C + head * a.dh;
  const float* __restrict__ vb = a.v + (long long)img * a.tk * C + head * a.dh;
  float s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = 0.f;
  for (int d = 0; d < a.dh; d += 8) {
    float qf[8];
    unpack8<T>(ld16(q + d), qf);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.tk) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[j] += qf[e] * kb[(long long)j * C + d + e];
      }
  }
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < a.tk) { s[j] *= a.scale; mx = fmaxf(mx, s[j]); }
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < a.tk) { s[j] = __expf(s[j] - mx); den += s[j]; }
  const float inv = 1.f / den;
  for (int d = 0; d < a.dh; d += 8) {
    float of[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) of[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.tk) {
        const float pj = s[j] * inv;
#pragma unroll
        for (int e = 0; e < 8; ++e) of[e] += pj * vb[(long long)j * C + d + e];
      }
    st16(o + d, pack8<T>(of));
  }
}

// row softmax fp32 -> T, one wave per row
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* src, T* out, long long rows, int cols) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = src + row * cols;
  T* o = out + row * cols;
  float mx = -INFINITY;
  for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, s[c]);
  mx = wave_max(mx);
  float den = 0.f;
  for (int c = lane; c < cols; c += 64) den += __expf(s[c] - mx);
  den = wave_sum(den);
  const float inv = 1.f / den;
  for (int c = lane; c < cols; c += 64) o[c] = from_f32<T>(__expf(s[c] - mx) * inv);
}

// ---------------------------------------------------------------------------------------
// direct conv, few input channels: fp32 NCHW sources -> T channels-last or fp32 NCHW.
// one thread per (pixel, 8 output channels)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_small_cin_kernel(const mobi_conv_small_cin_params a) {
  const int ncg = (a.cout + 7) >> 3;
  const long long hw = (long long)a.h * a.w;
  const long long total = (long long)a.batch * hw * ncg;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % ncg);
  const long long pix = idx / ncg;
  const int img = (int)(pix / hw);
  const int rem = (int)(pix - (long long)img * hw);
  const int y = rem / a.w, x = rem - y * a.w;
  const int cin = a.c[0] + a.c[1] + a.c[2];
  const int taps = a.kh * a.kw;
  const int K = cin * taps;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int co = cg * 8 + j;
    acc[j] = (a.bias && co < a.cout) ? a.bias[co] : 0.f;
  }
  int cbase = 0;
  for (int s = 0; s < 3; ++s) {
    if (!a.src[s] || a.c[s] <= 0) continue;
    const float* __restrict__ sp = a.src[s] + (long long)img * a.c[s] * hw;
    for (int c = 0; c < a.c[s]; ++c) {
      for (int ky = 0; ky < a.kh; ++ky) {
        const int yy = y + ky - a.pad_h;
        if (yy < 0 || yy >= a.h) continue;
        for (int kx = 0; kx < a.kw; ++kx) {
          const int xx = x + kx - a.pad_w;
          if (xx < 0 || xx >= a.w) continue;
          const float v = sp[(long long)c * hw + (long long)yy * a.w + xx];
          const int k = ((cbase + c) * a.kh + ky) * a.kw + kx;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int co = cg * 8 + j;
            if (co < a.cout) acc[j] += v * a.weight[(long long)co * K + k];
          }
        }
      }
    }
    cbase += a.c[s];
  }
  if (a.out_f32_nchw) {
    float* o = reinterpret_cast<float*>(a.out);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int co = cg * 8 + j;
      if (co < a.cout) o[((long long)img * a.cout + co) * hw + rem] = acc[j];
    }
  } else {
    T* o = reinterpret_cast<T*>(a.out);
    st16(o + pix * a.cout + cg * 8, pack8<T>(acc));
  }
}

// direct conv, few output channels: T channels-last -> fp32 NCHW; one wave per output pixel
template <typename T>
__global__ __launch_bounds__(256) void conv_small_cout_kernel(const mobi_conv_small_cout_params a) {
  const int lane = threadIdx.x & 63;
  const long long hw = (long long)a.h * a.w;
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (long long)a.batch * hw) return;
  const int img = (int)(pix / hw);
  const int rem = (int)(pix - (long long)img * hw);
  const int y = rem / a.w, x = rem - y * a.w;
  const int vpt = a.cin >> 3;                 // 8-channel vectors per tap
  const int taps = a.kh * a.kw;
  const int K = taps * a.cin;
  const T* __restrict__ src = reinterpret_cast<const T*>(a.src) + (long long)img * hw * a.cin;
  const T* __restrict__ w = reinterpret_cast<const T*>(a.weight);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (int v = lane; v < taps * vpt; v += 64) {
    const int tap = v / vpt, c = (v - tap * vpt) * 8;
    const int ky = tap / a.kw, kx = tap - ky * a.kw;
    const int yy = y + ky - a.pad_h, xx = x + kx - a.pad_w;
    if (yy < 0 || yy >= a.h || xx < 0 || xx >= a.w) continue;
    float xf[8];
    unpack8<T>(ld16(src + ((long long)yy * a.w + xx) * a.cin + c), xf);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.cout) {
        float wf[8];
        unpack8<T>(ld16(w + (long long)j * K + tap * a.cin + c), wf);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j] += xf[e] * wf[e];
      }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (j < a.cout) {
      float s = wave_sum(acc[j]);
      if (lane == 0) {
        s += a.bias ? a.bias[j] : 0.f;
        if (a.clamp) s = fminf(fmaxf(s, a.clamp_lo), a.clamp_hi);
        a.out[((long long)img * a.cout + j) * hw + rem] = s;
      }
    }
}

// The same convolution on the matrix cores (cin % 32 == 0, w % 16 == 0, at most 8 output channels: the output convolutions
// of the UNet, 320 -> 4 at 64 x 64, and of the VAE decoders, 128 -> 3 / 2 at 512 x 512).  The kernel above reads the 9 x cin
// weights of all output channels once per PIXEL (1.5 GB of cache traffic per UNet call, 119 us); here they sit in LDS as
// MFMA 16x16x32 A-operand images (8 weight rows + a zero row that the upper 8 lanes read), a wave owns 16 consecutive pixels
// of an image row per tile, and a lane loads its pixel's 32-channel pieces of each tap straight from global memory as the B
// fragments (taps outside the image: zeros).  D[channel][pixel] lands with 16 consecutive pixels of a channel in 16 lanes:
// 64-byte runs of the fp32 NCHW output.
template <typename T, int KSTEPS>                         // KSTEPS = cin / 32 (10: cin 320, 4: cin 128)
__global__ __launch_bounds__(256) void conv_small_cout_mfma_kernel(const mobi_conv_small_cout_params a, int tiles_per_wave) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int CIN = 32 * KSTEPS;
  constexpr int ROW = 64;                                  // bytes per weight row and 32-deep step
  extern __shared__ __attribute__((aligned(16))) unsigned char csm_lds[];     // [taps][KSTEPS][9 rows][64 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  const int taps = a.kh * a.kw, K = taps * CIN;
  {
    const T* __restrict__ w = reinterpret_cast<const T*>(a.weight);
    const int pieces = taps * KSTEPS * 9 * 4;               // 16-byte pieces
    for (int i = tid; i < pieces; i += 256) {
      const int ch = i & 3, row = (i >> 2) % 9, st = (i >> 2) / 9;          // st = tap * KSTEPS + step
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < a.cout) v = ld16(w + (long long)row * K + st * 32 + ch * 8);
      st16(csm_lds + (st * 9 + row) * ROW + ch * 16, v);
    }
  }
  __syncthreads();
  const int tiles_x = a.w >> 4;
  const long long hw = (long long)a.h * a.w;
  const long long tiles = (long long)a.batch * a.h * tiles_x;
  const unsigned char* wrd = csm_lds + min(r16, 8) * ROW + g4 * 16;
  const long long t_begin = ((long long)blockIdx.x * 4 + wave) * tiles_per_wave;
  for (long long t = t_begin; t < t_begin + tiles_per_wave && t < tiles; ++t) {
    const int tx = (int)(t % tiles_x);
    const long long ry = t / tiles_x;
    const int y = (int)(ry % a.h), img = (int)(ry / a.h);
    const int x = tx * 16 + r16;
    const T* __restrict__ src = reinterpret_cast<const T*>(a.src) + (long long)img * hw * CIN + 8 * g4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    u32x4 cur[KSTEPS], nxt[KSTEPS];
    auto fetch = [&](int tap, u32x4 (&dst)[KSTEPS]) {
      const int ky = tap / a.kw, kx = tap - ky * a.kw;
      const int yy = y + ky - a.pad_h, xx = x + kx - a.pad_w;
      const bool ok = tap < taps && yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
      const T* __restrict__ row = src + ((long long)(ok ? yy : y) * a.w + (ok ? xx : x)) * CIN;
#pragma unroll
      for (int s_ = 0; s_ < KSTEPS; ++s_) {
        dst[s_] = ld16(row + 32 * s_);
        if (!ok) dst[s_] = u32x4{0u, 0u, 0u, 0u};
      }
    };
    fetch(0, cur);
    for (int tap = 0; tap < taps; ++tap) {
      fetch(tap + 1, nxt);                                  // (past the last tap: a valid address, zeros, unused)
      const unsigned char* wt = wrd + tap * (KSTEPS * 9 * ROW);
#pragma unroll
      for (int s_ = 0; s_ < KSTEPS; ++s_)
        acc = mfma16(__builtin_bit_cast(frag_t, ld16(wt + s_ * 9 * ROW)), __builtin_bit_cast(frag_t, cur[s_]), acc);
#pragma unroll
      for (int s_ = 0; s_ < KSTEPS; ++s_) cur[s_] = nxt[s_];
    }
    // lane (pixel r16, g4) holds output channels 4 g4 .. 4 g4 + 3
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch = 4 * g4 + j;
      if (ch < a.cout) {
        float v = acc[j] + (a.bias ? a.bias[ch] : 0.f);
        if (a.clamp) v = fminf(fmaxf(v, a.clamp_lo), a.clamp_hi);
        a.out[((long long)img * a.cout + ch) * hw + (long long)y * a.w + x] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// layout converters
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* src, T* out, int batch, int c, int hw) {
  const long long total = (long long)batch * c * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const long long r = i / c;
    const int p = (int)(r % hw);
    const int b = (int)(r / hw);
    out[i] = from_f32<T>(src[((long long)b * c + ch) * hw + p]);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* src, float* out, int batch, int c, int hw) {
  const long long total = (long long)batch * c * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(i % hw);
    const long long r = i / hw;
    const int ch = (int)(r % c);
    const int b = (int)(r / c);
    out[i] = to_f32(src[((long long)b * hw + p) * c + ch]);
  }
}

// up to 3 fp32 NCHW sources concatenated on channels -> T channels-last, zero-padded to c_pad channels
template <typename T>
__global__ __launch_bounds__(256) void pack_sources_kernel(const float* s0, const float* s1, const float* s2, int c0,
                                                           int c1, int c2, int batch, int hw, int c_pad, T* out) {
  const long long total = (long long)batch * hw * (c_pad >> 3);
  const int vpp = c_pad >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % vpp);
    const long long pix = i / vpp;
    const int b = (int)(pix / hw);
    const int p = (int)(pix - (long long)b * hw);
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = v * 8 + j;
      float val = 0.f;
      if (c < c0) val = s0[((long long)b * c0 + c) * hw + p];
      else if (c < c0 + c1) val = s1[((long long)b * c1 + (c - c0)) * hw + p];
      else if (c < c0 + c1 + c2) val = s2[((long long)b * c2 + (c - c0 - c1)) * hw + p];
      f[j] = val;
    }
    st16(out + i * 8, pack8<T>(f));
  }
}

static inline unsigned grid_for(long long total, int block, long long cap = 65536) {
  long long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}


// ---------------------------------------------------------------------------------------
// two-key adapter (include/mobi_engine.h): LayerNorm statistics + `heads` gate logits + gated per-image vectors
// in one pass over the tokens.  One wave per token row (8 channels per lane and chunk), the per-image tables a / u / b
// in LDS, the 2 + heads row sums reduced together by a halving butterfly (17 shuffles instead of 6 per sum).
// ---------------------------------------------------------------------------------------
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void two_key_adapter_kernel(const mobi_two_key_adapter_params p, int rows_per_block) {
  __shared__ __attribute__((aligned(16))) float tk_lds[19 * 512 * MAXV];      // C <= 512 * MAXV (17 C tables + 2 C for the second result)
  const int C = p.channels, H = p.heads, V = C >> 3;
  float* s_a = tk_lds;                     // [H][C]
  float* s_u = tk_lds + 8 * C;             // [H][C]
  float* s_b = tk_lds + 16 * C;            // [C]
  float* s_lg = tk_lds + 17 * C;           // [C] gamma, [C] beta of the second result's LayerNorm (natural channel order)
  float* s_lb = tk_lds + 18 * C;
  const bool ln2 = p.ln_out[0] != nullptr; // (uniform)
  const int img = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // LDS image of a table row: the FIRST four floats of every 8-channel group, then the SECOND four (a lane reads
  // 16 bytes at lane * 16 of each half: conflict-free; the natural [channel] order puts lanes 32 bytes apart).
  // ALWAYS eight head rows (rows >= heads are zero: gate 0.5 x vector 0), so the row loop has no head-count branch
  // and the compiler batches the 34 LDS reads of a row instead of waiting per head.
  {
    const f32x4* ga = reinterpret_cast<const f32x4*>(p.a + (long long)img * H * C);
    const f32x4* gu = reinterpret_cast<const f32x4*>(p.u + (long long)img * H * C);
    const f32x4* gb = reinterpret_cast<const f32x4*>(p.b + (long long)img * C);
    const int Q = C >> 2;                                    // float4 per table row
    for (int i = tid; i < 8 * Q; i += 256) {
      const int h = i / Q, q = i - h * Q;
      const int d = h * Q + (q & 1) * (Q >> 1) + (q >> 1);
      const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
      reinterpret_cast<f32x4*>(s_a)[d] = h < H ? ga[i] : z;
      reinterpret_cast<f32x4*>(s_u)[d] = h < H ? gu[i] : z;
    }
    for (int q = tid; q < Q; q += 256) reinterpret_cast<f32x4*>(s_b)[(q & 1) * (Q >> 1) + (q >> 1)] = gb[q];
    if (ln2) {
      const float* lg = (img & 1) ? p.ln_gamma[1] : p.ln_gamma[0];
      const float* lb = (img & 1) ? p.ln_beta[1] : p.ln_beta[0];
      for (int c = tid; c < C; c += 256) { s_lg[c] = lg[c]; s_lb[c] = lb[c]; }
    }
  }
  float a_sum[8], cc[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    a_sum[h] = h < H ? p.a_sum[img * H + h] : 0.f;
    cc[h] = h < H ? p.c[img * H + h] : 0.f;
  }
  __syncthreads();
  const long long ximg = p.x_img_stride ? p.x_img_stride : (long long)p.rows_per_image * C;
  const long long oimg = p.out_img_stride ? p.out_img_stride : (long long)p.rows_per_image * C;
  const T* __restrict__ xb = reinterpret_cast<const T*>(p.x) + img * ximg;
  T* __restrict__ ob = reinterpret_cast<T*>(p.out) + img * oimg;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(p.rows_per_image, r_begin + rows_per_block);
  const float inv_c = 1.0f / (float)C;
  // the next row's 16-byte loads are in flight while this row is reduced, gated and stored (the per-row chain --
  // load, six shuffle levels, sigmoid, LDS-fed update -- is otherwise exposed once per row at 8-16 waves per CU)
  u32x4 cur[MAXV];
  auto load_row = [&](int r, u32x4 (&raw)[MAXV]) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + 64 * i;
      raw[i] = u32x4{0u, 0u, 0u, 0u};
      if (v < V && r < r_end) raw[i] = ld16(xb + (long long)r * C + v * 8);
    }
  };
  load_row(r_begin + wave, cur);
  for (int r = r_begin + wave; r < r_end; r += 4) {
    u32x4 nxt[MAXV];
    load_row(r + 4, nxt);
    float x[MAXV][8];
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + 64 * i;
      if (v < V) {
        unpack8<T>(cur[i], x[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[0] += x[i][j]; acc[1] += x[i][j] * x[i][j]; }
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const f32x4 a0 = reinterpret_cast<const f32x4*>(s_a)[h * (C >> 2) + v];
          const f32x4 a1 = reinterpret_cast<const f32x4*>(s_a)[h * (C >> 2) + (C >> 3) + v];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[2 + h] += x[i][j] * a0[j] + x[i][4 + j] * a1[j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[i][j] = 0.f;
      }
    }
    // halving butterfly: after the step with distance d a lane keeps the half of its sums selected by its bit d
    // (distance 32 and 16 by gfx950's lane-row swaps: v_permlane32_swap exchanges the upper half of its first operand
    //  with the lower half of the second, so first + second afterwards holds sum i in the lower and sum i + 8 in the
    //  upper half -- no LDS crossbar trip, no select; same for 16-lane rows)
    float v8[8], v4[4], v2[2], v1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float lo = acc[i], hi = acc[i + 8];
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
      v8[i] = lo + hi;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float lo = v8[i], hi = v8[i + 4];
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
      v4[i] = lo + hi;
    }
    {
      const bool hi = lane & 8;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float keep = hi ? v4[i + 2] : v4[i], send = hi ? v4[i] : v4[i + 2];
        v2[i] = keep + __shfl_xor(send, 8, 64);
      }
    }
    {
      const bool hi = lane & 4;
      const float keep = hi ? v2[1] : v2[0], send = hi ? v2[0] : v2[1];
      v1 = keep + __shfl_xor(send, 4, 64);
    }
    v1 += __shfl_xor(v1, 2, 64);
    v1 += __shfl_xor(v1, 1, 64);
    // sum k = 8 b5 + 4 b4 + 2 b3 + b2 sits in the lanes whose bits 5..2 are (b5, b4, b3, b2)
    auto total = [&](int k) {
      const int src = ((k >> 3) & 1) * 32 + ((k >> 2) & 1) * 16 + ((k >> 1) & 1) * 8 + (k & 1) * 4;
      return __builtin_amdgcn_readlane(__builtin_bit_cast(int, v1), src);
    };
    const float mean = __builtin_bit_cast(float, total(0)) * inv_c;
    float var = __builtin_bit_cast(float, total(1)) * inv_c - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + p.eps);
    float g[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      const float z = rstd * (__builtin_bit_cast(float, total(2 + h)) - mean * a_sum[h]) + cc[h];
      g[h] = 1.0f / (1.0f + __expf(-z));
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = lane + 64 * i;
      if (v < V) {
        float o[8];
        const f32x4 b0 = reinterpret_cast<const f32x4*>(s_b)[v], b1 = reinterpret_cast<const f32x4*>(s_b)[(C >> 3) + v];
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = x[i][j] + b0[j]; o[4 + j] = x[i][4 + j] + b1[j]; }
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const f32x4 u0 = reinterpret_cast<const f32x4*>(s_u)[h * (C >> 2) + v];
          const f32x4 u1 = reinterpret_cast<const f32x4*>(s_u)[h * (C >> 2) + (C >> 3) + v];
#pragma unroll
          for (int j = 0; j < 4; ++j) { o[j] += g[h] * u0[j]; o[4 + j] += g[h] * u1[j]; }
        }
        const u32x4 packed = pack8<T>(o);
        st16(ob + (long long)r * C + v * 8, packed);
        cur[i] = packed;                                     // the stored (rounded) row: what a LayerNorm launch would read
      }
    }
    if (ln2) {
      // second result (see two_key_adapter_regs_kernel): LayerNorm of the stored row, mean then variance about the mean
      float s1 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < V) {
          float f[8];
          unpack8<T>(cur[i], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) s1 += f[j];
        }
      const float mean2 = wave_sum(s1) * inv_c;
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i)
        if (lane + 64 * i < V) {
          float f[8];
          unpack8<T>(cur[i], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float d = f[j] - mean2; q2 += d * d; }
        }
      const float rstd2 = rsqrtf(wave_sum(q2) * inv_c + p.ln_eps);
      T* __restrict__ lrow = reinterpret_cast<T*>((img & 1) ? p.ln_out[1] : p.ln_out[0]) +
                             ((long long)(img >> 1) * p.rows_per_image + r) * C;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int v = lane + 64 * i;
        if (v < V) {
          float f[8], o[8];
          unpack8<T>(cur[i], f);
          const f32x4 g0 = *reinterpret_cast<const f32x4*>(s_lg + v * 8), g1 = *reinterpret_cast<const f32x4*>(s_lg + v * 8 + 4);
          const f32x4 h0 = *reinterpret_cast<const f32x4*>(s_lb + v * 8), h1 = *reinterpret_cast<const f32x4*>(s_lb + v * 8 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[j] = (f[j] - mean2) * rstd2 * g0[j] + h0[j];
            o[4 + j] = (f[4 + j] - mean2) * rstd2 * g1[j] + h1[j];
          }
          st16(lrow + v * 8, pack8<T>(o));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) cur[i] = nxt[i];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Two-key adapter on the matrix cores.  The VALU kernel above spends ~280 instructions per token row on two skinny
// products (8 gate logits = LN(x) . a_h, and the gated sum of 8 per-image vectors) and 34 LDS table reads; here both
// are MFMA 16x16x32 products on 16-row groups:
//   logits  D[h][token]  = sum_c  A[h][c] x[token][c]                (first operand: table rows, second: x rows)
//   update  D[ch][token] = sum_h  U^T[ch][h] g[token][h]             (k = 8 heads, zero-padded to 32)
// The fp32 tables are split into hi + lo parts of the storage type (two MFMAs each), so the products carry ~16
// mantissa bits whatever the storage type; a_sum is re-derived from the split table, so the LayerNorm mean term
// cancels against the SAME numbers the logits were formed with.  x is staged once through LDS (coalesced 16-byte
// loads), read as MFMA fragments (LN statistics come from the same registers) and as 8-byte pieces in the
// accumulator layout for the final add; the result goes back through the same LDS tile for coalesced stores.
// Per 16 rows and wave: 60 MFMAs, ~500 vector instructions (31 per row instead of 280).
// ---------------------------------------------------------------------------------------------------------
template <typename T, int CMAX>
__global__ __launch_bounds__(256) void two_key_adapter_mfma_kernel(const mobi_two_key_adapter_params p, int rows_per_block) {
  typedef typename Vec8<T>::type frag_t;
  // static LDS, sized for CMAX channels (320: 64 KB, two blocks per CU; 640: 127 KB)
  __shared__ __attribute__((aligned(16))) unsigned char tka_lds[2 * 8 * (CMAX * 2 + 16) + 2 * CMAX * 16 + (CMAX + 8) * 4 +
                                                                4 * 16 * (CMAX * 2 + 16)];
  const int C = p.channels, H = p.heads;
  const int AST = C * 2 + 16;                       // bytes per table row (hi or lo), 16-byte aligned, odd multiple of 16 ...
  const int XST = C * 2 + 16;                       // ... and per x row: conflict-free fragment reads
  unsigned char* s_ahi = tka_lds;                   // T [8][C] (+pad): rows >= H are zero (MFMA rows 8..15 read zeros)
  unsigned char* s_alo = s_ahi + 8 * AST;
  unsigned char* s_uhi = s_alo + 8 * AST;           // T [C][8]: U^T, heads contiguous
  unsigned char* s_ulo = s_uhi + C * 16;
  float* s_b = reinterpret_cast<float*>(s_ulo + C * 16);          // [C]
  float* s_asum = s_b + C;                          // [8] sums of the split table rows
  unsigned char* s_x = reinterpret_cast<unsigned char*>(s_asum + 8);
  const int img = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  unsigned char* xt = s_x + wave * 16 * XST;        // this wave's 16-row tile

  // ---- tables (once per block) --------------------------------------------------------------------------------
  {
    const float* ga = p.a + (long long)img * H * C;
    const float* gu = p.u + (long long)img * H * C;
    for (int i = tid; i < 8 * C; i += 256) {
      const int h = i / C, c = i - h * C;
      const float v = h < H ? ga[i] : 0.f;
      const T hi = (T)v;
      const T lo = (T)(v - (float)hi);
      *reinterpret_cast<T*>(s_ahi + h * AST + c * 2) = hi;
      *reinterpret_cast<T*>(s_alo + h * AST + c * 2) = lo;
    }
    for (int i = tid; i < 8 * C; i += 256) {
      const int h = i / C, c = i - h * C;
      const float v = h < H ? gu[i] : 0.f;
      const T hi = (T)v;
      const T lo = (T)(v - (float)hi);
      *reinterpret_cast<T*>(s_uhi + c * 16 + h * 2) = hi;
      *reinterpret_cast<T*>(s_ulo + c * 16 + h * 2) = lo;
    }
    for (int c = tid; c < C; c += 256) s_b[c] = p.b[(long long)img * C + c];
  }
  __syncthreads();
  {                                                 // sums of the numbers the matrix product will use: 32 threads per head
    const int h = tid >> 5, l = tid & 31;
    float sum = 0.f;
    for (int c = l; c < C; c += 32)
      sum += (float)*reinterpret_cast<const T*>(s_ahi + h * AST + c * 2) + (float)*reinterpret_cast<const T*>(s_alo + h * AST + c * 2);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (l == 0) s_asum[h] = sum;
  }
  __syncthreads();
  float a_sum[4], cc[4];                            // the lane's four heads: 4 g4 .. 4 g4 + 3 (g4 < 2)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int h = 4 * g4 + j;
    a_sum[j] = h < 8 ? s_asum[h] : 0.f;
    cc[j] = h < H ? p.c[img * H + h] : 0.f;
  }
  const long long ximg = p.x_img_stride ? p.x_img_stride : (long long)p.rows_per_image * C;
  const long long oimg = p.out_img_stride ? p.out_img_stride : (long long)p.rows_per_image * C;
  const T* __restrict__ xb = reinterpret_cast<const T*>(p.x) + img * ximg;
  T* __restrict__ ob = reinterpret_cast<T*>(p.out) + img * oimg;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(p.rows_per_image, r_begin + rows_per_block);
  const float inv_c = 1.0f / (float)C;
  const int pieces = 16 * (C >> 3);                 // 16-byte pieces of a 16-row tile (rows are contiguous in memory)
  const int ppr = C >> 3;                           // pieces per row

  // the next tile's 16-byte pieces are in flight (in registers) while this one is multiplied
  constexpr int NP = (16 * (CMAX >> 3) + 63) / 64;  // pieces per lane
  u32x4 nxt[NP];
  auto fetch = [&](int r0) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = lane + 64 * j;
      const int row = i / ppr, pc = i - row * ppr;
      nxt[j] = u32x4{0u, 0u, 0u, 0u};
      if (i < pieces && r0 + row < r_end) nxt[j] = ld16(xb + (long long)(r0 + row) * C + pc * 8);
    }
  };
  fetch(r_begin + 16 * wave);
  for (int r0 = r_begin + 16 * wave; r0 < r_end; r0 += 64) {
    // ---- x tile: registers -> LDS, next tile requested -----------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = lane + 64 * j;
      const int row = i / ppr, pc = i - row * ppr;
      if (i < pieces) st16(xt + row * XST + pc * 16, nxt[j]);
    }
    fetch(r0 + 64);
    __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0); the tile is private to the wave
    __builtin_amdgcn_wave_barrier();
    // ---- logits + LayerNorm sums --------------------------------------------------------------------------------
    f32x4 lg = f32x4{0.f, 0.f, 0.f, 0.f};
    float sx = 0.f, sxx = 0.f;
    for (int k0 = 0; k0 < C; k0 += 32) {
      const frag_t xf = __builtin_bit_cast(frag_t, ld16(xt + r16 * XST + (k0 + 8 * g4) * 2));
      frag_t ah, al;
#pragma unroll
      for (int j = 0; j < 8; ++j) { ah[j] = (T)0.0f; al[j] = (T)0.0f; }
      if (r16 < 8) {
        ah = __builtin_bit_cast(frag_t, ld16(s_ahi + r16 * AST + (k0 + 8 * g4) * 2));
        al = __builtin_bit_cast(frag_t, ld16(s_alo + r16 * AST + (k0 + 8 * g4) * 2));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = (float)xf[j]; sx += v; sxx += v * v; }
      lg = mfma16(ah, xf, lg);
      lg = mfma16(al, xf, lg);
    }
    sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);            // the four k-chunk lanes of a token
    sxx += __shfl_xor(sxx, 16, 64); sxx += __shfl_xor(sxx, 32, 64);
    const float mean = sx * inv_c;
    float var = sxx * inv_c - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + p.eps);
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z = rstd * (lg[j] - mean * a_sum[j]) + cc[j];
      g[j] = 1.0f / (1.0f + __expf(-z));
    }
    // gates of heads 4..7 live 16 lanes further on: bring them to the g4 = 0 lanes, which supply k = 0..7
    float gh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gh[j] = __shfl_down(g[j], 16, 64);
    frag_t gf;
#pragma unroll
    for (int j = 0; j < 4; ++j) { gf[j] = g4 == 0 ? (T)g[j] : (T)0.0f; gf[4 + j] = g4 == 0 ? (T)gh[j] : (T)0.0f; }
    // the gates are rounded to the storage type: carry the remainder as a second k-block so the product stays ~16-bit
    frag_t gl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gl[j] = g4 == 0 ? (T)(g[j] - (float)gf[j]) : (T)0.0f;
      gl[4 + j] = g4 == 0 ? (T)(gh[j] - (float)gf[4 + j]) : (T)0.0f;
    }
    // ---- update, final add, back into the tile -------------------------------------------------------------------
    for (int c0 = 0; c0 < C; c0 += 16) {
      frag_t uh, ul;
#pragma unroll
      for (int j = 0; j < 8; ++j) { uh[j] = (T)0.0f; ul[j] = (T)0.0f; }
      if (g4 == 0) {
        uh = __builtin_bit_cast(frag_t, ld16(s_uhi + (c0 + r16) * 16));
        ul = __builtin_bit_cast(frag_t, ld16(s_ulo + (c0 + r16) * 16));
      }
      f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
      d = mfma16(uh, gf, d);
      d = mfma16(ul, gf, d);
      d = mfma16(uh, gl, d);
      const int c = c0 + 4 * g4;                    // the lane's four channels of this tile, token r16
      const u32x2 xr = *reinterpret_cast<const u32x2*>(xt + r16 * XST + c * 2);
      typename Vec8<T>::half_type x4 = __builtin_bit_cast(typename Vec8<T>::half_type, xr);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(s_b + c);
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (float)x4[j] + b4[j] + d[j];
      *reinterpret_cast<u32x2*>(xt + r16 * XST + c * 2) = pack4<T>(o);
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // ---- LDS -> global ------------------------------------------------------------------------------------------
    for (int i = lane; i < pieces; i += 64) {
      const int row = i / ppr, pc = i - row * ppr;
      if (r0 + row < r_end) st16(ob + (long long)(r0 + row) * C + pc * 8, ld16(xt + row * XST + pc * 16));
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same two products with the token rows in REGISTERS (C = 320 / 640: the 64 x 64 and 32 x 32 levels).  A lane
// loads its token's row as the MFMA fragments it is multiplied as -- token lane & 15, channels 32 f + 8 (lane >> 4) .. + 7
// of k-block f -- straight from global memory (64-byte segments, each row read once, every load of a 16-token tile in
// flight at once) and keeps them.  The update product is issued as two 16-row halves per 32 channels whose table rows are
// ordered so that the accumulator registers a lane receives are the channels of the fragment it holds: A row m of half s
// is channel 32 f + 8 (m >> 2) + 4 s + (m & 3), so lane (token, q) gets channels 32 f + 8 q + 4 s + j.  x + b + update is
// then packed in place and stored as the same 16-byte pieces: no x tile in LDS (the kernel above: 128 C bytes of its
// 196 C), which leaves 68 C bytes of tables -- 23 / 46 KB: three / two blocks per CU.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int NF>
__global__ __launch_bounds__(256) void two_key_adapter_regs_kernel(const mobi_two_key_adapter_params p, int rows_per_block) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int C = 32 * NF;
  constexpr int AST = C * 2 + 16;                   // bytes per table row (hi or lo): odd multiple of 16, conflict-free reads
  // a: T [9][C] hi, lo (rows >= heads are zero; row 8 is what MFMA rows 8..15 read); u: T [C + 1][8] hi, lo (U^T, heads
  // contiguous; entry C is zero: what the lanes that supply k >= 8 read)
  __shared__ __attribute__((aligned(16))) unsigned char tkr_lds[2 * 9 * AST + 2 * (C + 1) * 16 + (C + 8) * 4 + 2 * C * 4];
  const int H = p.heads;
  unsigned char* s_ahi = tkr_lds;
  unsigned char* s_alo = s_ahi + 9 * AST;
  unsigned char* s_uhi = s_alo + 9 * AST;
  unsigned char* s_ulo = s_uhi + (C + 1) * 16;
  float* s_b = reinterpret_cast<float*>(s_ulo + (C + 1) * 16);    // [C]
  float* s_asum = s_b + C;                          // [8] sums of the split table rows
  float* s_lg = s_asum + 8;                         // [C] gamma, [C] beta of this image's second-result LayerNorm
  float* s_lb = s_lg + C;
  const bool ln2 = p.ln_out[0] != nullptr;          // (uniform)
  const int img = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r16 = lane & 15, g4 = lane >> 4;
  {
    const float* ga = p.a + (long long)img * H * C;
    const float* gu = p.u + (long long)img * H * C;
    // every load of the block in flight at once (the staging is a latency chain otherwise: 64 C bytes per block behind one
    // or two loads per thread); conflict-free LDS writes: a as four channels of one head per thread (8 bytes, consecutive
    // lanes consecutive), u as the eight heads of one channel per thread (one 16-byte entry)
#pragma unroll
    for (int i0 = 0; i0 < 2 * C; i0 += 256) {
      const int i = i0 + tid;                       // (2 C is a multiple of 64, not of 256: the last step is partial)
      const int h = i / (C / 4), c = (i - h * (C / 4)) * 4;
      if (i < 2 * C) {
        f32x4 va = f32x4{0.f, 0.f, 0.f, 0.f};
        if (h < H) va = *reinterpret_cast<const f32x4*>(ga + h * C + c);
        float lo[4], hi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { const T t = (T)va[e]; hi[e] = (float)t; lo[e] = va[e] - hi[e]; }
        *reinterpret_cast<u32x2*>(s_ahi + h * AST + c * 2) = pack4<T>(hi);
        *reinterpret_cast<u32x2*>(s_alo + h * AST + c * 2) = pack4<T>(lo);
      }
    }
#pragma unroll
    for (int c0 = 0; c0 < C; c0 += 256) {
      const int c = c0 + tid;
      if (c < C) {
        float hi[8], lo[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          const float v = h < H ? gu[h * C + c] : 0.f;
          const T t = (T)v;
          hi[h] = (float)t;
          lo[h] = v - hi[h];
        }
        st16(s_uhi + c * 16, pack8<T>(hi));
        st16(s_ulo + c * 16, pack8<T>(lo));
      }
    }
    for (int c = tid; c < C; c += 256) {
      s_b[c] = p.b[(long long)img * C + c];
      *reinterpret_cast<T*>(s_ahi + 8 * AST + c * 2) = (T)0.0f;
      *reinterpret_cast<T*>(s_alo + 8 * AST + c * 2) = (T)0.0f;
    }
    if (ln2) {
      const float* lg = (blockIdx.y & 1) ? p.ln_gamma[1] : p.ln_gamma[0];
      const float* lb = (blockIdx.y & 1) ? p.ln_beta[1] : p.ln_beta[0];
      for (int c = tid; c < C; c += 256) { s_lg[c] = lg[c]; s_lb[c] = lb[c]; }
    }
    if (tid < 8) {
      *reinterpret_cast<T*>(s_uhi + C * 16 + tid * 2) = (T)0.0f;
      *reinterpret_cast<T*>(s_ulo + C * 16 + tid * 2) = (T)0.0f;
    }
  }
  __syncthreads();
  {                                                 // sums of the numbers the matrix product will use: 32 threads per head
    const int h = tid >> 5, l = tid & 31;
    float sum = 0.f;
    for (int c = l; c < C; c += 32)
      sum += (float)*reinterpret_cast<const T*>(s_ahi + h * AST + c * 2) + (float)*reinterpret_cast<const T*>(s_alo + h * AST + c * 2);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (l == 0) s_asum[h] = sum;
  }
  __syncthreads();
  float a_sum[4], cc[4];                            // the lane's four heads: 4 g4 .. 4 g4 + 3 (g4 < 2)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int h = 4 * g4 + j;
    a_sum[j] = h < 8 ? s_asum[h] : 0.f;
    cc[j] = h < H ? p.c[img * H + h] : 0.f;
  }
  const long long ximg = p.x_img_stride ? p.x_img_stride : (long long)p.rows_per_image * C;
  const long long oimg = p.out_img_stride ? p.out_img_stride : (long long)p.rows_per_image * C;
  const T* __restrict__ xb = reinterpret_cast<const T*>(p.x) + img * ximg;
  T* __restrict__ ob = reinterpret_cast<T*>(p.out) + img * oimg;
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(p.rows_per_image, r_begin + rows_per_block);
  const float inv_c = 1.0f / (float)C;
  const int a_off = min(r16, 8) * AST + 16 * g4;    // this lane's table row of the logits product (byte offset)
  // update row r16 of half 0 is channel 8 (r16 >> 2) + (r16 & 3) of the 32-channel block; lanes beyond k = 7 read zeros
  const int u_off = g4 == 0 ? (8 * (r16 >> 2) + (r16 & 3)) * 16 : C * 16;
  const int u_step = g4 == 0 ? 16 : 0;              // (bytes per channel)

  for (int r0 = r_begin + 16 * wave; r0 < r_end; r0 += 64) {
    const int row = r0 + r16;
    const bool live = row < r_end;
    const T* __restrict__ xr = xb + (long long)(live ? row : r_end - 1) * C + 8 * g4;
    unsigned xf[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const u32x4 v = ld16(xr + 32 * f);
#pragma unroll
      for (int e = 0; e < 4; ++e) xf[f][e] = v[e];
    }
    auto frag = [&](int f) { return __builtin_bit_cast(frag_t, u32x4{xf[f][0], xf[f][1], xf[f][2], xf[f][3]}); };
    // ---- logits + LayerNorm sums --------------------------------------------------------------------------------
    f32x4 lg = f32x4{0.f, 0.f, 0.f, 0.f};
    float sx = 0.f, sxx = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const frag_t x8 = frag(f);
      const frag_t ah = __builtin_bit_cast(frag_t, ld16(s_ahi + a_off + 64 * f));
      const frag_t al = __builtin_bit_cast(frag_t, ld16(s_alo + a_off + 64 * f));
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = (float)x8[j]; sx += v; sxx += v * v; }
      lg = mfma16(ah, x8, lg);
      lg = mfma16(al, x8, lg);
    }
    // (the packed rows are what stays in registers: without the pins the compiler keeps the unpacked floats of the
    //  statistics alive for the final add, three times the registers)
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(xf[f][e]));
    sx += __shfl_xor(sx, 16, 64); sx += __shfl_xor(sx, 32, 64);            // the four k-chunk lanes of a token
    sxx += __shfl_xor(sxx, 16, 64); sxx += __shfl_xor(sxx, 32, 64);
    const float mean = sx * inv_c;
    float var = sxx * inv_c - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + p.eps);
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z = rstd * (lg[j] - mean * a_sum[j]) + cc[j];
      g[j] = 1.0f / (1.0f + __expf(-z));
    }
    // gates of heads 4..7 live 16 lanes further on: bring them to the g4 = 0 lanes, which supply k = 0..7
    float gh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) gh[j] = __shfl_down(g[j], 16, 64);
    frag_t gf, gl;                                  // gates and their rounding remainders (the product stays ~16-bit)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gf[j] = g4 == 0 ? (T)g[j] : (T)0.0f;
      gf[4 + j] = g4 == 0 ? (T)gh[j] : (T)0.0f;
      gl[j] = g4 == 0 ? (T)(g[j] - (float)gf[j]) : (T)0.0f;
      gl[4 + j] = g4 == 0 ? (T)(gh[j] - (float)gf[4 + j]) : (T)0.0f;
    }
    // ---- update, final add, store ---------------------------------------------------------------------------------
    T* __restrict__ orow = ob + (long long)row * C + 8 * g4;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      f32x4 d[2];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const frag_t uh = __builtin_bit_cast(frag_t, ld16(s_uhi + u_off + (32 * f + 4 * s_) * u_step));
        const frag_t ul = __builtin_bit_cast(frag_t, ld16(s_ulo + u_off + (32 * f + 4 * s_) * u_step));
        d[s_] = f32x4{0.f, 0.f, 0.f, 0.f};
        d[s_] = mfma16(uh, gf, d[s_]);
        d[s_] = mfma16(ul, gf, d[s_]);
        d[s_] = mfma16(uh, gl, d[s_]);
      }
      const frag_t x8 = frag(f);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_b + 32 * f + 8 * g4);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(s_b + 32 * f + 8 * g4 + 4);
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (float)x8[j] + b0[j] + d[0][j];
        o[4 + j] = (float)x8[4 + j] + b1[j] + d[1][j];
      }
      const u32x4 packed = pack8<T>(o);
      if (live) st16(orow + 32 * f, packed);
#pragma unroll
      for (int e = 0; e < 4; ++e) xf[f][e] = packed[e];          // the stored (rounded) row: what a LayerNorm launch would read
    }
    if (!ln2) continue;
    // ---- second result: LayerNorm of the stored row (mean, variance about the mean: layernorm_kernel's arithmetic) ------
    float s1 = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const frag_t x8 = frag(f);
#pragma unroll
      for (int j = 0; j < 8; ++j) s1 += (float)x8[j];
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    const float mean2 = s1 * inv_c;
    float q2 = 0.f;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const frag_t x8 = frag(f);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float dd = (float)x8[j] - mean2; q2 += dd * dd; }
    }
    q2 += __shfl_xor(q2, 16, 64); q2 += __shfl_xor(q2, 32, 64);
    const float rstd2 = rsqrtf(q2 * inv_c + p.ln_eps);
    T* __restrict__ lrow = reinterpret_cast<T*>((img & 1) ? p.ln_out[1] : p.ln_out[0]) +
                           ((long long)(img >> 1) * p.rows_per_image + row) * C + 8 * g4;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const frag_t x8 = frag(f);
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(s_lg + 32 * f + 8 * g4), g1 = *reinterpret_cast<const f32x4*>(s_lg + 32 * f + 8 * g4 + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(s_lb + 32 * f + 8 * g4), h1 = *reinterpret_cast<const f32x4*>(s_lb + 32 * f + 8 * g4 + 4);
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = ((float)x8[j] - mean2) * rstd2 * g0[j] + h0[j];
        o[4 + j] = ((float)x8[4 + j] - mean2) * rstd2 * g1[j] + h1[j];
      }
      if (live) st16(lrow + 32 * f, pack8<T>(o));
    }
  }
}

// fp32 x fp32 -> fp32 small GEMM (x [m][k], w [n][k]): 16 x 16 output tile per block (one output per thread, so that the
// few rows of these folds still spread over hundreds of blocks), k in LDS slabs of 32, plain fp32 FMA chains in k order.
// For the per-run context folds only.
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ out, int m,
                                                         int n, int k, int xs, int ws, int os) {
  __shared__ float lx[16][33], lw[16][33];
  const int tid = threadIdx.x, r = tid >> 4, c = tid & 15;
  const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16;
  float acc = 0.f;
  for (int k0 = 0; k0 < k; k0 += 32) {
    for (int i = tid; i < 16 * 32; i += 256) {
      const int rr = i >> 5, kk = i & 31;
      lx[rr][kk] = (m0 + rr < m && k0 + kk < k) ? x[(long long)(m0 + rr) * xs + k0 + kk] : 0.f;
      lw[rr][kk] = (n0 + rr < n && k0 + kk < k) ? w[(long long)(n0 + rr) * ws + k0 + kk] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) acc = __builtin_fmaf(lx[r][kk], lw[c][kk], acc);
    __syncthreads();
  }
  if (m0 + r < m && n0 + c < n) out[(long long)(m0 + r) * os + n0 + c] = acc + (bias ? bias[n0 + c] : 0.f);
}
}  // namespace mobi

using namespace mobi;
#define ST(stream) reinterpret_cast<hipStream_t>(stream)
#define DT_OK(d) ((d) == MOBI_F16 || (d) == MOBI_BF16)

extern "C" int mobi_skinny_linear(const mobi_skinny_linear_params* p, void* stream) {
  if (!p || !p->x || !p->weight || !p->out) return MOBI_ERR_ARG;
  if (!DT_OK(p->dtype) || p->m <= 0 || p->m > 16 || p->n <= 0 || p->k <= 0) return MOBI_ERR_ARG;
  if ((p->k & 7) || (p->x_row_stride & 3)) return MOBI_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p->x) | reinterpret_cast<uintptr_t>(p->weight)) & 15) return MOBI_ERR_ALIGN;
  const unsigned blocks = (unsigned)((p->n + 15) / 16);
  if ((p->k & 31) == 0 && mobi::tuning().skinny_mfma != 0) {
    if (p->n >= 8192) {                               // 64 columns per block: at least 128 blocks
      const unsigned wblocks = (unsigned)((p->n + 63) / 64);
      if (p->dtype == MOBI_F16) hipLaunchKernelGGL((skinny_linear_mfma_kernel<f16_t, true>), dim3(wblocks), dim3(256), 0, ST(stream), *p);
      else hipLaunchKernelGGL((skinny_linear_mfma_kernel<bf16_t, true>), dim3(wblocks), dim3(256), 0, ST(stream), *p);
    } else {
      if (p->dtype == MOBI_F16) hipLaunchKernelGGL((skinny_linear_mfma_kernel<f16_t, false>), dim3(blocks), dim3(256), 0, ST(stream), *p);
      else hipLaunchKernelGGL((skinny_linear_mfma_kernel<bf16_t, false>), dim3(blocks), dim3(256), 0, ST(stream), *p);
    }
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  if (p->dtype == MOBI_F16) hipLaunchKernelGGL((skinny_linear_kernel<f16_t>), dim3(blocks), dim3(256), 0, ST(stream), *p);
  else hipLaunchKernelGGL((skinny_linear_kernel<bf16_t>), dim3(blocks), dim3(256), 0, ST(stream), *p);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_linear_f32(const float* x, const float* w, const float* bias, float* out, int32_t m, int32_t n, int32_t k,
                               int32_t x_stride, int32_t w_stride, int32_t out_stride, void* stream) {
  if (!x || !w || !out || m <= 0 || n <= 0 || k <= 0) return MOBI_ERR_ARG;
  if (x_stride < k || w_stride < k || out_stride < n) return MOBI_ERR_ARG;
  if ((m + 15) / 16 > 65535) return MOBI_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(linear_f32_kernel, dim3((n + 15) / 16, (m + 15) / 16), dim3(256), 0, ST(stream), x, w, bias, out, m, n,
                     k, x_stride, w_stride, out_stride);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_timestep_embedding(const int64_t* t, const float* freqs, float* out, int32_t n, int32_t half,
                                       void* stream) {
  if (!t || !freqs || !out || n <= 0 || half <= 0) return MOBI_ERR_ARG;
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((long long)n * half, 256)), dim3(256), 0, ST(stream),
                     t, freqs, out, n, half);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_ctx_attention(const mobi_ctx_attention_params* p, void* stream) {
  if (!p || !p->q || !p->out || !p->k || !p->v) return MOBI_ERR_ARG;
  if (!DT_OK(p->dtype) || p->images <= 0 || p->heads <= 0 || p->tq <= 0) return MOBI_ERR_ARG;
  if (p->tk <= 0 || p->tk > 8 || p->dh <= 0 || (p->dh & 7)) return MOBI_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p->q) | reinterpret_cast<uintptr_t>(p->out)) & 15) return MOBI_ERR_ALIGN;
  const long long total = (long long)p->images * p->tq * p->heads;
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (p->dtype == MOBI_F16) hipLaunchKernelGGL((ctx_attention_kernel<f16_t>), dim3(blocks), dim3(256), 0, ST(stream), *p);
  else hipLaunchKernelGGL((ctx_attention_kernel<bf16_t>), dim3(blocks), dim3(256), 0, ST(stream), *p);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

// the register kernel's launches: C = 320 / 640 and enough token rows to spread its 64-row blocks over the chip (measured,
// tools/tka_lab.py --nusc256: 8 x 256 tokens at C = 640: 11.3-12.7 us against 8.4 us on the vector-ALU kernel; 8 x 1024 at C = 320:
// 8.9 against 12.3)
static int tka_regs_launch(int32_t channels, int64_t total_rows) {
  const int tka = mobi::tuning().tka_mfma;
  if (tka == 0 || tka == 1 || tka == 2) return 0;
  return (channels == 320 && total_rows >= 1024) || (channels == 640 && total_rows >= 4096);
}
// the second result (ln_out) is written by the register kernel and by the vector-ALU kernel, i.e. by every launch of the
// default routing; not by the LDS-tile kernel the A/B settings MOBI_TKA_MFMA=1 / 2 route C <= 640 / 320 to
extern "C" int mobi_two_key_adapter_fuses_ln(int32_t channels, int64_t total_rows) {
  if (channels <= 0 || (channels & 7) || channels > 1536 || total_rows <= 0) return 0;
  const int tka = mobi::tuning().tka_mfma;
  if ((tka == 1 || tka == 2) && (channels & 31) == 0 && channels <= (tka == 1 ? 640 : 320)) return 0;
  return 1;
}

extern "C" int mobi_two_key_adapter(const mobi_two_key_adapter_params* p, void* stream) {
  using namespace mobi;
  if (!p || !p->x || !p->out || !p->a || !p->a_sum || !p->c || !p->u || !p->b) return MOBI_ERR_ARG;
  if (!DT_OK(p->dtype) || p->images <= 0 || p->images > 65535 || p->rows_per_image <= 0) return MOBI_ERR_ARG;
  if (p->heads <= 0 || p->heads > 8 || p->channels <= 0 || (p->channels & 7) || p->channels > 1536) return MOBI_ERR_UNSUPPORTED;
  if ((p->x_img_stride & 7) || (p->out_img_stride & 7)) return MOBI_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(p->x) | reinterpret_cast<uintptr_t>(p->out) | reinterpret_cast<uintptr_t>(p->a) |
       reinterpret_cast<uintptr_t>(p->u) | reinterpret_cast<uintptr_t>(p->b)) & 15) return MOBI_ERR_ALIGN;
  // (measured, tools/kbench.py tka: C = 320 at 64 x 64 x 16 37.0 vs 67.3 us; at C = 640 its 127 KB of LDS leave one block
  //  per CU and it loses, 39.8 vs 36.8 us: the vector-ALU kernel keeps the wider levels; MOBI_TKA_MFMA=1 forces it to 640)
  const int tka = mobi::tuning().tka_mfma;          // unset: token rows in registers; 2 / 1: the LDS-tile kernel (to 320 / 640); 0: VALU
  if (p->ln_out[0]) {
    if (!mobi_two_key_adapter_fuses_ln(p->channels, (int64_t)p->images * p->rows_per_image)) return MOBI_ERR_UNSUPPORTED;
    if (!p->ln_out[1] || !p->ln_gamma[0] || !p->ln_gamma[1] || !p->ln_beta[0] || !p->ln_beta[1] || (p->images & 1)) return MOBI_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(p->ln_out[0]) | reinterpret_cast<uintptr_t>(p->ln_out[1])) & 15) return MOBI_ERR_ALIGN;
  }
  // (C = 1280 stays on the vector-ALU kernel: 16 x 16 and 8 x 8 tokens per image are a fixed cost of table staging plus
  //  one tile per wave either way -- 20.8 us there, 21.5 us as a 40-block unrolled register kernel, profiles/r03_tka_lab.txt)
  if (tka_regs_launch(p->channels, (int64_t)p->images * p->rows_per_image)) {
    // 16 rows per wave and tile, 64 per block at least; about 512 blocks over the launch (the tables are staged per block)
    long long rows = ((long long)p->rows_per_image * p->images + 511) / 512;
    rows = (rows + 15) / 16 * 16;
    if (rows < 64) rows = 64;
    if (mobi::tuning().tka_rows > 0) rows = (mobi::tuning().tka_rows + 15) / 16 * 16;
    if (rows > p->rows_per_image) rows = (p->rows_per_image + 15) / 16 * 16;
    const dim3 g2((unsigned)((p->rows_per_image + rows - 1) / rows), (unsigned)p->images);
#define MOBI_TKR(T_, NF_) hipLaunchKernelGGL((two_key_adapter_regs_kernel<T_, NF_>), g2, dim3(256), 0, ST(stream), *p, (int)rows)
    const int NF = p->channels / 32;
    if (p->dtype == MOBI_F16) { if (NF == 10) MOBI_TKR(f16_t, 10); else MOBI_TKR(f16_t, 20); }
    else                      { if (NF == 10) MOBI_TKR(bf16_t, 10); else MOBI_TKR(bf16_t, 20); }
#undef MOBI_TKR
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  if ((tka == 1 || tka == 2) && (p->channels & 31) == 0 && p->channels <= (tka == 1 ? 640 : 320)) {      // A/B only
    // matrix-core kernel: blocks of 64 rows (16 per wave) x k; about 512 blocks over the launch (tables re-staged per block)
    const int C = p->channels;
    long long rows = ((long long)p->rows_per_image * p->images + 511) / 512;
    rows = (rows + 63) / 64 * 64;
    if (rows > p->rows_per_image) rows = (p->rows_per_image + 63) / 64 * 64;
    const dim3 g2((unsigned)((p->rows_per_image + rows - 1) / rows), (unsigned)p->images);
#define MOBI_TKM(T_, CM_) hipLaunchKernelGGL((two_key_adapter_mfma_kernel<T_, CM_>), g2, dim3(256), 0, ST(stream), *p, (int)rows)
    if (p->dtype == MOBI_F16) { if (C <= 320) MOBI_TKM(f16_t, 320); else MOBI_TKM(f16_t, 640); }
    else                      { if (C <= 320) MOBI_TKM(bf16_t, 320); else MOBI_TKM(bf16_t, 640); }
#undef MOBI_TKM
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  // rows per block: about 1024 blocks over the whole launch (four 35-KB-LDS blocks per CU), at least 8 rows (the
  // tables are re-staged per block)
  // (beyond 1024 channels a block's tables take the CU's LDS: one round of 256 blocks -- 17.0 vs 20.6 us at 16 x 256 tokens, C = 1280)
  const long long target_blocks = p->channels > 1024 ? 256 : 1024;
  long long rpb = ((long long)p->rows_per_image * p->images + target_blocks - 1) / target_blocks;
  rpb = (rpb + 3) / 4 * 4;
  if (rpb < 8) rpb = 8;
  if (mobi::tuning().tka_rows > 0) rpb = (mobi::tuning().tka_rows + 3) / 4 * 4;
  if (rpb > p->rows_per_image) rpb = (p->rows_per_image + 3) / 4 * 4;
  const dim3 grid((unsigned)((p->rows_per_image + rpb - 1) / rpb), (unsigned)p->images);
  const int V = p->channels >> 3;
#define MOBI_TKA(T_, MV_) hipLaunchKernelGGL((two_key_adapter_kernel<T_, MV_>), grid, dim3(256), 0, ST(stream), *p, (int)rpb)
  if (p->dtype == MOBI_F16) { if (V <= 64) MOBI_TKA(f16_t, 1); else if (V <= 128) MOBI_TKA(f16_t, 2); else MOBI_TKA(f16_t, 3); }
  else { if (V <= 64) MOBI_TKA(bf16_t, 1); else if (V <= 128) MOBI_TKA(bf16_t, 2); else MOBI_TKA(bf16_t, 3); }
#undef MOBI_TKA
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_layernorm_rows_f32(const float* x, const float* gamma, const float* beta, float* out, int32_t rows,
                                       int32_t cols, int32_t x_stride, int32_t out_stride, float eps, void* stream) {
  if (!x || !gamma || !beta || !out || rows <= 0 || cols <= 0 || x_stride < cols || out_stride < cols) return MOBI_ERR_ARG;
  hipLaunchKernelGGL(layernorm_rows_f32_kernel, dim3((rows + 3) / 4), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x,
                     gamma, beta, out, rows, cols, x_stride, out_stride, eps);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_quick_gelu(const void* src, void* out, int64_t n, int32_t dtype, void* stream) {
  if (!src || !out || n <= 0 || (n & 7) || !DT_OK(dtype)) return MOBI_ERR_ARG;
  const long long vecs = n >> 3;
  long long blocks = (vecs + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == MOBI_F16)
    hipLaunchKernelGGL((quick_gelu_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const f16_t*>(src), reinterpret_cast<f16_t*>(out), vecs);
  else
    hipLaunchKernelGGL((quick_gelu_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(src), reinterpret_cast<bf16_t*>(out), vecs);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_softmax_rows(const float* src, void* out, int64_t rows, int32_t cols, int32_t dtype, void* stream) {
  if (!src || !out || rows <= 0 || cols <= 0 || !DT_OK(dtype)) return MOBI_ERR_ARG;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  if (dtype == MOBI_F16) hipLaunchKernelGGL((softmax_rows_kernel<f16_t>), dim3(blocks), dim3(256), 0, ST(stream), src, (f16_t*)out, (long long)rows, cols);
  else hipLaunchKernelGGL((softmax_rows_kernel<bf16_t>), dim3(blocks), dim3(256), 0, ST(stream), src, (bf16_t*)out, (long long)rows, cols);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_conv_small_cin(const mobi_conv_small_cin_params* p, void* stream) {
  if (!p || !p->src[0] || !p->weight || !p->out) return MOBI_ERR_ARG;
  if (!DT_OK(p->dtype) || p->batch <= 0 || p->h <= 0 || p->w <= 0 || p->cout <= 0) return MOBI_ERR_ARG;
  const int cin = p->c[0] + p->c[1] + p->c[2];
  if (cin <= 0 || cin > 16 || p->kh <= 0 || p->kw <= 0) return MOBI_ERR_UNSUPPORTED;
  if (!p->out_f32_nchw && (p->cout & 7)) return MOBI_ERR_UNSUPPORTED;
  const long long total = (long long)p->batch * p->h * p->w * ((p->cout + 7) / 8);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (p->dtype == MOBI_F16) hipLaunchKernelGGL((conv_small_cin_kernel<f16_t>), dim3(blocks), dim3(256), 0, ST(stream), *p);
  else hipLaunchKernelGGL((conv_small_cin_kernel<bf16_t>), dim3(blocks), dim3(256), 0, ST(stream), *p);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_conv_small_cout(const mobi_conv_small_cout_params* p, void* stream) {
  if (!p || !p->src || !p->weight || !p->out) return MOBI_ERR_ARG;
  if (!DT_OK(p->dtype) || p->batch <= 0 || p->h <= 0 || p->w <= 0) return MOBI_ERR_ARG;
  if (p->cout <= 0 || p->cout > 8 || p->cin <= 0 || (p->cin & 7)) return MOBI_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p->src) | reinterpret_cast<uintptr_t>(p->weight)) & 15) return MOBI_ERR_ALIGN;
  const long long pixels = (long long)p->batch * p->h * p->w;
  if ((p->cin == 320 || p->cin == 128 || p->cin == 256) && (p->w & 15) == 0 && p->kh * p->kw <= 9 && mobi::tuning().cout_mfma != 0) {
    // matrix-core form: 16-pixel tiles, about 512 blocks of four waves over the launch (the weights are staged per block)
    const long long tiles = pixels / 16;
    long long tpw = (tiles + 2047) / 2048;
    if (tpw < 1) tpw = 1;
    const unsigned blocks = (unsigned)((tiles + 4 * tpw - 1) / (4 * tpw));
    const size_t lds = (size_t)p->kh * p->kw * (p->cin / 32) * 9 * 64;
#define MOBI_CSM(T_, KS_) \
  hipLaunchKernelGGL((conv_small_cout_mfma_kernel<T_, KS_>), dim3(blocks), dim3(256), lds, ST(stream), *p, (int)tpw)   /* < 64 KB */
    // (256: the lidar decoder's output convolution on a hi | lo pair of its 128 channels, weights duplicated -- model.py's precise tail)
    if (p->dtype == MOBI_F16) { if (p->cin == 320) MOBI_CSM(f16_t, 10); else if (p->cin == 256) MOBI_CSM(f16_t, 8); else MOBI_CSM(f16_t, 4); }
    else                      { if (p->cin == 320) MOBI_CSM(bf16_t, 10); else if (p->cin == 256) MOBI_CSM(bf16_t, 8); else MOBI_CSM(bf16_t, 4); }
#undef MOBI_CSM
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  const unsigned blocks = (unsigned)((pixels + 3) / 4);
  if (p->dtype == MOBI_F16) hipLaunchKernelGGL((conv_small_cout_kernel<f16_t>), dim3(blocks), dim3(256), 0, ST(stream), *p);
  else hipLaunchKernelGGL((conv_small_cout_kernel<bf16_t>), dim3(blocks), dim3(256), 0, ST(stream), *p);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_pack_nchw_sources(const float* s0, const float* s1, const float* s2, int32_t c0, int32_t c1,
                                      int32_t c2, int32_t batch, int32_t hw, int32_t c_pad, void* out, int32_t dtype,
                                      void* stream) {
  if (!s0 || !out || c0 <= 0 || c1 < 0 || c2 < 0 || (c1 > 0 && !s1) || (c2 > 0 && !s2)) return MOBI_ERR_ARG;
  if (batch <= 0 || hw <= 0 || !DT_OK(dtype)) return MOBI_ERR_ARG;
  if ((c_pad & 7) || c_pad < c0 + c1 + c2) return MOBI_ERR_UNSUPPORTED;
  const unsigned g = grid_for((long long)batch * hw * (c_pad >> 3), 256, 8192);
  if (dtype == MOBI_F16) hipLaunchKernelGGL((pack_sources_kernel<f16_t>), dim3(g), dim3(256), 0, ST(stream), s0, s1, s2, c0, c1, c2, batch, hw, c_pad, (f16_t*)out);
  else hipLaunchKernelGGL((pack_sources_kernel<bf16_t>), dim3(g), dim3(256), 0, ST(stream), s0, s1, s2, c0, c1, c2, batch, hw, c_pad, (bf16_t*)out);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_nchw_f32_to_nhwc(const float* src, void* out, int32_t batch, int32_t c, int32_t hw, int32_t dtype,
                                     void* stream) {
  if (!src || !out || batch <= 0 || c <= 0 || hw <= 0 || !DT_OK(dtype)) return MOBI_ERR_ARG;
  const unsigned g = grid_for((long long)batch * c * hw, 256, 8192);
  if (dtype == MOBI_F16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<f16_t>), dim3(g), dim3(256), 0, ST(stream), src, (f16_t*)out, batch, c, hw);
  else hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t>), dim3(g), dim3(256), 0, ST(stream), src, (bf16_t*)out, batch, c, hw);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_nhwc_to_nchw_f32(const void* src, float* out, int32_t batch, int32_t c, int32_t hw, int32_t dtype,
                                     void* stream) {
  if (!src || !out || batch <= 0 || c <= 0 || hw <= 0 || !DT_OK(dtype)) return MOBI_ERR_ARG;
  const unsigned g = grid_for((long long)batch * c * hw, 256, 8192);
  if (dtype == MOBI_F16) hipLaunchKernelGGL((nhwc_to_nchw_kernel<f16_t>), dim3(g), dim3(256), 0, ST(stream), (const f16_t*)src, out, batch, c, hw);
  else hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t>), dim3(g), dim3(256), 0, ST(stream), (const bf16_t*)src, out, batch, c, hw);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

// ---------------------------------------------------------------------------------------------------------
// fp32 residual trunk (the VAE decoder's precise mode): trunk (f32, in place) += inc (T); x16 = T(trunk).  With inc == NULL
// only the conversion.  8 elements per thread, 16-byte accesses.
// ---------------------------------------------------------------------------------------------------------
namespace mobi {
template <typename T>
__global__ __launch_bounds__(256) void trunk_add_kernel(float* __restrict__ trunk, const T* __restrict__ inc, T* __restrict__ x16, long long vecs) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vecs; i += (long long)gridDim.x * 256) {
    f32x4 a = *reinterpret_cast<const f32x4*>(trunk + 8 * i), b = *reinterpret_cast<const f32x4*>(trunk + 8 * i + 4);
    float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    if (inc) {
      float g[8];
      unpack8<T>(ld16(inc + 8 * i), g);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] += g[j];
      *reinterpret_cast<f32x4*>(trunk + 8 * i) = f32x4{f[0], f[1], f[2], f[3]};
      *reinterpret_cast<f32x4*>(trunk + 8 * i + 4) = f32x4{f[4], f[5], f[6], f[7]};
    }
    st16(x16 + 8 * i, pack8<T>(f));
  }
}
}  // namespace mobi

// fp32 rows -> their hi | lo (| hi) split in the storage type (the operand form of mobi_groupnorm's out_mode 1 / 3 for tensors no
// GroupNorm stands in front of: the inputs of the decoders' upsampling convolutions and 1 x 1 shortcuts).  8 channels per thread.
namespace mobi {
template <typename T>
__global__ __launch_bounds__(256) void split_f32_kernel(const float* __restrict__ x, T* __restrict__ out, long long vecs, int C, int parts) {
  const int V = C >> 3;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vecs; i += (long long)gridDim.x * 256) {
    const long long row = i / V;
    const int c = (int)(i - row * V) * 8;
    float f[8], hi[8], lo[8];
    ld8f(x + row * C + c, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { hi[j] = (float)(T)f[j]; lo[j] = f[j] - hi[j]; }
    T* o = out + row * (long long)(parts * C) + c;
    const u32x4 h8 = pack8<T>(hi);
    st16(o, h8);
    st16(o + C, pack8<T>(lo));
    if (parts == 3) st16(o + 2 * C, h8);
  }
}
}  // namespace mobi

extern "C" int mobi_split_f32(const float* x, void* out, int64_t rows, int32_t channels, int32_t parts, int32_t dtype, void* stream) {
  using namespace mobi;
  if (!x || !out || rows <= 0 || channels <= 0 || (channels & 7) || (parts != 2 && parts != 3) || !DT_OK(dtype)) return MOBI_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 15) return MOBI_ERR_ALIGN;
  const long long vecs = rows * (channels >> 3);
  const unsigned g = grid_for(vecs, 256, 16384);
  if (dtype == MOBI_F16) hipLaunchKernelGGL((split_f32_kernel<f16_t>), dim3(g), dim3(256), 0, ST(stream), x, (f16_t*)out, vecs, channels, parts);
  else hipLaunchKernelGGL((split_f32_kernel<bf16_t>), dim3(g), dim3(256), 0, ST(stream), x, (bf16_t*)out, vecs, channels, parts);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_trunk_add(float* trunk, const void* inc, void* x16, int64_t n, int32_t dtype, void* stream) {
  using namespace mobi;
  if (!trunk || !x16 || n <= 0 || (n & 7) || !DT_OK(dtype)) return MOBI_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(trunk) | reinterpret_cast<uintptr_t>(inc) | reinterpret_cast<uintptr_t>(x16)) & 15) return MOBI_ERR_ALIGN;
  const unsigned g = grid_for(n / 8, 256, 16384);
  if (dtype == MOBI_F16) hipLaunchKernelGGL((trunk_add_kernel<f16_t>), dim3(g), dim3(256), 0, ST(stream), trunk, (const f16_t*)inc, (f16_t*)x16, (long long)(n / 8));
  else hipLaunchKernelGGL((trunk_add_kernel<bf16_t>), dim3(g), dim3(256), 0, ST(stream), trunk, (const bf16_t*)inc, (bf16_t*)x16, (long long)(n / 8));
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}
