// Backward-pass kernels of the transformer block (the training step of the adapter parameters,
// ldm/models/diffusion/ddpm.py:356-370, 1616-1669 of the reference -- `cond_adapter*`, `cross_modal*` of every
// BasicTransformerBlock, ldm/modules/attention.py:197-266).  FIRST SLICE: correct and deterministic, not tuned -- fp32
// vector arithmetic on LDS tiles; the data-gradient and weight-gradient PRODUCTS of the linear layers run on mobi_igemm
// (dx = dy W: the layer's weight read as [in][out]; dW = dy^T x: both operands transposed by transpose_kernel, the token
// axis as k).  What is here: LayerNorm backward (dx, per-block partial d gamma / d beta), GEGLU forward / backward on the
// un-fused projection, attention backward (log-sum-exp and row dots, dQ, dK | dV) for any head dim <= 160 and any token
// counts (self / cross-modal attention; two context tokens for the bbox adapter), column sums (bias gradients), a 16-bit
// transpose, the reduction of per-block partials (fixed order: bit-reproducible).
#include "common.h"
#include "tuning.h"

namespace mobi {
namespace {

constexpr int BW_TILE = 32;          // attention backward: queries / keys per tile
constexpr int BW_DH = 160;           // largest head dim
constexpr int BW_LD = BW_DH + 1;     // LDS row pitch in floats (odd: conflict-free column walks)

template <typename T>
__device__ __forceinline__ void load_tile(float* dst, const T* src, long long row_stride, int rows_valid, int dh, int tid) {
  // dst [BW_TILE][BW_LD] <- src [rows][dh] (zero rows beyond rows_valid)
  for (int i = tid; i < BW_TILE * dh; i += 256) {
    const int r = i / dh, d = i - r * dh;
    dst[r * BW_LD + d] = r < rows_valid ? (float)src[(long long)r * row_stride + d] : 0.f;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// 16-bit transpose: src [rows][cols] (row stride in elements) -> out [cols][rows] dense
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ src, long long src_stride, T* __restrict__ out,
                                                        int rows, int cols) {
  __shared__ T tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    if (r0 + r < rows && c0 + c < cols) tile[r][c] = src[(long long)(r0 + r) * src_stride + c0 + c];
  }
  __syncthreads();
  for (int i = tid; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < rows && c0 + c < cols) out[(long long)(c0 + c) * rows + r0 + r] = tile[r][c];
  }
}

// W [n][k] (16-bit) -> the ring kernels' 1-KiB request images (mobi_igemm_params.weight_tiled): block (p, s) = rows 16 p .. 16 p + 15,
// k 32 s .. 32 s + 31, row r's 16-byte chunk c at slot c ^ P[(r >> 2) & 3], P = {0, 2, 3, 1}.  One 16-byte chunk per thread.
// (ops.tile_weights did this with torch.gather over an int64 index tensor: 11 ms of every training step, whose 432 adapter
// tensors are re-packed after each optimizer update.)
__global__ __launch_bounds__(256) void tile_weights_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int n, int k) {
  const long long chunks = (long long)n * (k >> 3);
  const int steps = k >> 5;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) {
    // destination chunk i = ((p * steps + s) * 16 + r) * 4 + slot
    const int slot = (int)(i & 3), r = (int)((i >> 2) & 15);
    const long long ps = i >> 6;
    const int s = (int)(ps % steps);
    const long long p = ps / steps;
    const int perm = (0x1320 >> (4 * ((r >> 2) & 3))) & 3;          // P[(r >> 2) & 3] of {0, 2, 3, 1}
    const int c = slot ^ perm;
    dst[i] = src[((p * 16 + r) * (long long)k + 32 * s + 8 * c) >> 3];
  }
}

// column sums of dy [rows][cols] (T) -> partial [gridDim.x][cols] fp32 (fixed order within a block)
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ dy, long long stride, float* __restrict__ partial,
                                                     long long rows, int cols, int rows_per_block) {
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  for (int c = threadIdx.x; c < cols; c += 256) {
    float s = 0.f;
    for (long long r = r0; r < r1; ++r) s += (float)dy[r * stride + c];
    partial[(long long)blockIdx.x * cols + c] = s;
  }
}

// out[i] = sum_b partial[b][i]: blockIdx.y owns a contiguous range of the nblk rows (ascending b inside it); gridDim.y == 1 writes the
// result, otherwise a second call sums the gridDim.y partial rows -- fixed order either way
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk,
                                                              long long len) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= len) return;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b0 = blockIdx.y * per, b1 = b0 + per < nblk ? b0 + per : nblk;
  float s = 0.f;
  for (int b = b0; b < b1; ++b) s += partial[(long long)b * len + i];
  out[(long long)blockIdx.y * len + i] = s;
}

static void reduce_partials(const float* partial, float* scratch, float* out, int nblk, long long len, hipStream_t st) {
  // scratch: >= 64 * len floats when nblk > 64
  const unsigned gx = (unsigned)((len + 255) / 256);
  if (nblk <= 64) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, partial, out, nblk, len);
    return;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 64), dim3(256), 0, st, partial, scratch, nblk, len);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, st, scratch, out, 64, len);
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm backward over the last axis (nn.LayerNorm, attention.py:213-223): one wave per row
//   xh = (x - mean) rstd,  g = dy gamma,  dx = rstd (g - mean(g) - xh mean(g xh)) (+ dx_add)
//   partial[block][0][c] = sum_rows dy xh,  partial[block][1][c] = sum_rows dy   (the block's rows, fixed order)
// MAXV = channels per lane (C <= 64 MAXV): 5 / 10 / 20 for the UNet's widths, 24 for anything else up to 1536 -- the loops are
// unrolled over it, and at C = 320 nineteen of twenty-four predicated iterations were the kernel's time
template <typename T, int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ x, long long x_stride, const T* __restrict__ dy,
                                                            long long dy_stride, const float* __restrict__ gamma, float eps,
                                                            const T* __restrict__ dx_add, T* __restrict__ dx, float* __restrict__ partial,
                                                            long long rows, int C, int rows_per_block) {
  __shared__ float red[4][2][64 * MAXV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float ag[MAXV], ab[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  const float inv_c = 1.0f / (float)C;
  for (long long r = r0 + wave; r < r1; r += 4) {
    float xv[MAXV], gv[MAXV], dv[MAXV];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      xv[i] = c < C ? (float)x[r * x_stride + c] : 0.f;
      dv[i] = c < C ? (float)dy[r * dy_stride + c] : 0.f;
      s1 += xv[i];
    }
    const float mean = wave_sum(s1) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      const float d = c < C ? xv[i] - mean : 0.f;
      q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      const float xh = c < C ? (xv[i] - mean) * rstd : 0.f;
      gv[i] = c < C ? dv[i] * gamma[c] : 0.f;
      sg += gv[i];
      sgx += gv[i] * xh;
      ag[i] += dv[i] * xh;
      ab[i] += dv[i];
      xv[i] = xh;
    }
    const float mg = wave_sum(sg) * inv_c, mgx = wave_sum(sgx) * inv_c;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = lane + 64 * i;
      if (c < C) {
        float v = rstd * (gv[i] - mg - xv[i] * mgx);
        if (dx_add) v += (float)dx_add[r * (long long)C + c];
        dx[r * (long long)C + c] = (T)v;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = lane + 64 * i;
    if (c < C) { red[wave][0][c] = ag[i]; red[wave][1][c] = ab[i]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    partial[((long long)blockIdx.x * 2 + 0) * C + c] = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
    partial[((long long)blockIdx.x * 2 + 1) * C + c] = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// GEGLU on the UN-fused projection (attention.py:38-46): pre [rows][2 inner] = [value | gate];  h = value gelu_erf(gate)
template <typename T>
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const T* __restrict__ pre, T* __restrict__ h, long long rows, int inner) {
  const long long n = rows * inner;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / inner;
    const int c = (int)(i - r * inner);
    const float v = (float)pre[r * 2 * inner + c], g = (float)pre[r * 2 * inner + inner + c];
    h[i] = (T)(v * gelu_erf_f(g));
  }
}
// d value = dh gelu(g);  d gate = dh value gelu'(g),  gelu'(g) = Phi(g) + g phi(g)
template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const T* __restrict__ pre, const T* __restrict__ dh, T* __restrict__ dpre,
                                                        long long rows, int inner) {
  const long long n = rows * inner;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const long long r = i / inner;
    const int c = (int)(i - r * inner);
    const float v = (float)pre[r * 2 * inner + c], g = (float)pre[r * 2 * inner + inner + c], d = (float)dh[i];
    const float cdf = 0.5f * (1.0f + erf_as_f(g * 0.70710678118654752440f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * g * g);
    dpre[r * 2 * inner + c] = (T)(d * g * cdf);
    dpre[r * 2 * inner + inner + c] = (T)(d * v * (cdf + g * pdf));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention backward (CrossAttention.forward, attention.py:171-194: softmax(q k^T scale) v per head).
//   pass 1 (per query tile): L = log-sum-exp of the scaled scores, D = do . o           -> fp32 [image][head][tq] each
//   pass 2 (per query tile): P = exp(s - L), dS = P (dO V^T - D) scale, dQ = dS K
//   pass 3 (per key tile):   dV = P^T dO,  dK = dS^T Q
// Tiles of 32 x 32 scores, operands as fp32 in LDS, a thread owns row t / 8, columns 4 (t % 8) .. + 3 of the score tile
// and rows t / 8, channels t % 8 + 8 i of the [32][dh] results.
struct AttnBwdArgs {
  const void *q, *k, *v, *o, *dout;
  long long q_is, q_rs, k_is, k_rs, v_is, v_rs, o_is, o_rs, do_is, do_rs;   // image / row strides, elements
  void *dq, *dk, *dv;                                                         // T [image][t][heads * dh] dense
  float *lse, *dvec;                                                          // fp32 [image][head][tq]
  int heads, dh, tq, tk;
  float scale;
  int exact_d;     // 1: D = sum_j P_ij dP_ij from the pass's own P and dP (fp32); 0: D = do . o with the STORED o (A/B)
};

// D, the row term of the softmax backward (dS = P (dP - D)): the textbook form do . o reads the output as it was STORED -- rounded
// to 16 bits.  When the values share a common component (every token the same offset: LayerNorm biases, smooth feature maps) that
// rounding is an error of D proportional to the offset, dS = P (dP - D) picks it up on EVERY key with weight P_ij, and the products
// with K then add it up instead of cancelling it: sum_j P_ij K_j is the keys' own common component.  Measured (tests/attn_bwd_err.py,
// bf16, offsets 3 / 10 of unit-variance keys and values): dQ 28 % / 315 % off, fp16 3.5 % / 39 %.  With D = sum_j P_ij dP_ij from the
// SAME P and dP the later passes compute, sum_j dS_ij = 0 holds to fp32 rounding and the offset cancels as it does in exact arithmetic.
// Cost: the statistics pass stages V as well and multiplies dO V^T (what the dQ pass does anyway).

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_stats_kernel(const AttnBwdArgs a) {
  __shared__ float sQ[BW_TILE * BW_LD], sK[BW_TILE * BW_LD], sO[BW_TILE * BW_LD], sV[BW_TILE * BW_LD];
  const int tid = threadIdx.x, r = tid >> 3, cg = tid & 7;
  const int q0 = blockIdx.x * BW_TILE, h = blockIdx.y, img = blockIdx.z;
  const int qv = min(BW_TILE, a.tq - q0);
  const T* qp = reinterpret_cast<const T*>(a.q) + img * a.q_is + (long long)q0 * a.q_rs + h * a.dh;
  load_tile(sQ, qp, a.q_rs, qv, a.dh, tid);
  if (a.exact_d)
    load_tile(sO, reinterpret_cast<const T*>(a.dout) + img * a.do_is + (long long)q0 * a.do_rs + h * a.dh, a.do_rs, qv, a.dh, tid);
  float m = -3.0e38f, l = 0.f, da = 0.f;           // da = sum_j exp(s_j - m) dP_j, rescaled with l
  for (int k0 = 0; k0 < a.tk; k0 += BW_TILE) {
    const int kv = min(BW_TILE, a.tk - k0);
    __syncthreads();
    load_tile(sK, reinterpret_cast<const T*>(a.k) + img * a.k_is + (long long)k0 * a.k_rs + h * a.dh, a.k_rs, kv, a.dh, tid);
    if (a.exact_d)
      load_tile(sV, reinterpret_cast<const T*>(a.v) + img * a.v_is + (long long)k0 * a.v_rs + h * a.dh, a.v_rs, kv, a.dh, tid);
    __syncthreads();
    float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < a.dh; ++d) {
      const float qd = sQ[r * BW_LD + d];
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += qd * sK[(4 * cg + j) * BW_LD + d];
    }
    if (a.exact_d)
      for (int d = 0; d < a.dh; ++d) {
        const float od = sO[r * BW_LD + d];
#pragma unroll
        for (int j = 0; j < 4; ++j) dp[j] += od * sV[(4 * cg + j) * BW_LD + d];
      }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (4 * cg + j < kv) {
        const float sv = s[j] * a.scale;
        const float mn = fmaxf(m, sv);
        const float c = __expf(m - mn), e = __expf(sv - mn);
        l = l * c + e;
        da = da * c + e * dp[j];
        m = mn;
      }
    }
  }
  // the eight threads of a row (consecutive lanes) combine their (m, l, da)
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) {
    const float m2 = __shfl_xor(m, o, 64), l2 = __shfl_xor(l, o, 64), d2 = __shfl_xor(da, o, 64);
    const float mn = fmaxf(m, m2);
    const float c1 = __expf(m - mn), c2 = __expf(m2 - mn);
    l = l * c1 + l2 * c2;
    da = da * c1 + d2 * c2;
    m = mn;
  }
  float dd = da / l;
  if (!a.exact_d) {                                // D = do . o over the head's channels (the stored output)
    const T* op = reinterpret_cast<const T*>(a.o) + img * a.o_is + (long long)(q0 + r) * a.o_rs + h * a.dh;
    const T* dp = reinterpret_cast<const T*>(a.dout) + img * a.do_is + (long long)(q0 + r) * a.do_rs + h * a.dh;
    dd = 0.f;
    if (r < qv)
      for (int d = cg; d < a.dh; d += 8) dd += (float)op[d] * (float)dp[d];
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) dd += __shfl_xor(dd, o, 64);
  }
  if (cg == 0 && r < qv) {
    const long long idx = ((long long)img * a.heads + h) * a.tq + q0 + r;
    a.lse[idx] = m + __logf(l);
    a.dvec[idx] = dd;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnBwdArgs a) {
  __shared__ float sQ[BW_TILE * BW_LD], sO[BW_TILE * BW_LD], sK[BW_TILE * BW_LD], sV[BW_TILE * BW_LD], sS[BW_TILE * 33];
  const int tid = threadIdx.x, r = tid >> 3, cg = tid & 7;
  const int q0 = blockIdx.x * BW_TILE, h = blockIdx.y, img = blockIdx.z;
  const int qv = min(BW_TILE, a.tq - q0);
  load_tile(sQ, reinterpret_cast<const T*>(a.q) + img * a.q_is + (long long)q0 * a.q_rs + h * a.dh, a.q_rs, qv, a.dh, tid);
  load_tile(sO, reinterpret_cast<const T*>(a.dout) + img * a.do_is + (long long)q0 * a.do_rs + h * a.dh, a.do_rs, qv, a.dh, tid);
  const long long sidx = ((long long)img * a.heads + h) * a.tq + q0 + r;
  const float L = r < qv ? a.lse[sidx] : 0.f, D = r < qv ? a.dvec[sidx] : 0.f;
  float acc[BW_DH / 8];
#pragma unroll
  for (int i = 0; i < BW_DH / 8; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < a.tk; k0 += BW_TILE) {
    const int kv = min(BW_TILE, a.tk - k0);
    __syncthreads();
    load_tile(sK, reinterpret_cast<const T*>(a.k) + img * a.k_is + (long long)k0 * a.k_rs + h * a.dh, a.k_rs, kv, a.dh, tid);
    load_tile(sV, reinterpret_cast<const T*>(a.v) + img * a.v_is + (long long)k0 * a.v_rs + h * a.dh, a.v_rs, kv, a.dh, tid);
    __syncthreads();
    float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < a.dh; ++d) {
      const float qd = sQ[r * BW_LD + d], od = sO[r * BW_LD + d];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[j] += qd * sK[(4 * cg + j) * BW_LD + d];
        dp[j] += od * sV[(4 * cg + j) * BW_LD + d];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p = (4 * cg + j < kv && r < qv) ? __expf(s[j] * a.scale - L) : 0.f;
      sS[r * 33 + 4 * cg + j] = p * (dp[j] - D) * a.scale;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BW_DH / 8; ++i) {
      const int d = cg + 8 * i;
      if (d < a.dh) {
        float v = acc[i];
        for (int c = 0; c < BW_TILE; ++c) v += sS[r * 33 + c] * sK[c * BW_LD + d];
        acc[i] = v;
      }
    }
  }
  if (r < qv) {
    T* out = reinterpret_cast<T*>(a.dq) + ((long long)img * a.tq + q0 + r) * (a.heads * a.dh) + h * a.dh;
#pragma unroll
    for (int i = 0; i < BW_DH / 8; ++i) {
      const int d = cg + 8 * i;
      if (d < a.dh) out[d] = (T)acc[i];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnBwdArgs a) {
  __shared__ float sQ[BW_TILE * BW_LD], sO[BW_TILE * BW_LD], sK[BW_TILE * BW_LD], sV[BW_TILE * BW_LD], sP[BW_TILE * 33], sS[BW_TILE * 33];
  __shared__ float sL[BW_TILE], sD[BW_TILE];
  const int tid = threadIdx.x, r = tid >> 3, cg = tid & 7;
  const int k0 = blockIdx.x * BW_TILE, h = blockIdx.y, img = blockIdx.z;
  const int kv = min(BW_TILE, a.tk - k0);
  load_tile(sK, reinterpret_cast<const T*>(a.k) + img * a.k_is + (long long)k0 * a.k_rs + h * a.dh, a.k_rs, kv, a.dh, tid);
  load_tile(sV, reinterpret_cast<const T*>(a.v) + img * a.v_is + (long long)k0 * a.v_rs + h * a.dh, a.v_rs, kv, a.dh, tid);
  float ak[BW_DH / 8], av[BW_DH / 8];              // this thread: key row r, channels cg + 8 i
#pragma unroll
  for (int i = 0; i < BW_DH / 8; ++i) { ak[i] = 0.f; av[i] = 0.f; }
  for (int q0 = 0; q0 < a.tq; q0 += BW_TILE) {
    const int qv = min(BW_TILE, a.tq - q0);
    __syncthreads();
    load_tile(sQ, reinterpret_cast<const T*>(a.q) + img * a.q_is + (long long)q0 * a.q_rs + h * a.dh, a.q_rs, qv, a.dh, tid);
    load_tile(sO, reinterpret_cast<const T*>(a.dout) + img * a.do_is + (long long)q0 * a.do_rs + h * a.dh, a.do_rs, qv, a.dh, tid);
    if (tid < BW_TILE) {
      const long long sidx = ((long long)img * a.heads + h) * a.tq + q0 + tid;
      sL[tid] = tid < qv ? a.lse[sidx] : 0.f;
      sD[tid] = tid < qv ? a.dvec[sidx] : 0.f;
    }
    __syncthreads();
    // score tile: this thread owns query row r, keys 4 cg .. + 3
    float s[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < a.dh; ++d) {
      const float qd = sQ[r * BW_LD + d], od = sO[r * BW_LD + d];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[j] += qd * sK[(4 * cg + j) * BW_LD + d];
        dp[j] += od * sV[(4 * cg + j) * BW_LD + d];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p = (4 * cg + j < kv && r < qv) ? __expf(s[j] * a.scale - sL[r]) : 0.f;
      sP[r * 33 + 4 * cg + j] = p;
      sS[r * 33 + 4 * cg + j] = p * (dp[j] - sD[r]) * a.scale;
    }
    __syncthreads();
    // accumulate: key row r (of this tile), channels cg + 8 i, over the 32 queries
#pragma unroll
    for (int i = 0; i < BW_DH / 8; ++i) {
      const int d = cg + 8 * i;
      if (d < a.dh) {
        float vk = ak[i], vv = av[i];
        for (int qq = 0; qq < BW_TILE; ++qq) {
          vv += sP[qq * 33 + r] * sO[qq * BW_LD + d];
          vk += sS[qq * 33 + r] * sQ[qq * BW_LD + d];
        }
        ak[i] = vk;
        av[i] = vv;
      }
    }
  }
  if (r < kv) {
    const long long off = ((long long)img * a.tk + k0 + r) * (a.heads * a.dh) + h * a.dh;
    T* outk = reinterpret_cast<T*>(a.dk) + off;
    T* outv = reinterpret_cast<T*>(a.dv) + off;
#pragma unroll
    for (int i = 0; i < BW_DH / 8; ++i) {
      const int d = cg + 8 * i;
      if (d < a.dh) { outk[d] = (T)ak[i]; outv[d] = (T)av[i]; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same two passes on the matrix cores (MFMA 32x32x16; head dims whose 16-padded width is 16 KD, KD in {1..5, 10}).
// Layouts as in the forward attention / feed-forward kernels: the wave's own 32 rows (queries in the dQ pass, keys in the
// dK | dV pass) are the B operand and stay in registers, the other side's 32-row tiles go through LDS as the A operand;
// D^T[row of A][own row] accumulates in fp32, and an accumulator turned into the next product's B operand keeps its own
// k order (k slot (h, j) of step s = row 16 s + 8 (j >> 2) + 4 h + (j & 3)), which the A side reads accordingly from a
// TRANSPOSED copy of the tile.
//   dQ pass, per 32-key tile:  S^T = K Q^T, dP^T = V dO^T (A = K / V rows);  dS^T = P^T (dP^T - D) scale;  dQ^T += K^T dS^T
//   dK | dV pass, per 32-query tile:  S = Q K^T, dP = dO V^T (A = Q / dO rows);  dV^T += dO^T P,  dK^T += Q^T dS
// Rows beyond the tensors are staged as zeros (and L = +1e30 for missing queries, so their P is 0); missing keys are masked in P.
constexpr int BM_PITCH_PAD = 8;      // row-major tiles: dhp + 8 elements per row

// Staging of a 32-row tile, split in two so that the NEXT tile's rows are in flight while the current tile is multiplied:
// bm_fetch requests 16-byte pieces (a thread: piece tid + 256 j of the tile's 32 x 2 KD pieces, zeros beyond the rows / the
// head width), bm_put writes them to LDS (row-major only: the transposed operands are READ transposed, bm_frag_t below -- the
// first form kept transposed copies, eight 2-byte LDS writes per piece).
// (The first form staged element by element and requested a tile only once the previous one was done: 128 exposed load
// latencies per 4096-row pass, 2.1 + 2.8 ms per launch at 64 x 64 x 16.)
template <int KD>
struct BmRegs {
  static constexpr int NP = (32 * 2 * KD + 255) / 256;
  u32x4 v[NP];
};
template <typename T, int KD>
__device__ __forceinline__ void bm_fetch(BmRegs<KD>& g, const T* src, long long row_stride, int rows_valid, int dh, int tid) {
#pragma unroll
  for (int j = 0; j < BmRegs<KD>::NP; ++j) {
    const int p = tid + 256 * j, r = p / (2 * KD), c = p - r * (2 * KD);
    g.v[j] = (p < 32 * 2 * KD && r < rows_valid && 8 * c < dh) ? ld16(src + (long long)r * row_stride + 8 * c) : u32x4{0u, 0u, 0u, 0u};
  }
}
template <typename T, int KD, int PITCH>
__device__ __forceinline__ void bm_put(const BmRegs<KD>& g, T* row_major, int tid) {
#pragma unroll
  for (int j = 0; j < BmRegs<KD>::NP; ++j) {
    const int p = tid + 256 * j, r = p / (2 * KD), c = p - r * (2 * KD);
    if (p < 32 * 2 * KD) st16(row_major + r * PITCH + 8 * c, g.v[j]);
  }
}
// A^T fragments straight from a ROW-MAJOR tile (ds_read_b64_tr_b16: per 16-lane group a block of 4 rows x 16 columns comes back
// column-major): the A operand of a product that consumes an accumulator -- row = channel 32 m + (lane & 31), its k slots
// (half, j) = tile rows 16 s2 + 8 (j >> 2) + 4 half + (j & 3), the accumulator's own order.  Lane 4 q + p of a group addresses
// tile row q of the block, columns 4 p .. 4 p + 3; two reads (rows + 0..3 and + 8..11).  The tile's pitch covers 32 MD columns
// (columns beyond the head width are zeros or padding: they only reach output rows that are never stored).
template <typename T, int PITCH>
__device__ __forceinline__ typename Vec8<T>::type bm_frag_t(const T* tile, int s2, int m, int lane) {
  typedef __attribute__((address_space(3))) s16x4* lds4_t;
  const int l16 = lane & 15, grp = lane >> 4, half = lane >> 5;
  const T* tb = tile + (16 * s2 + 4 * half + (l16 >> 2)) * PITCH + 32 * m + 16 * (grp & 1) + 4 * (l16 & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(tb));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(tb + 8 * PITCH));
  return __builtin_bit_cast(typename Vec8<T>::type, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// pass 1 on the matrix cores: L = log-sum-exp of the scaled scores (online maximum / sum over the key tiles, a query per lane),
// D = sum_j P_ij dP_ij (EXACT: the values are staged too and dP^T = V dO^T is multiplied beside S^T = K Q^T) or do . o
template <typename T, int KD, bool EXACT>
__global__ __launch_bounds__(256) void attn_bwd_stats_mfma_kernel(const AttnBwdArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int DHP = 16 * KD, PITCH = DHP + BM_PITCH_PAD;
  __shared__ __attribute__((aligned(16))) T sK2[2][32 * PITCH];          // two tiles: tile t + 1 is written while tile t is read
  __shared__ __attribute__((aligned(16))) T sV2[EXACT ? 2 : 1][EXACT ? 32 * PITCH : 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, img = blockIdx.z;
  const int q = blockIdx.x * 128 + wave * 32 + ql;
  const bool qok = q < a.tq;
  const T* qp = reinterpret_cast<const T*>(a.q) + img * a.q_is + (long long)(qok ? q : 0) * a.q_rs + h * a.dh;
  const T* dp_ = reinterpret_cast<const T*>(a.dout) + img * a.do_is + (long long)(qok ? q : 0) * a.do_rs + h * a.dh;
  const T* op_ = reinterpret_cast<const T*>(a.o) + img * a.o_is + (long long)(qok ? q : 0) * a.o_rs + h * a.dh;
  frag_t qf[KD], dof[EXACT ? KD : 1];
  float dd = 0.f;
#pragma unroll
  for (int s = 0; s < KD; ++s) {
    const int d0 = 16 * s + 8 * half;
    const bool ok = qok && d0 < a.dh;
    qf[s] = ok ? __builtin_bit_cast(frag_t, ld16(qp + d0)) : __builtin_bit_cast(frag_t, u32x4{0u, 0u, 0u, 0u});
    if constexpr (EXACT) {
      dof[s] = ok ? __builtin_bit_cast(frag_t, ld16(dp_ + d0)) : __builtin_bit_cast(frag_t, u32x4{0u, 0u, 0u, 0u});
    } else if (ok) {
      float x[8], y[8];
      unpack8<T>(ld16(dp_ + d0), x);
      unpack8<T>(ld16(op_ + d0), y);
#pragma unroll
      for (int j = 0; j < 8; ++j) dd += x[j] * y[j];
    }
  }
  dd += __shfl_xor(dd, 32, 64);
  float m = -3.0e38f, l = 0.f, da = 0.f;            // da = sum_j exp(s_j - m) dP_j, rescaled with l
  const T* kbase = reinterpret_cast<const T*>(a.k) + img * a.k_is + h * a.dh;
  const T* vbase = reinterpret_cast<const T*>(a.v) + img * a.v_is + h * a.dh;
  BmRegs<KD> gk, gv;
  bm_fetch<T, KD>(gk, kbase, a.k_rs, min(32, a.tk), a.dh, tid);
  if constexpr (EXACT) bm_fetch<T, KD>(gv, vbase, a.v_rs, min(32, a.tk), a.dh, tid);
  bm_put<T, KD, PITCH>(gk, sK2[0], tid);
  if constexpr (EXACT) bm_put<T, KD, PITCH>(gv, sV2[0], tid);
  __syncthreads();
  for (int k0 = 0, it = 0; k0 < a.tk; k0 += 32, ++it) {
    const int kv = min(32, a.tk - k0);
    const T* sK = sK2[it & 1];
    const bool more = k0 + 32 < a.tk;
    if (more) {
      bm_fetch<T, KD>(gk, kbase + (long long)(k0 + 32) * a.k_rs, a.k_rs, min(32, a.tk - k0 - 32), a.dh, tid);
      if constexpr (EXACT) bm_fetch<T, KD>(gv, vbase + (long long)(k0 + 32) * a.v_rs, a.v_rs, min(32, a.tk - k0 - 32), a.dh, tid);
    }
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < KD; ++s) {
      st = mfma32(__builtin_bit_cast(frag_t, ld16(sK + ql * PITCH + 16 * s + 8 * half)), qf[s], st);
      if constexpr (EXACT) {
        const T* sV = sV2[it & 1];
        dpt = mfma32(__builtin_bit_cast(frag_t, ld16(sV + ql * PITCH + 16 * s + 8 * half)), dof[s], dpt);
      }
    }
    float tm = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
      st[r] = key < kv ? st[r] * a.scale : -3.0e38f;
      tm = fmaxf(tm, st[r]);
    }
    const float mn = fmaxf(m, tm);
    float ts = 0.f, td = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = __expf(st[r] - mn);
      ts += e;
      if constexpr (EXACT) td += e * dpt[r];
    }
    const float c = __expf(m - mn);
    l = l * c + ts;
    da = da * c + td;
    m = mn;
    if (more) {                                                      // the other tile: everyone was done with it a barrier ago
      bm_put<T, KD, PITCH>(gk, sK2[(it + 1) & 1], tid);
      if constexpr (EXACT) bm_put<T, KD, PITCH>(gv, sV2[(it + 1) & 1], tid);
    }
    __syncthreads();
  }
  {                                                 // the two lanes of a query combine their halves of the keys
    const float m2 = __shfl_xor(m, 32, 64), l2 = __shfl_xor(l, 32, 64), d2 = __shfl_xor(da, 32, 64);
    const float mn = fmaxf(m, m2);
    const float c1 = __expf(m - mn), c2 = __expf(m2 - mn);
    l = l * c1 + l2 * c2;
    da = da * c1 + d2 * c2;
    m = mn;
  }
  if constexpr (EXACT) dd = da / l;
  if (qok && half == 0) {
    const long long idx = ((long long)img * a.heads + h) * a.tq + q;
    a.lse[idx] = m + __logf(l);
    a.dvec[idx] = dd;
  }
}

template <typename T, int KD>
__global__ __launch_bounds__(256) void attn_bwd_dq_mfma_kernel(const AttnBwdArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int DHP = 16 * KD, MD = (KD + 1) / 2, PITCH = 32 * MD + BM_PITCH_PAD;      // 32 MD >= DHP columns: see bm_frag_t
  __shared__ __attribute__((aligned(16))) T sK2[2][32 * PITCH], sV2[2][32 * PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ql = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, img = blockIdx.z;
  const int q = blockIdx.x * 128 + wave * 32 + ql;
  const bool qok = q < a.tq;
  for (int i = tid; i < 2 * 32 * PITCH; i += 256) (&sK2[0][0])[i] = (T)0.0f;             // columns >= DHP stay zero
  const T* qp = reinterpret_cast<const T*>(a.q) + img * a.q_is + (long long)(qok ? q : 0) * a.q_rs + h * a.dh;
  const T* dp_ = reinterpret_cast<const T*>(a.dout) + img * a.do_is + (long long)(qok ? q : 0) * a.do_rs + h * a.dh;
  frag_t qf[KD], dof[KD];
#pragma unroll
  for (int s = 0; s < KD; ++s) {
    const int d0 = 16 * s + 8 * half;
    const bool ok = qok && d0 < a.dh;
    qf[s] = ok ? __builtin_bit_cast(frag_t, ld16(qp + d0)) : __builtin_bit_cast(frag_t, u32x4{0u, 0u, 0u, 0u});
    dof[s] = ok ? __builtin_bit_cast(frag_t, ld16(dp_ + d0)) : __builtin_bit_cast(frag_t, u32x4{0u, 0u, 0u, 0u});
  }
  const long long sidx = ((long long)img * a.heads + h) * a.tq + (qok ? q : 0);
  const float L = qok ? a.lse[sidx] : 1.0e30f, D = qok ? a.dvec[sidx] : 0.f;
  f32x16 acc[MD];
#pragma unroll
  for (int m = 0; m < MD; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  const T* kbase = reinterpret_cast<const T*>(a.k) + img * a.k_is + h * a.dh;
  const T* vbase = reinterpret_cast<const T*>(a.v) + img * a.v_is + h * a.dh;
  BmRegs<KD> gk, gv;
  bm_fetch<T, KD>(gk, kbase, a.k_rs, min(32, a.tk), a.dh, tid);
  bm_fetch<T, KD>(gv, vbase, a.v_rs, min(32, a.tk), a.dh, tid);
  __syncthreads();                                  // the zeroed transposed tiles
  bm_put<T, KD, PITCH>(gk, sK2[0], tid);
  bm_put<T, KD, PITCH>(gv, sV2[0], tid);
  __syncthreads();
  // ONE barrier per tile: tile t + 1 goes into the other LDS set behind the products of tile t (its rows were requested
  // before them), which every wave stopped reading at the previous barrier
  for (int k0 = 0, it = 0; k0 < a.tk; k0 += 32, ++it) {
    const int kv = min(32, a.tk - k0);
    const T* sK = sK2[it & 1];
    const T* sV = sV2[it & 1];
    const bool more = k0 + 32 < a.tk;
    if (more) {
      const int kn = min(32, a.tk - k0 - 32);
      bm_fetch<T, KD>(gk, kbase + (long long)(k0 + 32) * a.k_rs, a.k_rs, kn, a.dh, tid);
      bm_fetch<T, KD>(gv, vbase + (long long)(k0 + 32) * a.v_rs, a.v_rs, kn, a.dh, tid);
    }
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < KD; ++s) {
      const frag_t kf = __builtin_bit_cast(frag_t, ld16(sK + ql * PITCH + 16 * s + 8 * half));
      const frag_t vf = __builtin_bit_cast(frag_t, ld16(sV + ql * PITCH + 16 * s + 8 * half));
      st = mfma32(kf, qf[s], st);
      dpt = mfma32(vf, dof[s], dpt);
    }
    // lane (query, half), register r: key k0 + (r & 3) + 8 (r >> 2) + 4 half
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float p = key < kv ? __expf(st[r] * a.scale - L) : 0.f;
      ds[r] = p * (dpt[r] - D) * a.scale;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float v8[8] = {ds[8 * s2], ds[8 * s2 + 1], ds[8 * s2 + 2], ds[8 * s2 + 3], ds[8 * s2 + 4], ds[8 * s2 + 5], ds[8 * s2 + 6], ds[8 * s2 + 7]};
      const frag_t bf = __builtin_bit_cast(frag_t, pack8<T>(v8));
#pragma unroll
      for (int m = 0; m < MD; ++m) acc[m] = mfma32(bm_frag_t<T, PITCH>(sK, s2, m, lane), bf, acc[m]);   // K^T: row = channel 32 m + ql
    }
    if (more) {
      bm_put<T, KD, PITCH>(gk, sK2[(it + 1) & 1], tid);
      bm_put<T, KD, PITCH>(gv, sV2[(it + 1) & 1], tid);
    }
    __syncthreads();
  }
  if (qok) {
    T* out = reinterpret_cast<T*>(a.dq) + ((long long)img * a.tq + q) * (a.heads * a.dh) + h * a.dh;
#pragma unroll
    for (int m = 0; m < MD; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * m + 8 * g + 4 * half;
        if (d0 < a.dh) {
          const float v4[4] = {acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
          *reinterpret_cast<u32x2*>(out + d0) = pack4<T>(v4);
        }
      }
  }
}

template <typename T, int KD>
__global__ __launch_bounds__(256) void attn_bwd_dkv_mfma_kernel(const AttnBwdArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int DHP = 16 * KD, MD = (KD + 1) / 2, PITCH = 32 * MD + BM_PITCH_PAD;
  __shared__ __attribute__((aligned(16))) T sQ2[2][32 * PITCH], sO2[2][32 * PITCH];
  __shared__ __attribute__((aligned(16))) float sL2[2][32], sD2[2][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kl = lane & 31, half = lane >> 5;
  const int h = blockIdx.y, img = blockIdx.z;
  const int key = blockIdx.x * 128 + wave * 32 + kl;
  const bool kok = key < a.tk;
  for (int i = tid; i < 2 * 32 * PITCH; i += 256) { (&sQ2[0][0])[i] = (T)0.0f; (&sO2[0][0])[i] = (T)0.0f; }
  const T* kp = reinterpret_cast<const T*>(a.k) + img * a.k_is + (long long)(kok ? key : 0) * a.k_rs + h * a.dh;
  const T* vp = reinterpret_cast<const T*>(a.v) + img * a.v_is + (long long)(kok ? key : 0) * a.v_rs + h * a.dh;
  frag_t kf[KD], vf[KD];
#pragma unroll
  for (int s = 0; s < KD; ++s) {
    const int d0 = 16 * s + 8 * half;
    const bool ok = kok && d0 < a.dh;
    kf[s] = ok ? __builtin_bit_cast(frag_t, ld16(kp + d0)) : __builtin_bit_cast(frag_t, u32x4{0u, 0u, 0u, 0u});
    vf[s] = ok ? __builtin_bit_cast(frag_t, ld16(vp + d0)) : __builtin_bit_cast(frag_t, u32x4{0u, 0u, 0u, 0u});
  }
  f32x16 ak[MD], av[MD];
#pragma unroll
  for (int m = 0; m < MD; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ak[m][r] = 0.f; av[m][r] = 0.f; }
  const T* qbase = reinterpret_cast<const T*>(a.q) + img * a.q_is + h * a.dh;
  const T* obase = reinterpret_cast<const T*>(a.dout) + img * a.do_is + h * a.dh;
  const long long sbase = ((long long)img * a.heads + h) * a.tq;
  BmRegs<KD> gq, go;
  bm_fetch<T, KD>(gq, qbase, a.q_rs, min(32, a.tq), a.dh, tid);
  bm_fetch<T, KD>(go, obase, a.do_rs, min(32, a.tq), a.dh, tid);
  float gl = (tid < 32 && tid < a.tq) ? a.lse[sbase + tid] : 1.0e30f, gd = (tid < 32 && tid < a.tq) ? a.dvec[sbase + tid] : 0.f;
  __syncthreads();                                  // the zeroed transposed tiles
  bm_put<T, KD, PITCH>(gq, sQ2[0], tid);
  bm_put<T, KD, PITCH>(go, sO2[0], tid);
  if (tid < 32) { sL2[0][tid] = gl; sD2[0][tid] = gd; }
  __syncthreads();
  for (int q0 = 0, it = 0; q0 < a.tq; q0 += 32, ++it) {      // one barrier per tile, as in the dQ pass
    const T* sQ = sQ2[it & 1];
    const T* sO = sO2[it & 1];
    const float* sL = sL2[it & 1];
    const float* sD = sD2[it & 1];
    const bool more = q0 + 32 < a.tq;
    if (more) {
      const int qn = min(32, a.tq - q0 - 32);
      bm_fetch<T, KD>(gq, qbase + (long long)(q0 + 32) * a.q_rs, a.q_rs, qn, a.dh, tid);
      bm_fetch<T, KD>(go, obase + (long long)(q0 + 32) * a.do_rs, a.do_rs, qn, a.dh, tid);
      gl = (tid < qn) ? a.lse[sbase + q0 + 32 + tid] : 1.0e30f;
      gd = (tid < qn) ? a.dvec[sbase + q0 + 32 + tid] : 0.f;
    }
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { st[r] = 0.f; dpt[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < KD; ++s) {
      const frag_t qf = __builtin_bit_cast(frag_t, ld16(sQ + kl * PITCH + 16 * s + 8 * half));
      const frag_t of = __builtin_bit_cast(frag_t, ld16(sO + kl * PITCH + 16 * s + 8 * half));
      st = mfma32(qf, kf[s], st);
      dpt = mfma32(of, vf[s], dpt);
    }
    // lane (key, half), register r: query q0 + (r & 3) + 8 (r >> 2) + 4 half
    float pr[16], ds[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + 8 * g + 4 * half), d4 = *reinterpret_cast<const f32x4*>(sD + 8 * g + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float p = __expf(st[4 * g + j] * a.scale - l4[j]);
        pr[4 * g + j] = p;
        ds[4 * g + j] = p * (dpt[4 * g + j] - d4[j]) * a.scale;
      }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const float p8[8] = {pr[8 * s2], pr[8 * s2 + 1], pr[8 * s2 + 2], pr[8 * s2 + 3], pr[8 * s2 + 4], pr[8 * s2 + 5], pr[8 * s2 + 6], pr[8 * s2 + 7]};
      const float d8[8] = {ds[8 * s2], ds[8 * s2 + 1], ds[8 * s2 + 2], ds[8 * s2 + 3], ds[8 * s2 + 4], ds[8 * s2 + 5], ds[8 * s2 + 6], ds[8 * s2 + 7]};
      const frag_t pf = __builtin_bit_cast(frag_t, pack8<T>(p8)), df = __builtin_bit_cast(frag_t, pack8<T>(d8));
#pragma unroll
      for (int m = 0; m < MD; ++m) {
        av[m] = mfma32(bm_frag_t<T, PITCH>(sO, s2, m, lane), pf, av[m]);          // dO^T P
        ak[m] = mfma32(bm_frag_t<T, PITCH>(sQ, s2, m, lane), df, ak[m]);          // Q^T dS
      }
    }
    if (more) {
      const int nx = (it + 1) & 1;
      bm_put<T, KD, PITCH>(gq, sQ2[nx], tid);
      bm_put<T, KD, PITCH>(go, sO2[nx], tid);
      if (tid < 32) { sL2[nx][tid] = gl; sD2[nx][tid] = gd; }
    }
    __syncthreads();
  }
  if (kok) {
    const long long off = ((long long)img * a.tk + key) * (a.heads * a.dh) + h * a.dh;
    T* outk = reinterpret_cast<T*>(a.dk) + off;
    T* outv = reinterpret_cast<T*>(a.dv) + off;
#pragma unroll
    for (int m = 0; m < MD; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = 32 * m + 8 * g + 4 * half;
        if (d0 < a.dh) {
          const float k4[4] = {ak[m][4 * g], ak[m][4 * g + 1], ak[m][4 * g + 2], ak[m][4 * g + 3]};
          const float v4[4] = {av[m][4 * g], av[m][4 * g + 1], av[m][4 * g + 2], av[m][4 * g + 3]};
          *reinterpret_cast<u32x2*>(outk + d0) = pack4<T>(k4);
          *reinterpret_cast<u32x2*>(outv + d0) = pack4<T>(v4);
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm (32 groups) (+ SiLU) backward, data gradient only (the affine parameters of the UNet's GroupNorms are frozen):
//   z = xh gamma + beta, y = act(z);  dxh = dy act'(z) gamma;  dx = rstd (dxh - mean_g(dxh) - xh mean_g(dxh xh)) (+ dx_add)
// One block per (image, group); statistics re-derived from x (mean, then variance about the mean), four passes over the
// group's [hw][C / 32] slab.  x, dy, dx: T [image][hw][C] dense.
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int silu,
                                                            const T* __restrict__ dx_add, T* __restrict__ dx, int hw, int C) {
  __shared__ float red[8];
  const int G = C / 32, g = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  const long long base = (long long)img * hw * C + g * G;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  const float inv_m = 1.0f / ((float)G * (float)hw);
  float s = 0.f;
  for (int p = tid; p < hw; p += 256)
    for (int c = 0; c < G; ++c) s += (float)x[base + (long long)p * C + c];
  const float mean = block_sum(s) * inv_m;
  s = 0.f;
  for (int p = tid; p < hw; p += 256)
    for (int c = 0; c < G; ++c) { const float d = (float)x[base + (long long)p * C + c] - mean; s += d * d; }
  const float rstd = rsqrtf(block_sum(s) * inv_m + eps);
  auto dxh_of = [&](long long off, int c, float& xh) {
    xh = ((float)x[off] - mean) * rstd;
    const float ga = gamma[g * G + c];
    float d = (float)dy[off];
    if (silu) {
      const float z = xh * ga + beta[g * G + c];
      const float sg = 1.0f / (1.0f + __expf(-z));
      d *= sg * (1.0f + z * (1.0f - sg));
    }
    return d * ga;
  };
  float s1 = 0.f, s2 = 0.f;
  for (int p = tid; p < hw; p += 256)
    for (int c = 0; c < G; ++c) {
      float xh;
      const float d = dxh_of(base + (long long)p * C + c, c, xh);
      s1 += d;
      s2 += d * xh;
    }
  const float m1 = block_sum(s1) * inv_m;
  const float m2 = block_sum(s2) * inv_m;
  for (int p = tid; p < hw; p += 256)
    for (int c = 0; c < G; ++c) {
      const long long off = base + (long long)p * C + c;
      float xh;
      const float d = dxh_of(off, c, xh);
      float v = rstd * (d - m1 - xh * m2);
      if (dx_add) v += (float)dx_add[off];
      dx[off] = (T)v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same GroupNorm backward as three COALESCED passes (the one-block-per-group kernel above walks 20-byte pieces at a
// C-byte stride four times: 0.9 ms per launch at 64 x 64 x 16 x 320, 56 ms of a training step).  A block owns a chunk of
// GNB_PIX pixels of one image with ALL channels: thread (piece, r) keeps one 8-channel piece (16-byte loads) and walks the
// chunk's pixels r, r + R, ...; per-channel sums go through LDS in a fixed order to partial[image][chunk][k][C]; a small
// kernel folds them per (image, group) -- fixed order everywhere: bit-reproducible.
//   pass 1: sum x, sum x^2            -> mean, rstd
//   pass 2: sum dxh, sum dxh xh       -> m1, m2         (dxh = dy act'(z) gamma, xh = (x - mean) rstd)
//   pass 3: dx = rstd (dxh - m1 - xh m2) (+ dx_add)
constexpr int GNB_PIX = 256;
struct GnbArgs {
  const void *x, *dy, *dx_add;
  void* dx;
  const float *gamma, *beta;
  float* partial;        // [image][chunks][2][C]
  float* stats;          // [image][32][4]: mean, rstd, m1, m2
  int hw, C, chunks, silu;
  float eps;
};

template <typename T, int PASS>
__global__ __launch_bounds__(256) void gnb_pass_kernel(const GnbArgs a) {
  __shared__ float red[2][256][8];
  const int C = a.C, G = C / 32, pieces = C / 8, R = 256 / pieces;
  const int tid = threadIdx.x, piece = tid % pieces, r = tid / pieces;
  const int img = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * GNB_PIX, p1 = min(a.hw, p0 + GNB_PIX);
  const bool active = r < R;
  const T* xp = reinterpret_cast<const T*>(a.x) + (long long)img * a.hw * C + piece * 8;
  const T* dyp = reinterpret_cast<const T*>(a.dy) + (long long)img * a.hw * C + piece * 8;
  float mean[8], rstd[8], ga[8], be[8], m1[8], m2[8];
  if (PASS >= 2 && active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = piece * 8 + j;
      const float* st = a.stats + ((long long)img * 32 + c / G) * 4;
      mean[j] = st[0]; rstd[j] = st[1]; m1[j] = st[2]; m2[j] = st[3];
      ga[j] = a.gamma[c]; be[j] = a.beta[c];
    }
  }
  float s0[8], s1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  if (active) {
    for (int p = p0 + r; p < p1; p += R) {
      float xv[8];
      unpack8<T>(ld16(xp + (long long)p * C), xv);
      if (PASS == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0[j] += xv[j]; s1[j] += xv[j] * xv[j]; }
      } else {
        float dv[8];
        unpack8<T>(ld16(dyp + (long long)p * C), dv);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (xv[j] - mean[j]) * rstd[j];
          float d = dv[j];
          if (a.silu) {
            const float z = xh * ga[j] + be[j];
            const float sg = 1.0f / (1.0f + __expf(-z));
            d *= sg * (1.0f + z * (1.0f - sg));
          }
          d *= ga[j];
          if (PASS == 2) { s0[j] += d; s1[j] += d * xh; }
          else o[j] = rstd[j] * (d - m1[j] - xh * m2[j]);
        }
        if (PASS == 3) {
          const long long off = ((long long)img * a.hw + p) * C + piece * 8;
          if (a.dx_add) {
            float e[8];
            unpack8<T>(ld16(reinterpret_cast<const T*>(a.dx_add) + off), e);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += e[j];
          }
          st16(reinterpret_cast<T*>(a.dx) + off, pack8<T>(o));
        }
      }
    }
  }
  if (PASS == 3) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][tid][j] = s0[j]; red[1][tid][j] = s1[j]; }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {             // channel c: piece c / 8, element c % 8; rows r = 0 .. R - 1 in order
    float t0 = 0.f, t1 = 0.f;
    for (int rr = 0; rr < R; ++rr) { t0 += red[0][rr * pieces + c / 8][c & 7]; t1 += red[1][rr * pieces + c / 8][c & 7]; }
    float* out = a.partial + (((long long)img * a.chunks + chunk) * 2) * C;
    out[c] = t0;
    out[C + c] = t1;
  }
}

// one wave per (image, group): fold the chunk partials of the group's channels (fixed order), write the statistics
template <int PASS>
__global__ __launch_bounds__(64) void gnb_fold_kernel(const GnbArgs a) {
  // (the chunk partials meet in fp64, as in the forward's two-launch form (csrc/norm.hip, gn_apply_kernel): E[x^2] - mean^2 of
  //  a group whose mean is large against its spread then loses the partials' fp32 rounding only, not the difference's)
  const int g = blockIdx.x, img = blockIdx.y, lane = threadIdx.x, G = a.C / 32;
  double t0 = 0.0, t1 = 0.0;
  for (int i = lane; i < a.chunks * G; i += 64) {
    const int ch = i / G, c = g * G + (i - ch * G);
    const float* pp = a.partial + (((long long)img * a.chunks + ch) * 2) * a.C;
    t0 += (double)pp[c];
    t1 += (double)pp[a.C + c];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    t0 += __shfl_xor(t0, o, 64);
    t1 += __shfl_xor(t1, o, 64);
  }
  if (lane == 0) {
    float* st = a.stats + ((long long)img * 32 + g) * 4;
    const double inv_m = 1.0 / ((double)G * (double)a.hw);
    if (PASS == 1) {
      const double mean = t0 * inv_m;
      double var = t1 * inv_m - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      st[0] = (float)mean;
      st[1] = (float)(1.0 / sqrt(var + (double)a.eps));
    } else {
      st[2] = (float)(t0 * inv_m);
      st[3] = (float)(t1 * inv_m);
    }
  }
}

// out = a + b (gradients meeting at a fork of the graph: a skip connection's two consumers)
template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    out[i] = (T)((float)a[i] + (float)b[i]);
}

// dx = dy * silu'(z), fp32 (the bbox embedder's MLP, modules.py:77-83 of the reference: a handful of rows)
__global__ __launch_bounds__(256) void silu_bwd_f32_kernel(const float* __restrict__ z, const float* __restrict__ dy, float* __restrict__ dx, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = z[i], sg = 1.0f / (1.0f + __expf(-v));
    dx[i] = dy[i] * sg * (1.0f + v * (1.0f - sg));
  }
}

// AdamW (torch.optim.AdamW's update, ddpm.py:1649 of the reference), fp32 master parameters updated in place:
//   p *= 1 - lr wd;  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  p -= lr / bc1 * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi;
  }
}

// sum of every 2 x 2 block of pixels: src T [image][2h][2w][C] -> out T [image][h][w][C] (the backward of nearest x2)
template <typename T>
__global__ __launch_bounds__(256) void sumpool2_kernel(const T* __restrict__ src, T* __restrict__ out, int n, int h, int w, int C) {
  const long long total = (long long)n * h * w * C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long long r = i / C;
    const int xo = (int)(r % w);
    r /= w;
    const int yo = (int)(r % h), im = (int)(r / h);
    const long long b = (((long long)im * 2 * h + 2 * yo) * 2 * w + 2 * xo) * C + c;
    const float v = ((float)src[b] + (float)src[b + C]) + ((float)src[b + (long long)2 * w * C] + (float)src[b + (long long)2 * w * C + C]);
    out[i] = (T)v;
  }
}

}  // namespace mobi

#define ST(s) reinterpret_cast<hipStream_t>(s)
#define BW_DISPATCH(dtype, KERNEL, grid, ...)                                                                      \
  do {                                                                                                             \
    if ((dtype) == MOBI_F16) hipLaunchKernelGGL((KERNEL<mobi::f16_t>), grid, dim3(256), 0, ST(stream), __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<mobi::bf16_t>), grid, dim3(256), 0, ST(stream), __VA_ARGS__);                  \
  } while (0)

extern "C" int mobi_transpose(const void* src, int64_t src_row_stride, void* out, int32_t rows, int32_t cols, int32_t dtype,
                              void* stream) {
  using namespace mobi;
  if (!src || !out || rows <= 0 || cols <= 0 || src_row_stride < cols) return MOBI_ERR_ARG;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (dtype == MOBI_F16)
    hipLaunchKernelGGL((transpose_kernel<f16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const f16_t*>(src), src_row_stride,
                       reinterpret_cast<f16_t*>(out), rows, cols);
  else
    hipLaunchKernelGGL((transpose_kernel<bf16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(src), src_row_stride,
                       reinterpret_cast<bf16_t*>(out), rows, cols);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_tile_weights(const void* w, void* out, int32_t n, int32_t k, void* stream) {
  using namespace mobi;
  if (!w || !out || n <= 0 || k <= 0) return MOBI_ERR_ARG;
  if ((n & 15) || (k & 31)) return MOBI_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(out)) & 15) return MOBI_ERR_ALIGN;
  const long long chunks = (long long)n * (k >> 3);
  long long blocks = (chunks + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(tile_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, ST(stream), reinterpret_cast<const u32x4*>(w),
                     reinterpret_cast<u32x4*>(out), n, k);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int32_t mobi_backward_partial_blocks(int64_t rows) {
  // 16 rows per block (4 per wave) up to 4,096 blocks: with 256 rows per block a [65536, 320] LayerNorm backward ran as
  // 1,024 latency chains of 64 rows each -- 0.9 ms, 0.14 TB/s (rocprofv3 of tools/train_bench.py)
  const int64_t b = (rows + 15) / 16;
  return (int32_t)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

extern "C" int mobi_colsum(const void* dy, int64_t row_stride, int64_t rows, int32_t cols, int32_t dtype, float* partial,
                           float* out, void* stream) {
  using namespace mobi;
  if (!dy || !partial || !out || rows <= 0 || cols <= 0) return MOBI_ERR_ARG;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  const int nblk = mobi_backward_partial_blocks(rows);
  const int rpb = (int)((rows + nblk - 1) / nblk);
  if (dtype == MOBI_F16)
    hipLaunchKernelGGL((colsum_kernel<f16_t>), dim3(nblk), dim3(256), 0, ST(stream), reinterpret_cast<const f16_t*>(dy), row_stride, partial, rows, cols, rpb);
  else
    hipLaunchKernelGGL((colsum_kernel<bf16_t>), dim3(nblk), dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(dy), row_stride, partial, rows, cols, rpb);
  reduce_partials(partial, partial + (long long)nblk * cols, out, nblk, (long long)cols, ST(stream));
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_layernorm_bwd(const mobi_layernorm_bwd_params* p, void* stream) {
  using namespace mobi;
  if (!p || !p->x || !p->dy || !p->gamma || !p->dx || !p->partial || !p->dgamma_dbeta) return MOBI_ERR_ARG;
  if (p->dtype != MOBI_F16 && p->dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (p->rows <= 0 || p->channels <= 0 || p->channels > 1536) return MOBI_ERR_UNSUPPORTED;
  const int nblk = mobi_backward_partial_blocks(p->rows);
  const int rpb = (int)((p->rows + nblk - 1) / nblk);
  const long long xs = p->x_row_stride ? p->x_row_stride : p->channels, ds = p->dy_row_stride ? p->dy_row_stride : p->channels;
#define MOBI_LNB_LAUNCH(T_, MAXV_)                                                                                                   \
  hipLaunchKernelGGL((layernorm_bwd_kernel<T_, MAXV_>), dim3(nblk), dim3(256), 0, ST(stream), reinterpret_cast<const T_*>(p->x), xs, \
                     reinterpret_cast<const T_*>(p->dy), ds, p->gamma, p->eps, reinterpret_cast<const T_*>(p->dx_add),                \
                     reinterpret_cast<T_*>(p->dx), p->partial, (long long)p->rows, p->channels, rpb)
#define MOBI_LNB_BY_C(T_)                                                                                       \
  do {                                                                                                          \
    if (p->channels <= 320) MOBI_LNB_LAUNCH(T_, 5); else if (p->channels <= 640) MOBI_LNB_LAUNCH(T_, 10);       \
    else if (p->channels <= 1280) MOBI_LNB_LAUNCH(T_, 20); else MOBI_LNB_LAUNCH(T_, 24);                        \
  } while (0)
  if (p->dtype == MOBI_F16) MOBI_LNB_BY_C(f16_t); else MOBI_LNB_BY_C(bf16_t);
#undef MOBI_LNB_BY_C
#undef MOBI_LNB_LAUNCH
  // partial is [nblk][2][C]: d gamma = sum of the [.][0][.] planes, d beta of the [.][1][.] planes
  reduce_partials(p->partial, p->partial + (long long)nblk * 2 * p->channels, p->dgamma_dbeta, nblk, (long long)2 * p->channels, ST(stream));
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_geglu_fwd(const void* pre, void* h, int64_t rows, int32_t inner, int32_t dtype, void* stream) {
  using namespace mobi;
  if (!pre || !h || rows <= 0 || inner <= 0) return MOBI_ERR_ARG;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  const long long n = rows * inner;
  const dim3 grid((unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256));
  if (dtype == MOBI_F16) hipLaunchKernelGGL((geglu_fwd_kernel<f16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const f16_t*>(pre), reinterpret_cast<f16_t*>(h), (long long)rows, inner);
  else hipLaunchKernelGGL((geglu_fwd_kernel<bf16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(pre), reinterpret_cast<bf16_t*>(h), (long long)rows, inner);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_geglu_bwd(const void* pre, const void* dh, void* dpre, int64_t rows, int32_t inner, int32_t dtype, void* stream) {
  using namespace mobi;
  if (!pre || !dh || !dpre || rows <= 0 || inner <= 0) return MOBI_ERR_ARG;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  const long long n = rows * inner;
  const dim3 grid((unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256));
  if (dtype == MOBI_F16) hipLaunchKernelGGL((geglu_bwd_kernel<f16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const f16_t*>(pre), reinterpret_cast<const f16_t*>(dh), reinterpret_cast<f16_t*>(dpre), (long long)rows, inner);
  else hipLaunchKernelGGL((geglu_bwd_kernel<bf16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(pre), reinterpret_cast<const bf16_t*>(dh), reinterpret_cast<bf16_t*>(dpre), (long long)rows, inner);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_attention_bwd(const mobi_attention_bwd_params* p, void* stream) {
  using namespace mobi;
  if (!p || !p->q || !p->k || !p->v || !p->o || !p->dout || !p->dq || !p->dk || !p->dv || !p->lse || !p->dvec) return MOBI_ERR_ARG;
  if (p->dtype != MOBI_F16 && p->dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (p->images <= 0 || p->heads <= 0 || p->dh <= 0 || p->tq <= 0 || p->tk <= 0) return MOBI_ERR_ARG;
  if (p->dh > BW_DH) return MOBI_ERR_UNSUPPORTED;
  AttnBwdArgs a;
  a.q = p->q; a.k = p->k; a.v = p->v; a.o = p->o; a.dout = p->dout;
  a.q_is = p->q_img_stride; a.q_rs = p->q_row_stride; a.k_is = p->k_img_stride; a.k_rs = p->k_row_stride;
  a.v_is = p->v_img_stride; a.v_rs = p->v_row_stride; a.o_is = p->o_img_stride; a.o_rs = p->o_row_stride;
  a.do_is = p->dout_img_stride; a.do_rs = p->dout_row_stride;
  a.dq = p->dq; a.dk = p->dk; a.dv = p->dv; a.lse = p->lse; a.dvec = p->dvec;
  a.heads = p->heads; a.dh = p->dh; a.tq = p->tq; a.tk = p->tk; a.scale = p->scale;
  a.exact_d = tuning().attn_bwd_exact_d != 0;       // MOBI_ATTN_BWD_EXACT_D=0: D = do . o with the stored output (A/B)
  const dim3 gq((p->tq + BW_TILE - 1) / BW_TILE, p->heads, p->images), gk((p->tk + BW_TILE - 1) / BW_TILE, p->heads, p->images);
  // the matrix-core passes need 8-element pieces of a head row to be 16-byte aligned loads: dh % 8 == 0, strides % 8 == 0
  const int kd = (p->dh + 15) / 16;
  const bool al = p->dh % 8 == 0 && ((p->q_row_stride | p->k_row_stride | p->v_row_stride | p->dout_row_stride | p->q_img_stride |
                                      p->k_img_stride | p->v_img_stride | p->dout_img_stride) & 7) == 0 &&
                  ((p->o_row_stride | p->o_img_stride) & 7) == 0 &&
                  ((reinterpret_cast<uintptr_t>(p->q) | reinterpret_cast<uintptr_t>(p->k) | reinterpret_cast<uintptr_t>(p->v) |
                    reinterpret_cast<uintptr_t>(p->dout) | reinterpret_cast<uintptr_t>(p->o)) & 15) == 0;
  const dim3 gq4((p->tq + 127) / 128, p->heads, p->images), gk4((p->tk + 127) / 128, p->heads, p->images);
#define BM_CASE(KD_)                                                                                                       \
  case KD_:                                                                                                                \
    if (p->dtype == MOBI_F16) {                                                                                            \
      if (a.exact_d) hipLaunchKernelGGL((attn_bwd_stats_mfma_kernel<f16_t, KD_, true>), gq4, dim3(256), 0, ST(stream), a);  \
      else hipLaunchKernelGGL((attn_bwd_stats_mfma_kernel<f16_t, KD_, false>), gq4, dim3(256), 0, ST(stream), a);          \
      hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<f16_t, KD_>), gq4, dim3(256), 0, ST(stream), a);                         \
      hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<f16_t, KD_>), gk4, dim3(256), 0, ST(stream), a);                        \
    } else {                                                                                                               \
      if (a.exact_d) hipLaunchKernelGGL((attn_bwd_stats_mfma_kernel<bf16_t, KD_, true>), gq4, dim3(256), 0, ST(stream), a); \
      else hipLaunchKernelGGL((attn_bwd_stats_mfma_kernel<bf16_t, KD_, false>), gq4, dim3(256), 0, ST(stream), a);         \
      hipLaunchKernelGGL((attn_bwd_dq_mfma_kernel<bf16_t, KD_>), gq4, dim3(256), 0, ST(stream), a);                        \
      hipLaunchKernelGGL((attn_bwd_dkv_mfma_kernel<bf16_t, KD_>), gk4, dim3(256), 0, ST(stream), a);                       \
    }                                                                                                                      \
    break;
  bool done = false;
  if (al && !p->force_vector) {
    done = true;
    switch (kd) {
      BM_CASE(1) BM_CASE(2) BM_CASE(3) BM_CASE(4) BM_CASE(5) BM_CASE(10)
      default: done = false;
    }
  }
#undef BM_CASE
  if (!done) {
    BW_DISPATCH(p->dtype, attn_bwd_stats_kernel, gq, a);
    BW_DISPATCH(p->dtype, attn_bwd_dq_kernel, gq, a);
    BW_DISPATCH(p->dtype, attn_bwd_dkv_kernel, gk, a);
  }
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" size_t mobi_groupnorm_bwd_workspace_floats(int32_t images, int32_t hw, int32_t channels) {
  if (images <= 0 || hw <= 0 || channels <= 0) return 0;
  return (size_t)images * 32 * 4 + (size_t)images * ((hw + mobi::GNB_PIX - 1) / mobi::GNB_PIX) * 2 * channels;
}

extern "C" int mobi_groupnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, float eps, int32_t silu,
                                  const void* dx_add, void* dx, int32_t images, int32_t hw, int32_t channels, int32_t dtype,
                                  float* ws, void* stream) {
  using namespace mobi;
  if (!x || !dy || !gamma || !beta || !dx || images <= 0 || hw <= 0) return MOBI_ERR_ARG;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (channels <= 0 || channels % 32) return MOBI_ERR_UNSUPPORTED;
  if (ws && channels % 8 == 0 && channels / 8 <= 256 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dx_add)) & 15) == 0) {
    GnbArgs a;
    a.x = x; a.dy = dy; a.dx_add = dx_add; a.dx = dx; a.gamma = gamma; a.beta = beta;
    a.hw = hw; a.C = channels; a.chunks = (hw + GNB_PIX - 1) / GNB_PIX; a.silu = silu; a.eps = eps;
    a.stats = ws;
    a.partial = ws + (size_t)images * 32 * 4;
    const dim3 gp(a.chunks, images), gf(32, images);
    if (dtype == MOBI_F16) {
      hipLaunchKernelGGL((gnb_pass_kernel<f16_t, 1>), gp, dim3(256), 0, ST(stream), a);
      hipLaunchKernelGGL((gnb_fold_kernel<1>), gf, dim3(64), 0, ST(stream), a);
      hipLaunchKernelGGL((gnb_pass_kernel<f16_t, 2>), gp, dim3(256), 0, ST(stream), a);
      hipLaunchKernelGGL((gnb_fold_kernel<2>), gf, dim3(64), 0, ST(stream), a);
      hipLaunchKernelGGL((gnb_pass_kernel<f16_t, 3>), gp, dim3(256), 0, ST(stream), a);
    } else {
      hipLaunchKernelGGL((gnb_pass_kernel<bf16_t, 1>), gp, dim3(256), 0, ST(stream), a);
      hipLaunchKernelGGL((gnb_fold_kernel<1>), gf, dim3(64), 0, ST(stream), a);
      hipLaunchKernelGGL((gnb_pass_kernel<bf16_t, 2>), gp, dim3(256), 0, ST(stream), a);
      hipLaunchKernelGGL((gnb_fold_kernel<2>), gf, dim3(64), 0, ST(stream), a);
      hipLaunchKernelGGL((gnb_pass_kernel<bf16_t, 3>), gp, dim3(256), 0, ST(stream), a);
    }
    MOBI_CHECK_LAUNCH();
    return MOBI_OK;
  }
  const dim3 grid(32, images);
  if (dtype == MOBI_F16)
    hipLaunchKernelGGL((groupnorm_bwd_kernel<f16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const f16_t*>(x), reinterpret_cast<const f16_t*>(dy),
                       gamma, beta, eps, silu, reinterpret_cast<const f16_t*>(dx_add), reinterpret_cast<f16_t*>(dx), hw, channels);
  else
    hipLaunchKernelGGL((groupnorm_bwd_kernel<bf16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(x), reinterpret_cast<const bf16_t*>(dy),
                       gamma, beta, eps, silu, reinterpret_cast<const bf16_t*>(dx_add), reinterpret_cast<bf16_t*>(dx), hw, channels);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_sumpool2(const void* src, void* out, int32_t images, int32_t h, int32_t w, int32_t channels, int32_t dtype,
                             void* stream) {
  using namespace mobi;
  if (!src || !out || images <= 0 || h <= 0 || w <= 0 || channels <= 0) return MOBI_ERR_ARG;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  const long long total = (long long)images * h * w * channels;
  const dim3 grid((unsigned)((total + 255) / 256 > 65536 ? 65536 : (total + 255) / 256));
  if (dtype == MOBI_F16) hipLaunchKernelGGL((sumpool2_kernel<f16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const f16_t*>(src), reinterpret_cast<f16_t*>(out), images, h, w, channels);
  else hipLaunchKernelGGL((sumpool2_kernel<bf16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(src), reinterpret_cast<bf16_t*>(out), images, h, w, channels);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_add(const void* a, const void* b, void* out, int64_t n, int32_t dtype, void* stream) {
  using namespace mobi;
  if (!a || !b || !out || n <= 0) return MOBI_ERR_ARG;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  const dim3 grid((unsigned)((n + 255) / 256 > 65536 ? 65536 : (n + 255) / 256));
  if (dtype == MOBI_F16) hipLaunchKernelGGL((add_kernel<f16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const f16_t*>(a), reinterpret_cast<const f16_t*>(b), reinterpret_cast<f16_t*>(out), (long long)n);
  else hipLaunchKernelGGL((add_kernel<bf16_t>), grid, dim3(256), 0, ST(stream), reinterpret_cast<const bf16_t*>(a), reinterpret_cast<const bf16_t*>(b), reinterpret_cast<bf16_t*>(out), (long long)n);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_silu_bwd_f32(const float* z, const float* dy, float* dx, int64_t n, void* stream) {
  using namespace mobi;
  if (!z || !dy || !dx || n <= 0) return MOBI_ERR_ARG;
  const dim3 grid((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256));
  hipLaunchKernelGGL(silu_bwd_f32_kernel, grid, dim3(256), 0, ST(stream), z, dy, dx, (long long)n);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  using namespace mobi;
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || step <= 0) return MOBI_ERR_ARG;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  const dim3 grid((unsigned)((n + 255) / 256 > 16384 ? 16384 : (n + 255) / 256));
  hipLaunchKernelGGL(adamw_kernel, grid, dim3(256), 0, ST(stream), param, grad, exp_avg, exp_avg_sq, (long long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, sqrtf(bc2));
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}
