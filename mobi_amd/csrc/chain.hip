// Row-resident chain of C -> C linear layers for gfx950 (C = 320), include/mobi_engine.h mobi_row_chain:
// the launches BETWEEN the attention kernels of a transformer block (BasicTransformerBlock._forward, attention.py:230-266
// of the reference) as one kernel each -- to_out (+ residual + attn2 vector) -> two-key bbox adapter -> cross-modal
// LayerNorm -> to_q / to_k / to_v, and connector o to_out (+ residual) -> the partner's to_k / to_v -- so that the
// [rows][C] intermediates never make an HBM round trip.
//
// Structure = ff.hip's: a wave holds 32 token rows as the B fragments of MFMA 32x32x16 (lane l: row l & 31, channels
// 16 ks + 8 (l >> 5) + 0..7 of fragment ks), weights are the A operand, streamed through an LDS ring as ready-made 1-KiB
// fragment images by LDS-DMA (no VGPR staging), D^T[channel][row] accumulates in fp32.  The rows of every weight image
// are permuted (tau: bits 2 and 3 of the row index swapped) so that the accumulator registers a lane receives ARE the
// channels of the fragments it holds: acc[m][8 (ks & 1) + j] <-> fragment ks = 2 m + .., element j.  A product's result
// therefore becomes the next product's operand by a conversion in registers, and everything row-wise (residual, bias,
// LayerNorm statistics, the adapter's gates) is lane-local plus one exchange between lanes l and l + 32.
//
// LayerNorm in front of a projection is FOLDED into it (exact algebra): LN(x) W^T = rstd (x (W diag gamma)^T - mean s) +
// W beta, s = row sums of the rounded W diag gamma -- the product reads the raw rows, the epilogue applies two scalars per
// row; no normalised copy of the row exists, not even in registers.
//
// A block = 4 waves (one per SIMD: ~400 registers) = 128 rows of ONE image; the program (a short list of operations,
// mobi_chain_op) is interpreted with uniform branches, every product is the same unrolled 200-MFMA body.
// Weight ring: NSLOT slots of one chunk (20 KiB = 2 k-steps x 10 output tiles); the requests of chunk g + NSLOT - 1 are
// issued behind the barrier that opens chunk g (all waves are then done with chunk g - 1, whose slot it takes); a wave
// waits for its OWN requests with a counted vmcnt, the barrier publishes everybody's.
#include "common.h"

namespace mobi {
namespace {

constexpr int CH_C = 320, CH_KS = CH_C / 16, CH_MT = CH_C / 32;
constexpr int CH_NCH = CH_KS / 2;                 // chunks per product
constexpr int CH_FR = 2 * CH_MT;                  // fragments per chunk
constexpr int CH_CHUNK = CH_FR * 1024;
constexpr int CH_NSLOT = 6, CH_D = CH_NSLOT - 1;
constexpr int CH_RPW = CH_FR / 4;                 // requests per wave and chunk
constexpr int CH_AST = CH_C * 2 + 16;             // bytes per row of the adapter's logit tables
constexpr int CH_TAB_A = 9 * CH_AST, CH_TAB_U = (CH_C + 1) * 16;
constexpr int CH_TAB = 2 * CH_TAB_A + 2 * CH_TAB_U + CH_C * 4 + 64;
constexpr int CH_TAB_PAD = (CH_TAB + 1023) / 1024 * 1024;
constexpr int CH_QD = 5;                          // fragment queue depth: divides CH_FR (the queue runs on across chunks)
static_assert(CH_FR % CH_QD == 0, "queue slot of fragment f is f % QD in every chunk");
constexpr int CH_MAXP = 4;                        // products per program
constexpr int CH_VEC = (CH_MAXP * 2 + 2) * CH_C * 4;   // their bias / row-sum vectors, + the AFFINE_S scale / shift (fp32) in LDS
static_assert(CH_FR % 4 == 0, "chunk pieces are dealt to four waves");

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void wait_vm(int groups) {
  // all but the `groups` youngest request groups (CH_RPW requests each) of this wave have landed
  switch (groups) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * CH_RPW) : "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CH_RPW) : "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CH_RPW) : "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * CH_RPW) : "memory"); break;
  }
}

#ifndef MOBI_CHAIN_DBG
#define MOBI_CHAIN_DBG 0     // timing-only ablations (WRONG results), lab builds: 1 no epilogue arithmetic / stores, 2 no adapter,
#endif                       // 4 no row statistics, 8 no MFMAs
#ifdef MOBI_CHAIN_STAMPS     // lab builds only (tools/chain_stamps.py): shader-clock stamps of one camera and one lidar block
__device__ unsigned long long g_chain_stamps[2][32];
#define CH_STAMP(i)                                                                                         \
  do {                                                                                                      \
    if (tile == 0 && img < 2 && tid == 0) g_chain_stamps[img][(i)] = __builtin_readcyclecounter();          \
  } while (0)
#else
#define CH_STAMP(i) do {} while (0)
#endif

}  // namespace

// The adapter's tables of one image as the LDS image the chain kernel copies in by LDS-DMA (mobi_row_chain_adapter_image):
//   [a hi | a lo] 9 rows of CH_AST bytes each (T [head][C], row 8 zero: what MFMA rows 8..31 read), [u hi | u lo] (C + 1)
//   entries of 8 T (U^T, heads contiguous; entry C zero: what the lanes that supply k >= 8 read), b f32 [C],
//   a_sum f32 [8] (sums of the SPLIT table rows: the numbers the logit product uses), c f32 [8]; padded to whole KiB.
template <typename T>
__global__ __launch_bounds__(256) void chain_adapter_image_kernel(const float* __restrict__ a, const float* __restrict__ cvec,
                                                                  const float* __restrict__ u, const float* __restrict__ b,
                                                                  int H, unsigned char* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned char tab[CH_TAB_PAD];
  unsigned char* s_ahi = tab;
  unsigned char* s_alo = s_ahi + CH_TAB_A;
  unsigned char* s_uhi = s_alo + CH_TAB_A;
  unsigned char* s_ulo = s_uhi + CH_TAB_U;
  float* s_b = reinterpret_cast<float*>(s_ulo + CH_TAB_U);
  float* s_asum = s_b + CH_C;
  float* s_cc = s_asum + 8;
  const int img = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < CH_TAB_PAD / 4; i += 256) reinterpret_cast<unsigned*>(tab)[i] = 0u;
  __syncthreads();
  const float* ga = a + (long long)img * H * CH_C;
  const float* gu = u + (long long)img * H * CH_C;
  for (int i = tid; i < H * (CH_C / 4); i += 256) {
    const int hh = i / (CH_C / 4), c = (i - hh * (CH_C / 4)) * 4;
    const f32x4 va = *reinterpret_cast<const f32x4*>(ga + hh * CH_C + c);
    float lo[4], hi[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const T t = (T)va[e]; hi[e] = (float)t; lo[e] = va[e] - hi[e]; }
    *reinterpret_cast<u32x2*>(s_ahi + hh * CH_AST + c * 2) = pack4<T>(hi);
    *reinterpret_cast<u32x2*>(s_alo + hh * CH_AST + c * 2) = pack4<T>(lo);
  }
  for (int c = tid; c < CH_C; c += 256) {
    float hi[8], lo[8];
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) {
      const float v = hh < H ? gu[hh * CH_C + c] : 0.f;
      const T t = (T)v;
      hi[hh] = (float)t;
      lo[hh] = v - hi[hh];
    }
    st16(s_uhi + c * 16, pack8<T>(hi));
    st16(s_ulo + c * 16, pack8<T>(lo));
    s_b[c] = b[(long long)img * CH_C + c];
  }
  if (tid < 8) s_cc[tid] = tid < H ? cvec[img * H + tid] : 0.f;
  __syncthreads();
  {                                                 // sums of the numbers the logit product will use: 32 threads per head
    const int hh = tid >> 5, l = tid & 31;
    float sum = 0.f;
    for (int c = l; c < CH_C; c += 32)
      sum += (float)*reinterpret_cast<const T*>(s_ahi + hh * CH_AST + c * 2) + (float)*reinterpret_cast<const T*>(s_alo + hh * CH_AST + c * 2);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (l == 0) s_asum[hh] = sum;
  }
  __syncthreads();
  for (int i = tid; i < CH_TAB_PAD / 16; i += 256) st16(out + (long long)img * CH_TAB_PAD + i * 16, ld16(tab + i * 16));
}

// GroupNorm (32 groups) of x [image][hw][C] folded to two per-image vectors for MOBI_CH_AFFINE_S: scale = rstd gamma,
// shift = beta - mean rstd gamma (mean, then the variance about the mean: gn_stats_kernel's arithmetic).  One block per
// (group, image); a thread reads whole pixels of the group (C / 32 consecutive channels).
template <typename T>
__global__ __launch_bounds__(256) void gn_scale_shift_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float eps, float* __restrict__ scale, float* __restrict__ shift, int hw, int C) {
  __shared__ float red[4];
  const int G = C / 32, g = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  const T* base = x + (long long)img * hw * C + g * G;
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
    for (int c = 0; c < G; c += 2) {
      const unsigned w = *reinterpret_cast<const unsigned*>(base + (long long)p * C + c);
      typedef T T2 __attribute__((ext_vector_type(2)));
      const T2 v = __builtin_bit_cast(T2, w);
      s += (float)v[0] + (float)v[1];
    }
  const float mean = block_sum(s) * inv_m;
  s = 0.f;
  for (int p = tid; p < hw; p += 256)
    for (int c = 0; c < G; c += 2) {
      const unsigned w = *reinterpret_cast<const unsigned*>(base + (long long)p * C + c);
      typedef T T2 __attribute__((ext_vector_type(2)));
      const T2 v = __builtin_bit_cast(T2, w);
      const float d0 = (float)v[0] - mean, d1 = (float)v[1] - mean;
      s += d0 * d0 + d1 * d1;
    }
  const float rstd = rsqrtf(block_sum(s) * inv_m + eps);
  if (tid < G) {
    const int c = g * G + tid;
    const float sc = rstd * gamma[c];
    scale[(long long)img * C + c] = sc;
    shift[(long long)img * C + c] = beta[c] - mean * sc;
  }
}

template <typename T>
__global__ __launch_bounds__(256, 1) void row_chain_kernel(const mobi_row_chain_params a) {
  typedef typename Vec8<T>::type frag_t;
  __shared__ __attribute__((aligned(16))) unsigned char lds[CH_NSLOT * CH_CHUNK + CH_TAB_PAD + CH_VEC];
  unsigned char* ring = lds;
  unsigned char* s_ahi = lds + CH_NSLOT * CH_CHUNK;
  unsigned char* s_alo = s_ahi + CH_TAB_A;
  unsigned char* s_uhi = s_alo + CH_TAB_A;
  unsigned char* s_ulo = s_uhi + CH_TAB_U;
  float* s_b = reinterpret_cast<float*>(s_ulo + CH_TAB_U);      // [C]
  float* s_asum = s_b + CH_C;                                   // [8]
  float* s_cc = s_asum + 8;                                     // [8]
  float* s_vec = reinterpret_cast<float*>(lds + CH_NSLOT * CH_CHUNK + CH_TAB_PAD);   // [product][bias | svec][C]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, half = lane >> 5;
  const int tiles = a.rows_per_image / 128;
  const int img = blockIdx.x / tiles, tile = blockIdx.x - img * tiles;
  const int kind = a.nprog == 2 ? (img & 1) : 0;
  const int nops = a.nops[kind];
  const long long row = (long long)tile * 128 + wave * 32 + ql;          // within the image
  const unsigned lane16 = (unsigned)lane * 16u;
  const int ch_lane = 8 * half;                                           // + 16 ks + j
  CH_STAMP(0);

  auto row_ptr = [&](const void* base, long long img_stride, long long row_stride, int div) -> const T* {
    const int ii = div > 1 ? img / div : img;
    return reinterpret_cast<const T*>(base) + ii * img_stride + row * row_stride + ch_lane;
  };

  // ---- the rows first: a program's leading LOAD operations are issued before anything else -------------------------------
  // `xr` (the residual) lives until the FIRST product's epilogue consumes it -- that product is run outside the operation
  // loop below so that the registers are known to be free afterwards (inside the loop they would stay allocated: 80 of 256)
  frag_t xs[CH_KS], xr[CH_KS];
#pragma unroll
  for (int ks = 0; ks < CH_KS; ++ks) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { xs[ks][j] = (T)0.0f; xr[ks][j] = (T)0.0f; }
  }
  int op0 = 0, op_r = -1;                           // first operation of the loop below; the LOAD_R among the leading two
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (op0 == i && i < nops) {
      const mobi_chain_op& op = a.prog[kind][i];
      if (op.code == MOBI_CH_LOAD_S) {
        const T* p = row_ptr(op.p0, op.img_stride, op.row_stride, op.img_div);
#pragma unroll
        for (int ks = 0; ks < CH_KS; ++ks) xs[ks] = __builtin_bit_cast(frag_t, ld16(p + 16 * ks));
        op0 = i + 1;
      } else if (op.code == MOBI_CH_LOAD_R) {
        op_r = i;                                   // issued LAST (below): only the first epilogue needs these rows, their
        op0 = i + 1;                                // latency hides behind the first product
      }
    }
  }

  // ---- weight ring state (uniform) ------------------------------------------------------------------------------
  int pend = 0;            // chunks requested and not yet consumed
  int slot_c = 0;          // slot of the next chunk to consume
  int slot_q = 0;          // slot of the next chunk to request
  auto request = [&](const __amdgpu_buffer_rsrc_t& r, int chunk) {
#pragma unroll
    for (int i = 0; i < CH_RPW; ++i) {
      const int p = wave + 4 * i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(ring + slot_q * CH_CHUNK + p * 1024), 16, lane16,
                                               chunk * CH_CHUNK + p * 1024, 0, 0);
    }
    slot_q = slot_q + 1 == CH_NSLOT ? 0 : slot_q + 1;
    ++pend;
  };

  // ---- staging: the adapter's table image by LDS-DMA, the first product's first chunks, every product's bias / row-sum
  // vectors through registers (an epilogue that loaded them from global memory waited, load by load, behind its own stores)
  if (a.ad_image) {
    const __amdgpu_buffer_rsrc_t rt = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(a.ad_image)) + (long long)img * CH_TAB_PAD, 0, CH_TAB_PAD, 0x00020000);
#pragma unroll
    for (int i = 0; i < (CH_TAB_PAD / 1024 + 3) / 4; ++i) {
      const int p = wave + 4 * i;
      if (p < CH_TAB_PAD / 1024)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rt, (lds_ptr_t)(s_ahi + p * 1024), 16, lane16, p * 1024, 0, 0);
    }
  }
  {
    int k = 0;
    for (int oi = 0; oi < nops; ++oi) {
      const mobi_chain_op& op = a.prog[kind][oi];
      if (op.code != MOBI_CH_PRODUCT) continue;
      if (k == 0) {
        const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(op.p0), 0, CH_NCH * CH_CHUNK, 0x00020000);
        for (int j = 0; j < CH_D; ++j) request(r0, j);
      }
      if (wave == (k & 3)) {
        // product k's bias (and row sums) -> LDS by DMA too, 1280 bytes each: one full request and one of 16 lanes (masked
        // lanes leave their LDS bytes alone); wave k issues them
        const float* bias = op.bias + (long long)(op.bias_img_div > 1 ? img / op.bias_img_div : img) * op.bias_img_stride;
        const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, CH_C * 4, 0x00020000);
        float* dstv = s_vec + (2 * k) * CH_C;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)dstv, 16, lane16, 0, 0, 0);
        if (lane < (CH_C * 4 - 1024) / 16)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(dstv + 256), 16, lane16, 1024, 0, 0);
        if (op.flags & MOBI_CH_FOLD) {
          const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.svec), 0, CH_C * 4, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dstv + CH_C), 16, lane16, 0, 0, 0);
          if (lane < (CH_C * 4 - 1024) / 16)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(dstv + CH_C + 256), 16, lane16, 1024, 0, 0);
        }
      }
      ++k;
    }
  }
  if (op0 < nops && a.prog[kind][op0].code == MOBI_CH_AFFINE_S && wave == 3) {
    const mobi_chain_op& op = a.prog[kind][op0];
    float* dstv = s_vec + (2 * CH_MAXP) * CH_C;
    const __amdgpu_buffer_rsrc_t r_sc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.bias + (long long)img * CH_C), 0, CH_C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_sh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(op.svec + (long long)img * CH_C), 0, CH_C * 4, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_sc, (lds_ptr_t)dstv, 16, lane16, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r_sh, (lds_ptr_t)(dstv + CH_C), 16, lane16, 0, 0, 0);
    if (lane < (CH_C * 4 - 1024) / 16) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_sc, (lds_ptr_t)(dstv + 256), 16, lane16, 1024, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r_sh, (lds_ptr_t)(dstv + CH_C + 256), 16, lane16, 1024, 0, 0);
    }
  }
  // Everything requested so far (the operand rows, tables, vectors, the first chunks) has landed for every wave behind
  // this barrier -- a product starts with ITS chunk 0 in LDS, the invariant of the chunk loop -- except the residual rows,
  // which are requested last and left in flight (CH_KS loads per lane, the youngest).
  const bool r_pending = op_r >= 0;
  if (r_pending) {
    const mobi_chain_op& op = a.prog[kind][op_r];
    const T* p = row_ptr(op.p0, op.img_stride, op.row_stride, op.img_div);
#pragma unroll
    for (int ks = 0; ks < CH_KS; ++ks) xr[ks] = __builtin_bit_cast(frag_t, ld16(p + 16 * ks));
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(CH_KS) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  CH_STAMP(1);

  float rs = 1.0f, cs = 0.0f;

  // ---- a product's epilogue --------------------------------------------------------------------------------------
  auto epilogue = [&](auto fold_, auto resid_, auto to_s_, auto store_, const mobi_chain_op& op, const f32x16 (&acc)[CH_MT], int pk,
                      const frag_t* res) {
    constexpr bool FOLD = decltype(fold_)::value, RESID = decltype(resid_)::value, TO_S = decltype(to_s_)::value,
                   STORE = decltype(store_)::value;
    const float* bias = s_vec + (2 * pk) * CH_C + ch_lane;
    const float* sv = s_vec + (2 * pk + 1) * CH_C + ch_lane;
    if (MOBI_CHAIN_DBG & 1) return;
    T* dst = STORE ? const_cast<T*>(row_ptr(op.dst, op.dst_img_stride, op.dst_row_stride, op.dst_img_div)) : nullptr;
    // the vectors of step ks + 1 are read from LDS while step ks is computed: one step of prefetch, no more (left alone the
    // scheduler hoists every read of the loop: 160 registers the kernel does not have)
    f32x4 nb0 = *reinterpret_cast<const f32x4*>(bias), nb1 = *reinterpret_cast<const f32x4*>(bias + 4);
    f32x4 ns0 = f32x4{0.f, 0.f, 0.f, 0.f}, ns1 = ns0;
    if (FOLD) { ns0 = *reinterpret_cast<const f32x4*>(sv); ns1 = *reinterpret_cast<const f32x4*>(sv + 4); }
#pragma unroll
    for (int ks = 0; ks < CH_KS; ++ks) {
      const int m = ks >> 1, o = 8 * (ks & 1);
      const f32x4 b0 = nb0, b1 = nb1, s0 = ns0, s1 = ns1;
      if (ks + 1 < CH_KS) {
        nb0 = *reinterpret_cast<const f32x4*>(bias + 16 * (ks + 1));
        nb1 = *reinterpret_cast<const f32x4*>(bias + 16 * (ks + 1) + 4);
        if (FOLD) {
          ns0 = *reinterpret_cast<const f32x4*>(sv + 16 * (ks + 1));
          ns1 = *reinterpret_cast<const f32x4*>(sv + 16 * (ks + 1) + 4);
        }
      }
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = acc[m][o + j];
      if (FOLD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = rs * v[j] + cs * s0[j];
          v[4 + j] = rs * v[4 + j] + cs * s1[j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] += b0[j];
        v[4 + j] += b1[j];
      }
      if (RESID) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)res[ks][j];
      }
      const u32x4 packed = pack8<T>(v);
      if (TO_S) xs[ks] = __builtin_bit_cast(frag_t, packed);
      if (STORE) st16(dst + 16 * ks, packed);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- the two-key adapter on the row state ------------------------------------------------------------------------
  auto adapter = [&](auto store_, const mobi_chain_op& op) {
    constexpr bool STORE = decltype(store_)::value;
    if (MOBI_CHAIN_DBG & 2) return;
    // x + b + sum_h sigmoid(rstd (x . a_h - mean sum a_h) + c_h) u_h  (mobi_two_key_adapter; tables split hi + lo)
    f32x16 lgh, lgl;
#pragma unroll
    for (int r = 0; r < 16; ++r) { lgh[r] = 0.f; lgl[r] = 0.f; }
    float sx = 0.f, sxx = 0.f;
    // (the table offsets are re-derived HERE, behind an opaque move: hoisted out of the operation loop the per-step
    //  addresses stay allocated through every product -- and spill)
    int qlo = ql, hl = half;
    asm volatile("" : "+v"(qlo), "+v"(hl));
    const int a_off = (qlo < 8 ? qlo : 8) * CH_AST + 16 * hl;
    frag_t nah = __builtin_bit_cast(frag_t, ld16(s_ahi + a_off)), nal = __builtin_bit_cast(frag_t, ld16(s_alo + a_off));
#pragma unroll
    for (int ks = 0; ks < CH_KS; ++ks) {
      const frag_t ah = nah, al = nal;
      if (ks + 1 < CH_KS) {                         // one step of prefetch, no more (see the epilogue)
        nah = __builtin_bit_cast(frag_t, ld16(s_ahi + a_off + 32 * (ks + 1)));
        nal = __builtin_bit_cast(frag_t, ld16(s_alo + a_off + 32 * (ks + 1)));
      }
      lgh = mfma32(ah, xs[ks], lgh);
      lgl = mfma32(al, xs[ks], lgl);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float v = (float)xs[ks][j]; sx += v; sxx += v * v; }
      __builtin_amdgcn_sched_barrier(0);
    }
    // (the PACKED rows are what stays in registers: without the pins the compiler keeps the unpacked floats of the
    //  statistics alive for the final add -- twice the registers)
#pragma unroll
    for (int ks = 0; ks < CH_KS; ++ks) {
      u32x4 w = __builtin_bit_cast(u32x4, xs[ks]);
      asm volatile("" : "+v"(w));
      xs[ks] = __builtin_bit_cast(frag_t, w);
    }
    sx += __shfl_xor(sx, 32, 64);
    sxx += __shfl_xor(sxx, 32, 64);
    const float mean = sx * (1.0f / CH_C);
    float var = sxx * (1.0f / CH_C) - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = rsqrtf(var + a.ad_eps);
    // lane (ql, half) holds heads 4 half + j in registers j < 4
    float g[4], gh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float z = rstd * ((lgh[j] + lgl[j]) - mean * s_asum[4 * hl + j]) + s_cc[4 * hl + j];
      g[j] = 1.0f / (1.0f + __expf(-z));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) gh[j] = __shfl_down(g[j], 32, 64);          // heads 4..7 to the half-0 lanes (k = 0..7)
    frag_t gf, gl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gf[j] = hl == 0 ? (T)g[j] : (T)0.0f;
      gf[4 + j] = hl == 0 ? (T)gh[j] : (T)0.0f;
      gl[j] = hl == 0 ? (T)(g[j] - (float)gf[j]) : (T)0.0f;
      gl[4 + j] = hl == 0 ? (T)(gh[j] - (float)gf[4 + j]) : (T)0.0f;
    }
    // update: A row i of tile m is channel 32 m + tau(i); k = heads (lanes of half 1 supply zeros)
    const int ti = (qlo & 0x13) | ((qlo & 4) << 1) | ((qlo & 8) >> 1);
    const int u_off = hl == 0 ? ti * 16 : CH_C * 16;
    const int u_step = hl == 0 ? 32 * 16 : 0;
    T* dst = STORE ? const_cast<T*>(row_ptr(op.dst, op.dst_img_stride, op.dst_row_stride, op.dst_img_div)) : nullptr;
    auto upd = [&](int m) {
      const frag_t uh = __builtin_bit_cast(frag_t, ld16(s_uhi + u_off + m * u_step));
      const frag_t ul = __builtin_bit_cast(frag_t, ld16(s_ulo + u_off + m * u_step));
      f32x16 d;
#pragma unroll
      for (int r = 0; r < 16; ++r) d[r] = 0.f;
      d = mfma32(uh, gf, d);
      d = mfma32(ul, gf, d);
      d = mfma32(uh, gl, d);
      return d;
    };
    f32x16 dn = upd(0);
#pragma unroll
    for (int m = 0; m < CH_MT; ++m) {
      const f32x16 d = dn;
      if (m + 1 < CH_MT) dn = upd(m + 1);           // the next tile's products run under this tile's vector work
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const int ks = 2 * m + k2;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(s_b + 16 * ks + 8 * hl);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(s_b + 16 * ks + 8 * hl + 4);
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[j] = (float)xs[ks][j] + b0[j] + d[8 * k2 + j];
          v[4 + j] = (float)xs[ks][4 + j] + b1[j] + d[8 * k2 + 4 + j];
        }
        const u32x4 packed = pack8<T>(v);
        xs[ks] = __builtin_bit_cast(frag_t, packed);
        if (STORE) st16(dst + 16 * ks, packed);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- one product: the chunk loop and its epilogue ---------------------------------------------------------------------
  int pk = 0;                                       // products done
  auto product = [&](auto first_, const mobi_chain_op& op, const frag_t* res) {
    constexpr bool FIRST = decltype(first_)::value;  // the program's first product: the only one that may take the residual
    // Invariant on entry: this product's chunk 0 has landed for every wave.  At the top of chunk c one barrier says
    // "chunk c + 1 has landed everywhere" (each wave first waits for its own requests of it) and "chunk c - 1 has been
    // read by everyone", so the fragment queue runs on ACROSS the chunk boundary (no exposed LDS round trip per chunk)
    // and the requests of chunk c + NSLOT - 1 go into chunk c - 1's slot.
    const __amdgpu_buffer_rsrc_t r0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(op.p0), 0, CH_NCH * CH_CHUNK, 0x00020000);
    const bool has_next = op.p1 != nullptr;
    const __amdgpu_buffer_rsrc_t r1 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(has_next ? op.p1 : op.p0), 0, CH_NCH * CH_CHUNK, 0x00020000);
    f32x16 acc[CH_MT];
#pragma unroll
    for (int m = 0; m < CH_MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const unsigned char* base = ring + slot_c * CH_CHUNK + lane16;
    frag_t fq[CH_QD];
#pragma unroll
    for (int i = 0; i < CH_QD; ++i) fq[i] = __builtin_bit_cast(frag_t, ld16(base + i * 1024));
#pragma unroll
    for (int c = 0; c < CH_NCH; ++c) {
      // my requests of the next chunk in the stream have landed: in steady state (a product follows) the five chunks
      // c .. c + 4 are out, of which c + 2 .. c + 4 may stay in flight; the last product counts down
      if (FIRST && c < CH_D - 1 && r_pending && has_next) {
        // (issue order: chunks 0 .. D - 1, the residual rows, then chunk D + c' behind barrier c': until chunk D is the one
        //  waited for, the CH_KS residual loads are among the youngest and stay in flight)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CH_RPW + CH_KS) : "memory");
      } else if (has_next) wait_vm(3);
      else if (c + 1 < CH_NCH) wait_vm(CH_NCH - c - 2 < 3 ? CH_NCH - c - 2 : 3);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(CH_QD) : "memory");
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      --pend;
      const bool req = c + CH_D < CH_NCH || has_next;             // chunk c + D of the stream exists: request it, spread
      const __amdgpu_buffer_rsrc_t& rq = c + CH_D < CH_NCH ? r0 : r1;   // over the step (a request costs 60 - 100 issue cycles)
      const int rq_chunk = c + CH_D < CH_NCH ? c + CH_D : c + CH_D - CH_NCH;
      const int rq_slot = slot_q;
      slot_c = slot_c + 1 == CH_NSLOT ? 0 : slot_c + 1;
      const unsigned char* nbase = ring + slot_c * CH_CHUNK + lane16;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < CH_FR; ++f) {
        const int kk = f / CH_MT, m = f - kk * CH_MT;
        if (!(MOBI_CHAIN_DBG & 8)) acc[m] = mfma32(fq[f % CH_QD], xs[2 * c + kk], acc[m]);
        if (f + CH_QD < CH_FR) fq[f % CH_QD] = __builtin_bit_cast(frag_t, ld16(base + (f + CH_QD) * 1024));
        else if (c + 1 < CH_NCH) fq[f % CH_QD] = __builtin_bit_cast(frag_t, ld16(nbase + (f + CH_QD - CH_FR) * 1024));
        if (req && f % 4 == 1 && f / 4 < CH_RPW) {
          const int pz = wave + 4 * (f / 4);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_ptr_t)(ring + rq_slot * CH_CHUNK + pz * 1024), 16, lane16,
                                                   rq_chunk * CH_CHUNK + pz * 1024, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (req) {
        slot_q = slot_q + 1 == CH_NSLOT ? 0 : slot_q + 1;
        ++pend;
      }
      base = nbase;
    }
    CH_STAMP(20 + pk);
    typedef std::true_type Y;
    typedef std::false_type N;
    switch (op.flags & 15) {
      case MOBI_CH_STORE: epilogue(N{}, N{}, N{}, Y{}, op, acc, pk, res); break;
      case MOBI_CH_FOLD | MOBI_CH_STORE: epilogue(Y{}, N{}, N{}, Y{}, op, acc, pk, res); break;
      case MOBI_CH_RESID | MOBI_CH_TO_S: if constexpr (FIRST) epilogue(N{}, Y{}, Y{}, N{}, op, acc, pk, res); break;
      case MOBI_CH_RESID | MOBI_CH_TO_S | MOBI_CH_STORE: if constexpr (FIRST) epilogue(N{}, Y{}, Y{}, Y{}, op, acc, pk, res); break;
      case MOBI_CH_RESID | MOBI_CH_STORE: if constexpr (FIRST) epilogue(N{}, Y{}, N{}, Y{}, op, acc, pk, res); break;
      case MOBI_CH_TO_S | MOBI_CH_STORE: epilogue(N{}, N{}, Y{}, Y{}, op, acc, pk, res); break;
      default: break;                             // (mobi_row_chain rejects every other combination)
    }
    CH_STAMP(16 + pk);                            // (the product's MFMA loop ended at stamp 20 + pk)
    ++pk;
  };

  // ---- the program ---------------------------------------------------------------------------------------------------
  // head: [AFFINE_S] and the first product, straight-line (see `xr` above)
  if (op0 < nops && a.prog[kind][op0].code == MOBI_CH_AFFINE_S) {
    const mobi_chain_op& op = a.prog[kind][op0];
    const float* sc = s_vec + (2 * CH_MAXP) * CH_C + ch_lane;            // this image's scale / shift, staged above
    const float* sh = sc + CH_C;
    (void)op;
#pragma unroll
    for (int ks = 0; ks < CH_KS; ++ks) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(sc + 16 * ks), a1 = *reinterpret_cast<const f32x4*>(sc + 16 * ks + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh + 16 * ks), h1 = *reinterpret_cast<const f32x4*>(sh + 16 * ks + 4);
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = (float)xs[ks][j] * a0[j] + h0[j];
        v[4 + j] = (float)xs[ks][4 + j] * a1[j] + h1[j];
      }
      xs[ks] = __builtin_bit_cast(frag_t, pack8<T>(v));
    }
    CH_STAMP(2 + op0);
    ++op0;
  }
  if (op0 < nops && a.prog[kind][op0].code == MOBI_CH_PRODUCT) {
    product(std::true_type{}, a.prog[kind][op0], xr);
    CH_STAMP(2 + op0);
    ++op0;
  }
  for (int oi = op0; oi < nops; ++oi) {
    const mobi_chain_op& op = a.prog[kind][oi];
    const int code = op.code;
    if (code == MOBI_CH_LOAD_S) {
      const T* p = row_ptr(op.p0, op.img_stride, op.row_stride, op.img_div);
#pragma unroll
      for (int ks = 0; ks < CH_KS; ++ks) xs[ks] = __builtin_bit_cast(frag_t, ld16(p + 16 * ks));
    } else if (code == MOBI_CH_STORE_S) {
      T* dst = const_cast<T*>(row_ptr(op.dst, op.dst_img_stride, op.dst_row_stride, op.dst_img_div));
#pragma unroll
      for (int ks = 0; ks < CH_KS; ++ks) st16(dst + 16 * ks, __builtin_bit_cast(u32x4, xs[ks]));
    } else if (code == MOBI_CH_ROWSTATS && !(MOBI_CHAIN_DBG & 4)) {
      // mean, then the variance about the mean from the registers (layernorm_kernel's arithmetic); a row's channels lie
      // in lanes ql and ql + 32
      float s1 = 0.f;
#pragma unroll
      for (int ks = 0; ks < CH_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) s1 += (float)xs[ks][j];
      s1 += __shfl_xor(s1, 32, 64);
      const float mean = s1 * (1.0f / CH_C);
#pragma unroll
      for (int ks = 0; ks < CH_KS; ++ks) {          // (pins: see the adapter)
        u32x4 w = __builtin_bit_cast(u32x4, xs[ks]);
        asm volatile("" : "+v"(w));
        xs[ks] = __builtin_bit_cast(frag_t, w);
      }
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < CH_KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (float)xs[ks][j] - mean; q += d * d; }
      q += __shfl_xor(q, 32, 64);
      rs = rsqrtf(q * (1.0f / CH_C) + op.eps);
      cs = -rs * mean;
    } else if (code == MOBI_CH_ADAPTER) {
      adapter(std::true_type{}, op);                // (always with its store: a second instantiation costs the registers)
    } else if (code == MOBI_CH_PRODUCT) {
      product(std::false_type{}, op, nullptr);
    }
    CH_STAMP(2 + oi);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace mobi

#ifdef MOBI_CHAIN_STAMPS
extern "C" int mobi_chain_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mobi::g_chain_stamps), sizeof(unsigned long long) * 64);
}
#endif

extern "C" size_t mobi_row_chain_weight_bytes(int32_t channels) {
  return channels == mobi::CH_C ? (size_t)mobi::CH_NCH * mobi::CH_CHUNK : 0;
}

extern "C" size_t mobi_row_chain_adapter_image_bytes(int32_t channels) {
  return channels == mobi::CH_C ? (size_t)mobi::CH_TAB_PAD : 0;
}

extern "C" int mobi_row_chain_adapter_image(const float* a, const float* c, const float* u, const float* b, int32_t images,
                                            int32_t heads, int32_t channels, int32_t dtype, void* out, void* stream) {
  using namespace mobi;
  if (!a || !c || !u || !b || !out || images <= 0 || heads <= 0 || heads > 8) return MOBI_ERR_ARG;
  if (channels != CH_C) return MOBI_ERR_UNSUPPORTED;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(out)) & 15) return MOBI_ERR_ALIGN;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned char* o = reinterpret_cast<unsigned char*>(out);
  if (dtype == MOBI_F16) hipLaunchKernelGGL((chain_adapter_image_kernel<f16_t>), dim3(images), dim3(256), 0, st, a, c, u, b, heads, o);
  else hipLaunchKernelGGL((chain_adapter_image_kernel<bf16_t>), dim3(images), dim3(256), 0, st, a, c, u, b, heads, o);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_groupnorm_scale_shift(const void* x, const float* gamma, const float* beta, float eps, float* scale, float* shift,
                                          int32_t images, int32_t hw, int32_t channels, int32_t dtype, void* stream) {
  using namespace mobi;
  if (!x || !gamma || !beta || !scale || !shift || images <= 0 || hw <= 0) return MOBI_ERR_ARG;
  if (dtype != MOBI_F16 && dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (channels <= 0 || channels % 64) return MOBI_ERR_UNSUPPORTED;        // (pairs of channels per 4-byte load: C / 32 even)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(32, images);
  if (dtype == MOBI_F16) hipLaunchKernelGGL((gn_scale_shift_kernel<f16_t>), grid, dim3(256), 0, st, reinterpret_cast<const f16_t*>(x), gamma, beta, eps, scale, shift, hw, channels);
  else hipLaunchKernelGGL((gn_scale_shift_kernel<bf16_t>), grid, dim3(256), 0, st, reinterpret_cast<const bf16_t*>(x), gamma, beta, eps, scale, shift, hw, channels);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

extern "C" int mobi_row_chain_supported(int32_t channels, int32_t rows_per_image) {
  return channels == mobi::CH_C && rows_per_image > 0 && rows_per_image % 128 == 0;
}

extern "C" int mobi_row_chain(const mobi_row_chain_params* p, void* stream) {
  using namespace mobi;
  if (!p) return MOBI_ERR_ARG;
  if (p->dtype != MOBI_F16 && p->dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (!mobi_row_chain_supported(p->channels, p->rows_per_image)) return MOBI_ERR_UNSUPPORTED;
  if (p->images <= 0 || (p->nprog != 1 && p->nprog != 2)) return MOBI_ERR_ARG;
  if (p->nprog == 2 && (p->images & 1)) return MOBI_ERR_ARG;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  for (int k = 0; k < p->nprog; ++k) {
    if (p->nops[k] <= 0 || p->nops[k] > MOBI_CHAIN_MAX_OPS) return MOBI_ERR_ARG;
    const mobi_chain_op* next_product = nullptr;
    int products = 0;
    // program shape the kernel runs: [LOAD_S | LOAD_R in the first two places] [AFFINE_S] [first PRODUCT, the only one that
    // may take the residual] then LOAD_S / ROWSTATS / ADAPTER / PRODUCT / STORE_S in any order
    int head = 0;
    while (head < 2 && head < p->nops[k] && (p->prog[k][head].code == MOBI_CH_LOAD_S || p->prog[k][head].code == MOBI_CH_LOAD_R)) ++head;
    if (head < p->nops[k] && p->prog[k][head].code == MOBI_CH_AFFINE_S) ++head;
    const int first_product = head < p->nops[k] && p->prog[k][head].code == MOBI_CH_PRODUCT ? head : -1;
    for (int i = p->nops[k] - 1; i >= 0; --i) {
      const mobi_chain_op& op = p->prog[k][i];
      if ((op.code == MOBI_CH_LOAD_R && i >= 2) || (op.code == MOBI_CH_AFFINE_S && i >= head) ||
          (op.code == MOBI_CH_PRODUCT && (op.flags & MOBI_CH_RESID) && i != first_product)) return MOBI_ERR_UNSUPPORTED;
      switch (op.code) {
        case MOBI_CH_LOAD_S: case MOBI_CH_LOAD_R:
          if (!op.p0) return MOBI_ERR_ARG;
          if (!al16(op.p0) || (op.img_stride & 7) || (op.row_stride & 7) || op.row_stride < CH_C) return MOBI_ERR_ALIGN;
          break;
        case MOBI_CH_AFFINE_S:
          if (!op.bias || !op.svec) return MOBI_ERR_ARG;
          if (!al16(op.bias) || !al16(op.svec)) return MOBI_ERR_ALIGN;
          break;
        case MOBI_CH_ROWSTATS: break;
        case MOBI_CH_STORE_S:
          if (!op.dst) return MOBI_ERR_ARG;
          if (!al16(op.dst) || (op.dst_img_stride & 7) || (op.dst_row_stride & 7) || op.dst_row_stride < CH_C) return MOBI_ERR_ALIGN;
          break;
        case MOBI_CH_ADAPTER:
          if (!p->ad_image) return MOBI_ERR_ARG;
          if (!al16(p->ad_image)) return MOBI_ERR_ALIGN;
          if (!(op.flags & MOBI_CH_STORE) || !op.dst) return MOBI_ERR_ARG;          // the adapter writes its result
          if (!al16(op.dst) || (op.dst_img_stride & 7) || (op.dst_row_stride & 7) || op.dst_row_stride < CH_C) return MOBI_ERR_ALIGN;
          break;
        case MOBI_CH_PRODUCT: {
          if (!op.p0 || !al16(op.p0) || !al16(op.p1)) return op.p0 ? MOBI_ERR_ALIGN : MOBI_ERR_ARG;
          // the prefetch pointer must be the next product's image (or NULL on the last one): the ring is consumed in order
          if (op.p1 != (next_product ? next_product->p0 : nullptr)) return MOBI_ERR_ARG;
          const int f = op.flags & 15;
          if (f != MOBI_CH_STORE && f != (MOBI_CH_FOLD | MOBI_CH_STORE) && f != (MOBI_CH_RESID | MOBI_CH_TO_S) &&
              f != (MOBI_CH_RESID | MOBI_CH_TO_S | MOBI_CH_STORE) && f != (MOBI_CH_RESID | MOBI_CH_STORE) &&
              f != (MOBI_CH_TO_S | MOBI_CH_STORE)) return MOBI_ERR_UNSUPPORTED;
          if ((f & MOBI_CH_FOLD) && (!op.svec || !al16(op.svec))) return MOBI_ERR_ARG;
          if (!op.bias) return MOBI_ERR_ARG;                      // (a vector of zeros where the layer has none)
          if (!al16(op.bias) || (op.bias_img_stride & 3)) return MOBI_ERR_ALIGN;
          if ((f & MOBI_CH_STORE) && (!op.dst || !al16(op.dst) || (op.dst_img_stride & 7) || (op.dst_row_stride & 7) ||
                                      op.dst_row_stride < CH_C)) return op.dst ? MOBI_ERR_ALIGN : MOBI_ERR_ARG;
          next_product = &op;
          if (++products > CH_MAXP) return MOBI_ERR_UNSUPPORTED;
          break;
        }
        default: return MOBI_ERR_ARG;
      }
    }
  }
  const long long blocks = (long long)p->images * (p->rows_per_image / 128);
  if (blocks > 0x7fffffffLL) return MOBI_ERR_UNSUPPORTED;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p->dtype == MOBI_F16) hipLaunchKernelGGL((row_chain_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, st, *p);
  else hipLaunchKernelGGL((row_chain_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, *p);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}
