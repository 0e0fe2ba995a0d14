// Fused attention for gfx950:  out = softmax(q k^T * scale) v  without materialising
// the score matrix (CrossAttention.forward, attention.py:171-194).
//
// Block = 4 waves, each wave owns 32 query rows; the block walks the keys in tiles
// of 64 that all four waves share through LDS.  MFMA 32x32x16, fp32 accumulate.
//   S^T[key][q]  = K[key][:] . Q[q][:]      (A = K rows from LDS, B = Q held in registers)
// so every lane owns ONE query column: the running max / sum of the online softmax
// are per-lane scalars and the only cross-lane step is one xor-32 shuffle.
//   O^T[d][q]   += V^T[d][key] . P^T[key][q]
// P^T is the S^T accumulator itself, converted to T in registers and fed straight back
// as the B operand (no LDS round trip), so the A fragments of V^T must follow the accumulator's
// permuted key order
//   key(j, half) = 16 s + 8 (j >> 2) + 4 half + (j & 3).
// Two V layouts:
//   v_layout 1 (production): V row-major [token][channel], e.g. the v columns of a stacked q|k|v projection.
//     The tile is staged like K ([key][d] rows) and the A fragments come from gfx950's transposing LDS read
//     ds_read_b64_tr_b16 (a 4-key x 16-channel block per 16 lanes, delivered channel-major).
//   v_layout 0: V already transposed [channel][token] (written by a projection's transposed epilogue); the LDS
//     image is key-permuted so that a fragment is one 16-byte read.
// LDS row strides are odd multiples of the access width (bank-conflict free reads).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "tuning.h"

#ifndef MOBI_ATTN_DBG
#define MOBI_ATTN_DBG 0    // diagnosis only (wrong results): bit 0 = no exp, bit 1 = K / V tiles loaded once, bit 2 = no P.V,
#endif                     // bit 3 = no S = K.Q^T
#ifndef MOBI_ATTN_PRIO
#define MOBI_ATTN_PRIO 1   // s_setprio 1 around the two MFMA clusters of a key tile (measured -5 %: 649 vs 683 us)
#endif
#ifndef MOBI_ATTN_LOAD_LATE
#define MOBI_ATTN_LOAD_LATE 1 // next tile requested behind the S MFMAs instead of at the top of the loop (-3.7 %: 653 vs 678 us)
#endif
#ifndef MOBI_ATTN_STORE_LATE
#define MOBI_ATTN_STORE_LATE 0
#endif
#ifndef MOBI_ATTN_FUSE_EXP
#define MOBI_ATTN_FUSE_EXP 0  // 1: exponentials of a 16-key group issued right before the P.V MFMAs that consume them
                              // (measured SLOWER at 4 waves per SIMD: 676-681 vs 632-639 us on [16, 4096 x 4096, 8 x 40])
#endif
#ifndef MOBI_ATTN_DBUF
#define MOBI_ATTN_DBUF 1   // two LDS images of the K / V tile: one barrier per key tile, the next tile is written while
#endif                     // this one is still being multiplied (A/B: -DMOBI_ATTN_DBUF=0)

#ifndef MOBI_ATTN_RDBG
#define MOBI_ATTN_RDBG 0   // attention_rows_kernel, diagnosis only (wrong results): bit 0 = no exp / pack / OR, bit 1 = K / V tiles
#endif                     // staged once, bit 2 = no P.V (reads + MFMAs), bit 3 = no S (reads + MFMAs), bit 4 = no barrier
#ifndef MOBI_ATTN_PDBG
#define MOBI_ATTN_PDBG 0   // attention_pipe_kernel, diagnosis only (wrong results): bit 0 = tiles staged in the prologue only,
#endif                     // bit 1 = no barrier, bit 2 = no OR test / exact path
#ifndef MOBI_ATTN_HP_FD
#define MOBI_ATTN_HP_FD 4  // attention_hp_kernel: operand fragments requested this many MFMA gaps ahead
#endif
#ifndef MOBI_ATTN_HP_NW6
#define MOBI_ATTN_HP_NW6 1 // attention_hp_kernel: six-wave blocks (three waves per SIMD) for the big launches; 0: eight-wave
#endif
#ifndef MOBI_ATTN_RVAR
#define MOBI_ATTN_RVAR 0   // attention_rows_kernel A/B variants: bit 1 = bf16 storage keeps the per-half OR test
#endif

namespace mobi {

// Workgroup -> (query block, head, image).  The hardware deals consecutive workgroup ids (x fastest) round-robin to the eight
// XCDs; left alone, the query blocks of one (head, image) -- which all stream the same K / V panel -- land on eight different
// L2s and each fetches the panel from HBM (measured: 378 MB per [16, 4096 x 4096, 8 x 40] launch against 168 MB of
// operands).  xcd_remap hands every XCD a contiguous range of the (image, head, query block) list instead.
__device__ __forceinline__ void attn_block(int xcd_map, int& qblk, int& head, int& img) {
  const int gx = (int)gridDim.x, gy = (int)gridDim.y;
  const int lin = (int)blockIdx.x + gx * ((int)blockIdx.y + gy * (int)blockIdx.z);
  const int L = xcd_map ? xcd_remap(lin, gx * gy * (int)gridDim.z) : lin;
  qblk = L % gx;
  const int hi = L / gx;
  head = hi % gy;
  img = hi / gy;
}

struct AttnArgs {
  const void* q; long long q_img; int q_row;
  const void* k; long long k_img; int k_row;
  const void* vt; long long vt_img; int vt_row;      // v_layout 1: V rows [tk][vt_row]; 0: V^T rows [C][vt_row]
  void* out; long long out_img; int out_row;
  int heads, dh, tq, tk;
  float cexp;                // scale * log2(e), or 1 when q already carries it (mobi_attention_params.q_log2_scaled)
  int xcd_map;               // workgroup ids re-dealt so that the query blocks of a (head, image) share one XCD's L2
};

// VVEC: every V^T row start is 16-byte aligned (tk % 8 == 0 rows), the production case; the generic
// variant loads ragged V^T rows element-wise.  WPS = waves per SIMD the register budget is held to.
// VMODE 0: V^T, generic (ragged rows, element loads)   1: V^T, 16-byte aligned rows   2: V row-major (tr reads)
// NW = waves per block (4 or 8): 32 NW queries share every staged K / V tile, so with 8 a thread requests and stores half
// as many 16-byte pieces per key tile (two 8-wave blocks per CU instead of four 4-wave blocks: same occupancy).
template <typename T, int KS, int VMODE, int WPS, int NW = 4>
__global__ __launch_bounds__(64 * NW, (WPS * 4) / NW) void attention_kernel(const AttnArgs a) {
  constexpr int NTHR = 64 * NW;
  typedef typename Vec8<T>::type frag_t;
  constexpr bool VVEC = VMODE == 1;
  constexpr bool VROWS = VMODE == 2;
  constexpr int DT = (KS + 1) / 2;                 // 32-row tiles of the head dim for P.V
  constexpr int KSTR = KS * 32 + 16;               // bytes per K row in LDS (odd multiple of 16)
  // V^T image: 144-byte rows of 64 permuted keys.  V-rows image: [key][DT*32 channels], row stride an odd
  // multiple of 64 bytes: the 4 key rows x 64 bytes a 32-lane half reads per ds_read_b64_tr_b16 then cover all
  // 64 banks exactly once.
  constexpr int VSTR = VROWS ? ((DT & 1) ? DT * 64 : DT * 64 + 64) : 144;
  constexpr int K_BYTES = 64 * KSTR;
  constexpr int V_BYTES = VROWS ? 64 * VSTR : DT * 32 * VSTR;
  constexpr int KP = (64 * KS * 2 + NTHR - 1) / NTHR;    // 16-byte K pieces per thread
  constexpr int VP = VROWS ? KP : (DT * 32 * 8 + NTHR - 1) / NTHR;    // 16-byte V pieces per thread
  constexpr int NBUF = MOBI_ATTN_DBUF ? 2 : 1;
  constexpr int IMG_BYTES = K_BYTES + V_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NBUF * IMG_BYTES];
  unsigned char* ldsK = lds;
  unsigned char* ldsV = lds + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  int qblk, head, img;
  attn_block(a.xcd_map, qblk, head, img);
  const int q0 = qblk * (32 * NW) + wave * 32;
  const int dh = a.dh;
  // When the head dim leaves a padded row in the P.V tile (dh < 32*DT: dh = 40, 80, ...; not 64, 160) that
  // row of V^T is filled with ones, so the matrix core accumulates the softmax denominator
  // l[q] = sum_k P[k][q] for free (row dh of O^T) instead of 32 VALU adds per key tile.
  const bool ONES = dh < DT * 32;

  const T* __restrict__ qp = reinterpret_cast<const T*>(a.q) + img * a.q_img + head * dh;
  const T* __restrict__ kp = reinterpret_cast<const T*>(a.k) + img * a.k_img + head * dh;
  const T* __restrict__ vp = reinterpret_cast<const T*>(a.vt) + img * a.vt_img +
                             (VROWS ? (long long)head * dh : (long long)head * dh * a.vt_row);
  T* __restrict__ op = reinterpret_cast<T*>(a.out) + img * a.out_img + head * dh;

  // Q fragments: lane (q = ql, half) holds Q[q][ks*16 + 8*half .. +8)
  frag_t qf[KS];
  {
    const int qrow = q0 + ql;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + half * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qrow < a.tq && c < dh) v = ld16(qp + (long long)qrow * a.q_row + c);
      qf[ks] = __builtin_bit_cast(frag_t, v);
    }
  }

  // K / V^T staging.
  // VVEC (production) path: buffer loads with a per-block descriptor (SGPRs) and a 32-bit byte offset that is a
  // per-thread constant plus a wave-uniform term for the key tile -- one VALU add per 16-byte piece; rows past
  // the last key, padded head-dim columns and padded V^T rows fall outside the descriptor's extent (or carry an
  // out-of-range constant) and read as zero in hardware.  Loads are unconditional, so vmcnt can count them.
  // V^T keys are stored PERMUTED inside every group of 16 (key 8a+4b+c -> position 8b+4a+c): the 8 keys a lane
  // needs for one k-step of P.V (the accumulator's key order) are then 16 contiguous bytes.
  u32x4 kr[KP], vr[VP];
  unsigned koff[KP], voff[VP];
  unsigned ones_m = 0;                                  // pieces of the all-ones row (softmax denominator)
  constexpr unsigned OOB = 0x80000000u;
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    const int p = tid + NTHR * i;
    const int row = p / (KS * 2), pc = p - row * (KS * 2);
    koff[i] = (row < 64 && pc * 8 < dh) ? (unsigned)(row * a.k_row + pc * 8) * 2u : OOB;
  }
#pragma unroll
  for (int i = 0; i < VP; ++i) {
    const int p = tid + NTHR * i;
    if (VROWS) {                                        // same piece map as K: (key row, 8 channels)
      const int row = p / (KS * 2), pc = p - row * (KS * 2);
      voff[i] = (row < 64 && pc * 8 < dh) ? (unsigned)(row * a.vt_row + pc * 8) * 2u : OOB;
    } else {
      const int row = p >> 3, pc = p & 7;
      voff[i] = row < dh ? (unsigned)(row * a.vt_row + pc * 8) * 2u : OOB;
      if (ONES && row == dh) ones_m |= 1u << i;
    }
  }
  const int k_bytes = ((a.tk - 1) * a.k_row + dh) * 2;
  const int v_bytes = VROWS ? ((a.tk - 1) * a.vt_row + dh) * 2 : ((dh - 1) * a.vt_row + a.tk) * 2;
  if (VROWS) {
    // channels [dh, DT*32) of every key row are written ONCE: zero, except channel dh = 1.0 (the denominator
    // column); the tile stores below only touch channels < dh
    for (int p = tid; p < 64 * DT * 4; p += NTHR) {
      const int row = p / (DT * 4), pc = p - row * (DT * 4);
      if (pc * 8 >= dh) {
        typename Vec8<T>::type e;
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = (T)0.0f;
        if (pc * 8 == dh) e[0] = (T)1.0f;
#pragma unroll
        for (int b = 0; b < NBUF; ++b) st16(ldsV + b * IMG_BYTES + row * VSTR + pc * 16, __builtin_bit_cast(u32x4, e));
      }
    }
  }
  auto load_tile = [&](int key0) {
    if (VROWS) {
      const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(kp), 0, k_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(vp), 0, v_bytes, 0x00020000);
      const unsigned ku = (unsigned)key0 * (unsigned)a.k_row * 2u, vu = (unsigned)key0 * (unsigned)a.vt_row * 2u;
#pragma unroll
      for (int i = 0; i < KP; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, koff[i] + ku, 0, 0);
#pragma unroll
      for (int i = 0; i < VP; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, voff[i] + vu, 0, 0);
    } else if (VVEC) {
      const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(kp), 0, k_bytes, 0x00020000);
      const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(vp), 0, v_bytes, 0x00020000);
      const unsigned ku = (unsigned)key0 * (unsigned)a.k_row * 2u, vu = (unsigned)key0 * 2u;    // wave-uniform
#pragma unroll
      for (int i = 0; i < KP; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, koff[i] + ku, 0, 0);
#pragma unroll
      for (int i = 0; i < VP; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, voff[i] + vu, 0, 0);
      if (key0 + 64 > a.tk) {             // ragged last tile: keys >= tk of a V^T row alias the next row -> zero them
#pragma unroll
        for (int i = 0; i < VP; ++i) {
          const int pc = (tid + NTHR * i) & 7;
          if (key0 + pc * 8 >= a.tk) vr[i] = u32x4{0u, 0u, 0u, 0u};
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < KP; ++i) {
        const int p = tid + NTHR * i;
        const int row = p / (KS * 2), pc = p - row * (KS * 2);
        const bool ok = row < 64 && key0 + row < a.tk && pc * 8 < dh;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ok) v = ld16(kp + (long long)(key0 + row) * a.k_row + pc * 8);
        kr[i] = v;
      }
#pragma unroll
      for (int i = 0; i < VP; ++i) {
        const int p = tid + NTHR * i;
        const int row = p >> 3, pc = p & 7;
        typename Vec8<T>::type e;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = row < dh && key0 + pc * 8 + j < a.tk;
          e[j] = ok ? vp[(long long)row * a.vt_row + key0 + pc * 8 + j] : (T)0.0f;
        }
        vr[i] = __builtin_bit_cast(u32x4, e);
      }
    }
  };
  auto store_tile = [&](int boff) {
    typename Vec8<T>::type one8;
#pragma unroll
    for (int j = 0; j < 8; ++j) one8[j] = (T)1.0f;
    const u32x4 ones = __builtin_bit_cast(u32x4, one8);
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int p = tid + NTHR * i;
      const int row = p / (KS * 2), pc = p - row * (KS * 2);
      if (row < 64) st16(ldsK + boff + row * KSTR + pc * 16, kr[i]);
    }
    if (VROWS) {
#pragma unroll
      for (int i = 0; i < VP; ++i) {
        const int p = tid + NTHR * i;
        const int row = p / (KS * 2), pc = p - row * (KS * 2);
        if (row < 64 && pc * 8 < dh) st16(ldsV + boff + row * VSTR + pc * 16, vr[i]);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) {
      const int p = tid + NTHR * i;
      const int row = p >> 3, pc = p & 7;
      if (row < DT * 32) {
        // the ones row may also cover keys >= tk: their P is exactly 0, so the denominator is unaffected
        const u32x4 v = (ones_m >> i) & 1u ? ones : vr[i];
        unsigned char* d = ldsV + boff + row * VSTR + (pc >> 1) * 32 + (pc & 1) * 8;
        *reinterpret_cast<u32x2*>(d) = u32x2{v[0], v[1]};             // keys 8a + (0..3)     -> pos 4a + ..
        *reinterpret_cast<u32x2*>(d + 16) = u32x2{v[2], v[3]};        // keys 8a + 4 + (0..3) -> pos 8 + 4a + ..
      }
    }
  };

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float cexp = a.cexp;

  const int ntiles = (a.tk + 63) / 64;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * 64;
    const bool more = t + 1 < ntiles;
    const int boff = MOBI_ATTN_DBUF ? (t & 1) * IMG_BYTES : 0;       // image of this tile
#if !MOBI_ATTN_LOAD_LATE
#if MOBI_ATTN_DBG & 2
    if (more && t == 0) load_tile(key0 + 64);
#else
    if (more) load_tile(key0 + 64);
#endif
#endif

    // ---- S^T = K . Q^T for two 32-key sub-tiles --------------------------------
    f32x16 s[2];
#if MOBI_ATTN_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
      const unsigned char* kb = ldsK + boff + (kt * 32 + ql) * KSTR + half * 16;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        frag_t kf = __builtin_bit_cast(frag_t, ld16(kb + ks * 32));
#if MOBI_ATTN_DBG & 8
        asm volatile("" :: "v"(kf));
        s[kt][ks] += (float)kf[0];
#else
        s[kt] = mfma32(kf, qf[ks], s[kt]);
#endif
      }
    }
#if MOBI_ATTN_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#if MOBI_ATTN_LOAD_LATE
    if (more) load_tile(key0 + 64);          // behind the S MFMAs: the requests do not delay the K fragment reads
#endif
    // ---- online softmax (one query column per lane) -----------------------------
    // p = exp2(s * c - m * c), c = scale * log2(e): one FMA + one v_exp per score; the running max is
    // kept in the raw score domain (scale > 0).  Masking only on the ragged last tile; O / l are rescaled
    // only when some lane's max actually moved (wave-uniform branch).
    float mx = -INFINITY;
    if (key0 + 64 <= a.tk) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kt][r]);
    } else {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float v = key < a.tk ? s[kt][r] : -INFINITY;
          s[kt][r] = v;
          mx = fmaxf(mx, v);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    if (!__all(m_new == m_run)) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cexp);   // first tile: exp2(-inf) = 0
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      m_run = m_new;
    }
    const float mc = m_run * cexp;
#if !MOBI_ATTN_FUSE_EXP
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
#if MOBI_ATTN_DBG & 1
      for (int r = 0; r < 16; ++r) s[kt][r] = __builtin_fmaf(s[kt][r], cexp, -mc);
#else
      for (int r = 0; r < 16; ++r) s[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][r], cexp, -mc));
#endif
    if (!ONES) {
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) psum += s[kt][r];
      l_run += psum;
    }
#endif

#if MOBI_ATTN_DBUF && !MOBI_ATTN_STORE_LATE
    // the next tile goes into the OTHER image: every wave left it at the barrier that ended the previous step, and
    // its loads were issued a whole S / softmax phase ago
    if (more) store_tile(IMG_BYTES - boff);
#endif
    // ---- O^T += V^T . P^T --------------------------------------------------------
#if MOBI_ATTN_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        frag_t pf;
#if MOBI_ATTN_FUSE_EXP
        // A/B: the exponentials of a 16-key group right before the MFMAs that consume them, so that groups 1..3 issue
        // behind the P.V MFMAs of the group before (tools/probes/mfma_valu_overlap.hip); slower in this kernel
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#if MOBI_ATTN_DBG & 1
          const float pv = __builtin_fmaf(s[kt][st * 8 + j], cexp, -mc);
#else
          const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kt][st * 8 + j], cexp, -mc));
#endif
          if (!ONES) l_run += pv;
          pf[j] = (T)pv;
        }
#else
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (T)s[kt][st * 8 + j];
#endif
        if constexpr (VROWS) {
          // lane 4q+p of each 16-lane group addresses key row q, channels 4p..4p+3 of the group's 16-channel block
          // (block = channels 32 d + 16 (group & 1)); it receives channel (lane & 15) of the 4 keys.  Two reads:
          // keys base + 4 half + (0..3) and + 8, the accumulator's key order.
          typedef __attribute__((address_space(3))) s16x4* lds4_t;
          const int l16 = lane & 15, grp = lane >> 4;
          const unsigned char* vb = ldsV + boff + (kt * 32 + st * 16 + 4 * half + (l16 >> 2)) * VSTR +
                                    (16 * (grp & 1) + 4 * (l16 & 3)) * 2;
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64 + 8 * VSTR));
            const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#if MOBI_ATTN_DBG & 4
            asm volatile("" :: "v"(both), "v"(pf));
#else
            o[d] = mfma32(__builtin_bit_cast(frag_t, both), pf, o[d]);
#endif
          }
        } else {
          const unsigned char* vb = ldsV + boff + ql * VSTR + (kt * 32 + st * 16 + half * 8) * 2;
#pragma unroll
          for (int d = 0; d < DT; ++d) {
            frag_t vf = __builtin_bit_cast(frag_t, ld16(vb + d * 32 * VSTR));
            o[d] = mfma32(vf, pf, o[d]);
          }
        }
      }
#if MOBI_ATTN_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#if MOBI_ATTN_DBUF && MOBI_ATTN_STORE_LATE
    if (more) store_tile(IMG_BYTES - boff);   // behind the P.V MFMAs: the loads had the whole softmax to land
#endif
    __syncthreads();                  // every wave is done with this tile's LDS image (and wrote the next one)
#if !MOBI_ATTN_DBUF
    if (more) {
      store_tile(0);
      __syncthreads();
    }
#endif
  }

  // ---- normalise and store: lane holds O^T[d][q] for d = 32 dt + 8 g + 4 half + (0..3) -------
  if (ONES) {      // row dh of O^T sits in lane-half 0, register (dh % 32) / 2 of tile dh / 32
    float lsum = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (d * 32 + g * 8 == dh) lsum = o[d][g * 4];
    l_run = half == 0 ? lsum : 0.f;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + ql;
  if (qrow < a.tq) {
    T* orow = op + (long long)qrow * a.out_row;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = d * 32 + g * 8 + half * 4;
        if (d0 < dh) {
          float f[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) f[j] = o[d][g * 4 + j] * inv;
          *reinterpret_cast<u32x2*>(orow + d0) = pack4<T>(f);
        }
      }
  }
}

// =========================================================================================================
// attention_rows_kernel: the production kernel for V row-major (v_layout 1), head dims up to 80.
//
// Why a second kernel: SQ counters of attention_kernel on [16, 4096 x 4096, 8 x 40] show a 32-query x 64-key wave tile
// costing ~1,100 cycles of instruction ISSUE on its SIMD (16 v_max3, 31 v_fma, 33 v_exp, 16 v_cvt_pk, 16 v_mov_b64, 18
// s_nop, 22 LDS reads, staging) around 448 cycles of matrix pipe, and tools/probes/mfma_fill.hip shows that on this chip a
// wave's vector instructions do issue in the shadow of its MFMAs: the loop is bound by the NUMBER of vector instructions.
// This kernel keeps the tiling and the two LDS images of attention_kernel and removes vector instructions:
//   * Q is multiplied by scale * log2(e) once, so an MFMA result is already an exponent of two;
//   * the shift by the running maximum rides on the MFMA's C operand (sixteen registers holding -shift), so a score needs
//     no FMA and the S accumulators no zeroing;
//   * the shift is NOT the exact running maximum: it stays B below it... (B = bias) and is only raised when some
//     probability of the tile reaches 2.0 -- one bit (bit 14) of the packed 16-bit word in BOTH storage types, so the test is
//     an OR over the packed words (v_or3_b32: 8 per tile instead of 16 v_max3 + the per-tile rescale decision).  Softmax is
//     shift-invariant, so the result is the same function; the shifted probabilities sit at most at 2.0 and typically at
//     2^-B (bf16 keeps its 8 bits over the whole f32 exponent range; f16 stays normal down to 2^-14, B = 4).
//   * the exact path (maximum of the tile, rescale of O, probabilities recomputed from the kept scores) runs on the first
//     tile and whenever the test fires (a wave-uniform branch).
// Per wave tile: 14 MFMA, 32 v_exp, 16 v_cvt_pk, 8 v_or3 + the LDS reads and the staging of the next tile.
// =========================================================================================================
template <typename T> struct AttnBias;
template <> struct AttnBias<bf16_t> { static constexpr float value = 8.0f; };
template <> struct AttnBias<f16_t> { static constexpr float value = 4.0f; };

template <typename T>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef T T2 __attribute__((ext_vector_type(2)));
  T2 v;
  v[0] = (T)lo;
  v[1] = (T)hi;
  return __builtin_bit_cast(unsigned, v);
}

// QSH: the head dim leaves a padded channel in the last k-step (dh % 16 == 8: 8, 24, 40, 72); the shift then rides in
// that channel of Q' against a column of ones in the K image instead of sixteen C-operand registers.
// H16 (padded head dims with an odd number of k-steps, i.e. dh = 40 / 72: the last 32-channel block of O^T holds 8 channels and
// the denominator): that block's P.V runs on MFMA 16x16x32 -- D[16 channels][16 queries], two per 32-key half, 16 cycles each
// instead of two 32-cycle 32x32x16 on a block that is 3/4 padding (12 instead of 14 MFMA-equivalents per wave tile at dh = 40).
// P arrives in the score product's lane order (lane = query l & 31, its 16 keys of the half in 8 packed registers); the
// 16x16x32 B operand wants lane = query l & 15 with four lane groups of 8 keys: v_permlane16_swap of the half's first four
// packed registers against its last four gives both operands (queries 0-15 and 16-31), each lane group holding the keys
// 16 (G & 1) + 4 (G >> 1) + (e & 3) + 8 (e >> 2) -- the rows the V^T fragment's transposed reads then address.
// MEASURED (tools/ab_attn_h16.sh, profiles/r04_ab_attn_h16.txt): correct (every attention test passes on it) and SLOWER --
// 485-487 against 474-476 us on [16, 4096 x 4096, 8 x 40], 20.72 against 20.52 ms per step: the loop's vector time equals its
// matrix time (r03 counters), and the eight swaps (+ the moves around them) cost the vector side more than the two
// MFMA-equivalents save the matrix side.  Off by default (MOBI_ATTN_H16=1 selects it), kept for the record and the A/B.
template <typename T, int KS, int NW, bool QSH, bool H16 = false>
__global__ __launch_bounds__(64 * NW, (KS <= 2 && (QSH || (KS & 1))) ? 4 : KS == 2 ? 3 : (KS == 3 && NW == 8 && QSH) ? 4 : KS == 3 ? 3 : 2)
void attention_rows_kernel(const AttnArgs a) {
  constexpr int NTHR = 64 * NW;
  typedef typename Vec8<T>::type frag_t;
  static_assert(!H16 || (QSH && (KS & 1) && KS >= 3), "H16: dh % 32 == 8, at least one full 32-channel block");
  constexpr int DT = (KS + 1) / 2;
  constexpr int DT32 = H16 ? DT - 1 : DT;      // 32-channel blocks of O^T on MFMA 32x32x16
  constexpr int KSTR = KS * 32 + 16;
  constexpr int VSTR = (DT & 1) ? DT * 64 : DT * 64 + 64;
  constexpr int K_BYTES = 64 * KSTR, V_BYTES = 64 * VSTR, IMG_BYTES = K_BYTES + V_BYTES;
  constexpr int KP = (64 * KS * 2 + NTHR - 1) / NTHR;      // 16-byte pieces per thread, K tile and V tile alike
  constexpr float BIAS = AttnBias<T>::value;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * IMG_BYTES];
  unsigned char* ldsK = lds;
  unsigned char* ldsV = lds + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  int qblk, head, img;
  attn_block(a.xcd_map, qblk, head, img);
  const int q0 = qblk * (32 * NW) + wave * 32;
  const int dh = a.dh;
  // a padded channel of the P.V tile carries the denominator: dh is KS*16 - 8 (QSH) or KS*16, DT*32 is (KS + 1)/2 * 32
  constexpr bool ONES = QSH || (KS & 1);
  // (no test on padded head dims, both storage types: the first pass keeps the shift of the first 32 keys and an overflow --
  //  a probability beyond fp32's / fp16's range -- shows in the denominator; fp16 has 2^20 of headroom above that shift,
  //  i.e. a later score may exceed the first 32 keys' maximum by 13.9 nats before the block repeats its pass with the test.
  //  The other instantiations would not fit their register budget with both passes; RVAR bit 1 keeps the test everywhere, A/B)
  constexpr bool TEST = !QSH || (MOBI_ATTN_RVAR & 2) || (std::is_same<T, f16_t>::value && (MOBI_ATTN_RVAR & 4));

  const T* __restrict__ qp = reinterpret_cast<const T*>(a.q) + img * a.q_img + head * dh;
  const T* __restrict__ kp = reinterpret_cast<const T*>(a.k) + img * a.k_img + head * dh;
  const T* __restrict__ vp = reinterpret_cast<const T*>(a.vt) + img * a.vt_img + (long long)head * dh;
  T* __restrict__ op = reinterpret_cast<T*>(a.out) + img * a.out_img + head * dh;

  // Q fragments, scaled once: an MFMA result is then the exponent of two of the probability
  frag_t qf[KS];
  {
    const float cexp = a.cexp;
    const int qrow = q0 + ql;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + half * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qrow < a.tq && c < dh) v = ld16(qp + (long long)qrow * a.q_row + c);
      frag_t f = __builtin_bit_cast(frag_t, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (T)((float)f[j] * cexp);          // cexp == 1 (q_log2_scaled): q bit for bit
      qf[ks] = f;
    }
  }

  u32x4 kr[KP], vr[KP];
  unsigned koff[KP], voff[KP];
  constexpr unsigned OOB = 0x80000000u;
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    const int p = tid + NTHR * i;
    const int row = p / (KS * 2), pc = p - row * (KS * 2);
    const bool on = row < 64 && pc * 8 < dh;
    koff[i] = on ? (unsigned)(row * a.k_row + pc * 8) * 2u : OOB;
    voff[i] = on ? (unsigned)(row * a.vt_row + pc * 8) * 2u : OOB;
  }
  const int k_bytes = ((a.tk - 1) * a.k_row + dh) * 2;
  const int v_bytes = ((a.tk - 1) * a.vt_row + dh) * 2;
  // channels [dh, KS*16) of every K row are written ONCE: zero, except channel dh = 1.0 when the shift rides in Q'
  for (int p = tid; p < 64 * KS * 2; p += NTHR) {
    const int row = p / (KS * 2), pc = p - row * (KS * 2);
    if (pc * 8 >= dh) {
      frag_t e;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (T)0.0f;
      if (QSH && pc * 8 == dh) e[0] = (T)1.0f;
#pragma unroll
      for (int b = 0; b < 2; ++b) st16(ldsK + b * IMG_BYTES + row * KSTR + pc * 16, __builtin_bit_cast(u32x4, e));
    }
  }
  // channels [dh, DT*32) of every V row are written ONCE: zero, except channel dh = 1.0 (the denominator column)
  for (int p = tid; p < 64 * DT * 4; p += NTHR) {
    const int row = p / (DT * 4), pc = p - row * (DT * 4);
    if (pc * 8 >= dh) {
      frag_t e;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (T)0.0f;
      if (pc * 8 == dh) e[0] = (T)1.0f;
#pragma unroll
      for (int b = 0; b < 2; ++b) st16(ldsV + b * IMG_BYTES + row * VSTR + pc * 16, __builtin_bit_cast(u32x4, e));
    }
  }
  auto load_tile = [&](int key0) {
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(kp), 0, k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(vp), 0, v_bytes, 0x00020000);
    // the key tile's byte offset rides in the scalar offset of the request (wave-uniform): no per-tile address arithmetic
    const unsigned ku = (unsigned)key0 * (unsigned)a.k_row * 2u, vu = (unsigned)key0 * (unsigned)a.vt_row * 2u;
#pragma unroll
    for (int i = 0; i < KP; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, koff[i], ku, 0);
#pragma unroll
    for (int i = 0; i < KP; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, voff[i], vu, 0);
  };
  auto store_tile = [&](int boff) {
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int p = tid + NTHR * i;
      const int row = p / (KS * 2), pc = p - row * (KS * 2);
      if (row < 64 && pc * 8 < dh) {
        st16(ldsK + boff + row * KSTR + pc * 16, kr[i]);
        st16(ldsV + boff + row * VSTR + pc * 16, vr[i]);
      }
    }
  };

  f32x16 o[DT32];
#pragma unroll
  for (int d = 0; d < DT32; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  // H16: channels 32 DT32 + 4 (lane >> 4) + (0..3) of query (lane & 15) [ox] / 16 + (lane & 15) [oy]
  f32x4 ox = {0.f, 0.f, 0.f, 0.f}, oy = {0.f, 0.f, 0.f, 0.f};
  f32x16 cm;                               // minus the shift of this lane's query column, in all sixteen registers
#pragma unroll
  for (int r = 0; r < 16; ++r) cm[r] = 0.f;
  // QSH: lanes of the half that holds channel dh keep minus the shift in element dh % 8 (= 0) of the last Q' fragment
  const bool qsh_lane = QSH && half == ((dh >> 3) & 1);
  float l_run = 0.f;

  const int ntiles = (a.tk + 63) / 64;
  typedef __attribute__((address_space(3))) s16x4* lds4_t;
  const int l16 = lane & 15, grp = lane >> 4;
  const int k_lane = ql * KSTR + half * 16;
  const int v_lane = (4 * half + (l16 >> 2)) * VSTR + (16 * (grp & 1) + 4 * (l16 & 3)) * 2;
  const int v16_lane = (16 * (grp & 1) + 4 * (grp >> 1) + (l16 >> 2)) * VSTR + (32 * DT32 + 4 * (l16 & 3)) * 2;
  // V^T fragment of the 16-channel block for the 32 keys of half kt (A operand of MFMA 16x16x32)
  auto read_v16 = [&](int boff, int kt) -> frag_t {
    const unsigned char* vb = ldsV + boff + v16_lane + kt * 32 * VSTR;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + 8 * VSTR));
    return __builtin_bit_cast(frag_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  // the half's packed P as the two 16x16x32 B operands (queries 0-15, 16-31)
  auto swap_p = [&](const unsigned (&pwh)[8], u32x4& px, u32x4& py) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const auto r = __builtin_amdgcn_permlane16_swap(pwh[i], pwh[4 + i], false, false);
      px[i] = r[0];
      py[i] = r[1];
    }
  };
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // S'^T of one 32-key half from its LDS image (KS MFMAs), keys past the end masked
  auto mask_half = [&](int key0, int kt, f32x16& sx) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= a.tk) sx[r] = -INFINITY;
    }
  };
  auto scores_half = [&](auto ragged_tag, int boff, int key0, int kt, f32x16& sx) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const frag_t kf = __builtin_bit_cast(frag_t, ld16(ldsK + boff + k_lane + kt * 32 * KSTR + ks * 32));
      sx = mfma32(kf, qf[ks], ks == 0 ? (QSH ? zero16 : cm) : sx);
    }
    if (decltype(ragged_tag)::value) mask_half(key0, kt, sx);
  };
  // the exact path for one half: its maximum per query column, the shift raised to (maximum - BIAS) -- set outright when
  // nothing has been accumulated yet --, O rescaled, the other half's pending scores moved, the half's probabilities
  auto raise_shift = [&](f32x16& sx, auto first_tag, auto pending_tag, f32x16& pending, unsigned (&pwh)[8]) {
    constexpr bool first = decltype(first_tag)::value, has_pending = decltype(pending_tag)::value;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sx[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float d = mx + BIAS;
    if (!first) d = fmaxf(d, 0.f);           // the shift only rises
    if (QSH) {
      // the shift is a value of the storage type: take the step its rounding actually makes
      const float old_q = __shfl(qsh_lane ? (float)qf[KS - 1][0] : 0.f, ql + 32, 64);
      const T new_q = (T)(old_q - d);
      d = old_q - (float)new_q;
      if (qsh_lane) qf[KS - 1][0] = new_q;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) cm[r] -= d;
    }
    if (!first) {
      const float alpha = __builtin_amdgcn_exp2f(-d);
      l_run *= alpha;
#pragma unroll
      for (int dd = 0; dd < DT32; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
      if constexpr (H16) {
        const float ax = __shfl(alpha, l16, 64), ay = __shfl(alpha, 16 + l16, 64);      // lanes 0-31 hold the queries' factors
#pragma unroll
        for (int r = 0; r < 4; ++r) { ox[r] *= ax; oy[r] *= ay; }
      }
    }
    if (has_pending) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pending[r] -= d;
    }
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e0 = __builtin_amdgcn_exp2f(sx[2 * i] - d), e1 = __builtin_amdgcn_exp2f(sx[2 * i + 1] - d);
      pwh[i] = pack2<T>(e0, e1);
      if (!ONES) psum += e0 + e1;
    }
    if (!ONES) l_run += psum;
  };

  // A wave that waits on the busy matrix pipe holds up the vector issue of the SIMD's other waves too
  // (tools/probes/mfma_fill.hip, 'split' rows), so MFMAs are issued with vector work of the SAME wave behind each:
  //   S'(half 0)                      KS MFMAs alone
  //   S'(half 1)      KS MFMAs, each followed by a share of half 0's exponentials / packing / OR test
  //   P.V(half 0)   2 DT MFMAs, each followed by a share of half 1's
  //   P.V(half 1)   2 DT MFMAs alone
  // Each half has its own OR test right behind its exponentials (same basic block: with a branch in between hipcc sinks
  // the exponentials out of the MFMA gaps) and before its P.V; O stays in the old shift's scale until a test fails.
  // The ragged last tile runs a second copy of the body (RAGGED) with the masking, so the common one has no branch for it.
  auto tile = [&](auto ragged_tag, auto test_tag, int t) {
    constexpr bool RAGGED = decltype(ragged_tag)::value, TESTED = decltype(test_tag)::value;
    const int key0 = t * 64;
    const int boff = (t & 1) * IMG_BYTES;
    const bool more = t + 1 < ntiles;
    if (!QSH) asm volatile("" : "+v"(cm)); // sixteen live registers, not one value re-broadcast per MFMA
    f32x16 s[2];
    unsigned pw[2][8];
    if constexpr (RAGGED) {
      // the last, ragged tile: one half at a time on the exact path (masked scores, their maximum, probabilities), no
      // speculation and no interleaving -- it runs once
      s16x8 vr_[2][DT32];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        scores_half(ragged_tag, boff, key0, kt, s[0]);
        raise_shift(s[0], std::false_type{}, std::false_type{}, s[0], pw[0]);
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const unsigned char* vb = ldsV + boff + v_lane + (kt * 32 + st * 16) * VSTR;
#pragma unroll
          for (int d = 0; d < DT32; ++d) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64 + 8 * VSTR));
            vr_[st][d] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const u32x4 pu = {pw[0][st * 4], pw[0][st * 4 + 1], pw[0][st * 4 + 2], pw[0][st * 4 + 3]};
#pragma unroll
          for (int d = 0; d < DT32; ++d)
            o[d] = mfma32(__builtin_bit_cast(frag_t, vr_[st][d]), __builtin_bit_cast(frag_t, pu), o[d]);
        }
        if constexpr (H16) {
          u32x4 px, py;
          swap_p(pw[0], px, py);
          const frag_t va = read_v16(boff, kt);
          ox = mfma16(va, __builtin_bit_cast(frag_t, px), ox);
          oy = mfma16(va, __builtin_bit_cast(frag_t, py), oy);
        }
      }
      return;
    }
    frag_t kf1[KS];
    scores_half(ragged_tag, boff, key0, 0, s[0]);
    auto read_k1 = [&](int ks) { kf1[ks] = __builtin_bit_cast(frag_t, ld16(ldsK + boff + k_lane + 32 * KSTR + ks * 32)); };
    read_k1(0);
    if (KS > 1) read_k1(1);
    __builtin_amdgcn_sched_barrier(0);
    unsigned orr = 0u;
    float psum = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s[1] = mfma32(kf1[ks], qf[ks], ks == 0 ? (QSH ? zero16 : cm) : s[1]);
      if (ks + 2 < KS) read_k1(ks + 2);
#pragma unroll
      for (int i = (8 * ks) / KS; i < (8 * (ks + 1)) / KS; ++i) {
        const float e0 = __builtin_amdgcn_exp2f(s[0][2 * i]), e1 = __builtin_amdgcn_exp2f(s[0][2 * i + 1]);
        pw[0][i] = pack2<T>(e0, e1);
        if (TESTED) orr |= pw[0][i];
        if (!ONES) psum += e0 + e1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TESTED && !__all((orr & 0x40004000u) == 0u)) {
      scores_half(ragged_tag, boff, key0, 0, s[0]);
      raise_shift(s[0], std::false_type{}, std::true_type{}, s[1], pw[0]);
    } else if (!ONES) {
      l_run += psum;
    }
    if (more && !(MOBI_ATTN_RDBG & 2)) load_tile(key0 + 64);
    // V^T fragments one 16-key group (2 DT reads) ahead of the MFMAs that use them
    s16x8 vf[2][DT32];
    auto read_v = [&](int kt, int st) {
      const unsigned char* vb = ldsV + boff + v_lane + (kt * 32 + st * 16) * VSTR;
#pragma unroll
      for (int d = 0; d < DT32; ++d) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64 + 8 * VSTR));
        vf[st][d] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    read_v(0, 0);
    frag_t v16a, v16b;                       // H16: the 16-channel block's V^T fragments of the two halves
    u32x4 px, py;
    if constexpr (H16) v16a = read_v16(boff, 0);
    __builtin_amdgcn_sched_barrier(0);
    orr = 0u;
    psum = 0.f;
    // 2 DT MFMA slots in both forms; H16: the last two are the 16x16x32 pair (queries 0-15, 16-31) of the 16-channel block
#pragma unroll
    for (int m = 0; m < 2 * DT; ++m) {
      if (!H16 || m < 2 * DT32) {
        const int st = m / DT32, d = m - st * DT32;
        const u32x4 pu = {pw[0][st * 4], pw[0][st * 4 + 1], pw[0][st * 4 + 2], pw[0][st * 4 + 3]};
        o[d] = mfma32(__builtin_bit_cast(frag_t, vf[st][d]), __builtin_bit_cast(frag_t, pu), o[d]);
        if (m == 0) read_v(0, 1);
      } else if (m == 2 * DT32) {
        ox = mfma16(v16a, __builtin_bit_cast(frag_t, px), ox);
        v16b = read_v16(boff, 1);
      } else {
        oy = mfma16(v16a, __builtin_bit_cast(frag_t, py), oy);
      }
#pragma unroll
      for (int i = (8 * m) / (2 * DT); i < (8 * (m + 1)) / (2 * DT); ++i) {
        const float e0 = __builtin_amdgcn_exp2f(s[1][2 * i]), e1 = __builtin_amdgcn_exp2f(s[1][2 * i + 1]);
        pw[1][i] = pack2<T>(e0, e1);
        if (TESTED) orr |= pw[1][i];
        if (!ONES) psum += e0 + e1;
      }
      if (m == DT32) read_v(1, 0);
      if (H16 && m == 2 * DT32 - 1) swap_p(pw[0], px, py);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TESTED && !__all((orr & 0x40004000u) == 0u)) {
      scores_half(ragged_tag, boff, key0, 1, s[1]);
      raise_shift(s[1], std::false_type{}, std::false_type{}, s[1], pw[1]);
    } else if (!ONES) {
      l_run += psum;
    }
    if (more && !((MOBI_ATTN_RDBG & 2) && t > 0)) store_tile(IMG_BYTES - boff);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (H16) swap_p(pw[1], px, py);
#pragma unroll
    for (int m = 0; m < 2 * DT; ++m) {
      if (!H16 || m < 2 * DT32) {
        const int st = m / DT32, d = m - st * DT32;
        const u32x4 pu = {pw[1][st * 4], pw[1][st * 4 + 1], pw[1][st * 4 + 2], pw[1][st * 4 + 3]};
        o[d] = mfma32(__builtin_bit_cast(frag_t, vf[st][d]), __builtin_bit_cast(frag_t, pu), o[d]);
        if (m == 0) read_v(1, 1);
      } else if (m == 2 * DT32) {
        ox = mfma16(v16b, __builtin_bit_cast(frag_t, px), ox);
      } else {
        oy = mfma16(v16b, __builtin_bit_cast(frag_t, py), oy);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!(MOBI_ATTN_RDBG & 16)) __syncthreads();
  };
  // TEST (fp16 storage): the OR test per half-tile keeps every probability below 2.0 (fp16 ends at 65504).  bf16 storage has
  // fp32's exponent range: the shift of the first 32 keys stays for the whole row (softmax is shift-invariant, a probability
  // may be 2^100), the per-tile test, its 8 OR instructions and its branches are gone, and an overflow -- a score more than
  // ~120 octaves above that first maximum -- shows in the denominator (inf / nan / 0) at the end: the block then repeats its
  // pass WITH the test (block-uniform decision), which raises the shift as it goes.
  const int nfull = a.tk / 64;
  float l_tot = 0.f;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool tested = TEST || attempt == 1;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    {
      // the first shift, from the exact maximum of the first 32 keys (the loop then only tests; its first half-tile is
      // computed once more there, so that the loop body has no first-step special case for the compiler to hoist on)
      f32x16 s0;
      unsigned pw0[8];
      if (a.tk < 32) scores_half(std::true_type{}, 0, 0, 0, s0); else scores_half(std::false_type{}, 0, 0, 0, s0);
      raise_shift(s0, std::true_type{}, std::false_type{}, s0, pw0);
      if (!ONES) l_run = 0.f;
    }
    if (tested) {
      for (int t = 0; t < nfull; ++t) tile(std::false_type{}, std::true_type{}, t);
    } else {
      for (int t = 0; t < nfull; ++t) tile(std::false_type{}, std::false_type{}, t);
    }
    if (nfull < ntiles) tile(std::true_type{}, std::true_type{}, nfull);
    // the denominator: row dh of O^T (lane-half 0, register (dh % 32) / 2 of tile dh / 32) or the running sum
    if constexpr (H16) {
      // channel dh = 32 DT32 + 8: register 0 of lane group 2 (lanes 32-47) of ox (queries 0-15) / oy (16-31)
      const float dx = __shfl(ox[0], 32 + l16, 64), dy = __shfl(oy[0], 32 + l16, 64);
      l_tot = ql < 16 ? dx : dy;
    } else {
      if (ONES) {
        float lsum = 0.f;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            if (d * 32 + g * 8 == dh) lsum = o[d][g * 4];
        l_run = half == 0 ? lsum : 0.f;
      }
      l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    if (tested) break;
    const bool bad = !(l_tot > 0.f && l_tot < 1.0e30f);
    if (!__syncthreads_or(bad ? 1 : 0)) break;
#pragma unroll
    for (int d = 0; d < DT32; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) { ox[r] = 0.f; oy[r] = 0.f; }
#pragma unroll
    for (int r = 0; r < 16; ++r) cm[r] = 0.f;
    if (qsh_lane) qf[KS - 1][0] = (T)0.0f;
    l_run = 0.f;
  }

  // ---- normalise and store: lane holds O^T[d][q] for d = 32 dt + 8 g + 4 half + (0..3) -------
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + ql;
  if (qrow < a.tq) {
    T* orow = op + (long long)qrow * a.out_row;
#pragma unroll
    for (int d = 0; d < DT32; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = d * 32 + g * 8 + half * 4;
        if (d0 < dh) {
          float f[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) f[j] = o[d][g * 4 + j] * inv;
          *reinterpret_cast<u32x2*>(orow + d0) = pack4<T>(f);
        }
      }
  }
  if constexpr (H16) {
    // the 16-channel block: lane holds channels 32 DT32 + 4 (lane >> 4) + (0..3) of query l16 (ox) and 16 + l16 (oy)
    const float invx = 1.0f / __shfl(l_tot, l16, 64), invy = 1.0f / __shfl(l_tot, 16 + l16, 64);
    const int d0 = 32 * DT32 + 4 * grp;
    if (d0 < dh) {
      const int qx = q0 + l16, qy = q0 + 16 + l16;
      if (qx < a.tq) {
        const float f[4] = {ox[0] * invx, ox[1] * invx, ox[2] * invx, ox[3] * invx};
        *reinterpret_cast<u32x2*>(op + (long long)qx * a.out_row + d0) = pack4<T>(f);
      }
      if (qy < a.tq) {
        const float f[4] = {oy[0] * invy, oy[1] * invy, oy[2] * invy, oy[3] * invy};
        *reinterpret_cast<u32x2*>(op + (long long)qy * a.out_row + d0) = pack4<T>(f);
      }
    }
  }
}


#ifdef MOBI_DEV   // development build only: two software-pipelined forms of attention_rows_kernel, both measured SLOWER
// (profiles/r03_attention_lab_variants.txt; selectable with MOBI_ATTN_V3=2 / 3 for A/B runs, covered by the parity tests there)
// =========================================================================================================
// attention_hp_kernel: attention_rows_kernel's arithmetic and tiling as a half-tile software pipeline (see the comment at its
// main loop).  The production kernel for V row-major, head dims up to 80.
// =========================================================================================================
template <typename T, int KS, int NW, bool QSH>
__global__ __launch_bounds__(64 * NW, NW == 6 ? 3 : (KS <= 2 && (QSH || (KS & 1))) ? 4 : KS <= 3 ? 3 : 2)
void attention_hp_kernel(const AttnArgs a) {
  constexpr int NTHR = 64 * NW;
  typedef typename Vec8<T>::type frag_t;
  constexpr int DT = (KS + 1) / 2;
  constexpr int KSTR = KS * 32 + 16;
  constexpr int VSTR = (DT & 1) ? DT * 64 : DT * 64 + 64;
  constexpr int K_BYTES = 64 * KSTR, V_BYTES = 64 * VSTR, IMG_BYTES = K_BYTES + V_BYTES;
  constexpr int KP = (64 * KS * 2 + NTHR - 1) / NTHR;      // 16-byte pieces per thread, K tile and V tile alike
  constexpr float BIAS = AttnBias<T>::value;
  constexpr int NIMG = 3;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NIMG * IMG_BYTES];
  unsigned char* ldsK = lds;
  unsigned char* ldsV = lds + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  int qblk, head, img;
  attn_block(a.xcd_map, qblk, head, img);
  const int q0 = qblk * (32 * NW) + wave * 32;
  const int dh = a.dh;
  // a padded channel of the P.V tile carries the denominator: dh is KS*16 - 8 (QSH) or KS*16, DT*32 is (KS + 1)/2 * 32
  constexpr bool ONES = QSH || (KS & 1);

  const T* __restrict__ qp = reinterpret_cast<const T*>(a.q) + img * a.q_img + head * dh;
  const T* __restrict__ kp = reinterpret_cast<const T*>(a.k) + img * a.k_img + head * dh;
  const T* __restrict__ vp = reinterpret_cast<const T*>(a.vt) + img * a.vt_img + (long long)head * dh;
  T* __restrict__ op = reinterpret_cast<T*>(a.out) + img * a.out_img + head * dh;

  // Q fragments, scaled once: an MFMA result is then the exponent of two of the probability
  frag_t qf[KS];
  {
    const float cexp = a.cexp;
    const int qrow = q0 + ql;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + half * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qrow < a.tq && c < dh) v = ld16(qp + (long long)qrow * a.q_row + c);
      frag_t f = __builtin_bit_cast(frag_t, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (T)((float)f[j] * cexp);          // cexp == 1 (q_log2_scaled): q bit for bit
      qf[ks] = f;
    }
  }

  u32x4 kr[KP], vr[KP];
  unsigned koff[KP], voff[KP];
  constexpr unsigned OOB = 0x80000000u;
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    const int p = tid + NTHR * i;
    const int row = p / (KS * 2), pc = p - row * (KS * 2);
    const bool on = row < 64 && pc * 8 < dh;
    koff[i] = on ? (unsigned)(row * a.k_row + pc * 8) * 2u : OOB;
    voff[i] = on ? (unsigned)(row * a.vt_row + pc * 8) * 2u : OOB;
  }
  const int k_bytes = ((a.tk - 1) * a.k_row + dh) * 2;
  const int v_bytes = ((a.tk - 1) * a.vt_row + dh) * 2;
  // channels [dh, KS*16) of every K row are written ONCE: zero, except channel dh = 1.0 when the shift rides in Q'
  for (int p = tid; p < 64 * KS * 2; p += NTHR) {
    const int row = p / (KS * 2), pc = p - row * (KS * 2);
    if (pc * 8 >= dh) {
      frag_t e;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (T)0.0f;
      if (QSH && pc * 8 == dh) e[0] = (T)1.0f;
#pragma unroll
      for (int b = 0; b < NIMG; ++b) st16(ldsK + b * IMG_BYTES + row * KSTR + pc * 16, __builtin_bit_cast(u32x4, e));
    }
  }
  // channels [dh, DT*32) of every V row are written ONCE: zero, except channel dh = 1.0 (the denominator column)
  for (int p = tid; p < 64 * DT * 4; p += NTHR) {
    const int row = p / (DT * 4), pc = p - row * (DT * 4);
    if (pc * 8 >= dh) {
      frag_t e;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (T)0.0f;
      if (pc * 8 == dh) e[0] = (T)1.0f;
#pragma unroll
      for (int b = 0; b < NIMG; ++b) st16(ldsV + b * IMG_BYTES + row * VSTR + pc * 16, __builtin_bit_cast(u32x4, e));
    }
  }
  auto load_tile = [&](int key0) {
    const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(kp), 0, k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(vp), 0, v_bytes, 0x00020000);
    // the key tile's byte offset rides in the scalar offset of the request (wave-uniform): no per-tile address arithmetic
    const unsigned ku = (unsigned)key0 * (unsigned)a.k_row * 2u, vu = (unsigned)key0 * (unsigned)a.vt_row * 2u;
#pragma unroll
    for (int i = 0; i < KP; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, koff[i], ku, 0);
#pragma unroll
    for (int i = 0; i < KP; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, voff[i], vu, 0);
  };
  auto store_tile = [&](int boff) {
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int p = tid + NTHR * i;
      const int row = p / (KS * 2), pc = p - row * (KS * 2);
      if (row < 64 && pc * 8 < dh) {
        st16(ldsK + boff + row * KSTR + pc * 16, kr[i]);
        st16(ldsV + boff + row * VSTR + pc * 16, vr[i]);
      }
    }
  };

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  f32x16 cm;                               // minus the shift of this lane's query column, in all sixteen registers
#pragma unroll
  for (int r = 0; r < 16; ++r) cm[r] = 0.f;
  // QSH: lanes of the half that holds channel dh keep minus the shift in element dh % 8 (= 0) of the last Q' fragment
  const bool qsh_lane = QSH && half == ((dh >> 3) & 1);
  float l_run = 0.f;

  const int ntiles = (a.tk + 63) / 64;
  typedef __attribute__((address_space(3))) s16x4* lds4_t;
  const int l16 = lane & 15, grp = lane >> 4;
  const int k_lane = ql * KSTR + half * 16;
  const int v_lane = (4 * half + (l16 >> 2)) * VSTR + (16 * (grp & 1) + 4 * (l16 & 3)) * 2;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // S'^T of one 32-key half from its LDS image (KS MFMAs), keys past the end masked
  auto mask_half = [&](int key0, int kt, f32x16& sx) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= a.tk) sx[r] = -INFINITY;
    }
  };
  auto scores_half = [&](auto ragged_tag, int boff, int key0, int kt, f32x16& sx) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const frag_t kf = __builtin_bit_cast(frag_t, ld16(ldsK + boff + k_lane + kt * 32 * KSTR + ks * 32));
      sx = mfma32(kf, qf[ks], ks == 0 ? (QSH ? zero16 : cm) : sx);
    }
    if (decltype(ragged_tag)::value) mask_half(key0, kt, sx);
  };
  // the exact path for one half: its maximum per query column, the shift raised to (maximum - BIAS) -- set outright when
  // nothing has been accumulated yet --, O rescaled, the other half's pending scores moved, the half's probabilities
  auto raise_shift = [&](f32x16& sx, auto first_tag, auto pending_tag, f32x16& pending, unsigned (&pwh)[8]) {
    constexpr bool first = decltype(first_tag)::value, has_pending = decltype(pending_tag)::value;
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sx[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float d = mx + BIAS;
    if (!first) d = fmaxf(d, 0.f);           // the shift only rises
    if (QSH) {
      // the shift is a value of the storage type: take the step its rounding actually makes
      const float old_q = __shfl(qsh_lane ? (float)qf[KS - 1][0] : 0.f, ql + 32, 64);
      const T new_q = (T)(old_q - d);
      d = old_q - (float)new_q;
      if (qsh_lane) qf[KS - 1][0] = new_q;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) cm[r] -= d;
    }
    if (!first) {
      const float alpha = __builtin_amdgcn_exp2f(-d);
      l_run *= alpha;
#pragma unroll
      for (int dd = 0; dd < DT; ++dd)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
    }
    if (has_pending) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pending[r] -= d;
    }
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float e0 = __builtin_amdgcn_exp2f(sx[2 * i] - d), e1 = __builtin_amdgcn_exp2f(sx[2 * i + 1] - d);
      pwh[i] = pack2<T>(e0, e1);
      if (!ONES) psum += e0 + e1;
    }
    if (!ONES) l_run += psum;
  };

  // Half-tile software pipeline: one HALF-STEP multiplies P.V of the previous 32-key half (2 DT MFMAs) and S' of the next
  // half (KS MFMAs) while the exponentials / packing / OR test of the current half run in their gaps -- no MFMA is issued
  // without vector work of the same wave behind it (a wave that waits on the busy matrix pipe holds up the vector issue of
  // the SIMD's other waves: tools/probes/mfma_fill.hip, 'split' rows; attention_rows_kernel leaves KS + 2 DT of its
  // 2 KS + 4 DT MFMAs per tile bare).  Same registers as attention_rows_kernel (two 16-register score halves, two packed
  // halves): four waves per SIMD.  Three LDS images: a half-step reads the previous tile's V and the next tile's K; ONE
  // barrier per tile; the next tile is requested from memory a whole tile ahead.
  auto first_shift = [&](f32x16& sx) {     // the first shift from the exact maximum of the first half; sx moved to it
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sx[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float d = mx + BIAS;
    if (QSH) {
      const T new_q = (T)(-d);
      d = -(float)new_q;
      if (qsh_lane) qf[KS - 1][0] = new_q;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) cm[r] = -d;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sx[r] -= d;
  };
#define MOBI_HP_BARRIER()                                \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
    __builtin_amdgcn_s_barrier();                        \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
  auto pv_half = [&](int boff, int kt, const unsigned (&pw)[8]) {         // O^T += V^T . P^T of one half, bare (pipeline ends)
    s16x8 vf[2][DT];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const unsigned char* vb = ldsV + boff + v_lane + (kt * 32 + st * 16) * VSTR;
#pragma unroll
      for (int d = 0; d < DT; ++d) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64 + 8 * VSTR));
        vf[st][d] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const u32x4 pu = {pw[st * 4], pw[st * 4 + 1], pw[st * 4 + 2], pw[st * 4 + 3]};
#pragma unroll
      for (int d = 0; d < DT; ++d) o[d] = mfma32(__builtin_bit_cast(frag_t, vf[st][d]), __builtin_bit_cast(frag_t, pu), o[d]);
    }
  };
  // half-step of (tile t, half KT): sc = S'(t, KT) in, sn = S' of the next half out (NEXT), pp = P of the previous half in
  // (PREV), pc = P(t, KT) out
  auto half_step = [&](auto kt_tag, auto prev_tag, auto next_tag, int t, f32x16& sc, f32x16& sn, const unsigned (&pp)[8],
                       unsigned (&pc)[8]) {
    constexpr int KT = decltype(kt_tag)::value;
    constexpr bool PREV = decltype(prev_tag)::value, NEXT = decltype(next_tag)::value;
    const int key0 = t * 64;
    const int bt = (t % 3) * IMG_BYTES;
    const int bprev = KT == 0 ? ((t + 2) % 3) * IMG_BYTES : bt;          // the tile of the previous half
    const int bnext = KT == 0 ? bt : ((t + 1) % 3) * IMG_BYTES;          // the tile of the next half
    if (!QSH) asm volatile("" : "+v"(cm));
    constexpr int NPV = PREV ? 2 * DT : 0, NS = NEXT ? KS : 0, NG = NPV + NS;
    // operand fragment of gap g (the KS S' MFMAs spread evenly among the P.V ones), requested FD gaps ahead of its MFMA: at
    // most FD + 1 fragments (4 registers each) are live -- with all of a half-step's fragments requested up front the
    // kernel needs ~190 registers instead of 128
    constexpr int FD = MOBI_ATTN_HP_FD;
    auto is_s = [&](int g) { return NS > 0 && ((g + 1) * NS) / NG != (g * NS) / NG; };
    // (ONE address register per operand and half-step, every fragment at an immediate offset from it: computed per
    //  fragment the addresses cost 36 vector instructions per tile -- SQ_INSTS_VALU 120 M against 84 M per launch)
    int kofs = bnext + k_lane, vofs = bprev + v_lane;
    asm volatile("" : "+v"(kofs), "+v"(vofs));
    const unsigned char* kbase = ldsK + kofs;
    const unsigned char* vbase = ldsV + vofs;
    auto read_frag = [&](int g) -> frag_t {
      if (is_s(g)) {
        const int ks = (g * NS) / NG;
        return __builtin_bit_cast(frag_t, ld16(kbase + ((KT ^ 1) * 32 * KSTR + ks * 32)));
      }
      const int m = g - (g * NS) / NG, st = m / DT, d = m - st * DT;
      const unsigned char* vb = vbase + (((KT ^ 1) * 32 + st * 16) * VSTR);
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64 + 8 * VSTR));
      return __builtin_bit_cast(frag_t, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    frag_t fq[FD + 1];
#pragma unroll
    for (int g = 0; g < FD && g < NG; ++g) fq[g] = read_frag(g);
    __builtin_amdgcn_sched_barrier(0);
    unsigned orr = 0u;
    float psum = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + FD < NG) fq[(g + FD) % (FD + 1)] = read_frag(g + FD);
      const frag_t f = fq[g % (FD + 1)];
      if (is_s(g)) {
        const int ks = (g * NS) / NG;
        sn = mfma32(f, qf[ks], ks == 0 ? (QSH ? zero16 : cm) : sn);
      } else {
        const int m = g - (g * NS) / NG, st = m / DT, d = m - st * DT;
        const u32x4 pu = {pp[st * 4], pp[st * 4 + 1], pp[st * 4 + 2], pp[st * 4 + 3]};
        o[d] = mfma32(f, __builtin_bit_cast(frag_t, pu), o[d]);
      }
#pragma unroll
      for (int i = (8 * g) / NG; i < (8 * (g + 1)) / NG; ++i) {
        const float e0 = __builtin_amdgcn_exp2f(sc[2 * i]), e1 = __builtin_amdgcn_exp2f(sc[2 * i + 1]);
        pc[i] = pack2<T>(e0, e1);
        orr |= pc[i];
        if (!ONES) psum += e0 + e1;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!__all((orr & 0x40004000u) == 0u)) {
      scores_half(std::false_type{}, bt, key0, KT, sc);
      raise_shift(sc, std::false_type{}, next_tag, sn, pc);
    } else if (!ONES) {
      l_run += psum;
    }
  };

  const int nfull = a.tk / 64;
  load_tile(0);
  store_tile(0);
  if (ntiles > 1) load_tile(64);
  __syncthreads();
  f32x16 sA, sB;
  unsigned pA[8], pB[8];
  if (a.tk < 32) scores_half(std::true_type{}, 0, 0, 0, sA); else scores_half(std::false_type{}, 0, 0, 0, sA);
  first_shift(sA);
  // tile-level staging at the top of a tile: tile t+1 (in registers since the previous tile) into its image, tile t+2 requested
  auto stage = [&](int t) {
    if (t + 1 < ntiles) store_tile(((t + 1) % 3) * IMG_BYTES);
    if (t + 2 < ntiles) load_tile((t + 2) * 64);
  };
  typedef std::integral_constant<int, 0> H0;
  typedef std::integral_constant<int, 1> H1;
  typedef std::true_type Y;
  typedef std::false_type N;
  // (first and last tile peeled: the steady-state loop has one straight-line version of each half-step)
  if (nfull > 0) {
    stage(0);
    half_step(H0{}, N{}, Y{}, 0, sA, sB, pB, pA);
    MOBI_HP_BARRIER();                       // tile t+1's image is complete; the image of tile t-1 is free
    if (nfull > 1) {
      half_step(H1{}, Y{}, Y{}, 0, sB, sA, pA, pB);
      int t = 1;
      for (; t + 1 < nfull; ++t) {
        stage(t);
        half_step(H0{}, Y{}, Y{}, t, sA, sB, pB, pA);
        MOBI_HP_BARRIER();
        half_step(H1{}, Y{}, Y{}, t, sB, sA, pA, pB);
      }
      stage(t);
      half_step(H0{}, Y{}, Y{}, t, sA, sB, pB, pA);
      MOBI_HP_BARRIER();
      half_step(H1{}, Y{}, N{}, t, sB, sA, pA, pB);
    } else {
      half_step(H1{}, Y{}, N{}, 0, sB, sA, pA, pB);
    }
  }
  if (nfull > 0) pv_half(((nfull - 1) % 3) * IMG_BYTES, 1, pB);           // the pipeline's last P.V
  if (nfull < ntiles) {
    // the ragged last tile: one half at a time on the exact path (masked scores, their maximum, probabilities)
    const int t = nfull, key0 = t * 64, boff = (t % 3) * IMG_BYTES;
    if (nfull > 0) MOBI_HP_BARRIER();        // (its image was written during the last full tile's first half-step)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      if (nfull > 0 || kt > 0) scores_half(std::true_type{}, boff, key0, kt, sA);
      raise_shift(sA, std::false_type{}, std::false_type{}, sA, pA);
      pv_half(boff, kt, pA);
    }
  }
#undef MOBI_HP_BARRIER

  // ---- normalise and store: lane holds O^T[d][q] for d = 32 dt + 8 g + 4 half + (0..3) -------
  if (ONES) {      // row dh of O^T sits in lane-half 0, register (dh % 32) / 2 of tile dh / 32
    float lsum = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (d * 32 + g * 8 == dh) lsum = o[d][g * 4];
    l_run = half == 0 ? lsum : 0.f;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + ql;
  if (qrow < a.tq) {
    T* orow = op + (long long)qrow * a.out_row;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = d * 32 + g * 8 + half * 4;
        if (d0 < dh) {
          float f[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) f[j] = o[d][g * 4 + j] * inv;
          *reinterpret_cast<u32x2*>(orow + d0) = pack4<T>(f);
        }
      }
  }
}


// =========================================================================================================
// attention_pipe_kernel: attention_rows_kernel's arithmetic (Q' = Q * scale * log2 e, shift on the MFMA's C operand or in a
// padded Q' channel, OR test on the packed probabilities) with the parts of DIFFERENT key tiles overlapped inside a wave.
//
// Why: ablations of attention_rows_kernel on [16, 4096 x 4096, 8 x 40] (tools/attn_lab.py, profiles/r03_attention_lab.txt):
// removing the P.V MFMAs saves exactly their matrix-pipe time, removing the exponentials theirs -- with S(t) -> softmax(t)
// -> P.V(t) in sequence a wave's time per tile is the SUM of its dependent latencies (~3,300 cycles at four waves per
// SIMD) and the matrix pipe idles during the vector phase of all four.  Here step t of a wave issues
//     S'(t+1) = K(t+1) . Q'^T   (2 KS MFMAs)   with the exponentials / packing / OR test of tile t dealt into their gaps,
//     O^T += V^T(t) . P^T(t)    (4 DT MFMAs)   back to back (the SIMD's other wave has the vector pipe meanwhile),
// all operand reads of the step requested up front, ONE barrier per key tile.  Two waves per SIMD.
// LDS: three images (K tile + V tile), tile t in image t % 3: step t reads K(t+1) and V(t) (and K(t) on the exact path) and
// writes tile t+2 -- requested from memory a whole step earlier -- into the image tile t-1 left at the last barrier.
// =========================================================================================================
template <typename T, int KS, int NW, bool QSH>
__global__ __launch_bounds__(64 * NW, 2) void attention_pipe_kernel(const AttnArgs a) {
  constexpr int NTHR = 64 * NW;
  typedef typename Vec8<T>::type frag_t;
  constexpr int DT = (KS + 1) / 2;
  constexpr int KSTR = KS * 32 + 16;
  constexpr int VSTR = (DT & 1) ? DT * 64 : DT * 64 + 64;
  constexpr int K_BYTES = 64 * KSTR, V_BYTES = 64 * VSTR, IMG_BYTES = K_BYTES + V_BYTES;
  constexpr int KP = (64 * KS * 2 + NTHR - 1) / NTHR;
  constexpr int NSLOT = 4;
  constexpr float BIAS = AttnBias<T>::value;
  constexpr bool ONES = QSH || (KS & 1);
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * IMG_BYTES];
  unsigned char* ldsK = lds;
  unsigned char* ldsV = lds + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  int qblk, head, img;
  attn_block(a.xcd_map, qblk, head, img);
  const int q0 = qblk * (32 * NW) + wave * 32;
  const int dh = a.dh;

  const T* __restrict__ qp = reinterpret_cast<const T*>(a.q) + img * a.q_img + head * dh;
  const T* __restrict__ kp = reinterpret_cast<const T*>(a.k) + img * a.k_img + head * dh;
  const T* __restrict__ vp = reinterpret_cast<const T*>(a.vt) + img * a.vt_img + (long long)head * dh;
  T* __restrict__ op = reinterpret_cast<T*>(a.out) + img * a.out_img + head * dh;

  frag_t qf[KS];
  {
    const float cexp = a.cexp;
    const int qrow = q0 + ql;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + half * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qrow < a.tq && c < dh) v = ld16(qp + (long long)qrow * a.q_row + c);
      frag_t f = __builtin_bit_cast(frag_t, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (T)((float)f[j] * cexp);          // cexp == 1 (q_log2_scaled): q bit for bit
      qf[ks] = f;
    }
  }

  u32x4 kr[KP], vr[KP];
  unsigned koff[KP], voff[KP];
  constexpr unsigned OOB = 0x80000000u;
#pragma unroll
  for (int i = 0; i < KP; ++i) {
    const int p = tid + NTHR * i;
    const int row = p / (KS * 2), pc = p - row * (KS * 2);
    const bool on = row < 64 && pc * 8 < dh;
    koff[i] = on ? (unsigned)(row * a.k_row + pc * 8) * 2u : OOB;
    voff[i] = on ? (unsigned)(row * a.vt_row + pc * 8) * 2u : OOB;
  }
  const int k_bytes = ((a.tk - 1) * a.k_row + dh) * 2;
  const int v_bytes = ((a.tk - 1) * a.vt_row + dh) * 2;
  // padded channels of the K rows (zero; channel dh = 1.0 when the shift rides in Q') and of the V rows (zero; channel
  // dh = 1.0: the denominator column) are written ONCE into every image; the tile stores only touch channels < dh
  for (int p = tid; p < 64 * KS * 2; p += NTHR) {
    const int row = p / (KS * 2), pc = p - row * (KS * 2);
    if (pc * 8 >= dh) {
      frag_t e;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (T)0.0f;
      if (QSH && pc * 8 == dh) e[0] = (T)1.0f;
#pragma unroll
      for (int b = 0; b < NSLOT; ++b) st16(ldsK + b * IMG_BYTES + row * KSTR + pc * 16, __builtin_bit_cast(u32x4, e));
    }
  }
  for (int p = tid; p < 64 * DT * 4; p += NTHR) {
    const int row = p / (DT * 4), pc = p - row * (DT * 4);
    if (pc * 8 >= dh) {
      frag_t e;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (T)0.0f;
      if (pc * 8 == dh) e[0] = (T)1.0f;
#pragma unroll
      for (int b = 0; b < NSLOT; ++b) st16(ldsV + b * IMG_BYTES + row * VSTR + pc * 16, __builtin_bit_cast(u32x4, e));
    }
  }
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(kp), 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(vp), 0, v_bytes, 0x00020000);
  auto load_tile = [&](int key0) {
    const unsigned ku = (unsigned)key0 * (unsigned)a.k_row * 2u, vu = (unsigned)key0 * (unsigned)a.vt_row * 2u;
#pragma unroll
    for (int i = 0; i < KP; ++i) kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rk, koff[i], ku, 0);
#pragma unroll
    for (int i = 0; i < KP; ++i) vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rv, voff[i], vu, 0);
  };
  auto store_tile = [&](int boff) {
#pragma unroll
    for (int i = 0; i < KP; ++i) {
      const int p = tid + NTHR * i;
      const int row = p / (KS * 2), pc = p - row * (KS * 2);
      if (row < 64 && pc * 8 < dh) {
        st16(ldsK + boff + row * KSTR + pc * 16, kr[i]);
        st16(ldsV + boff + row * VSTR + pc * 16, vr[i]);
      }
    }
  };
#define MOBI_PIPE_BARRIER()                              \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
    __builtin_amdgcn_s_barrier();                        \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  f32x16 cm;
#pragma unroll
  for (int r = 0; r < 16; ++r) cm[r] = 0.f;
  const bool qsh_lane = QSH && half == 1;     // channel dh = KS*16 - 8 sits in the upper half's fragment, element 0
  float l_run = 0.f;
  const int ntiles = (a.tk + 63) / 64;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  typedef __attribute__((address_space(3))) s16x4* lds4_t;
  const int l16 = lane & 15, grp = lane >> 4;
  const int k_lane = ql * KSTR + half * 16;                                        // K fragment of (key ql, half)
  const int v_lane = (4 * half + (l16 >> 2)) * VSTR + (16 * (grp & 1) + 4 * (l16 & 3)) * 2;

  auto read_k = [&](int boff, frag_t (&kf)[2][KS]) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        kf[kt][ks] = __builtin_bit_cast(frag_t, ld16(ldsK + boff + k_lane + kt * 32 * KSTR + ks * 32));
  };
  auto scores = [&](frag_t (&kf)[2][KS], f32x16 (&sx)[2]) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sx[kt] = mfma32(kf[kt][ks], qf[ks], ks == 0 ? (QSH ? zero16 : cm) : sx[kt]);
  };
  auto mask_ragged = [&](f32x16 (&sx)[2], int key0) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key >= a.tk) sx[kt][r] = -INFINITY;
      }
  };

  int slot0 = 0, slot1 = 1, slot2 = 2, slot3 = 3;                  // images of tiles t, t+1, t+2, t+3
  // Operand fragments live in registers ACROSS steps and are refilled behind their last use ("rolling"): right after the
  // P.V MFMA that consumed fragment i of V(t-1), fragment i of V(t) is requested into the same registers; right after an
  // S MFMA, the same fragment of K(t+2).  A step therefore starts multiplying at once -- with the reads requested at the
  // top of the step every wave of the block sat in the LDS latency together, the matrix pipe idle (stamps: 30 % of a step).
  s16x8 vf[2][2][DT];                                               // V^T fragments of the tile whose P.V comes next
  frag_t kf[2][KS];                                                 // K fragments of the tile whose S' comes next
  auto read_v1 = [&](int boff, int i) {                             // fragment i = (kt, st, d) of the V tile in image boff
    const int kt = i / (2 * DT), st = (i / DT) & 1, d = i % DT;
    const unsigned char* vb = ldsV + boff + v_lane + (kt * 32 + st * 16) * VSTR;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64 + 8 * VSTR));
    vf[kt][st][d] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto read_k1 = [&](int boff, int i) {
    const int kt = i / KS, ks = i - kt * KS;
    kf[kt][ks] = __builtin_bit_cast(frag_t, ld16(ldsK + boff + k_lane + kt * 32 * KSTR + ks * 32));
  };
  // one step: sc = S'(t) (in, complete), sn = S'(t+1) (out); pp = P(t-1) (in), pw = P(t) (out).
  // MFMA i of the step: i < 4 DT: O^T += V^T(t-1) . P^T(t-1) (one accumulator's MFMAs DT apart); then the 2 KS of S'(t+1).
  // EVERY MFMA is followed by its share of the tile's vector work in the wave's own stream: a wave that waits on the busy
  // matrix pipe holds up the vector issue of the SIMD's other wave as well (tools/probes/mfma_fill.hip, 'split' rows).
  auto step = [&](auto first_tag, int t, f32x16 (&sc)[2], f32x16 (&sn)[2], unsigned (&pp)[2][8], unsigned (&pw)[2][8]) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int key0 = t * 64;
    const int b0 = slot0 * IMG_BYTES, b2 = slot2 * IMG_BYTES;
    if (!(MOBI_ATTN_PDBG & 1)) {
      if (t + 3 < ntiles) store_tile(slot3 * IMG_BYTES);    // requested one step ago (or in the prologue)
      if (t + 4 < ntiles) load_tile(key0 + 256);
    }
    if (key0 + 64 > a.tk) mask_ragged(sc, key0);             // ragged last tile
    if (!QSH) asm volatile("" : "+v"(cm));
    __builtin_amdgcn_sched_barrier(0);
    unsigned orr = 0u;
    float psum = 0.f;
    constexpr int NPV = FIRST ? 0 : 4 * DT, NG = NPV + 2 * KS;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g < NPV) {
        const int kt = g / (2 * DT), st = (g / DT) & 1, d = g % DT;
        const u32x4 pu = {pp[kt][st * 4], pp[kt][st * 4 + 1], pp[kt][st * 4 + 2], pp[kt][st * 4 + 3]};
        o[d] = mfma32(__builtin_bit_cast(frag_t, vf[kt][st][d]), __builtin_bit_cast(frag_t, pu), o[d]);
      } else {
        const int kt = (g - NPV) / KS, ks = (g - NPV) - kt * KS;
        sn[kt] = mfma32(kf[kt][ks], qf[ks], ks == 0 ? (QSH ? zero16 : cm) : sn[kt]);
      }
#pragma unroll
      for (int pi = (16 * g) / NG; pi < (16 * (g + 1)) / NG; ++pi) {
        const int pk = pi >> 3, i = pi & 7;
        const float e0 = __builtin_amdgcn_exp2f(sc[pk][2 * i]), e1 = __builtin_amdgcn_exp2f(sc[pk][2 * i + 1]);
        pw[pk][i] = pack2<T>(e0, e1);
        orr |= pw[pk][i];
        if (!ONES) psum += e0 + e1;
      }
      // refill behind the MFMA: V(t) for the next step's P.V, K(t+2) for its S'
      if (g < NPV) {
        read_v1(b0, g);
      } else {
        read_k1(b2, g - NPV);                               // (past the last tile: a stale image, the result is not used)
        if (FIRST) {
#pragma unroll
          for (int i = (4 * DT * (g - NPV)) / (2 * KS); i < (4 * DT * (g - NPV + 1)) / (2 * KS); ++i) read_v1(b0, i);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const bool fix = !(MOBI_ATTN_PDBG & 4) && !__all((orr & 0x40004000u) == 0u);
    if (fix) {
      // exact path (a probability reached 2.0): S'(t) once more from its LDS image, the tile's maximum per query
      // column, the shift raised to (maximum - BIAS), O (tiles < t) rescaled, S'(t+1) moved to the new shift, P(t) recomputed
      frag_t kc[2][KS];
      read_k(b0, kc);
      scores(kc, sc);
      if (key0 + 64 > a.tk) mask_ragged(sc, key0);
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kt][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float d = fmaxf(mx + BIAS, 0.f);       // the shift only rises
      if (QSH) {
        // the shift is a value of the storage type: take the step its rounding actually makes
        const float old_q = __shfl(qsh_lane ? (float)qf[KS - 1][0] : 0.f, ql + 32, 64);
        const T new_q = (T)(old_q - d);
        d = old_q - (float)new_q;
        if (qsh_lane) qf[KS - 1][0] = new_q;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) cm[r] -= d;
      }
      {
        const float alpha = __builtin_amdgcn_exp2f(-d);
        l_run *= alpha;
#pragma unroll
        for (int dd = 0; dd < DT; ++dd)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dd][r] *= alpha;
      }
      psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[kt][r] -= d;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float e0 = __builtin_amdgcn_exp2f(sc[kt][2 * i] - d), e1 = __builtin_amdgcn_exp2f(sc[kt][2 * i + 1] - d);
          pw[kt][i] = pack2<T>(e0, e1);
          if (!ONES) psum += e0 + e1;
        }
      }
    }
    if (!ONES) l_run += psum;
    const int s_ = slot0;
    slot0 = slot1;
    slot1 = slot2;
    slot2 = slot3;
    slot3 = s_;
    if (MOBI_ATTN_PDBG & 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else MOBI_PIPE_BARRIER();
  };

  f32x16 sa[2], sb[2];
  load_tile(0);
  store_tile(0);
  if (ntiles > 1) {
    load_tile(64);
    store_tile(IMG_BYTES);
  }
  if (ntiles > 2) {
    load_tile(128);
    store_tile(2 * IMG_BYTES);
  }
  if (ntiles > 3) load_tile(192);                            // stays in registers until step 0
  MOBI_PIPE_BARRIER();
  {
    // S'(0), and the first shift from its exact maximum (every later step starts from a valid shift and only tests)
    read_k(0, kf);
    scores(kf, sa);
    if (64 > a.tk) mask_ragged(sa, 0);
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sa[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float d = mx + BIAS;
    if (QSH) {
      const T new_q = (T)(-d);
      d = -(float)new_q;
      if (qsh_lane) qf[KS - 1][0] = new_q;
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) cm[r] = -d;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sa[kt][r] -= d;
  }
  read_k(IMG_BYTES, kf);                                     // K(1) for step 0
  unsigned pa[2][8], pb[2][8];
  step(std::true_type{}, 0, sa, sb, pb, pa);
  int t = 1;
  for (; t + 1 < ntiles; t += 2) {
    step(std::false_type{}, t, sb, sa, pa, pb);
    step(std::false_type{}, t + 1, sa, sb, pb, pa);
  }
  const bool odd_left = t < ntiles;                        // one more step, on the (sb, pa) -> pb roles
  if (odd_left) step(std::false_type{}, t, sb, sa, pa, pb);
  auto last_pv = [&](unsigned (&pl)[2][8]) {                 // the last tile's P.V (its fragments came in during the last step)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const u32x4 pu = {pl[kt][st * 4], pl[kt][st * 4 + 1], pl[kt][st * 4 + 2], pl[kt][st * 4 + 3]};
#pragma unroll
        for (int d = 0; d < DT; ++d)
          o[d] = mfma32(__builtin_bit_cast(frag_t, vf[kt][st][d]), __builtin_bit_cast(frag_t, pu), o[d]);
      }
  };
  if (odd_left) last_pv(pb); else last_pv(pa);
#undef MOBI_PIPE_BARRIER

  if (ONES) {
    float lsum = 0.f;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (d * 32 + g * 8 == dh) lsum = o[d][g * 4];
    l_run = half == 0 ? lsum : 0.f;
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + ql;
  if (qrow < a.tq) {
    T* orow = op + (long long)qrow * a.out_row;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = d * 32 + g * 8 + half * 4;
        if (d0 < dh) {
          float f[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) f[j] = o[d][g * 4 + j] * inv;
          *reinterpret_cast<u32x2*>(orow + d0) = pack4<T>(f);
        }
      }
  }
}


#endif  // MOBI_DEV (pipelined variants)

#ifdef MOBI_DEV   // development build only: the software-pipelined A/B alternative (measured slower, MOBI_ATTN_SP=1)
// =========================================================================================================
// SOFTWARE-PIPELINED variant of the 8-wave kernel (V row-major, one 16-byte K and V piece per thread and key tile,
// odd KS so that the padded head-dim row carries the softmax denominator: dh 33..48 with KS = 3).
// An A/B ALTERNATIVE (MOBI_ATTN_SP=1), measured slower than the kernel above -- kept because the next step for this
// kernel starts from it (profiles/r01_attention_ablation.txt):
//
// In the kernel above a wave runs  S(t) MFMAs -> softmax(t) -> P.V(t) MFMAs  one after the other, and SQ counters on
// [16, 4096 x 4096, 8 x 40] show the matrix pipe busy 41 % and the vector pipe 62 % of the time with 18 % in both: the
// kernel time is the SUM of the two (tools/pmc_attn.sh).  A ping-pong split (one wave of a SIMD in an MFMA-only phase,
// its partner in the softmax phase, barrier-separated) was worse still (744-789 us): the partner's MFMAs stretched the
// softmax phase from ~750 to ~1250 cycles.  tools/probes/mfma_valu_overlap.hip says why: v_exp_f32 hides behind a
// wave's own MFMAs (14 MFMA + 42 v_exp: 480 cycles vs 468 for the MFMAs alone), plain vector instructions mostly do
// not (14 MFMA + 84 v_fma: 648 vs 468 + 284).
//
// Here the three parts of DIFFERENT key tiles share one instruction stream: step t of a wave issues
//     O^T += V^T(t-1) . P^T(t-1)    (8 MFMAs)      S^T(t+1) = K(t+1) . Q^T    (6 MFMAs)      softmax(t) -> P^T(t)
// with the softmax's vector instructions dealt into the MFMA gaps by hand, all operand reads of the step requested
// up front, ONE barrier per key tile.  The rescale of O by the moving maximum is applied one step late, before the
// P.V of the tile it belongs to (a wave-uniform, rare branch outside the interleaved region).
// LDS: four images (K tile + V tile), tile t in image t % 4: K(t) is read in step t-1, V(t) in step t+1, tile t+2 is
// written in step t into the image of tile t-2 (last read in step t-1).
// =========================================================================================================
template <typename T, int KS>
__global__ __launch_bounds__(512, 2) void attention_sp_kernel(const AttnArgs a) {
  static_assert(KS & 1, "odd KS: a padded head-dim row exists and carries the denominator");
  constexpr int NTHR = 512;
  typedef typename Vec8<T>::type frag_t;
  constexpr int DT = (KS + 1) / 2;
  constexpr int KSTR = KS * 32 + 16;
  constexpr int VSTR = (DT & 1) ? DT * 64 : DT * 64 + 64;
  constexpr int K_BYTES = 64 * KSTR, V_BYTES = 64 * VSTR, IMG_BYTES = K_BYTES + V_BYTES;
  constexpr int NSLOT = 4;
  static_assert(64 * KS * 2 <= NTHR, "one 16-byte piece per thread and key tile");
  __shared__ __attribute__((aligned(16))) unsigned char lds[NSLOT * IMG_BYTES];
  unsigned char* ldsK = lds;
  unsigned char* ldsV = lds + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ql = lane & 31, half = lane >> 5;
  int qblk, head, img;
  attn_block(a.xcd_map, qblk, head, img);
  const int q0 = qblk * 256 + wave * 32;
  const int dh = a.dh;

  const T* __restrict__ qp = reinterpret_cast<const T*>(a.q) + img * a.q_img + head * dh;
  const T* __restrict__ kp = reinterpret_cast<const T*>(a.k) + img * a.k_img + head * dh;
  const T* __restrict__ vp = reinterpret_cast<const T*>(a.vt) + img * a.vt_img + (long long)head * dh;
  T* __restrict__ op = reinterpret_cast<T*>(a.out) + img * a.out_img + head * dh;

  frag_t qf[KS];
  {
    const int qrow = q0 + ql;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = ks * 16 + half * 8;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (qrow < a.tq && c < dh) v = ld16(qp + (long long)qrow * a.q_row + c);
      qf[ks] = __builtin_bit_cast(frag_t, v);
    }
  }

  // staging: thread -> (key row, 8 channels), the same piece of the K tile and of the V tile
  constexpr unsigned OOB = 0x80000000u;
  const int s_row = tid / (KS * 2), s_pc = tid - s_row * (KS * 2);
  const bool s_on = s_row < 64, s_von = s_on && s_pc * 8 < dh;
  const unsigned koff = s_von ? (unsigned)(s_row * a.k_row + s_pc * 8) * 2u : OOB;
  const unsigned voff = s_von ? (unsigned)(s_row * a.vt_row + s_pc * 8) * 2u : OOB;
  const int k_bytes = ((a.tk - 1) * a.k_row + dh) * 2;
  const int v_bytes = ((a.tk - 1) * a.vt_row + dh) * 2;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(kp), 0, k_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(vp), 0, v_bytes, 0x00020000);
  u32x4 kr, vr;
  auto load_tile = [&](int key0) {
    kr = __builtin_amdgcn_raw_buffer_load_b128(rk, koff + (unsigned)key0 * (unsigned)a.k_row * 2u, 0, 0);
    vr = __builtin_amdgcn_raw_buffer_load_b128(rv, voff + (unsigned)key0 * (unsigned)a.vt_row * 2u, 0, 0);
  };
  auto store_tile = [&](int slot) {
    if (s_on) st16(ldsK + slot * IMG_BYTES + s_row * KSTR + s_pc * 16, kr);
    if (s_von) st16(ldsV + slot * IMG_BYTES + s_row * VSTR + s_pc * 16, vr);
  };
  // channels [dh, DT*32) of every V row are written ONCE: zero, except channel dh = 1.0 (the denominator column)
  for (int p = tid; p < 64 * DT * 4; p += NTHR) {
    const int row = p / (DT * 4), pc = p - row * (DT * 4);
    if (pc * 8 >= dh) {
      typename Vec8<T>::type e;
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = (T)0.0f;
      if (pc * 8 == dh) e[0] = (T)1.0f;
#pragma unroll
      for (int b = 0; b < NSLOT; ++b) st16(ldsV + b * IMG_BYTES + row * VSTR + pc * 16, __builtin_bit_cast(u32x4, e));
    }
  }

#define MOBI_ASP_BARRIER()                               \
  do {                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                   \
    __builtin_amdgcn_s_barrier();                        \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)

  f32x16 o[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  f32x16 sa[2], sb[2];
  frag_t pa[2][2], pb[2][2];
  float m_run = -INFINITY, alpha = 1.f;
  const float cexp = a.cexp;
  const int ntiles = (a.tk + 63) / 64;

  load_tile(0);
  store_tile(0);
  if (ntiles > 1) { load_tile(64); store_tile(1); }
  if (ntiles > 2) load_tile(128);                      // stays in registers until step 0
  __syncthreads();

  typedef __attribute__((address_space(3))) s16x4* lds4_t;
  const int l16 = lane & 15, grp = lane >> 4;
  frag_t kfr[2][KS];
  s16x8 vfr[2][2][DT];
  auto read_k = [&](int t) {
    const int boff = (t & 3) * IMG_BYTES;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const unsigned char* kb = ldsK + boff + (kt * 32 + ql) * KSTR + half * 16;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) kfr[kt][ks] = __builtin_bit_cast(frag_t, ld16(kb + ks * 32));
    }
  };
  auto read_v = [&](int t) {
    const int boff = (t & 3) * IMG_BYTES;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const unsigned char* vb = ldsV + boff + (kt * 32 + st * 16 + 4 * half + (l16 >> 2)) * VSTR +
                                  (16 * (grp & 1) + 4 * (l16 & 3)) * 2;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(vb + d * 64 + 8 * VSTR));
          vfr[kt][st][d] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
      }
  };
  auto do_pv = [&](frag_t (&pp)[2][2]) {                 // O^T += V^T . P^T of the previous tile
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int st = 0; st < 2; ++st)
          o[d] = mfma32(__builtin_bit_cast(frag_t, vfr[kt][st][d]), pp[kt][st], o[d]);
  };
  auto do_s = [&](f32x16 (&sn)[2]) {                     // S^T of the next tile = K . Q^T
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) sn[kt] = mfma32(kfr[kt][ks], qf[ks], sn[kt]);
    }
  };
  auto rescale = [&]() {                                 // O (and the denominator row in it) follows the moving maximum
    if (!__all(alpha == 1.f)) {
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
  };
  // one step: sc = S(t) (in), sn = S(t+1) (out), pp = P(t-1) (in), pn = P(t) (out)
  auto step = [&](auto first_tag, int t, f32x16 (&sc)[2], f32x16 (&sn)[2], frag_t (&pp)[2][2], frag_t (&pn)[2][2]) {
    constexpr bool FIRST = decltype(first_tag)::value;
    const int key0 = t * 64;
    if (!FIRST) rescale();                               // by the factor softmax(t-1) found, before P.V(t-1)
    if (key0 + 64 > a.tk) {                              // ragged last tile
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (key >= a.tk) sc[kt][r] = -INFINITY;
        }
    }
    if (t + 2 < ntiles) store_tile((t + 2) & 3);         // requested one step ago (or in the prologue)
    if (t + 3 < ntiles) load_tile((t + 3) * 64);
    __builtin_amdgcn_sched_barrier(0);
    // ---- interleaved region: 14 MFMA gaps, the vector work of softmax(t) dealt into them by hand (the gaps are
    // pinned with sched_barrier: left to the scheduler, sched_group_barrier or not, most gaps stayed empty and the
    // exponentials came in two bursts) ------------------------------------------------------------------------------
    // K(t+1) fragments now (first used in gap 8); the V(t) fragments of the NEXT step's P.V are requested behind this
    // step's last P.V MFMA into the registers it frees, so no MFMA waits at the head of a step for the read burst of
    // all eight waves
    read_k(t + 1);                                       // (past the last tile: a stale image, the result is not used)
    __builtin_amdgcn_sched_barrier(0);
    auto mf_pv = [&](int i) {                            // MFMA i of O^T += V^T . P^T: one accumulator's four in a row
      if (FIRST) return;
      const int d = i >> 2, kt = (i >> 1) & 1, st = i & 1;
      o[d] = mfma32(__builtin_bit_cast(frag_t, vfr[kt][st][d]), pp[kt][st], o[d]);
    };
    auto mf_s = [&](int i) {                             // MFMA i of S^T(t+1)
      const int kt = i / KS, ks = i - kt * KS;
      if (ks == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[kt][r] = 0.f;
      }
      sn[kt] = mfma32(kfr[kt][ks], qf[ks], sn[kt]);
    };
    float mx = -INFINITY;
#pragma unroll
    for (int g = 0; g < 4; ++g) {                        // gaps 0-3: the maximum of 8 scores each
      mf_pv(g);
#pragma unroll
      for (int r = 0; r < 8; ++r) mx = fmaxf(mx, sc[g >> 1][(g & 1) * 8 + r]);
      __builtin_amdgcn_sched_barrier(0);
    }
    mf_pv(4);                                            // gap 4: the column's maximum, the rescale factor
    {
      // both halves of a query column by gfx950's lane-row swap (no LDS crossbar trip): v_permlane32_swap exchanges the
      // upper half of its first operand with the lower half of its second, so max(first, second) afterwards is the
      // column's maximum in both halves.  Inline asm: with the builtin hipcc folds max(result[0], result[1]) of two
      // equal inputs into result[0].
      float lo = mx, hi = mx;
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(lo), "+v"(hi));
      mx = fmaxf(lo, hi);
    }
    const float m_new = fmaxf(m_run, mx);
    alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cexp);                       // first tile: exp2(-inf) = 0, O = 0
    m_run = m_new;
    const float mc = m_new * cexp;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {                        // gaps 5-12: four exponentials each
      if (g < 3) mf_pv(5 + g); else mf_s(g - 3);
      if (g == 3) read_v(t);
      const int kt = g >> 2, st = (g >> 1) & 1, j0 = (g & 1) * 4;
#pragma unroll
      for (int j = j0; j < j0 + 4; ++j)
        pn[kt][st][j] = (T)__builtin_amdgcn_exp2f(__builtin_fmaf(sc[kt][st * 8 + j], cexp, -mc));
      __builtin_amdgcn_sched_barrier(0);
    }
    mf_s(5);
    // (the step's results are pinned here: without it the compiler sinks the exponentials of this step past the
    //  barrier, next to their use in the next step's P.V)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int st = 0; st < 2; ++st) asm volatile("" : "+v"(pn[kt][st]));
    MOBI_ASP_BARRIER();
  };

  // S(0)
  read_k(0);
  do_s(sa);
  step(std::true_type{}, 0, sa, sb, pb, pa);
  int t = 1;
  for (; t + 1 < ntiles; t += 2) {
    step(std::false_type{}, t, sb, sa, pa, pb);
    step(std::false_type{}, t + 1, sa, sb, pb, pa);
  }
  const bool odd_left = t < ntiles;                      // one more step, on the (sb, pa) -> pb roles
  if (odd_left) step(std::false_type{}, t, sb, sa, pa, pb);
  // the last tile's P.V
  rescale();
  if (odd_left) do_pv(pb); else do_pv(pa);
#undef MOBI_ASP_BARRIER

  float lsum = 0.f;                // row dh of O^T sits in lane-half 0, register (dh % 32) / 2 of tile dh / 32
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (d * 32 + g * 8 == dh) lsum = o[d][g * 4];
  const float l_run = half == 0 ? lsum : 0.f;
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int qrow = q0 + ql;
  if (qrow < a.tq) {
    T* orow = op + (long long)qrow * a.out_row;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d0 = d * 32 + g * 8 + half * 4;
        if (d0 < dh) {
          float f[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) f[j] = o[d][g * 4 + j] * inv;
          *reinterpret_cast<u32x2*>(orow + d0) = pack4<T>(f);
        }
      }
  }
}

#endif  // MOBI_DEV

// (the H16 instantiation exists for KS = 3 only; the trait keeps the other KS_ expansions of the dispatch macro well-formed)
template <int KS_> struct mobi_attn_h16_ok { static constexpr bool value = KS_ == 3; static constexpr int ks = 3; };

template <typename T, int VVEC>
static int launch_attention_v(const mobi_attention_params* p, const AttnArgs& a, hipStream_t st) {
  dim3 grid((p->tq + 127) / 128, p->heads, p->images), block(256);
  const int ks = (p->dh + 15) / 16;
  if constexpr (VVEC == 2) {
    // 8-wave blocks (256 queries per staged K / V tile) for the big launches
    const long long blocks8 = (long long)((p->tq + 255) / 256) * p->heads * p->images;
    // measured -5.5 % on [16 | 8 images, 4096 x 4096, 8 x 40]; round 4 (tools/kbench.py attn, MOBI_ATTN_NW=4 | 8): from ONE block
    // per CU on -- [8, 1024 x 1024, 8 x 80] 30.8 against 33.0 us, [16, ...] 62.9 against 66.3, [16, 1024 x 1024, 8 x 40] 34.2
    // against 41.4, [2, 4096 x 4096, 8 x 40] 69.2 against 76.8; with 128 blocks (4 images at 1024 x 1024) they lose (26.3 vs 22.0)
    int nw8 = blocks8 >= (tuning().attn_nw8_blocks > 0 ? tuning().attn_nw8_blocks : 256) && ks <= 5;
    if (tuning().attn_nw > 0) nw8 = ks <= 5 && tuning().attn_nw == 8;       // tests / A-B: 8 forces, 4 forbids
    if (ks <= 5 && tuning().attn_v3 != 0) {           // the reduced-instruction kernel (MOBI_ATTN_V3=0: the kernel below)
      dim3 gridr((p->tq + (nw8 ? 255 : 127)) / (nw8 ? 256 : 128), p->heads, p->images), blockr(nw8 ? 512 : 256);
      const bool qsh = (p->dh & 15) != 0;             // a padded channel in the last k-step carries the shift
#ifdef MOBI_DEV
      if (tuning().attn_v3 == 2 && ks == 3 && qsh && nw8) {      // A/B: the software-pipelined kernel (MOBI_ATTN_V3=2)
        hipLaunchKernelGGL((attention_pipe_kernel<T, 3, 8, true>), gridr, blockr, 0, st, a);
        MOBI_CHECK_LAUNCH();
        return MOBI_OK;
      }
#endif
#define MOBI_ATTN_ROWS_K(KERNEL_, KS_)                                                                   \
  do {                                                                                                   \
    if (qsh && KS_ == 3 && tuning().attn_h16 == 1 && mobi_attn_h16_ok<KS_>::value) {                     \
      /* dh = 40, MOBI_ATTN_H16=1 (A/B, measured SLOWER): the last 16 channels of O^T on MFMA 16x16x32 */ \
      if (nw8) hipLaunchKernelGGL((attention_rows_kernel<T, mobi_attn_h16_ok<KS_>::ks, 8, true, true>), gridr, blockr, 0, st, a); \
      else hipLaunchKernelGGL((attention_rows_kernel<T, mobi_attn_h16_ok<KS_>::ks, 4, true, true>), gridr, blockr, 0, st, a);     \
    } else if (qsh) {                                                                                    \
      if (nw8) hipLaunchKernelGGL((KERNEL_<T, KS_, 8, true>), gridr, blockr, 0, st, a);                   \
      else hipLaunchKernelGGL((KERNEL_<T, KS_, 4, true>), gridr, blockr, 0, st, a);                       \
    } else {                                                                                             \
      if (nw8) hipLaunchKernelGGL((KERNEL_<T, KS_, 8, false>), gridr, blockr, 0, st, a);                  \
      else hipLaunchKernelGGL((KERNEL_<T, KS_, 4, false>), gridr, blockr, 0, st, a);                      \
    }                                                                                                    \
  } while (0)
#ifdef MOBI_DEV
#define MOBI_ATTN_ROWS(KS_)                                                                              \
  do {                                                                                                   \
    if (tuning().attn_v3 == 3 && qsh) {           /* A/B: the half-tile pipeline (padded head dims only) */ \
      if (nw8 && KS_ <= 3 && MOBI_ATTN_HP_NW6) {                                                         \
        dim3 grid6((p->tq + 191) / 192, p->heads, p->images), block6(384);                               \
        hipLaunchKernelGGL((attention_hp_kernel<T, KS_, 6, true>), grid6, block6, 0, st, a);             \
      } else if (nw8) {                                                                                  \
        hipLaunchKernelGGL((attention_hp_kernel<T, KS_, 8, true>), gridr, blockr, 0, st, a);             \
      } else {                                                                                           \
        hipLaunchKernelGGL((attention_hp_kernel<T, KS_, 4, true>), gridr, blockr, 0, st, a);             \
      }                                                                                                  \
    } else {                                                                                             \
      MOBI_ATTN_ROWS_K(attention_rows_kernel, KS_);                                                      \
    }                                                                                                    \
  } while (0)
#else
#define MOBI_ATTN_ROWS(KS_) MOBI_ATTN_ROWS_K(attention_rows_kernel, KS_)
#endif
      if (ks <= 1) MOBI_ATTN_ROWS(1);
      else if (ks == 2) MOBI_ATTN_ROWS(2);
      else if (ks == 3) MOBI_ATTN_ROWS(3);
      else if (ks == 4) MOBI_ATTN_ROWS(4);
      else MOBI_ATTN_ROWS(5);
#undef MOBI_ATTN_ROWS
#undef MOBI_ATTN_ROWS_K
      MOBI_CHECK_LAUNCH();
      return MOBI_OK;
    }
    if (nw8) {
      dim3 grid8((p->tq + 255) / 256, p->heads, p->images), block8(512);
      // software-pipelined kernel (head dims 33..48): measured SLOWER than the kernel above (693 vs 620 us on
      // [16, 4096 x 4096, 8 x 40], tools/ab_attn.sh), kept selectable for A/B runs and covered by the parity tests
#ifdef MOBI_DEV
      int sp = 0;
      if (tuning().attn_sp == 1) sp = ks == 3;
      if (sp) {
        hipLaunchKernelGGL((attention_sp_kernel<T, 3>), grid8, block8, 0, st, a);
        MOBI_CHECK_LAUNCH();
        return MOBI_OK;
      }
#endif
#define MOBI_ATTN_CASE8(KS_) hipLaunchKernelGGL((attention_kernel<T, KS_, 2, 2, 8>), grid8, block8, 0, st, a)
      if (ks <= 1) MOBI_ATTN_CASE8(1);
      else if (ks == 2) MOBI_ATTN_CASE8(2);
      else if (ks == 3) MOBI_ATTN_CASE8(3);
      else if (ks == 4) MOBI_ATTN_CASE8(4);
      else MOBI_ATTN_CASE8(5);
#undef MOBI_ATTN_CASE8
      MOBI_CHECK_LAUNCH();
      return MOBI_OK;
    }
  }
#define MOBI_ATTN_CASE(KS_, WPS_) hipLaunchKernelGGL((attention_kernel<T, KS_, VVEC, WPS_>), grid, block, 0, st, a)
  if (ks <= 1) MOBI_ATTN_CASE(1, 2);
  else if (ks == 2) MOBI_ATTN_CASE(2, 2);
  else if (ks == 3) MOBI_ATTN_CASE(3, 2);
  else if (ks == 4) MOBI_ATTN_CASE(4, 2);
  else if (ks == 5) MOBI_ATTN_CASE(5, 2);
  else if (ks == 6) MOBI_ATTN_CASE(6, 2);
  else if (ks <= 8) MOBI_ATTN_CASE(8, 1);
  else if (ks <= 10) MOBI_ATTN_CASE(10, 1);
  else return MOBI_ERR_UNSUPPORTED;
#undef MOBI_ATTN_CASE
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

template <typename T>
static int launch_attention(const mobi_attention_params* p, const AttnArgs& a, hipStream_t st) {
  if (p->v_layout == 1) return launch_attention_v<T, 2>(p, a, st);
  const bool vvec = (p->vt_row_stride % 8 == 0) && (p->vt_img_stride % 8 == 0) && (p->tk % 8 == 0) &&
                    ((reinterpret_cast<uintptr_t>(p->vt) & 15) == 0);
  return vvec ? launch_attention_v<T, 1>(p, a, st) : launch_attention_v<T, 0>(p, a, st);
}

}  // namespace mobi

extern "C" int mobi_attention(const mobi_attention_params* p, void* stream) {
  using namespace mobi;
  if (!p || !p->q || !p->k || !p->vt || !p->out) return MOBI_ERR_ARG;
  if (p->dtype != MOBI_F16 && p->dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (p->images <= 0 || p->heads <= 0 || p->tq <= 0 || p->tk <= 0 || p->dh <= 0) return MOBI_ERR_ARG;
  if ((p->dh & 7) || p->dh > 160) return MOBI_ERR_UNSUPPORTED;
  if (p->v_layout != 0 && p->v_layout != 1) return MOBI_ERR_ARG;
  if (p->v_layout == 1 && ((p->vt_row_stride & 7) || (p->vt_img_stride & 7) ||
                           (reinterpret_cast<uintptr_t>(p->vt) & 15))) return MOBI_ERR_ALIGN;
  if ((p->q_row_stride & 7) || (p->k_row_stride & 7) || (p->out_row_stride & 3)) return MOBI_ERR_ALIGN;
  if ((p->q_img_stride & 7) || (p->k_img_stride & 7) || (p->out_img_stride & 3)) return MOBI_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(p->q) | reinterpret_cast<uintptr_t>(p->k) |
       reinterpret_cast<uintptr_t>(p->out)) & 15) return MOBI_ERR_ALIGN;
  if (p->heads > 65535 || p->images > 65535) return MOBI_ERR_UNSUPPORTED;
  AttnArgs a;
  a.q = p->q; a.q_img = p->q_img_stride; a.q_row = p->q_row_stride;
  a.k = p->k; a.k_img = p->k_img_stride; a.k_row = p->k_row_stride;
  a.vt = p->vt; a.vt_img = p->vt_img_stride; a.vt_row = p->vt_row_stride;
  a.out = p->out; a.out_img = p->out_img_stride; a.out_row = p->out_row_stride;
  a.heads = p->heads; a.dh = p->dh; a.tq = p->tq; a.tk = p->tk;
  a.cexp = p->q_log2_scaled ? 1.0f : p->scale * 1.4426950408889634f;
  a.xcd_map = tuning().attn_xcd != 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return p->dtype == MOBI_F16 ? launch_attention<f16_t>(p, a, st) : launch_attention<bf16_t>(p, a, st);
}
