// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//
//   out[m][n] = epi( scale * sum_k A[m][k] * W[n][k] ),  m = (image, y, x), k = (tap, channel)
//
// The activation operand is a GATHER: each row is an output pixel, each 64-channel k-tile belongs to one filter
// tap of one source, so 3x3 / 1x5 / strided / asymmetric-pad convolutions, nearest-x2 upsampling on the load side
// and the channel concat of two sources never materialise an im2col, an upsampled or a concatenated tensor.
// MFMA 16x16x32, fp32 accumulators, wave tile 64 pixels x 80 (64) channels, [row][64 k] LDS images with 128-byte
// rows and a 16-byte-slot XOR swizzle (slot ^= row & 7: conflict-free ds_read_b128 fragment reads).
//
// Three main loops (igemm_prepare picks one per launch, mobi_igemm_kernel_variant reports it):
//   igemm_pp_kernel    persistent 256-pixel tiles, operands by LDS-DMA into three stages, PING-PONG schedule (the
//                      two waves of a SIMD one phase apart: LOAD | MATRIX), deferred register epilogue.  Every
//                      launch whose tiles are full, whose output is row-major T and that has at most one of
//                      bias / per-image vector; also the split-K launches with long k ranges (fp32 slabs).
//   igemm_glds_kernel  same geometry in lockstep, LDS-staged epilogues (transposed / fp32 / ragged / bias AND vector)
//   igemm_ring_kernel    128-pixel tiles, four waves, two blocks per CU, operands by LDS-DMA into a ring of four 32-deep
//                      k-slots: small m, split-K ranges, ragged tiles, every output mode
//   igemm_kernel       128- / 256-pixel tiles staged through registers: operands beyond 2 GB, chunk-major k
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "tuning.h"
#include "igemm_small.h"

// A/B switch (build with -DMOBI_IGEMM_FENCE=1): pin the load / MFMA / LDS-write phases of a k step
#ifndef MOBI_FRAG2
#define MOBI_FRAG2 0
#endif
#ifndef MOBI_DMA_KS
#define MOBI_DMA_KS 1      // k-step of a tile in front of which the next-but-one tile's DMA is issued (A/B: 0 | 1)
#endif
#if defined(MOBI_IGEMM_FENCE) && MOBI_IGEMM_FENCE
#define MOBI_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define MOBI_SCHED_FENCE() ((void)0)
#endif

#ifndef MOBI_DBG_SKIP
#define MOBI_DBG_SKIP 0    // diagnosis only (wrong results): bit 0 = no activation DMA, bit 1 = no weight DMA, bit 2 = no MFMA
#endif
#ifndef MOBI_PP_DMA_KS
#define MOBI_PP_DMA_KS 1   // ping-pong kernel: LOAD phase (k-step) that carries the DMA requests
#endif
#ifndef MOBI_PP_DMA_SPLIT
#define MOBI_PP_DMA_SPLIT 1 // ping-pong kernel: activation pieces requested in LOAD(ks 0), weight pieces in LOAD(ks 1)
#endif
#ifndef MOBI_PP_DMA_AT
#define MOBI_PP_DMA_AT 0   // two-phase schedule: requests in the LOAD phase (0), between the MATRIX phase's k-steps (1),
#endif                     // activations in LOAD + weights in MATRIX (2)
#ifndef MOBI_PP_PHASES
#define MOBI_PP_PHASES 2   // ping-pong kernel: phases per k-tile (4: LOAD / MATRIX per 32-deep k-step; 2: per k-tile --
                           // measured 8 % faster on the 3x3 shapes: half the barriers, 640-cycle MATRIX phases)
#endif
#ifndef MOBI_STAGED_PRIO
#define MOBI_STAGED_PRIO 0 // register-staged kernel: s_setprio 1 around a k-step's MFMAs (A/B)
#endif
#ifndef MOBI_PP_PRIO
#define MOBI_PP_PRIO 1     // ping-pong kernel: s_setprio of the MATRIX phase
#endif
#ifndef MOBI_RING_STAGGER
#define MOBI_RING_STAGGER 1 // 256 x 320 ring kernel: waves 4-7 half a step behind waves 0-3 (0: lockstep, A/B)
#endif
#ifndef MOBI_STAMP
#define MOBI_STAMP 0       // 1: in-kernel phase stamps of the direct-to-LDS kernel (tools/stamp_igemm.py), never in a release build
#endif

namespace mobi {

#if MOBI_STAMP
__device__ unsigned long long* g_stamps = nullptr;      // [block][8]: entry, first tile landed, loop end, epilogue end, hw id, nk
#define MOBI_STAMP_AT(slot)                                                                          \
  do {                                                                                               \
    if (g_stamps && threadIdx.x == 0)                                                                \
      g_stamps[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (slot)] = \
          __builtin_amdgcn_s_memrealtime();                                                          \
  } while (0)
#else
#define MOBI_STAMP_AT(slot) ((void)0)
#endif
#if MOBI_STAMP >= 2
// per wave: accumulated 10-ns ticks of the five phases of a k-tile step (wait, barrier, k-step 0, DMA issue,
// k-step 1) + the number of steps; [block][wave][8]
__device__ unsigned long long* g_phase = nullptr;
#define MOBI_PHASE_DECL unsigned long long ph_t[6]; unsigned ph_acc[5] = {0u, 0u, 0u, 0u, 0u}; unsigned ph_n = 0
#define MOBI_PHASE(i) ph_t[i] = __builtin_amdgcn_s_memrealtime()
#define MOBI_PHASE_ACC()                                                                  \
  do {                                                                                    \
    for (int i_ = 0; i_ < 5; ++i_) ph_acc[i_] += (unsigned)(ph_t[i_ + 1] - ph_t[i_]);     \
    ++ph_n;                                                                               \
  } while (0)
#else
#define MOBI_PHASE_DECL ((void)0)
#define MOBI_PHASE(i) ((void)0)
#define MOBI_PHASE_ACC() ((void)0)
#endif

// 16 zero bytes in the code object: the source of every padded / out-of-range 16-byte piece of an operand tile
// (a plain load from here instead of a load + select; the library allocates nothing).
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

struct IgemmArgs {
  const void* src0; const void* src1;
  int c0, c1, C;
  int hin, win, up, hout, wout, hw_out;
  int kh, kw, stride, pad_h, pad_w;
  int img_pix_stride;      // pixels between images of the sources (elements / channels)
  const void* weight; long long w_group_stride;
  int n_packed, cout, ktot, nk;
  int M;                 // rows per group
  int imgs_per_group;
  const float* bias; const float* rowvec; int rowvec_stride;
  const void* residual; long long res_img_stride;
  void* out; long long out_img_stride;
  int out_mode, epilogue;
  float scale;
  int tiles_m, tiles_n;
  int n_major;             // work list walks the pixel tiles of one channel tile first: an XCD (a contiguous range of the
                           // list) then streams ITS slice of the weights instead of all of them (launches whose weights
                           // outweigh their activations: the 16 x 16 / 8 x 8 levels)
  int src0_bytes, src1_bytes, w_bytes, fast, glds;
  int k_order;             // 0: k = tap*C + c;  1: k = (c/64)*taps*64 + tap*64 + c%64 (FAST shapes only)   // buffer extents for the FAST path's descriptors
  int wm;                  // waves along the pixel axis (2 or 4): block tile = 64*wm pixels
  int splits, nk_per;      // split-K: blockIdx.y owns k-tiles [y*nk_per, (y+1)*nk_per)
  float* split_ws;         // f32 [splits][M][n_packed] partial sums (NULL: single pass)
  const float* ln_svec;    // LayerNorm folded into this 1 x 1 launch (see ln_fold_acc): row sums of the packed W diag(gamma), or NULL
  float ln_eps;
  int sync_mode;           // 0: no `sync`; 1: device-coherent slab traffic; 2: a tile's blocks share one XCD's L2 (see splitk_arrive_and_finish)
  int* sync;               // split-K finished inside the launch: one arrival counter per output tile (zero before and after
                           // the launch), or NULL: the slabs are summed by igemm_splitk_reduce_kernel (a second launch)
  int ring_direct;         // 256 x 320 ring tiles: register epilogue (full tiles, row-major T output, bias OR per-image vector)
  int sm_direct;           // 128 x 160 ring tiles: register epilogue / register slab stores (full tiles)
  const void* w_tiled;     // W as 1-KiB request images (ring kernels), or NULL
  int sm64;                // 128-pixel tiles on the 64-deep-step kernel (igemm_ring64_kernel)
  int lin_window;          // direct-to-LDS kernel: window pixels are linear in the tap (no upsampling, <= 16 taps)
  int epi_direct;          // direct-to-LDS kernel: register epilogue (full tiles, row-major T output)
  int pp;                  // register-epilogue launch on the ping-pong kernel
  int sm;                  // 128-pixel tiles on the LDS-DMA ring kernel (igemm_ring_kernel<.., false>)
  int wide;                // ring kernel with eight waves: 2 = 256 x 320 (256) tiles, 1 = 128 x 320 (256) tiles
  int hw_shift, w_shift;   // log2(hw_out), log2(wout) when both are powers of two (ping-pong kernel), else -1
  int small;               // 32 | 64: the operands-in-registers kernel of csrc/igemm_small.hip with that square tile (0: not)
};

// (a macro, not a function of `a`: a reference to the kernel-argument struct makes the compiler keep a copy of it in scratch)
// (the ping-pong kernel, at its register limit, always walks pixel-major: igemm_prepare leaves n_major 0 for it)
#define MOBI_TILE_OF_M(L_, tile_m_, tile_n_) const int tile_n_ = (L_) % a.tiles_n, tile_m_ = (L_) / a.tiles_n
#define MOBI_TILE_OF(L_, tile_m_, tile_n_)                                                     \
  const int tdiv_##tile_m_ = __builtin_amdgcn_readfirstlane(a.n_major ? a.tiles_m : a.tiles_n); \
  const int tq_##tile_m_ = (L_) / tdiv_##tile_m_, tr_##tile_m_ = (L_) - tq_##tile_m_ * tdiv_##tile_m_; \
  const int tile_m_ = a.n_major ? tr_##tile_m_ : tq_##tile_m_;                                 \
  const int tile_n_ = a.n_major ? tq_##tile_m_ : tr_##tile_m_


__device__ __forceinline__ void vm_store16_dev(void* p, const u32x4& v);      // device-coherent 16-byte store (below)

// ---------------------------------------------------------------------------------------------------------
// Epilogue shared by both main-loop variants: accumulators -> per-wave fp32 LDS tile -> coalesced 16-byte rows
// with scale / bias / per-image vector / residual / GEGLU, fp32 or transposed output, split-K partial slabs.
// `stage` is this wave's private LDS area (the caller guarantees every wave is past its last fragment read).
// RPP = pixel rows of the wave's 64 x WAVE_N tile that go through the stage per pass (32: 2 passes, 16: 4 passes;
// the direct-to-LDS kernel uses 16 so that all eight stages fit the one k-stage that is free at that point).
// Latency: the bias is added in the accumulator domain from registers loaded before anything else, and the
// residual rows of ALL passes are requested up front, so no global load sits between an LDS read and a store.
// ---------------------------------------------------------------------------------------------------------
template <int RPP, bool TR, int NT>
struct EpiGeom {
  static constexpr int WAVE_N = NT * 16;
  static constexpr int STRIDE = TR ? (RPP + 4) : (WAVE_N + 4);      // floats
  static constexpr int ROWS = TR ? WAVE_N : RPP;
  static constexpr int BYTES = ROWS * STRIDE * 4;                   // per wave
};

// RES_EARLY = false: the residual rows of a pass are requested at the start of THAT pass (the 256 x 320 ring kernel holds
// 160 accumulator registers through the epilogue: all passes' rows up front would spill)
template <typename T, int NT, bool TR, int RPP, bool RES_EARLY = true>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& a, float* stage, f32x4 (&acc)[NT][4], int lane,
                                               int group, int nw0, int mw0) {
  constexpr int WAVE_N = NT * 16;
  constexpr int STAGE_STRIDE = EpiGeom<RPP, TR, NT>::STRIDE;
  constexpr int MT = RPP / 16;                 // 16-pixel MFMA tiles per pass
  constexpr int NPASS = 4 / MT;
  const int r16 = lane & 15, g4 = lane >> 4;
  const float scale = a.scale;
  T* __restrict__ outT = reinterpret_cast<T*>(a.out);
  float* __restrict__ outF = reinterpret_cast<float*>(a.out);
  const T* __restrict__ resid = reinterpret_cast<const T*>(a.residual);
  const bool plain = !TR && !a.split_ws && a.epilogue != MOBI_EPI_GEGLU;
  const bool use_bias = a.bias != nullptr && !a.split_ws;

  // image / in-image index of the wave's first row (wave-uniform); rows of a task are found by a short walk
  const int mw0c = mw0 < a.M ? mw0 : 0;
  const int img0 = __builtin_amdgcn_readfirstlane(mw0c / a.hw_out);
  const int rem0 = mw0c - img0 * a.hw_out;
  auto locate = [&](int row_in_tile, int& img, int& rem) {
    img = img0; rem = rem0 + row_in_tile;
    while (rem >= a.hw_out) { rem -= a.hw_out; ++img; }
  };

  // ---- bias in the accumulator domain ----------------------------------------------------------------------
  f32x4 bias4[NT];
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    bias4[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (use_bias) {
      if (!TR) {
        const int n = nw0 + ni * 16 + g4 * 4;                  // D rows = channel: 4 consecutive channels per lane
        if (n < a.n_packed) bias4[ni] = *reinterpret_cast<const f32x4*>(a.bias + n);
      } else {
        const int n = nw0 + ni * 16 + r16;                     // D cols = channel: one channel per lane
        const float b = n < a.cout ? a.bias[n] : 0.f;
        bias4[ni] = f32x4{b, b, b, b};
      }
    }
  }

  // ---- residual rows of every pass (plain path) --------------------------------------------------------------
  constexpr int TPR_P = WAVE_N / 8;
  constexpr int NIT_P = (RPP * TPR_P + 63) / 64;
  u32x4 rres[RES_EARLY ? NPASS : 1][NIT_P];
  auto request_rows = [&](int pass, u32x4 (&dst)[NIT_P]) {
#pragma unroll
    for (int it = 0; it < NIT_P; ++it) {
      dst[it] = u32x4{0u, 0u, 0u, 0u};
      const int task = lane + 64 * it;
      const int row = task / TPR_P, cg = task - row * TPR_P;
      const int rt = pass * RPP + row;
      const int n = nw0 + cg * 8;
      if (task < RPP * TPR_P && mw0 + rt < a.M && n < a.cout) {
        int img, rem;
        locate(rt, img, rem);
        const long long gi = (long long)(group * a.imgs_per_group + img);
        dst[it] = ld16(resid + gi * a.res_img_stride + (long long)rem * a.cout + n);
      }
    }
  };
  if (RES_EARLY && plain && resid) {
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) request_rows(pass, rres[pass]);
  }

#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    if (!RES_EARLY && plain && resid) request_rows(pass, rres[0]);
    // (the caller's last barrier ordered every wave's fragment reads before these writes; the passes of one
    //  wave are ordered by the waits below)
    if (!TR) {
      // D rows = channel (g4*4 + r), cols = pixel (r16): lane owns 4 consecutive channels
#pragma unroll
      for (int ml = 0; ml < MT; ++ml)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          f32x4 v = acc[ni][pass * MT + ml];
          v = v * scale + bias4[ni];
          *reinterpret_cast<f32x4*>(stage + (ml * 16 + r16) * STAGE_STRIDE + ni * 16 + g4 * 4) = v;
        }
    } else {
      // D rows = pixel (g4*4 + r), cols = channel (r16): lane owns 4 consecutive pixels
#pragma unroll
      for (int ml = 0; ml < MT; ++ml)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
          f32x4 v = acc[ni][pass * MT + ml];
          v = v * scale + bias4[ni];
          *reinterpret_cast<f32x4*>(stage + (ni * 16 + r16) * STAGE_STRIDE + ml * 16 + g4 * 4) = v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): this wave's LDS writes are done
    __builtin_amdgcn_wave_barrier();

    const int rp0 = pass * RPP;              // first row (inside the wave tile) of this pass
    if (!TR) {
      if (a.split_ws) {
        // split-K: raw fp32 partial sums, finished by igemm_splitk_reduce_kernel
        constexpr int TPR = WAVE_N / 8;
        float* __restrict__ wsp = a.split_ws + (long long)blockIdx.y * a.M * a.n_packed;
        for (int task = lane; task < RPP * TPR; task += 64) {
          const int row = task / TPR, cg = task - row * TPR;
          const int m = mw0 + rp0 + row;
          const int n = nw0 + cg * 8;
          if (m >= a.M || n >= a.n_packed) continue;
          const float* sp = stage + row * STAGE_STRIDE + cg * 8;
          float* d = wsp + (long long)m * a.n_packed + n;
          if (a.sync_mode == 1) {                              // finished inside the launch, device-coherent slab stores
            vm_store16_dev(d, *reinterpret_cast<const u32x4*>(sp));
            vm_store16_dev(d + 4, *reinterpret_cast<const u32x4*>(sp + 4));
          } else {
            *reinterpret_cast<f32x4*>(d) = *reinterpret_cast<const f32x4*>(sp);
            *reinterpret_cast<f32x4*>(d + 4) = *reinterpret_cast<const f32x4*>(sp + 4);
          }
        }
      } else if (a.epilogue == MOBI_EPI_GEGLU) {
        // packed columns: per 16-column MFMA tile, 8 value columns then the 8 gate columns of the same outputs
        constexpr int TPR = WAVE_N / 16;
        for (int task = lane; task < RPP * TPR; task += 64) {
          const int row = task / TPR, t = task - row * TPR;
          const int m = mw0 + rp0 + row;
          const int oc = (nw0 >> 1) + t * 8;
          if (m >= a.M || oc >= a.cout) continue;
          const float* sp = stage + row * STAGE_STRIDE + t * 16;
          float av[8], gv[8], o[8];
          ld8f(sp, av);
          ld8f(sp + 8, gv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = av[j] * gelu_erf_f(gv[j]);
          int img, rem;
          locate(rp0 + row, img, rem);
          const long long gi = (long long)(group * a.imgs_per_group + img);
          st16(outT + gi * a.out_img_stride + (long long)rem * a.cout + oc, pack8<T>(o));
        }
      } else {
#pragma unroll
        for (int it = 0; it < NIT_P; ++it) {
          const int task = lane + 64 * it;
          const int row = task / TPR_P, cg = task - row * TPR_P;
          const int m = mw0 + rp0 + row;
          const int n = nw0 + cg * 8;
          if (task >= RPP * TPR_P || m >= a.M || n >= a.cout) continue;
          const float* sp = stage + row * STAGE_STRIDE + cg * 8;
          float o[8];
          ld8f(sp, o);
          int img, rem;
          locate(rp0 + row, img, rem);
          const long long gi = (long long)(group * a.imgs_per_group + img);
          if (a.rowvec) {
            float rv[8];
            ld8f(a.rowvec + gi * a.rowvec_stride + n, rv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += rv[j];
          }
          if (resid) {
            float rf[8];
            unpack8<T>(rres[RES_EARLY ? pass : 0][it], rf);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += rf[j];
          }
          const long long off = gi * a.out_img_stride + (long long)rem * a.cout + n;
          if (a.out_mode == MOBI_OUT_ROWS_F32) {
            *reinterpret_cast<f32x4*>(outF + off) = f32x4{o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4*>(outF + off + 4) = f32x4{o[4], o[5], o[6], o[7]};
          } else {
            st16(outT + off, pack8<T>(o));
          }
        }
      }
    } else {
      // transposed output [image][cout][hw]
      constexpr int PG = RPP / 8;            // 8-pixel groups per staged row
      const bool vec_ok = (a.hw_out & 7) == 0;
      for (int task = lane; task < WAVE_N * PG; task += 64) {
        const int crow = task / PG, pg = task - crow * PG;
        const int n = nw0 + crow;
        const int m = mw0 + rp0 + pg * 8;
        if (n >= a.cout || m >= a.M) continue;
        const float* sp = stage + crow * STAGE_STRIDE + pg * 8;
        float o[8];
        ld8f(sp, o);
        if (vec_ok) {
          int img, rem;
          locate(rp0 + pg * 8, img, rem);
          const long long gi = (long long)(group * a.imgs_per_group + img);
          st16(outT + gi * a.out_img_stride + (long long)n * a.hw_out + rem, pack8<T>(o));
        } else {
          for (int j = 0; j < 8; ++j) {
            const int mj = m + j;
            if (mj >= a.M) break;
            const int img = mj / a.hw_out, rem = mj - img * a.hw_out;
            const long long gi = (long long)(group * a.imgs_per_group + img);
            outT[gi * a.out_img_stride + (long long)n * a.hw_out + rem] = from_f32<T>(o[j]);
          }
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Register epilogue of the direct-to-LDS kernel (full tiles, row-major T output, no per-image vector, no split):
// no LDS, no barrier.  A lane-row exchange (v_permlane16_swap / v_permlane32_swap, gfx950) turns the MFMA
// accumulator layout (4 consecutive channels of one pixel per lane and tile) into 8 consecutive channels of one
// pixel per lane (plain: one 16-byte store per lane and tile pair) or pairs every value with its gate (GEGLU:
// 4 outputs per lane, one 8-byte store).  Bias and residual rows were requested at the START of the output tile's
// k loop by explicit loads the compiler does not track; the kernel counts every vector-memory instruction it
// issues, so all waits are exact s_waitcnt vmcnt(N) immediates and never drain the DMA of the next k-tiles.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x4 vm_load16(const void* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}
// (the s_nop 1 is part of the statement: hipcc does not know that a 16-byte store reads its data registers for
//  two more wave cycles and would let its next VALU instruction overwrite them -- seen as garbage in the first two
//  of a lane's eight outputs, last four lanes of each 16-lane row)
__device__ __forceinline__ void vm_store16(void* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void vm_store8(void* p, const u32x2& v) {
  asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

// =========================================================================================================
// Split-K finished INSIDE the launch (no reduce launch, no host round trip between the partial sums and the epilogue).
// Every block of a split launch writes its fp32 slab tile, then ARRIVES at its output tile's counter; the block that
// arrives last sums the tile's slabs in ascending split order -- the same sums in the same order as
// igemm_splitk_reduce_kernel, so the result is bit-identical to the two-launch form and does not depend on who
// arrived last -- applies the epilogue (bias, per-image vector, residual, one rounding) and returns the counter to zero.
// Nobody waits for anybody: a block that is not last simply ends.
//
// Coherence without cache-wide fences.  The eight XCDs' L2s are not coherent with each other, and an agent-scope
// release / acquire pair (buffer_wbl2 + buffer_inv: what __threadfence() emits) writes back and invalidates WHOLE L2s --
// round 2 measured that form at 13.55 against 7.58 ms per mobi_nusc_256 step.  Here only the slab traffic itself is made
// device-coherent: slab stores and slab loads carry sc1 (device scope: written through to / fetched from the memory
// side, which is where the counter's device-scope atomic lives too), every wave drains its stores (vmcnt(0)) before the
// block's single arrival, and nothing else of the launch (operands, outputs) changes its cache policy.
// =========================================================================================================
__device__ __forceinline__ void vm_store16_dev(void* p, const u32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// one 16-byte slab store: device-coherent when the launch finishes its own split (wave-uniform choice)
__device__ __forceinline__ void slab_store16(bool dev, void* p, const u32x4& v) {
  if (dev) vm_store16_dev(p, v); else vm_store16(p, v);
}

// The arithmetic of igemm_splitk_reduce_kernel on rows [m0, m0 + bm) x columns [n0, n0 + bn) of the output, by the
// THREADS threads of one block, for at most FOUR slabs (the host keeps launches with more splits on the reduce launch: one
// block summing 8-16 slabs of its tile is a chain of memory round trips that the other 255 CUs spend idle -- measured
// 54.6 against 22.3 us at 16 splits, profiles/r05_fused_split_lab.txt).  Four items (16-byte column groups) per thread
// and pass, every slab piece of the pass requested before the first is used.  DEV: slab loads with sc1.
template <typename T, int THREADS, bool DEV>
__device__ __forceinline__ void splitk_finish_tile(const IgemmArgs& a, int m0, int bm, int n0, int bn) {
  const int m1 = min(a.M, m0 + bm), n1 = min(a.cout, n0 + bn);
  if (m1 <= m0 || n1 <= n0) return;
  const int vpr = (n1 - n0) >> 3;                            // cout % 8 == 0 (host check), tiles start at multiples of 8
  const int total = (m1 - m0) * vpr;
  T* __restrict__ outT = reinterpret_cast<T*>(a.out);
  float* __restrict__ outF = reinterpret_cast<float*>(a.out);
  const T* __restrict__ resid = reinterpret_cast<const T*>(a.residual);
  const int S = a.splits;                                    // 2 .. 4
  const unsigned slab_bytes = (unsigned)a.M * (unsigned)a.n_packed * 4u;     // S * slab_bytes < 2^32 (host check)
  // slab pieces through a buffer descriptor: loads the compiler itself waits for (an inline-asm load's destination may be
  // copied -- spilled, moved to an accumulator register -- before a hand-placed wait), with the cache policy in the instruction
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.split_ws, 0, (int)(slab_bytes * (unsigned)S), 0x00020000);
  constexpr int AUX = DEV ? 16 : 0;                          // sc1
  constexpr int IB = 2;
  for (int i0 = threadIdx.x; i0 < total; i0 += IB * THREADS) {
    u32x4 v[IB][4][2];
    int mm[IB], nn[IB];
#pragma unroll
    for (int b = 0; b < IB; ++b) {
      const int i = min(i0 + b * THREADS, total - 1);        // (past the end: a harmless duplicate of the last item, not stored)
      const int mr = i / vpr;
      mm[b] = m0 + mr; nn[b] = n0 + (i - mr * vpr) * 8;
      const unsigned off = ((unsigned)mm[b] * (unsigned)a.n_packed + (unsigned)nn[b]) * 4u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < S) {
          v[b][u][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)u * slab_bytes, 0, AUX);
          v[b][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)u * slab_bytes + 16u, 0, AUX);
        }
      }
    }
#pragma unroll
    for (int b = 0; b < IB; ++b) {
      if (i0 + b * THREADS >= total) break;
      const int m = mm[b], n = nn[b];
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (u < S) {                                         // ascending split order: the reduce launch's sums
          const f32x4 x0 = __builtin_bit_cast(f32x4, v[b][u][0]), x1 = __builtin_bit_cast(f32x4, v[b][u][1]);
#pragma unroll
          for (int j = 0; j < 4; ++j) { o[j] += x0[j]; o[4 + j] += x1[j]; }
        }
      }
      const int img = m / a.hw_out, rem = m - img * a.hw_out;
      if (a.bias) {
        float bb[8];
        ld8f(a.bias + n, bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += bb[j];
      }
      if (a.rowvec) {
        float rv[8];
        ld8f(a.rowvec + (long long)img * a.rowvec_stride + n, rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += rv[j];
      }
      if (resid) {
        float rf[8];
        unpack8<T>(ld16(resid + (long long)img * a.res_img_stride + (long long)rem * a.cout + n), rf);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += rf[j];
      }
      const long long off = (long long)img * a.out_img_stride + (long long)rem * a.cout + n;
      if (a.out_mode == MOBI_OUT_ROWS_F32) {
        *reinterpret_cast<f32x4*>(outF + off) = f32x4{o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(outF + off + 4) = f32x4{o[4], o[5], o[6], o[7]};
      } else {
        st16(outT + off, pack8<T>(o));
      }
    }
  }
}

// arrive + finish: `sync_mode` 1 = device-coherent slab traffic (sc1 stores / loads, a device-scope counter: correct wherever
// the blocks of a tile run); 2 = the blocks of a tile share ONE XCD's L2 (the host has checked that the grid deals them so:
// consecutive workgroup ids go round the eight XCDs, so blocks (x, y) of a grid whose x extent is a multiple of 8 land on XCD
// x % 8 for every y): plain stores, an L2 atomic, loads that bypass only the CU's own vector cache
template <typename T, int THREADS>
__device__ __forceinline__ void splitk_arrive_and_finish(const IgemmArgs& a, int tile, int m0, int bm, int n0, int bn, int* s_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's slab stores have been acknowledged
  __syncthreads();                                           // ... and so have every other wave's of the block
  if (threadIdx.x == 0) {
    *s_flag = a.sync_mode == 2 ? __hip_atomic_fetch_add(a.sync + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                               : __hip_atomic_fetch_add(a.sync + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (*s_flag != a.splits - 1) return;                       // not the last to arrive: done
  if (threadIdx.x == 0) {
    if (a.sync_mode == 2) __hip_atomic_store(a.sync + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(a.sync + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (a.sync_mode == 2) splitk_finish_tile<T, THREADS, false>(a, m0, bm, n0, bn);
  else splitk_finish_tile<T, THREADS, true>(a, m0, bm, n0, bn);
}

// wait until at most n (wave-uniform; rounded DOWN to an encoded value, which only waits longer) vector-memory
// operations of this wave are outstanding
__device__ __forceinline__ void wait_vmcnt_le(int n) {
#define MOBI_VMW(k) else if (n >= k) asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory")
  if (n >= 37) asm volatile("s_waitcnt vmcnt(37)" ::: "memory");
  MOBI_VMW(32); MOBI_VMW(27); MOBI_VMW(26); MOBI_VMW(22); MOBI_VMW(18); MOBI_VMW(17); MOBI_VMW(14); MOBI_VMW(12);
  MOBI_VMW(10); MOBI_VMW(8); MOBI_VMW(7); MOBI_VMW(6); MOBI_VMW(5); MOBI_VMW(4);
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef MOBI_VMW
}

// the same with the steady-state count of a k loop tried first (the cascade above is ~11 scalar compare / branch pairs
// before it reaches a small count; the ping-pong kernel waits twice per k-tile with n = the requests of one k-tile)
#ifndef MOBI_PP_GEGLU_STORE16
#define MOBI_PP_GEGLU_STORE16 1   // GEGLU register epilogue: pairs of 16-channel tiles stored as 16 bytes per lane
#endif
#ifndef MOBI_PP_WDUP
#define MOBI_PP_WDUP 0
#endif
#ifndef MOBI_PP_FASTWAIT
#define MOBI_PP_FASTWAIT 1
#endif
__device__ __forceinline__ void wait_vmcnt_le_fast(int n) {
  if (!MOBI_PP_FASTWAIT) { wait_vmcnt_le(n); return; }
  if (n == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else wait_vmcnt_le(n);
}

// the four lane-row swaps of one accumulator tile pair as ONE statement: the hazard slots (VALU write -> swap read before,
// swap write -> VALU read after) are paid once per four independent swaps instead of once per swap
#define MOBI_SWAP4(INSN, X, Y)                                                                                       \
  asm volatile("s_nop 1\n\t" INSN " %0, %4\n\t" INSN " %1, %5\n\t" INSN " %2, %6\n\t" INSN " %3, %7\n\ts_nop 1" \
               : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(Y[0]), "+v"(Y[1]), "+v"(Y[2]), "+v"(Y[3]))

template <int NT>
struct DirectEpiRegs {
  u32x4 bias[NT];          // f32 x 4: channels 16 ni + 4 g4 + (0..3) of the wave's columns (bias OR per-image vector)
  u32x4 res[NT][2];        // T x 8: the lane's 8 output channels of its pixel in tile pair p
};

// request bias / residual of output tile (mw0, nw0); returns the number of vector-memory instructions issued
// VEC = false: the caller has put bias / per-image vector into the accumulators' initial value (ping-pong kernel)
template <typename T, int NT, bool VEC = true, bool RES = true>
__device__ __forceinline__ int direct_epilogue_request(const IgemmArgs& a, DirectEpiRegs<NT>& q, int lane, int group,
                                                       int nw0, int mw0) {
  const int r16 = lane & 15, g4 = lane >> 4;
  int n_issued = 0;
  if (VEC && (a.bias || a.rowvec)) {                                    // never both (checked on the host)
    const float* vec = a.bias;
    if (a.rowvec) {                                            // the wave's 64 pixels lie in one image: hw_out % 64 == 0
      const int img = __builtin_amdgcn_readfirstlane(mw0 / a.hw_out);
      vec = a.rowvec + (long long)(group * a.imgs_per_group + img) * a.rowvec_stride;
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) q.bias[ni] = vm_load16(vec + nw0 + ni * 16 + g4 * 4);
    n_issued += NT;
  }
  if (RES && a.residual) {                                     // plain epilogue only (checked on the host)
    const T* __restrict__ resid = reinterpret_cast<const T*>(a.residual);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int m = mw0 + (2 * p + (g4 & 1)) * 16 + r16;
      const int img = m / a.hw_out, rem = m - img * a.hw_out;
      const T* rowp = resid + (long long)(group * a.imgs_per_group + img) * a.res_img_stride + (long long)rem * a.cout +
                      nw0 + 8 * (g4 >> 1);
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) q.res[ni][p] = vm_load16(rowp + ni * 16);
    }
    n_issued += 2 * NT;
  }
  return n_issued;
}

// returns the number of vector-memory instructions issued (stores)
template <typename T, int NT, bool GEGLU, bool VEC = true>
__device__ __forceinline__ int direct_epilogue(const IgemmArgs& a, f32x4 (&acc)[NT][4], DirectEpiRegs<NT>& q, int lane,
                                               int group, int nw0, int mw0) {
  const int r16 = lane & 15, g4 = lane >> 4;
  const float scale = a.scale;
  T* __restrict__ outT = reinterpret_cast<T*>(a.out);
  // the requested registers are complete (caller waited): pin their uses behind that wait
  const bool has_vec = VEC && (a.bias || a.rowvec);
  if (has_vec) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) asm volatile("" : "+v"(q.bias[ni]));
  }
  if (!GEGLU && a.residual) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) { asm volatile("" : "+v"(q.res[ni][0])); asm volatile("" : "+v"(q.res[ni][1])); }
  }
  constexpr bool geglu = GEGLU;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // plain: after the 16-lane-row swap a lane holds pixel tile 2p + (g4 & 1), channels 16 ni + 8 (g4 >> 1) + 0..7
    // GEGLU: after the 32-lane swap a lane holds pixel tile 2p + (g4 >> 1), outputs 8 ni + 4 (g4 & 1) + 0..3
    const int mi = geglu ? 2 * p + (g4 >> 1) : 2 * p + (g4 & 1);
    const int m = mw0 + mi * 16 + r16;
    const int img = m / a.hw_out, rem = m - img * a.hw_out;
    T* rowp = outT + (long long)(group * a.imgs_per_group + img) * a.out_img_stride + (long long)rem * a.cout;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      f32x4 x = acc[ni][2 * p] * scale, y = acc[ni][2 * p + 1] * scale;
      if (has_vec) {
        const f32x4 b = __builtin_bit_cast(f32x4, q.bias[ni]);
        x += b; y += b;
      }
      // (inline asm: hipcc 7.2 merges the four __builtin_amdgcn_permlane*_swap calls of a tile into one and
      //  broadcasts its result; the s_nops cover the VALU-write -> permlane-swap -> VALU-read hazard slots the
      //  compiler would otherwise insert itself)
      float xs[4] = {x[0], x[1], x[2], x[3]}, ys[4] = {y[0], y[1], y[2], y[3]};
      if constexpr (geglu) MOBI_SWAP4("v_permlane32_swap_b32", xs, ys);
      else MOBI_SWAP4("v_permlane16_swap_b32", xs, ys);
      if constexpr (geglu) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = xs[r] * gelu_erf_f(ys[r]);
        vm_store8(rowp + (nw0 >> 1) + ni * 8 + 4 * (g4 & 1), pack4<T>(o));
      } else {
        float o[8] = {xs[0], xs[1], xs[2], xs[3], ys[0], ys[1], ys[2], ys[3]};
        if (a.residual) {
          float rf[8];
          unpack8<T>(q.res[ni][p], rf);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += rf[j];
        }
        vm_store16(rowp + nw0 + ni * 16 + 8 * (g4 >> 1), pack8<T>(o));
      }
    }
  }
  return 2 * NT;
}

// ---------------------------------------------------------------------------------------------------------
// Register epilogue of the 256 x 320 ring tiles: a wave holds 128 pixels x NT * 16 channels = 32 NT accumulator
// registers, so bias and residual cannot all sit in registers beside them.  Bias / per-image vector START the sums (the
// kernel loads them before its k loop); here four passes of 32 pixels (pass k = 16-pixel tile pair k): the residual rows
// of pass k + 1 are requested before pass k is converted and stored (two sets of NT registers).  Counted waits:
// vector-memory operations retire in order, the wait before pass k leaves only the NT younger requests out.
// ---------------------------------------------------------------------------------------------------------
template <typename T, int NT, bool GEGLU>
__device__ __forceinline__ void ring_register_epilogue(const IgemmArgs& a, f32x4 (&acc)[NT][8], int lane, int group, int nw0,
                                                       int mw0) {
  const int r16 = lane & 15, g4 = lane >> 4;
  T* __restrict__ outT = reinterpret_cast<T*>(a.out);
  const T* __restrict__ resid = reinterpret_cast<const T*>(a.residual);
  const bool has_res = !GEGLU && a.residual;
  u32x4 res[2][NT];                                            // (defined on every path: the untracked loads below are
#pragma unroll                                                 //  pinned by in / out asm operands, which must not see undef)
  for (int ni = 0; ni < NT; ++ni) res[0][ni] = res[1][ni] = u32x4{0u, 0u, 0u, 0u};
  auto out_row = [&](int k) {                                  // pixel this lane stores in pass k
    const int mi = GEGLU ? 2 * k + (g4 >> 1) : 2 * k + (g4 & 1);
    return mw0 + mi * 16 + r16;
  };
  auto request_res = [&](int k, u32x4 (&r)[NT]) {
    const int m = out_row(k);
    const int img = m / a.hw_out, rem = m - img * a.hw_out;
    const T* rowp = resid + (long long)(group * a.imgs_per_group + img) * a.res_img_stride + (long long)rem * a.cout + nw0 +
                    8 * (g4 >> 1);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) r[ni] = vm_load16(rowp + ni * 16);
  };
  if (has_res) request_res(0, res[0]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (has_res) {
      // in flight, oldest first: rows of pass k | the NT stores of pass k - 1 | rows of pass k + 1 (requested now)
      if (k + 1 < 4) request_res(k + 1, res[(k + 1) & 1]);
      const bool both = k > 0 && k + 1 < 4;                    // (k is a literal once the loop is unrolled)
      if (both) { if (NT == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
      else                  { if (NT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    }
    if constexpr (!GEGLU) {
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) asm volatile("" : "+v"(res[k & 1][ni]));
    }
    const int m = out_row(k);
    const int img = m / a.hw_out, rem = m - img * a.hw_out;
    T* rowp = outT + (long long)(group * a.imgs_per_group + img) * a.out_img_stride + (long long)rem * a.cout;
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const f32x4 x = acc[ni][2 * k], y = acc[ni][2 * k + 1];
      float xs[4] = {x[0], x[1], x[2], x[3]}, ys[4] = {y[0], y[1], y[2], y[3]};
      if constexpr (GEGLU) MOBI_SWAP4("v_permlane32_swap_b32", xs, ys);
      else MOBI_SWAP4("v_permlane16_swap_b32", xs, ys);
      if constexpr (GEGLU) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = xs[r] * gelu_erf_f(ys[r]);
        vm_store8(rowp + (nw0 >> 1) + ni * 8 + 4 * (g4 & 1), pack4<T>(o));
      } else {
        float o[8] = {xs[0], xs[1], xs[2], xs[3], ys[0], ys[1], ys[2], ys[3]};
        if (has_res) {
          float rf[8];
          unpack8<T>(res[k & 1][ni], rf);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += rf[j];
        }
        vm_store16(rowp + nw0 + ni * 16 + 8 * (g4 >> 1), pack8<T>(o));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Register epilogue of the ping-pong kernel.  Differences to the one above: bias / per-image vector are
// already in the sums (they started them); only the residual rows of tile pair p = 0 stay in registers over the
// k loop (NT x 4 VGPRs), those of pair p = 1 are requested at the START of the epilogue and land while pair 0 is
// converted and stored (their registers are the k loop's fragment registers, free at a tile boundary).
// ---------------------------------------------------------------------------------------------------------
template <int NT>
struct PpEpiRegs {
  u32x4 res0[NT];          // T x 8: the lane's 8 output channels of its pixel in tile pair 0
};
template <typename T, int NT>
__device__ __forceinline__ void pp_residual_request(const IgemmArgs& a, u32x4 (&r)[NT], int p, int lane, int group,
                                                    int nw0, int mw0) {
  const int r16 = lane & 15, g4 = lane >> 4;
  const T* __restrict__ resid = reinterpret_cast<const T*>(a.residual);
  const int m = mw0 + (2 * p + (g4 & 1)) * 16 + r16;
  const int img = m >> a.hw_shift, rem = m & (a.hw_out - 1);
  const T* rowp = resid + (long long)(group * a.imgs_per_group + img) * a.res_img_stride + (long long)rem * a.cout +
                  nw0 + 8 * (g4 >> 1);
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) r[ni] = vm_load16(rowp + ni * 16);
}
// request of pair 0 at the start of an output tile; returns the number of vector-memory instructions issued
#ifndef MOBI_PP_RES0_EARLY
#define MOBI_PP_RES0_EARLY 0   // 1: pair 0's residual rows are requested at the START of the tile's k loop and held in NT x 4
#endif                         //    VGPRs across it (the NT = 5 instance then needs 264 registers: 8 spilled to scratch)
template <typename T, int NT, bool GEGLU>
__device__ __forceinline__ int pp_epilogue_request(const IgemmArgs& a, PpEpiRegs<NT>& q, int lane, int group, int nw0,
                                                   int mw0) {
  if (!MOBI_PP_RES0_EARLY || GEGLU || !a.residual) return 0;
  pp_residual_request<T, NT>(a, q.res0, 0, lane, group, nw0, mw0);
  return NT;
}
// the caller has waited for pair 0's rows.  `vm_after` = vector-memory instructions the caller issued after its own
// last counted one ... not needed: this function only waits for what IT issued (pair 1's rows, behind NT stores).
template <typename T, int NT, bool GEGLU>
__device__ __forceinline__ int pp_epilogue(const IgemmArgs& a, f32x4 (&acc)[NT][4], PpEpiRegs<NT>& q, int lane, int group,
                                           int nw0, int mw0) {
  const int r16 = lane & 15, g4 = lane >> 4;
  T* __restrict__ outT = reinterpret_cast<T*>(a.out);
  const bool resid = !GEGLU && a.residual;
  u32x4 res1[NT];
  int n_issued = 0;
  if (resid) {
    if (!MOBI_PP_RES0_EARLY) {                               // both pairs now: pair 0's rows land first (issue order)
      pp_residual_request<T, NT>(a, q.res0, 0, lane, group, nw0, mw0);
      n_issued += NT;
    }
    pp_residual_request<T, NT>(a, res1, 1, lane, group, nw0, mw0);
    n_issued += NT;
    if (!MOBI_PP_RES0_EARLY) {
      if (NT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) asm volatile("" : "+v"(q.res0[ni]));
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // plain: after the 16-lane-row swap a lane holds pixel tile 2p + (g4 & 1), channels 16 ni + 8 (g4 >> 1) + 0..7
    // GEGLU: after the 32-lane swap a lane holds pixel tile 2p + (g4 >> 1), outputs 8 ni + 4 (g4 & 1) + 0..3
    const int mi = GEGLU ? 2 * p + (g4 >> 1) : 2 * p + (g4 & 1);
    const int m = mw0 + mi * 16 + r16;
    const int img = m >> a.hw_shift, rem = m & (a.hw_out - 1);
    T* rowp = outT + (long long)(group * a.imgs_per_group + img) * a.out_img_stride + (long long)rem * a.cout;
    if (p == 1 && resid) {                                   // pair 1's rows: only pair 0's NT stores are younger
      if (NT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) asm volatile("" : "+v"(res1[ni]));
    }
    u32x2 gk[NT];                                            // GEGLU: the lane's 4 packed outputs per 16-channel tile
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      const f32x4 x = acc[ni][2 * p], y = acc[ni][2 * p + 1];
      float xs[4] = {x[0], x[1], x[2], x[3]}, ys[4] = {y[0], y[1], y[2], y[3]};
      if constexpr (GEGLU) MOBI_SWAP4("v_permlane32_swap_b32", xs, ys);
      else MOBI_SWAP4("v_permlane16_swap_b32", xs, ys);
      if constexpr (GEGLU) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = xs[r] * gelu_erf_f(ys[r]);
        gk[ni] = pack4<T>(o);
        // 16-byte stores (the epilogue is store-ISSUE bound): tiles ni, ni + 1 trade halves across the 16-lane rows, so
        // that even rows hold outputs 8 ni + 0..7 and odd rows 8 (ni + 1) + 0..7 of their pixel
        if ((ni & 1) && MOBI_PP_GEGLU_STORE16) {
          unsigned a0 = gk[ni - 1][0], a1 = gk[ni - 1][1], b0 = gk[ni][0], b1 = gk[ni][1];
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1"
                       : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1));
          vm_store16(rowp + (nw0 >> 1) + (ni - 1 + (g4 & 1)) * 8, u32x4{a0, a1, b0, b1});
          ++n_issued;
        } else if (ni == NT - 1 || !MOBI_PP_GEGLU_STORE16) {
          vm_store8(rowp + (nw0 >> 1) + ni * 8 + 4 * (g4 & 1), gk[ni]);
          ++n_issued;
        }
      } else {
        float o[8] = {xs[0], xs[1], xs[2], xs[3], ys[0], ys[1], ys[2], ys[3]};
        if (resid) {
          float rf[8];
          unpack8<T>(p == 0 ? q.res0[ni] : res1[ni], rf);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += rf[j];
        }
        vm_store16(rowp + nw0 + ni * 16 + 8 * (g4 >> 1), pack8<T>(o));
        ++n_issued;
      }
    }
  }
  return n_issued;
}

// WM = waves along the pixel axis: 2 -> 128-pixel tile, 4 waves, two blocks per CU;
//                                  4 -> 256-pixel tile, 8 waves, one block per CU (weight tile shared by
//                                       twice the pixels: fewer LDS writes and L2 reads per FLOP).
// FAST: channel counts are multiples of 64 (and the concat boundary too), so one k-tile lies inside one filter tap
//       of one source.  Tap / source / channel bookkeeping is then wave-uniform (SALU), operands are fetched
//       with buffer loads (SGPR descriptor + 32-bit offset, out-of-range offset = hardware zero fill), and the
//       per-lane address work per k-tile drops to one add + one select per 16-byte piece.
template <typename T, int NT, bool TR, int WM, bool FAST>
__global__ __launch_bounds__(128 * WM, 2) void igemm_kernel(const IgemmArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int NTHREADS = 128 * WM;
  constexpr int NWAVES = 2 * WM;
  constexpr int BM = 64 * WM;
  constexpr int RP = NTHREADS / 8;               // tile rows staged per pass of the block
  constexpr int WAVE_N = NT * 16;
  constexpr int BN = 2 * WAVE_N;
  constexpr int WP = (BN + RP - 1) / RP;         // weight pieces per thread and k-tile
  constexpr int X_TILE = BM * 128;               // bytes per buffer
  constexpr int W_TILE = BN * 128;
  constexpr int STAGE_STRIDE = TR ? 36 : (WAVE_N + 4);      // floats
  constexpr int STAGE_ROWS = TR ? WAVE_N : 32;
  constexpr int STAGE_BYTES = STAGE_ROWS * STAGE_STRIDE * 4;
  constexpr int MAIN_BYTES = 2 * X_TILE + 2 * W_TILE;
  constexpr int LDS_BYTES = MAIN_BYTES > NWAVES * STAGE_BYTES ? MAIN_BYTES : NWAVES * STAGE_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int group = blockIdx.z;

  const int nblk = a.tiles_m * a.tiles_n;
  const int L = xcd_remap(blockIdx.x, nblk);
  MOBI_TILE_OF(L, tile_m, tile_n);
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const T* __restrict__ src0 = reinterpret_cast<const T*>(a.src0);
  const T* __restrict__ src1 = reinterpret_cast<const T*>(a.src1);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(a.weight) + (long long)group * a.w_group_stride;

  // ---- staging assignment -------------------------------------------------
  const int seg = tid & 7;               // 16-byte piece of the 128-byte k row
  const int slot = seg >> 2;             // which 32-channel chunk of the k-tile
  const int sub = (seg & 3) * 8;         // channel offset inside the chunk
  const int row_b = tid >> 3;            // 0..RP-1

  // per output row (fixed for the whole k loop): pixel index of the image origin, window origin
  int x_gp[4], x_h[4], x_w[4];
  unsigned x_okm = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + row_b + RP * i;
    const bool ok = m < a.M;
    x_okm |= ok ? (1u << i) : 0u;
    const int mm = ok ? m : 0;
    const int img = mm / a.hw_out;
    const int rem = mm - img * a.hw_out;
    const int ho = rem / a.wout;
    const int wo = rem - ho * a.wout;
    x_gp[i] = (group * a.imgs_per_group + img) * a.img_pix_stride;
    x_h[i] = ho * a.stride - a.pad_h;
    x_w[i] = wo * a.stride - a.pad_w;
  }
  // per weight row: pointer to its k = 0 element, validity
  const T* w_ptr[WP];
  unsigned w_okm = 0;
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int n = n0 + row_b + RP * i;
    const bool ok = (row_b + RP * i < BN) && n < a.n_packed;
    w_okm |= ok ? (1u << i) : 0u;
    w_ptr[i] = wgt + (long long)(ok ? n : 0) * a.ktot;
  }
  // per filter tap (changes every C/64 k-tiles): source pixel of each row and its validity
  int t_pix[4] = {0, 0, 0, 0};
  unsigned t_okm = 0;
  int t_tap = -1;
  const int cpt = a.C >> 5;              // 32-channel chunks per tap
  const int taps = a.kh * a.kw;
  const int hlog = a.hin << a.up, wlog = a.win << a.up;
  // running (tap, chunk-in-tap) of this thread's chunk slot
  const int kt_begin = blockIdx.y * a.nk_per;
  const int kt_end = min(a.nk, kt_begin + a.nk_per);
  int cc, tap, ky, kx;
  {
    const int c0 = 2 * kt_begin + slot;
    tap = c0 / cpt; cc = c0 - tap * cpt;
    ky = tap / a.kw; kx = tap - ky * a.kw;
  }

  // Two register sets: the loads of k-tile t+2 are in flight while tile t is multiplied and tile t+1
  // (loaded one step earlier) is written to LDS.  Every load is issued UNCONDITIONALLY (invalid pieces read
  // the zero block g_zero16) so that the compiler can count them and wait with vmcnt(N) for the older
  // tile only, instead of draining with vmcnt(0).
  u32x4 xr0[4], wr0[WP], xr1[4], wr1[WP];
  unsigned ok0 = 0, ok1 = 0;
  const T* const zsrc = reinterpret_cast<const T*>(g_zero16);

  // ---- FAST path state (all wave-uniform) ----------------------------------------------------------------
  int u_tap = 0, u_ky = 0, u_kx = 0, u_c = 0;           // tap and channel offset (multiple of 64) of the next tile
  unsigned f_row[4] = {0u, 0u, 0u, 0u};                 // byte offset of each row's pixel in the current source/tap
  unsigned f_w[WP];                                     // byte offset of each weight row (out of range if invalid)
  int f_tap = -1, f_src = -1;
  if constexpr (FAST) {
    const long long c_first = (long long)kt_begin * 64;
    if (a.k_order) {                       // channel-chunk-major k: tile = (64-channel chunk, tap), taps innermost
      const int taps_ = a.kh * a.kw;
      const int cc = kt_begin / taps_;
      u_tap = kt_begin - cc * taps_; u_c = cc * 64;
    } else {                               // tap-major k: tile = (tap, 64-channel chunk)
      u_tap = (int)(c_first / a.C); u_c = (int)(c_first - (long long)u_tap * a.C);
    }
    u_ky = u_tap / a.kw; u_kx = u_tap - u_ky * a.kw;
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int n = n0 + row_b + RP * i;
      const bool ok = (row_b + RP * i < BN) && n < a.n_packed;
      f_w[i] = ok ? (unsigned)n * (unsigned)a.ktot * 2u + (unsigned)seg * 16u : 0x80000000u;
    }
  }
  auto load_tile_fast = [&](int kt, u32x4 (&xr)[4], u32x4 (&wr)[WP]) {
    const bool tile_ok = kt < kt_end;                         // uniform
    const int src = u_c >= a.c0 ? 1 : 0;                       // uniform
    if (u_tap != f_tap || src != f_src) {                      // uniform branch, no loads inside
      f_tap = u_tap; f_src = src;
      const unsigned cs2 = (unsigned)(src ? a.c1 : a.c0) * 2u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int hi = x_h[i] + u_ky, wi = x_w[i] + u_kx;
        const bool ok = ((x_okm >> i) & 1u) && hi >= 0 && hi < hlog && wi >= 0 && wi < wlog;
        const unsigned pix = (unsigned)(x_gp[i] + (hi >> a.up) * a.win + (wi >> a.up));
        f_row[i] = ok ? pix * cs2 + (unsigned)seg * 16u : 0x80000000u;
      }
    }
    // descriptors are rebuilt from scalars every tile (a few SALU ops); a tile past the k range gets 0 records,
    // i.e. every piece of it reads as zero
    const void* xbase = src ? a.src1 : a.src0;
    const int xbytes = tile_ok ? (src ? a.src1_bytes : a.src0_bytes) : 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xbase), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wgt), 0, tile_ok ? a.w_bytes : 0, 0x00020000);
    const unsigned chb = (unsigned)(src ? u_c - a.c0 : u_c) * 2u;    // uniform channel byte offset
#pragma unroll
    for (int i = 0; i < 4; ++i) xr[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, f_row[i] + chb, 0, 0);
    const unsigned kb = (unsigned)kt * 128u;
#pragma unroll
    for (int i = 0; i < WP; ++i) wr[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, f_w[i] + kb, 0, 0);
    if (a.k_order) {
      ++u_tap;
      if (++u_kx == a.kw) { u_kx = 0; if (++u_ky == a.kh) { u_ky = 0; u_tap = 0; u_c += 64; } }
    } else {
      u_c += 64;
      if (u_c >= a.C) { u_c = 0; ++u_tap; if (++u_kx == a.kw) { u_kx = 0; ++u_ky; } }
    }
  };

  auto load_tile_generic = [&](int kt, u32x4 (&xr)[4], u32x4 (&wr)[WP], unsigned& okm) {
    if (tap != t_tap) {                       // no loads inside this branch
      t_tap = tap;
      t_okm = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int hi = x_h[i] + ky, wi = x_w[i] + kx;
        const bool ok = hi >= 0 && hi < hlog && wi >= 0 && wi < wlog;
        t_okm |= ok ? (1u << i) : 0u;
        t_pix[i] = (hi >> a.up) * a.win + (wi >> a.up);
      }
      t_okm &= x_okm;
    }
    const bool chunk_ok = (tap < taps) && (kt < kt_end);
    const int ch = cc * 32 + sub;
    const bool second = ch >= a.c0;
    const T* __restrict__ base = second ? src1 : src0;
    const unsigned cs = second ? a.c1 : a.c0;
    const unsigned chs = second ? ch - a.c0 : ch;
    const unsigned xm = chunk_ok ? t_okm : 0u;
    unsigned m = xm;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned long long off = (unsigned long long)(unsigned)(x_gp[i] + t_pix[i]) * cs + chs;
      const T* ptr = ((xm >> i) & 1u) ? base + off : zsrc;
      xr[i] = ld16(ptr);
    }
    const int kk = kt * 64 + seg * 8;
    const unsigned wmask = ((kk < a.ktot) && (kt < kt_end)) ? w_okm : 0u;
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const T* ptr = ((wmask >> i) & 1u) ? w_ptr[i] + kk : zsrc;
      wr[i] = ld16(ptr);
    }
    okm = m | (wmask << 4);
    // advance this thread's chunk by one k-tile (two chunks)
    cc += 2;
    while (cc >= cpt) { cc -= cpt; ++tap; if (++kx == a.kw) { kx = 0; ++ky; } }
  };
  auto load_tile = [&](int kt, u32x4 (&xr)[4], u32x4 (&wr)[WP], unsigned& okm) {
    if constexpr (FAST) load_tile_fast(kt, xr, wr);
    else load_tile_generic(kt, xr, wr, okm);
  };
  auto store_tile = [&](int buf, const u32x4 (&xr)[4], const u32x4 (&wr)[WP], unsigned okm) {
    unsigned char* xb = lds + buf * X_TILE;
    unsigned char* wb = lds + 2 * X_TILE + buf * W_TILE;
    (void)okm;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = row_b + RP * i;
      st16(xb + r * 128 + ((seg ^ (r & 7)) << 4), xr[i]);
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      const int r = row_b + RP * i;
      if (r < BN) st16(wb + r * 128 + ((seg ^ (r & 7)) << 4), wr[i]);
    }
  };

  f32x4 acc[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, g4 = lane >> 4;

  auto compute_tile = [&](int buf) {
    const unsigned char* xb = lds + buf * X_TILE + (wm * 64 + r16) * 128;
    const unsigned char* wb = lds + 2 * X_TILE + buf * W_TILE + (wn * WAVE_N + r16) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int sw = ((ks * 4 + g4) ^ (r16 & 7)) << 4;
      frag_t xf[4], wf[NT];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) xf[mi] = __builtin_bit_cast(frag_t, ld16(xb + mi * 16 * 128 + sw));
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) wf[ni] = __builtin_bit_cast(frag_t, ld16(wb + ni * 16 * 128 + sw));
#if MOBI_STAGED_PRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = TR ? mfma16(xf[mi], wf[ni], acc[ni][mi]) : mfma16(wf[ni], xf[mi], acc[ni][mi]);
#if MOBI_STAGED_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  };

  // prologue: tile kt_begin -> LDS[0], tile kt_begin+1 -> set 1 (loads past kt_end are harmless no-ops)
  load_tile(kt_begin, xr0, wr0, ok0);
  store_tile(0, xr0, wr0, ok0);
  load_tile(kt_begin + 1, xr1, wr1, ok1);
  __syncthreads();
#pragma clang loop unroll(disable)
  for (int kt = kt_begin; kt < kt_end; kt += 2) {
    // even step: tile kt in LDS[0]; tile kt+1 waits in set 1; tile kt+2 -> set 0
    load_tile(kt + 2, xr0, wr0, ok0);
    MOBI_SCHED_FENCE();
    compute_tile(0);
    MOBI_SCHED_FENCE();
    store_tile(1, xr1, wr1, ok1);
    __syncthreads();
    MOBI_SCHED_FENCE();
    // odd step: tile kt+1 in LDS[1]; tile kt+2 waits in set 0; tile kt+3 -> set 1
    load_tile(kt + 3, xr1, wr1, ok1);
    MOBI_SCHED_FENCE();
    if (kt + 1 < kt_end) compute_tile(1);
    MOBI_SCHED_FENCE();
    store_tile(0, xr0, wr0, ok0);
    __syncthreads();
    MOBI_SCHED_FENCE();
  }

  // ---- epilogue -----------------------------------------------------------
  float* stage = reinterpret_cast<float*>(lds) + wave * (STAGE_BYTES / 4);
  igemm_epilogue<T, NT, TR, 32>(a, stage, acc, lane, group, n0 + wn * WAVE_N, m0 + wm * 64);
}

#ifdef MOBI_DEV   // development build only (A/B partner of the ping-pong kernel, MOBI_IGEMM_PP=0): the shipped library routes
                  // every launch it used to take (LDS-staged epilogues of the 256-pixel geometry) to igemm_ring_kernel
// =========================================================================================================
// Direct-to-LDS main loop (FAST shapes only):  256 pixels x (2 * WAVE_N) channels, 8 waves (4 x 2), THREE LDS
// stages filled by global_load_lds_dwordx4 (no VGPR staging, no ds_write), one raw s_barrier per k-tile and a
// COUNTED s_waitcnt vmcnt: the tile being multiplied is complete while the next tile's DMA stays in flight.
//   step i:  wait(k-tile i landed) ; barrier ; issue DMA of k-tile i+2 into stage (i+2)%3 ; multiply stage i%3
// The barrier also proves every wave finished reading stage (i-1)%3 == (i+2)%3 before it is overwritten.
//
// PERSISTENT: one block per CU walks output tiles blockIdx.x, +gridDim.x, ...; the k-tile sequence runs on
// ACROSS output tiles, so the first two k-tiles of the next output tile are already in flight while this tile's
// epilogue runs (measured per output tile before: 3.1 us entry-to-first-tile + 1.2 us block turnaround, every
// tile; now once per block).  The epilogue stages through the LDS stage of the tile's LAST k-tile, the only one
// that is free at that point (the other two are DMA targets).
// LDS image = [row][8 x 16 B], lane-linear per wave instruction (8 rows = 1 KiB); the XOR swizzle is applied on
// the SOURCE side: the lane that fills slot s of row r fetches piece s ^ (r & 7).  Padded / out-of-range pieces
// are fetched from the 16-byte zero block.
// =========================================================================================================
// MODE 0: LDS-staged epilogue, row-major output    1: LDS-staged epilogue, transposed output
//      2: register epilogue (plain)                  3: register epilogue (GEGLU)
template <typename T, int NT, int MODE>
__global__ __launch_bounds__(512, 2) void igemm_glds_kernel(const IgemmArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr bool TR = MODE == 1;
  constexpr bool DIRECT = MODE >= 2;
  constexpr int BM = 256;
  constexpr int WAVE_N = NT * 16;
  constexpr int BN = 2 * WAVE_N;
  constexpr int X_TILE = BM * 128, W_TILE = BN * 128;
  constexpr int STAGE = X_TILE + W_TILE;
  constexpr int WJ = (BN + 63) / 64;                         // weight DMA instructions per thread and tile
  constexpr int PIECES = 4 + WJ;                             // DMA instructions per thread and tile
  constexpr int RPP = 16;                                    // epilogue rows per pass: 8 wave stages fit one k-stage
  constexpr int EPI_WAVE = EpiGeom<RPP, TR, NT>::BYTES;
  static_assert(8 * EPI_WAVE <= STAGE, "the epilogue must fit the one free k-stage");
  static_assert(PIECES == 6 || PIECES == 7, "vmcnt immediates below");
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const int group = blockIdx.z;
  const int nblk = a.tiles_m * a.tiles_n;
  MOBI_STAMP_AT(0);
  const T* __restrict__ wgt = reinterpret_cast<const T*>(a.weight) + (long long)group * a.w_group_stride;
  const unsigned char* const zsrc = reinterpret_cast<const unsigned char*>(g_zero16);

  const int rloc = lane >> 3;                                // row inside the wave's 8-row DMA piece
  const int sg = (lane & 7) ^ rloc;                          // source piece that lands in this lane's slot
  const int kt_begin = blockIdx.y * a.nk_per;
  const int kt_end = min(a.nk, kt_begin + a.nk_per);
  const int hlog = a.hin << a.up, wlog = a.win << a.up;

  // ---- fetch side: state of the output tile whose k-tiles are being requested --------------------------------
  int f_bid = blockIdx.x;                                    // output tile being fetched (>= nblk: nothing left)
  int f_kt = kt_begin;                                       // its next k-tile
  int f_slot = 0;                                            // LDS stage of the next request (0, 1, 2, 0, ...)
  int x_gp[4], x_h[4], x_w[4];                               // activation rows of this thread: 8 * wave + rloc + 64 j
  unsigned x_okm = 0;
  const unsigned char* w_src[WJ];
  int w_lds[WJ];
  int u_tap = 0, u_ky = 0, u_kx = 0, u_c = 0;                // wave-uniform (tap, channel offset) of the next k-tile
  unsigned f_row[4] = {0u, 0u, 0u, 0u};
  unsigned f_okm = 0;
  int f_tap = -1, f_src = -1;
  // linear window (no upsampling, <= 16 taps): pixel(tap) = x_pix0[j] + ky * win + kx, so a tap change costs one
  // wave-uniform offset and a mask shift instead of re-deriving four windows (what made the taps-innermost k order
  // slower than it had to be)
  const bool lin = a.lin_window != 0;
  int x_pix0[4] = {0, 0, 0, 0};                              // x_gp + x_h * win + x_w (may be "negative": only used when valid)
  unsigned long long x_tapok = 0;                            // bit 4 * tap + j: row j's window pixel of that tap is inside
  unsigned f_rowbase[4] = {0u, 0u, 0u, 0u};
  unsigned f_tapoff = 0;

  auto set_fetch_tile = [&]() {
    const int L = xcd_remap(f_bid, nblk);
    MOBI_TILE_OF(L, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    x_okm = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + 8 * wave + rloc + 64 * j;
      const bool ok = m < a.M;
      x_okm |= ok ? (1u << j) : 0u;
      const int mm = ok ? m : 0;
      const int img = mm / a.hw_out, rem = mm - img * a.hw_out;
      const int ho = rem / a.wout, wo = rem - ho * a.wout;
      x_gp[j] = (group * a.imgs_per_group + img) * a.img_pix_stride;
      x_h[j] = ho * a.stride - a.pad_h;
      x_w[j] = wo * a.stride - a.pad_w;
    }
    if (lin) {
      x_tapok = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x_pix0[j] = x_gp[j] + x_h[j] * a.win + x_w[j];
        if ((x_okm >> j) & 1u) {
          for (int ky = 0; ky < a.kh; ++ky)
            for (int kx = 0; kx < a.kw; ++kx) {
              const int hi = x_h[j] + ky, wi = x_w[j] + kx;
              if (hi >= 0 && hi < hlog && wi >= 0 && wi < wlog) x_tapok |= 1ull << ((ky * a.kw + kx) * 4 + j);
            }
        }
      }
    }
    // weight rows: j < WJ-1 (or all, when BN % 64 == 0): 8 * wave + rloc + 64 j; the last partial group of 32 rows
    // is fetched by waves 0-3 and (identically, benign duplicate) by waves 4-7 so that every wave issues PIECES DMAs
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const bool partial = (BN % 64 != 0) && (j == WJ - 1);
      const int r = partial ? 64 * j + 8 * (wave & 3) + rloc : 64 * j + 8 * wave + rloc;
      const int n = n0 + r;
      w_lds[j] = (partial ? 64 * j + 8 * (wave & 3) : 64 * j + 8 * wave) * 128;
      w_src[j] = n < a.n_packed ? reinterpret_cast<const unsigned char*>(wgt + (long long)n * a.ktot) + sg * 16 : nullptr;
    }
    const long long c_first = (long long)kt_begin * 64;
    if (a.k_order) {                       // channel-chunk-major k: tile = (64-channel chunk, tap), taps innermost
      const int taps_ = a.kh * a.kw;
      const int cc = kt_begin / taps_;
      u_tap = kt_begin - cc * taps_; u_c = cc * 64;
    } else {                               // tap-major k: tile = (tap, 64-channel chunk)
      u_tap = (int)(c_first / a.C); u_c = (int)(c_first - (long long)u_tap * a.C);
    }
    u_ky = u_tap / a.kw; u_kx = u_tap - u_ky * a.kw;
    f_tap = -1; f_src = -1;
  };

  // request the next k-tile of the sequence (caller checked f_bid < nblk) and advance the sequence
  // every vector-memory instruction this wave issues is counted (wave-uniform); mk0 / mk1 / mk2 = the count right
  // after the requests of the oldest / next / youngest k-tile that has not been multiplied yet
  int vm_issued = 0, mk0 = 0, mk1 = 0, mk2 = 0;
  auto issue_next = [&]() {
    unsigned char* st = lds + f_slot * STAGE;
    f_slot = f_slot == 2 ? 0 : f_slot + 1;
    const int src = u_c >= a.c0 ? 1 : 0;
    if (lin) {
      if (src != f_src) {
        f_src = src;
        const unsigned cs2 = (unsigned)(src ? a.c1 : a.c0) * 2u;
#pragma unroll
        for (int j = 0; j < 4; ++j) f_rowbase[j] = (unsigned)x_pix0[j] * cs2 + (unsigned)sg * 16u;
        f_tap = -1;
      }
      if (u_tap != f_tap) {
        f_tap = u_tap;
        f_tapoff = (unsigned)(u_ky * a.win + u_kx) * ((unsigned)(src ? a.c1 : a.c0) * 2u);     // wave-uniform
        f_okm = (unsigned)(x_tapok >> (u_tap * 4)) & 15u;
#pragma unroll
        for (int j = 0; j < 4; ++j) f_row[j] = f_rowbase[j] + f_tapoff;
      }
    } else if (u_tap != f_tap || src != f_src) {
      f_tap = u_tap; f_src = src; f_okm = 0;
      const unsigned cs2 = (unsigned)(src ? a.c1 : a.c0) * 2u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hi = x_h[j] + u_ky, wi = x_w[j] + u_kx;
        const bool ok = ((x_okm >> j) & 1u) && hi >= 0 && hi < hlog && wi >= 0 && wi < wlog;
        f_okm |= ok ? (1u << j) : 0u;
        f_row[j] = (unsigned)(x_gp[j] + (hi >> a.up) * a.win + (wi >> a.up)) * cs2 + (unsigned)sg * 16u;
      }
    }
    const unsigned char* xbase = reinterpret_cast<const unsigned char*>(src ? a.src1 : a.src0) +
                                 (unsigned)(src ? u_c - a.c0 : u_c) * 2u;
#if !(MOBI_DBG_SKIP & 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned char* g = ((f_okm >> j) & 1u) ? xbase + f_row[j] : zsrc;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(st + (8 * wave + 64 * j) * 128), 16, 0, 0);
    }
    vm_issued += 4;
#endif
    const unsigned kb = (unsigned)f_kt * 128u;
#if !(MOBI_DBG_SKIP & 2)
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const unsigned char* g = w_src[j] ? w_src[j] + kb : zsrc;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)(st + X_TILE + w_lds[j]), 16, 0, 0);
    }
    vm_issued += WJ;
#endif
    if (++f_kt == kt_end) {                                  // the sequence moves on to this block's next output tile
      f_kt = kt_begin;
      f_bid += gridDim.x;
      if (f_bid < nblk) set_fetch_tile();
    } else if (a.k_order) {
      ++u_tap;
      if (++u_kx == a.kw) { u_kx = 0; if (++u_ky == a.kh) { u_ky = 0; u_tap = 0; u_c += 64; } }
    } else {
      u_c += 64;
      if (u_c >= a.C) { u_c = 0; ++u_tap; if (++u_kx == a.kw) { u_kx = 0; ++u_ky; } }
    }
  };

  const int r16 = lane & 15, g4 = lane >> 4;
  f32x4 acc[NT][4];

  if (kt_begin >= kt_end) return;                            // never launched: the host trims empty split-K ranges

  set_fetch_tile();
  int ahead = 0;                                             // requested k-tiles not yet multiplied
  int c_slot = 0;                                            // LDS stage of the k-tile to multiply next
  issue_next(); ++ahead; mk0 = vm_issued;
  if (f_bid < nblk) { issue_next(); ++ahead; mk1 = vm_issued; }
  DirectEpiRegs<DIRECT ? NT : 1> dq;
  MOBI_PHASE_DECL;

  for (int bid = blockIdx.x; bid < nblk; bid += gridDim.x) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int L = xcd_remap(bid, nblk);
    MOBI_TILE_OF(L, tm_, tn_);
    const int nw0 = tn_ * BN + wn * WAVE_N, mw0 = tm_ * BM + wm * 64;
    int mk_req = 0;
    if constexpr (DIRECT) {                                  // bias / residual rows of this output tile: in flight
      vm_issued += direct_epilogue_request<T, NT>(a, dq, lane, group, nw0, mw0);      // for the whole k loop
      mk_req = vm_issued;
    }

    for (int kt = kt_begin; kt < kt_end; ++kt) {
      // the oldest requested k-tile has landed once only operations issued AFTER its requests are outstanding
      // (vector-memory operations retire in issue order).  LDS-staged epilogues issue loads / stores that are not
      // counted: under-counting only makes the wait stricter.
      MOBI_PHASE(0);
      wait_vmcnt_le(vm_issued - mk0);
      MOBI_PHASE(1);
      __builtin_amdgcn_s_barrier();
      MOBI_PHASE(2);
#if MOBI_STAMP
      if (kt == kt_begin && bid == (int)blockIdx.x) MOBI_STAMP_AT(1);
#endif
      const unsigned char* st = lds + c_slot * STAGE;
      const unsigned char* xb = st + (wm * 64 + r16) * 128;
      const unsigned char* wb = st + X_TILE + (wn * WAVE_N + r16) * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // the DMA of the k-tile two steps ahead is issued between the two k-steps: its (expensive) issue slots
        // then sit in the shadow of the first k-step's MFMAs instead of delaying the first fragment reads
        if (ks == MOBI_DMA_KS) MOBI_PHASE(3);
        if (ks == MOBI_DMA_KS && f_bid < nblk) {
          issue_next(); ++ahead;
          if (ahead == 2) mk1 = vm_issued; else mk2 = vm_issued;
        }
        if (ks == MOBI_DMA_KS) MOBI_PHASE(4);
        const int sw = ((ks * 4 + g4) ^ (r16 & 7)) << 4;
        frag_t xf[4], wf[NT];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) xf[mi] = __builtin_bit_cast(frag_t, ld16(xb + mi * 16 * 128 + sw));
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) wf[ni] = __builtin_bit_cast(frag_t, ld16(wb + ni * 16 * 128 + sw));
#if MOBI_DBG_SKIP & 4
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) asm volatile("" :: "v"(xf[mi]));
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) asm volatile("" :: "v"(wf[ni]));
#else
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi)
            acc[ni][mi] = TR ? mfma16(xf[mi], wf[ni], acc[ni][mi]) : mfma16(wf[ni], xf[mi], acc[ni][mi]);
#endif
      }
      MOBI_PHASE(5);
      MOBI_PHASE_ACC();
      --ahead; mk0 = mk1; mk1 = mk2;
      if (kt + 1 < kt_end) c_slot = c_slot == 2 ? 0 : c_slot + 1;
    }
    if constexpr (DIRECT) {
      // registers only: no barrier, the waves drift by at most one epilogue until the next k-tile's barrier
      MOBI_STAMP_AT(2);
      wait_vmcnt_le(vm_issued - mk_req);                     // bias / residual registers are complete
      vm_issued += direct_epilogue<T, NT, MODE == 3>(a, acc, dq, lane, group, nw0, mw0);
    } else {
      // every wave is past its last fragment read of stage c_slot: it is free until the k-tile three steps on is
      // requested, which happens after the next barrier, i.e. after every wave has left this epilogue
      __syncthreads();
      MOBI_STAMP_AT(2);
      float* stage = reinterpret_cast<float*>(lds + c_slot * STAGE) + wave * (EPI_WAVE / 4);
      igemm_epilogue<T, NT, TR, RPP>(a, stage, acc, lane, group, nw0, mw0);
    }
    c_slot = c_slot == 2 ? 0 : c_slot + 1;
  }
#if MOBI_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the stamp then covers this wave's stores too
  MOBI_STAMP_AT(3);
  if (g_stamps && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* d = g_stamps + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8;
    d[4] = ((unsigned long long)xcc << 32) | hw;
    d[5] = (unsigned long long)(kt_end - kt_begin);
    d[6] = (unsigned long long)((nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
  }
#if MOBI_STAMP == 2
  if (g_phase && lane == 0) {
    unsigned long long* d = g_phase + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wave) * 8;
    for (int i = 0; i < 5; ++i) d[i] = ph_acc[i];
    d[5] = ph_n;
  }
#endif
#endif
}

#endif  // MOBI_DEV

// =========================================================================================================
// PING-PONG main loop (register-epilogue launches of the direct-to-LDS geometry: full 256 x (2 * WAVE_N) tiles,
// row-major output, no split-K, window pixels linear in the tap).
//
// Why: with all eight waves in lockstep the three parts of a k-tile step ADD UP instead of overlapping (3x3,
// 320 -> 320, 64x64 x 16, 173 us: LDS fragment reads + barriers alone 74 us, + DMA requests 122 us, MFMAs alone
// 48 us at peak; tools/diag_ingest.sh).  The two waves of every SIMD (w and w + 4) run the same program, so both
// wait for their fragments, then both queue on the one matrix pipe, then both stall in the DMA issue.
//
// Here every k-step is split into a LOAD phase (fragment reads LDS -> registers, DMA requests of the k-tile two
// steps ahead, the previous output tile's epilogue stores) and a MATRIX phase (MFMAs only).  Every phase boundary
// is one raw s_barrier of all eight waves, and waves 4-7 run ONE PHASE BEHIND waves 0-3: while one wave of a SIMD
// multiplies, its partner reads / requests / stores.  The epilogue of an output tile is deferred into the first
// LOAD phase of the next tile (registers only, so it needs no LDS), where it overlaps the partner's MATRIX phase.
//
//   phase g = 4 i + { 0: early LOAD(i, ks 0)  | late MATRIX(i-1, ks 1)
//                     1: early MATRIX(i, ks 0) | late LOAD(i, ks 0)
//                     2: early LOAD(i, ks 1)   | late MATRIX(i, ks 0)
//                     3: early MATRIX(i, ks 1) | late LOAD(i, ks 1) }
// LDS hazards (three stages, k-tile i in stage i % 3):
//   RAW  k-tile i+1 is first read in phase 4(i+1) (early waves): every wave waits for ITS pieces of k-tile i+1 (a
//        counted vmcnt; vector-memory operations retire in issue order) before the barrier that ends phase 4i+3.
//   WAR  the DMA of k-tile i+2 overwrites stage (i-1) % 3, last read by the late waves in phase 4(i-1)+3; it is
//        requested in LOAD(i, ks MOBI_PP_DMA_KS), i.e. in phase >= 4i, and every LOAD phase ends with lgkmcnt(0).
// Operands arrive by buffer_load ... lds (SGPR descriptor + 32-bit lane offset): padded pieces use an offset
// beyond the descriptor's range and the hardware writes zeros.
// =========================================================================================================
// SLAB: split-K launch -- blockIdx.y owns a k range and leaves fp32 partial sums (no bias / vector / residual)
template <typename T, int NT, bool GEGLU, bool SLAB = false>
__global__ __launch_bounds__(512, 2) void igemm_pp_kernel(const IgemmArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int BM = 256;
  constexpr int WAVE_N = NT * 16;
  constexpr int BN = 2 * WAVE_N;
  constexpr int X_TILE = BM * 128, W_TILE = BN * 128;
  constexpr int STAGE = X_TILE + W_TILE;
  constexpr int WJ = (BN + 63) / 64;                         // weight DMA instructions per thread and tile
  constexpr int PIECES = 4 + WJ;
  constexpr unsigned OOB = 0x80000000u;                      // beyond every descriptor (extents < 2^31, host check)
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE];
  __shared__ int s_arrive;                                   // split-K finished inside the launch: the arrival order
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);   // provably wave-uniform copy: LDS-DMA bases and the
  const int wm = wave & 3, wn = wave >> 2;                   // early / late branches stay scalar
  const bool late = wave_s >= 4;                             // the half that runs one phase behind
  const int group = blockIdx.z;
  const int nblk = a.tiles_m * a.tiles_n;
  MOBI_STAMP_AT(0);
  // split-K: blockIdx.y owns k-tiles [kt_begin, kt_begin + nk) and leaves fp32 partial sums for the reduce launch
  const int kt_begin = SLAB ? blockIdx.y * a.nk_per : 0;
  const int nk = SLAB ? min(a.nk, kt_begin + a.nk_per) - kt_begin : a.nk;
  constexpr bool slab = SLAB;
  const T* wgt = reinterpret_cast<const T*>(a.weight) + (long long)group * a.w_group_stride;
  const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src0), 0, a.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src1 ? a.src1 : a.src0), 0, a.src1 ? a.src1_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wgt), 0, a.w_bytes, 0x00020000);

  const int rloc = lane >> 3;                                // row inside the wave's 8-row DMA piece
  const unsigned sg16 = (unsigned)((lane & 7) ^ rloc) * 16u; // source piece that lands in this lane's slot (swizzle)

  // ---- fetch side ----------------------------------------------------------------------------------------------
  int f_bid = blockIdx.x, f_kt = kt_begin, f_slot = 0;
  int x_hw0 = 0;                                             // window origin of this lane's first row (8 * wave + rloc):
                                                             // (row << 16) | (column & 0xffff), both may be < 0
                                                             // rows 64 j further on: a wave-uniform (row, column) step,
                                                             // see issue_x
  int x_img[4];                                              // first pixel of the four rows' images (wave-uniform: hw_out % 64 == 0)
  unsigned f_row[4] = {OOB, OOB, OOB, OOB};                  // byte offsets of the current (tap, source)
  unsigned w_off = 0;                                        // this lane's piece of weight row n0 + 8 * wave + rloc
  int w_rowstep = 0;                                         // bytes between weight rows 64 apart (wave-uniform)
  int w_half = 0;                                            // waves 4-7: back to rows 128.. of the partial group
  int u_tap = 0, u_ky = 0, u_kx = 0, u_c = 0, f_tap = -1, f_src = -1;

  auto set_fetch_tile = [&]() {
    const int L = xcd_remap(f_bid, nblk);
    MOBI_TILE_OF_M(L, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    {
      const int m = m0 + 8 * wave + rloc;                    // < M: every tile is full
      const int rem = m & (a.hw_out - 1);                    // hw_out, wout: powers of two (host check)
      const int ho = rem >> a.w_shift, wo = rem & (a.wout - 1);
      const int h0 = ho * a.stride - a.pad_h, w0 = wo * a.stride - a.pad_w;
      x_hw0 = (int)(((unsigned)h0 << 16) | ((unsigned)w0 & 0xffffu));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        x_img[j] = __builtin_amdgcn_readfirstlane((group * a.imgs_per_group + ((m + 64 * j) >> a.hw_shift)) * a.img_pix_stride);
    }
    w_off = (unsigned)(n0 + 8 * wave + rloc) * (unsigned)a.ktot * 2u + sg16;
    w_rowstep = a.ktot * 128;
    w_half = (wave_s >> 2) * (a.ktot * 64);
    if constexpr (SLAB) {
      if (a.k_order) {                     // channel-chunk-major k: taps innermost
        const int taps_ = a.kh * a.kw;
        const int cc = kt_begin / taps_;
        u_tap = kt_begin - cc * taps_; u_c = cc * 64;
      } else {                             // tap-major k
        const int c_first = kt_begin * 64;
        u_tap = c_first / a.C; u_c = c_first - u_tap * a.C;
      }
      u_ky = u_tap / a.kw; u_kx = u_tap - u_ky * a.kw;
    } else {
      u_tap = 0; u_ky = 0; u_kx = 0; u_c = 0;
    }
    f_tap = -1; f_src = -1;
  };

  // every vector-memory instruction this wave issues is counted (wave-uniform); mk1 / mk2 = the count right after
  // the requests of the next / next-but-one k-tile
  int vm_issued = 0, mk1 = 0, mk2 = 0;
  // A k-tile is requested in two parts so that the LDS-DMA path (measured: one 1-KiB request per ~32 cycles and CU,
  // the issuing wave stalls behind it) is fed in BOTH LOAD phases of a k-tile instead of in one burst:
  //   part X (4 activation pieces) in LOAD(ks 0), part W (WJ weight pieces) in LOAD(ks 1); the k-tile's mark is taken
  //   after part W.
  auto issue_x = [&]() {
    unsigned char* st = lds + f_slot * STAGE;
    const int src = u_c >= a.c0 ? 1 : 0;
    if (u_tap != f_tap || src != f_src) {
      f_tap = u_tap; f_src = src;
      const unsigned cs2 = (unsigned)(src ? a.c1 : a.c0) * 2u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // pixel 64 j after the lane's first one, inside its image: 64 j = dr * wout + dc with no carry into the
        // lane's own column (tiles start at multiples of 256, the lane's offset inside its 64-block is < 64)
        const int d = (64 * j) & (a.hw_out - 1);
        const int hi = (x_hw0 >> 16) + (d >> a.w_shift) * a.stride + u_ky;
        const int wi = (int)(short)x_hw0 + (d & (a.wout - 1)) * a.stride + u_kx;
        // (nearest x2 upsampling on the load side: window coordinates live on the upsampled grid, a.up = 1)
        const bool ok = (unsigned)hi < (unsigned)(a.hin << a.up) && (unsigned)wi < (unsigned)(a.win << a.up);
        f_row[j] = ok ? (unsigned)(x_img[j] + (hi >> a.up) * a.win + (wi >> a.up)) * cs2 + sg16 : OOB;
      }
    }
    const int soff = (src ? u_c - a.c0 : u_c) * 2;
#if !(MOBI_DBG_SKIP & 1)
    if (src) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (lds_ptr_t)(st + (8 * wave_s + 64 * j) * 128), 16, f_row[j], soff, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_ptr_t)(st + (8 * wave_s + 64 * j) * 128), 16, f_row[j], soff, 0, 0);
    }
    vm_issued += 4;
#endif
  };
  auto issue_w = [&]() {
    unsigned char* st = lds + f_slot * STAGE;
    f_slot = f_slot == 2 ? 0 : f_slot + 1;
#if !(MOBI_DBG_SKIP & 2)
    const int kb = f_kt * 128;
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      // rows 8 * wave + rloc + 64 j; the last partial group of 32 rows is fetched by waves 0-3 on even k-tiles and by
      // waves 4-7 on odd ones (MOBI_PP_WDUP = 1: by both, a benign duplicate, so that every wave issues PIECES requests)
      const bool partial = (BN % 64 != 0) && (j == WJ - 1);
      const int wl = (partial ? 64 * j + 8 * (wave_s & 3) : 64 * j + 8 * wave_s) * 128;
      const int so = kb + j * w_rowstep - (partial ? w_half : 0);
      if (MOBI_PP_WDUP || !partial || (((f_kt ^ (wave_s >> 2)) & 1) == 0)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(st + X_TILE + wl), 16, w_off, so, 0, 0);
        ++vm_issued;
      }
    }
#endif
    if (++f_kt == kt_begin + nk) {                           // the sequence moves on to this block's next output tile
      f_kt = kt_begin;
      f_bid += gridDim.x;
      if (f_bid < nblk) set_fetch_tile();
    } else if (a.k_order) {                                  // channel-chunk-major k: taps innermost
      ++u_tap;
      if (++u_kx == a.kw) { u_kx = 0; if (++u_ky == a.kh) { u_ky = 0; u_tap = 0; u_c += 64; } }
    } else {                                                 // tap-major k
      u_c += 64;
      if (u_c >= a.C) { u_c = 0; ++u_tap; if (++u_kx == a.kw) { u_kx = 0; ++u_ky; } }
    }
  };
  auto issue_next = [&]() { issue_x(); issue_w(); };

#define MOBI_PP_BARRIER()                     \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

  const int r16 = lane & 15, g4 = lane >> 4;
  f32x4 acc[NT][4];
  PpEpiRegs<NT> dq;                                      // residual rows only: bias / per-image vector start the sums

  // bias OR per-image vector (never both) of the wave tile at (mw0, nw0): f32 x 4 per 16-column MFMA tile, the
  // accumulator layout.  Requested BEFORE anything else of the phase so that the latency hides behind the
  // epilogue stores; lands in registers that are free at that point (no fragment is live at a tile boundary).
#ifdef MOBI_DBG_NOVEC                                      // timing only (wrong results): no bias / vector request and wait
  const bool has_vec = false;
#else
  const bool has_vec = !slab && (a.bias || a.rowvec);      // (split-K: the reduce launch adds them)
#endif
  auto request_vec = [&](u32x4 (&bv)[NT], int nw0, int mw0) {
    const float* vec = a.bias;
    if (a.rowvec) {                                          // the wave's 64 pixels lie in one image: hw_out % 64 == 0
      const int img = __builtin_amdgcn_readfirstlane(mw0 / a.hw_out);
      vec = a.rowvec + (long long)(group * a.imgs_per_group + img) * a.rowvec_stride;
    }
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) bv[ni] = vm_load16(vec + nw0 + ni * 16 + g4 * 4);
  };
  auto start_sums = [&](u32x4 (&bv)[NT]) {
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (has_vec) { asm volatile("" : "+v"(bv[ni])); v = __builtin_bit_cast(f32x4, bv[ni]); }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = v;
    }
  };

  // split-K partial sums straight from the accumulators: a lane holds 4 consecutive channels of pixel r16 per tile
  auto slab_epilogue = [&](int nw0, int mw0) {
    float* wsp = a.split_ws + (long long)blockIdx.y * a.M * a.n_packed;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      float* rowp = wsp + (long long)(mw0 + mi * 16 + r16) * a.n_packed + nw0 + g4 * 4;
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) slab_store16(a.sync_mode == 1, rowp + ni * 16, __builtin_bit_cast(u32x4, acc[ni][mi]));
    }
    return 4 * NT;
  };
  auto finish_tile = [&](int nw0, int mw0) {
    if constexpr (SLAB) return slab_epilogue(nw0, mw0);
    else return pp_epilogue<T, NT, GEGLU>(a, acc, dq, lane, group, nw0, mw0);
  };
  auto request_tile = [&](int nw0, int mw0) {
    if constexpr (SLAB) return 0;
    else return pp_epilogue_request<T, NT, GEGLU>(a, dq, lane, group, nw0, mw0);
  };

  int ahead = 0;                                             // requested k-tiles not yet multiplied
  {
    const int L0 = xcd_remap(blockIdx.x, nblk);
    MOBI_TILE_OF_M(L0, tm_, tn_);
    const int nw0 = tn_ * BN + wn * WAVE_N, mw0 = tm_ * BM + wm * 64;
    u32x4 bv[NT];
    if (has_vec) { request_vec(bv, nw0, mw0); vm_issued += NT; }
    set_fetch_tile();
    issue_next(); ++ahead;
    const int mk_first = vm_issued;
    if (f_bid < nblk) { issue_next(); ++ahead; mk1 = vm_issued; }
    wait_vmcnt_le(vm_issued - mk_first);                     // this wave's pieces of the first k-tile (and, older,
    MOBI_STAMP_AT(1);
    start_sums(bv);                                          // the vector) have landed
    vm_issued += request_tile(nw0, mw0);
  }
  int mk_req = vm_issued;
  int c_slot = 0;                                            // LDS stage of the k-tile to multiply next
  if (late) MOBI_PP_BARRIER();

  int p_nw0 = 0, p_mw0 = 0;
  bool f_more = false;
#if MOBI_STAMP == 3
  // per wave, in shader cycles: [ks][wait at the LOAD barrier, LOAD, wait at the MATRIX barrier, MATRIX issue]
  unsigned pp_acc[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, pp_n = 0;
  unsigned long long pp_t[6];
#define MOBI_PP_T(i) pp_t[i] = __builtin_amdgcn_s_memtime()
#else
#define MOBI_PP_T(i) ((void)0)
#endif
  for (int bid = blockIdx.x; bid < nblk; bid += gridDim.x) {
    const int L = xcd_remap(bid, nblk);
    MOBI_TILE_OF_M(L, tm_, tn_);
    const int nw0 = tn_ * BN + wn * WAVE_N, mw0 = tm_ * BM + wm * 64;
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* st = lds + c_slot * STAGE;
      const unsigned char* xb = st + (wm * 64 + r16) * 128;
      const unsigned char* wb = st + X_TILE + (wn * WAVE_N + r16) * 128;
#if MOBI_PP_PHASES == 2
      // ---- TWO phases per k-tile: LOAD = all 2 x (4 + NT) fragment reads + the requests, MATRIX = 8 NT MFMAs ----------
      {
        MOBI_PP_T(0);
        MOBI_PP_BARRIER();
        MOBI_PP_T(1);
        if (kt == 0 && bid != (int)blockIdx.x) {             // tile boundary: previous tile's registers -> memory
          // (the next tile's bias / per-image vector is requested AFTER the previous tile's epilogue: requested before it,
          //  its NT x 4 registers are live across the epilogue and the NT = 5 instance spills 8 VGPRs to scratch)
          wait_vmcnt_le(vm_issued - mk_req);                 // residual rows of the previous tile
          vm_issued += finish_tile(p_nw0, p_mw0);
          u32x4 bv[NT];
          if (has_vec) { request_vec(bv, nw0, mw0); vm_issued += NT; }
          const int mk_vec = vm_issued;
          wait_vmcnt_le(vm_issued - mk_vec);
          start_sums(bv);
          vm_issued += request_tile(nw0, mw0);
          mk_req = vm_issued;
        }
        frag_t xf[2][4], wf[2][NT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const int sw = ((ks * 4 + g4) ^ (r16 & 7)) << 4;
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) xf[ks][mi] = __builtin_bit_cast(frag_t, ld16(xb + mi * 16 * 128 + sw));
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) wf[ks][ni] = __builtin_bit_cast(frag_t, ld16(wb + ni * 16 * 128 + sw));
        }
        f_more = f_bid < nblk;
#if MOBI_PP_DMA_AT == 0
        if (f_more) {
          issue_next(); ++ahead;
          if (ahead == 2) mk1 = vm_issued; else mk2 = vm_issued;
        }
#elif MOBI_PP_DMA_AT == 2
        if (f_more) issue_x();
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if MOBI_STAMP == 3
        pp_t[5] = __builtin_amdgcn_s_memtime();
#endif
        if (late && ahead >= 2) wait_vmcnt_le_fast(vm_issued - mk1);
        MOBI_PP_T(2);
        MOBI_PP_BARRIER();
        MOBI_PP_T(3);
        __builtin_amdgcn_s_setprio(MOBI_PP_PRIO);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16(wf[ks][ni], xf[ks][mi], acc[ni][mi]);
#if MOBI_PP_DMA_AT != 0
          // requests in the shadow of the matrix work (the k-tile two steps ahead)
          if (ks == 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (f_more) {
#if MOBI_PP_DMA_AT == 1
              issue_x();
#endif
              issue_w(); ++ahead;
              if (ahead == 2) mk1 = vm_issued; else mk2 = vm_issued;
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#endif
        }
        __builtin_amdgcn_s_setprio(0);
        MOBI_PP_T(4);
#if MOBI_STAMP == 3
        if (kt != 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) pp_acc[0][i] += (unsigned)(pp_t[i + 1] - pp_t[i]);
          pp_acc[1][0] += (unsigned)(pp_t[2] - pp_t[5]);     // the counted vmcnt wait inside the LOAD phase (late waves)
          ++pp_n;
        }
        pp_t[5] = __builtin_amdgcn_s_memtime();
#endif
        if (!late && ahead >= 2) wait_vmcnt_le_fast(vm_issued - mk1);
#if MOBI_STAMP == 3
        if (kt != 0) pp_acc[1][1] += (unsigned)(__builtin_amdgcn_s_memtime() - pp_t[5]);   // the same wait, early waves
#endif
      }
#else
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // ---- LOAD phase ----------------------------------------------------------------------------------------
        MOBI_PP_T(0);
        MOBI_PP_BARRIER();
        MOBI_PP_T(1);
        if (ks == 0 && kt == 0 && bid != (int)blockIdx.x) {  // tile boundary: previous tile's registers -> memory
          // (the next tile's bias / per-image vector is requested AFTER the previous tile's epilogue: requested before it,
          //  its NT x 4 registers are live across the epilogue and the NT = 5 instance spills 8 VGPRs to scratch)
          wait_vmcnt_le(vm_issued - mk_req);                 // residual rows of the previous tile
          vm_issued += finish_tile(p_nw0, p_mw0);
          u32x4 bv[NT];
          if (has_vec) { request_vec(bv, nw0, mw0); vm_issued += NT; }
          const int mk_vec = vm_issued;
          wait_vmcnt_le(vm_issued - mk_vec);
          start_sums(bv);
          vm_issued += request_tile(nw0, mw0);
          mk_req = vm_issued;
        }
        const int sw = ((ks * 4 + g4) ^ (r16 & 7)) << 4;
        frag_t xf[4], wf[NT];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) xf[mi] = __builtin_bit_cast(frag_t, ld16(xb + mi * 16 * 128 + sw));
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) wf[ni] = __builtin_bit_cast(frag_t, ld16(wb + ni * 16 * 128 + sw));
        // k-tile two steps ahead, behind the reads' latency
#if MOBI_PP_DMA_SPLIT
        if (ks == 0) { f_more = f_bid < nblk; if (f_more) issue_x(); }
        if (ks == 1 && f_more) {
          issue_w(); ++ahead;
          if (ahead == 2) mk1 = vm_issued; else mk2 = vm_issued;
        }
#else
        if (ks == MOBI_PP_DMA_KS && f_bid < nblk) {
          issue_next(); ++ahead;
          if (ahead == 2) mk1 = vm_issued; else mk2 = vm_issued;
        }
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage may be overwritten after two more barriers
        if (ks == 1 && late && ahead >= 2) wait_vmcnt_le_fast(vm_issued - mk1);      // RAW rule above, late waves
        // ---- MATRIX phase --------------------------------------------------------------------------------------
        MOBI_PP_T(2);
        MOBI_PP_BARRIER();
        MOBI_PP_T(3);
        __builtin_amdgcn_s_setprio(MOBI_PP_PRIO);
#if MOBI_DBG_SKIP & 4
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) asm volatile("" :: "v"(xf[mi]));
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) asm volatile("" :: "v"(wf[ni]));
#else
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
          for (int mi = 0; mi < 4; ++mi) acc[ni][mi] = mfma16(wf[ni], xf[mi], acc[ni][mi]);
#endif
        __builtin_amdgcn_s_setprio(0);
        MOBI_PP_T(4);
#if MOBI_STAMP == 3
        if (kt != 0) {                                       // (tile boundaries excluded)
#pragma unroll
          for (int i = 0; i < 4; ++i) pp_acc[ks][i] += (unsigned)(pp_t[i + 1] - pp_t[i]);
          if (ks) ++pp_n;
        }
#endif
        if (ks == 1 && !late && ahead >= 2) wait_vmcnt_le_fast(vm_issued - mk1);     // RAW rule above, early waves
      }
#endif
      --ahead; mk1 = mk2;
      c_slot = c_slot == 2 ? 0 : c_slot + 1;
    }
    p_nw0 = nw0; p_mw0 = mw0;
  }
  if (!late) MOBI_PP_BARRIER();                              // every wave has passed the same number of barriers
  MOBI_STAMP_AT(2);
  wait_vmcnt_le(vm_issued - mk_req);
  finish_tile(p_nw0, p_mw0);
  if constexpr (SLAB) {
    // split-K finished inside the launch (the host sets `sync` only when every block owns exactly ONE output tile): the
    // block that arrives last at the tile sums its slabs and applies the epilogue
    if (a.sync) {
      const int L = xcd_remap(blockIdx.x, nblk);
      MOBI_TILE_OF_M(L, ftm, ftn);
      splitk_arrive_and_finish<T, 512>(a, L, ftm * BM, BM, ftn * BN, BN, &s_arrive);
    }
  }
#if MOBI_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the stamp then covers this wave's stores too
  MOBI_STAMP_AT(3);
  if (g_stamps && threadIdx.x == 0) {
    unsigned long long* d = g_stamps + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8;
    d[5] = (unsigned long long)nk;
    d[6] = (unsigned long long)((nblk - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
  }
#endif
#if MOBI_STAMP == 3
  if (g_phase && lane == 0) {
    unsigned long long* d = g_phase + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wave) * 16;
    for (int i = 0; i < 8; ++i) d[i] = pp_acc[i >> 2][i & 3];
    d[8] = pp_n;
  }
#endif
#undef MOBI_PP_T
#undef MOBI_PP_BARRIER
}

// =========================================================================================================
// LayerNorm folded into the 1 x 1 launch that consumes it (the ring kernels; attention.py:234 `attn1(norm1(x))` and :264
// `ff(norm3(x))` of the reference are LayerNorm -> Linear).  Exact algebra, as in csrc/chain.hip:
//     LN(x) W^T + b = rstd (x (W diag gamma)^T - mean s) + (W beta + b),      s = row sums of the ROUNDED W diag gamma
// The launch multiplies the RAW rows by the packed W diag(gamma); the row statistics come from the launch's own operand: a block
// sweeps the whole k range of its rows (1 x 1, one source, no split), and every A fragment a wave reads for the matrix cores holds
// 8 consecutive channels of one row per lane -- v_dot2c_f32 (two products per instruction, fp32 accumulate) sums them and their
// squares.  The MT fragments of a wave row are dealt to the NW / 2 waves that share it (SPW each: 8 instructions per fragment and
// k-step, issued between the step's MFMAs), the partial sums meet in 8 bytes of LDS per row at the end of the k loop, and the fold
// is applied in the accumulator domain before any epilogue runs -- no normalised copy of a row ever exists, and the LayerNorm
// launch (2 B read + 2 B written per element, 27 launches per mobi_nusc_512 step) is gone.
// =========================================================================================================
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rowstat_acc(const f16x8& x, float& sum, float& sq) {
  struct Q { f16x2_t p[4]; };
  const Q q = __builtin_bit_cast(Q, x);
  const f16x2_t one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sum = __builtin_amdgcn_fdot2(q.p[i], one, sum, false);
    sq = __builtin_amdgcn_fdot2(q.p[i], q.p[i], sq, false);
  }
}
__device__ __forceinline__ void rowstat_acc(const bf16x8& x, float& sum, float& sq) {
  struct Q { bf16x2_t p[4]; };
  const Q q = __builtin_bit_cast(Q, x);
  const bf16x2_t one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    sum = __builtin_amdgcn_fdot2_f32_bf16(q.p[i], one, sum, false);
    sq = __builtin_amdgcn_fdot2_f32_bf16(q.p[i], q.p[i], sq, false);
  }
}

// s_rowstat: BM x {mean, rstd} of the block's rows.  ls_sum / ls_sq: this wave's partial sums of fragments SPW * wn + i of its
// wave row (lane: row r16 of the fragment, k chunk g4).  add_bias: the epilogue that follows does not add the bias itself.
template <int NT, int MT, int SPW>
__device__ __forceinline__ void ln_fold_acc(const IgemmArgs& a, f32x4 (&acc)[NT][MT], const float (&ls_sum)[SPW],
                                            const float (&ls_sq)[SPW], float* s_rowstat, int lane, int wm, int wn, int nw0,
                                            bool add_bias) {
  const int r16 = lane & 15, g4 = lane >> 4;
  const float inv_k = 1.0f / (float)a.ktot;
#pragma unroll
  for (int i = 0; i < SPW; ++i) {
    float sm = ls_sum[i], sq = ls_sq[i];
    sm += __shfl_xor(sm, 16, 64); sq += __shfl_xor(sq, 16, 64);         // the four k-chunk lane groups of a row
    sm += __shfl_xor(sm, 32, 64); sq += __shfl_xor(sq, 32, 64);
    const float mean = sm * inv_k;
    const float var = fmaxf(sq * inv_k - mean * mean, 0.f);
    if (g4 == 0) {
      float* d = s_rowstat + ((wm * MT + SPW * wn + i) * 16 + r16) * 2;
      d[0] = mean; d[1] = rsqrtf(var + a.ln_eps);
    }
  }
  __syncthreads();
  // (few live registers: the 256 x 320 tiles hold 160 accumulators -- a row's two statistics are re-read per 16-pixel tile, the
  //  row sums / bias of a 16-channel tile per tile)
#pragma unroll
  for (int ni = 0; ni < NT; ++ni) {
    const int n = nw0 + ni * 16 + g4 * 4;
    f32x4 sv = f32x4{0.f, 0.f, 0.f, 0.f}, bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n < a.n_packed) {
      sv = *reinterpret_cast<const f32x4*>(a.ln_svec + n);
      if (add_bias && a.bias) bv = *reinterpret_cast<const f32x4*>(a.bias + n);
    }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const float* d = s_rowstat + ((wm * MT + mi) * 16 + r16) * 2;
      const float mean = d[0], rstd = d[1];
      acc[ni][mi] = rstd * (acc[ni][mi] - mean * sv) + bv;
    }
  }
}

// =========================================================================================================
// SMALL-M main loop (every shape the 256-pixel persistent kernels do not take: few rows, split-K ranges, ragged
// tiles, transposed / fp32 output): 128 pixels x (2 * WAVE_N) channels, FOUR waves (2 x 2, wave tile 64 x WAVE_N),
// TWO blocks per CU.  Operands arrive by LDS-DMA into a RING of four 32-deep k-slots (18 KB each): the requests of
// k-step s + 3 are issued while step s is multiplied, so three steps of prefetch cover the landing latency that
// dominates short launches (the register-staged loop it replaces had one k-tile of cover, measured 2.2 us per
// 64-deep k-tile at 17 % matrix-pipe use).  One raw s_barrier per k-step:
//     wait(own pieces of step s landed: counted vmcnt) ; barrier ; read fragments of slot s % 4 ;
//     request step s + 3 into slot (s + 3) % 4 == (s - 1) % 4 (every wave is past its reads of it) ; multiply
// The two co-resident blocks are not synchronised with each other: one multiplies while the other requests / waits.
// LDS image of a slot: [row][4 x 16 B] (64-byte rows); a 1-KiB request is 16 rows; the 16-byte slot of chunk c of
// row r is c ^ P[(r >> 2) & 3], P = {0, 2, 3, 1}: conflict-free for ds_read_b128's four lane groups (rows r16,
// chunk g4); applied on the SOURCE side of the request (the lane that fills slot s of row r fetches chunk s ^ P).
// Padded / out-of-range / past-the-k-range pieces use an offset beyond the descriptor: the hardware writes zeros.
// Every wave issues the same number of requests per step (R = 2 activation + 2 or 3 weight pieces), also past the
// end of its k range (all-zero pieces into slots nobody reads), so that every wait is the immediate vmcnt(2 R).
// Epilogue: the LDS-staged one (all output modes, split-K slabs, ragged tiles) through the drained ring.
// =========================================================================================================
// Geometries (NW waves as 2 along pixels x NW / 2 along channels; MT 16-pixel MFMA tiles per wave):
//   NW 4, MT 4  128 x (2 WAVE_N) tile, wave tile  64 x WAVE_N, two blocks per CU          (small m)
//   NW 8, MT 4  128 x (4 WAVE_N) tile, wave tile  64 x WAVE_N, one block per CU           (medium m: twice the blocks)
//   NW 8, MT 8  256 x (4 WAVE_N) tile, wave tile 128 x WAVE_N, one block per CU: 29 % fewer operand bytes per FLOP than
//               the 256 x 160 tile of the ping-pong kernel (the LDS-DMA path accepts ~42 B per clock and CU, which is
//               exactly what that tile needs at full matrix rate) and 13 instead of 18 fragment reads per 40 MFMAs
template <typename T, int NT, bool TR, int NW, int MT, bool LNF = false>
__global__ __launch_bounds__(NW * 64, 2) void igemm_ring_kernel(const IgemmArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr bool WIDE = NW == 8;                             // eight waves: the two halves run half a step apart
  constexpr int BM = 2 * MT * 16;
  constexpr int APW = BM / 16 / NW;                          // 1-KiB activation requests per wave and k-step (1 | 2)
  constexpr int WAVE_N = NT * 16;
  constexpr int BN = (NW / 2) * WAVE_N;
  constexpr int WPIECES = BN / 16;                           // 1-KiB weight requests per k-step
  constexpr int SLOT = (BM + BN) * 64;                       // bytes per 32-deep k-slot
  constexpr int RING = 4 * SLOT;
  constexpr int STAGE_BYTES = EpiGeom<32, TR, NT>::BYTES;    // per wave
  constexpr int LDS_BYTES = RING > NW * STAGE_BYTES ? RING : NW * STAGE_BYTES;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  __shared__ int s_arrive;                                   // split-K finished inside the launch: the arrival order
  __shared__ float s_rowstat[LNF ? BM * 2 : 2];              // LayerNorm folded into the launch: {mean, rstd} of the block's rows
  static_assert(!(LNF && TR), "the LayerNorm fold has no transposed-output form");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave & 1, wn = wave >> 1;
  const int group = blockIdx.z;
  const int nblk = a.tiles_m * a.tiles_n;
  const int L = xcd_remap(blockIdx.x, nblk);
  MOBI_TILE_OF(L, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  constexpr int SPW = MT / (NW / 2);                         // A fragments per wave whose row sums it accumulates (ln_fold_acc)
  constexpr bool lnf = LNF;                                  // (an instantiation of its own: the plain kernels keep their registers)
  float ls_sum[SPW], ls_sq[SPW];
#pragma unroll
  for (int i = 0; i < SPW; ++i) ls_sum[i] = ls_sq[i] = 0.f;
  MOBI_STAMP_AT(0);
  const bool w_tiled = a.w_tiled != nullptr;                 // (wave-uniform) weights as 1-KiB request images
  // (request images of stacked per-group matrices: a group's image has as many elements as its matrix and follows the previous one's)
  const T* wgt = (w_tiled ? reinterpret_cast<const T*>(a.w_tiled) : reinterpret_cast<const T*>(a.weight)) + (long long)group * a.w_group_stride;
  const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src0), 0, a.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src1 ? a.src1 : a.src0), 0, a.src1 ? a.src1_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wgt), 0, a.w_bytes, 0x00020000);

  // ---- request side: this lane fills slot (lane & 3) of row (lane >> 2) of each 16-row piece -----------------------
  const int rp = lane >> 2;
  const unsigned perm = (0x1320u >> (((rp >> 2) & 3) * 4)) & 3u;              // P = {0, 2, 3, 1}
  const unsigned chunk16 = (((unsigned)lane & 3u) ^ perm) * 16u;
  int x_base[APW], x_h[APW], x_w[APW];                       // activation rows 16 * (APW * wave + j) + rp of the tile
  bool x_ok[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int m = m0 + 16 * (APW * wave + j) + rp;
    x_ok[j] = m < a.M;
    const int mm = x_ok[j] ? m : 0;
    const int img = mm / a.hw_out, rem = mm - img * a.hw_out;
    const int ho = rem / a.wout, wo = rem - ho * a.wout;
    x_base[j] = (group * a.imgs_per_group + img) * a.img_pix_stride;
    x_h[j] = ho * a.stride - a.pad_h;
    x_w[j] = wo * a.stride - a.pad_w;
  }
  unsigned w_off[3];                                         // weight rows 16 * (wave + NW i) + rp of the tile
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int piece = wave + NW * i;
    const int n = n0 + 16 * piece + rp;
    w_off[i] = (piece < WPIECES && n < a.n_packed) ? (unsigned)n * (unsigned)a.ktot * 2u + chunk16 : OOB;
    // request images: piece (n0 / 16 + piece) starts at (k / 32) KiB times its index, a lane copies its own 16 bytes
    if (w_tiled) w_off[i] = (piece < WPIECES && n0 + 16 * piece < a.n_packed)
                                ? (unsigned)(n0 / 16 + piece) * (unsigned)(a.ktot >> 5) * 1024u + (unsigned)lane * 16u : OOB;
  }
  const int hlog = a.hin << a.up, wlog = a.win << a.up;
  // k range of this block in 32-deep steps (split-K: blockIdx.y owns 64-deep k-tiles [y * nk_per, (y + 1) * nk_per))
  const int ks_all = a.ktot >> 5;
  const int ks_begin = blockIdx.y * a.nk_per * 2;
  const int ks_end = min(ks_all, ks_begin + a.nk_per * 2);
  int u_c, u_tap, u_ky, u_kx;                                // (tap, channel) of the next step to request (tap-major k)
  {
    const long long c_first = (long long)ks_begin * 32;
    u_tap = (int)(c_first / a.C); u_c = (int)(c_first - (long long)u_tap * a.C);
    u_ky = u_tap / a.kw; u_kx = u_tap - u_ky * a.kw;
  }
  unsigned f_row[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) f_row[j] = OOB;
  int f_tap = -1, f_src = -1, f_ks = ks_begin;

  auto issue_step = [&]() {
    unsigned char* st = lds + (f_ks & 3) * SLOT;
    const bool live = f_ks < ks_end;                         // wave-uniform; past the range: all-zero pieces
    const int src = u_c >= a.c0 ? 1 : 0;
    if (live && (u_tap != f_tap || src != f_src)) {
      f_tap = u_tap; f_src = src;
      const unsigned cs2 = (unsigned)(src ? a.c1 : a.c0) * 2u;
#pragma unroll
      for (int j = 0; j < APW; ++j) {
        const int hi = x_h[j] + u_ky, wi = x_w[j] + u_kx;
        const bool ok = x_ok[j] && (unsigned)hi < (unsigned)hlog && (unsigned)wi < (unsigned)wlog;
        f_row[j] = ok ? (unsigned)(x_base[j] + (hi >> a.up) * a.win + (wi >> a.up)) * cs2 + chunk16 : OOB;
      }
    }
    const int soff = (src ? u_c - a.c0 : u_c) * 2;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      const unsigned vo = live ? f_row[j] : OOB;
      if (src) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (lds_ptr_t)(st + 16 * (APW * wave_s + j) * 64), 16, vo, soff, 0, 0);
      else     __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_ptr_t)(st + 16 * (APW * wave_s + j) * 64), 16, vo, soff, 0, 0);
    }
    const int kb = w_tiled ? f_ks * 1024 : f_ks * 64;        // byte offset of the step inside a weight row / a piece's images
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i == 2 && (NT != 5 || wave_s >= NW / 2)) break;    // a third piece exists for NT = 5 only (lower half of the waves)
      const unsigned vo = live ? w_off[i] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(st + BM * 64 + (wave_s + NW * i) * 1024), 16, vo, kb, 0, 0);
    }
    ++f_ks;
    if (live) {
      u_c += 32;
      if (u_c >= a.C) { u_c = 0; ++u_tap; if (++u_kx == a.kw) { u_kx = 0; ++u_ky; } }
    }
  };
  // own requests of the two youngest steps may stay in flight: R = APW + 2, + 1 for the lower half of the waves with NT = 5
  auto wait_step = [&]() {
    if (NT == 5 && wave_s < NW / 2) {
      if (APW == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      if (APW == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r16 = lane & 15, g4 = lane >> 4;
  if constexpr (WIDE && MT == 8 && !TR) {
    // register-epilogue launches: bias OR per-image vector start the sums (scale is 1; a wave's 128 pixels lie in one image)
    if (!lnf && a.ring_direct && (a.bias || a.rowvec)) {       // (LayerNorm fold: the bias is added behind the fold)
      const float* vec = a.bias;
      if (a.rowvec) {
        const int img = __builtin_amdgcn_readfirstlane((m0 + wm * 128) / a.hw_out);
        vec = a.rowvec + (long long)(group * a.imgs_per_group + img) * a.rowvec_stride;
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(vec + n0 + wn * WAVE_N + i * 16 + g4 * 4);
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = b;
      }
    }
  }
  const unsigned fsw = (((unsigned)g4 ^ ((0x1320u >> (((r16 >> 2) & 3) * 4)) & 3u)) << 4);
  const unsigned char* xrd = lds + (wm * MT * 16 + r16) * 64 + fsw;
  const unsigned char* wrd = lds + BM * 64 + (wn * WAVE_N + r16) * 64 + fsw;

  auto read_frags = [&](int s, frag_t (&xf)[MT], frag_t (&wf)[NT]) {
    const int so = (s & 3) * SLOT;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) xf[mi] = __builtin_bit_cast(frag_t, ld16(xrd + so + mi * 16 * 64));
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) wf[ni] = __builtin_bit_cast(frag_t, ld16(wrd + so + ni * 16 * 64));
  };
  auto multiply = [&](const frag_t (&xf)[MT], const frag_t (&wf)[NT]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi)
        acc[ni][mi] = TR ? mfma16(xf[mi], wf[ni], acc[ni][mi]) : mfma16(wf[ni], xf[mi], acc[ni][mi]);
    __builtin_amdgcn_s_setprio(0);
  };
  // LayerNorm fold: the wave's share of the A fragments (SPW of the MT of its wave row: fragments SPW * wn + i) are among the
  // fragments it multiplies anyway -- WHICH of them is a register index, so the k loop below is compiled once per wn (a generic
  // lambda over an integral constant, one scalar branch in front of the loop) and 8 v_dot2c per fragment sum the rows and their
  // squares between the step's MFMAs.  Two forms lost before this one (profiles/r05_ln_fold_launches.txt): a scalar branch per
  // candidate fragment INSIDE the loop (+20 % per launch), and a second LDS read of the share into registers of its own (+9 ... 14 %:
  // two more fragment reads in the LOAD phase, the longer one of the two).
  auto multiply_stats = [&](const frag_t (&xf)[MT], const frag_t (&wf)[NT], auto wnc) {
    constexpr int WNC = decltype(wnc)::value;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ni = 0; ni < NT; ++ni)
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) {
        acc[ni][mi] = TR ? mfma16(xf[mi], wf[ni], acc[ni][mi]) : mfma16(wf[ni], xf[mi], acc[ni][mi]);
        if (ni == 0 && mi < SPW) rowstat_acc(xf[SPW * WNC + mi], ls_sum[mi], ls_sq[mi]);
      }
    // the order the scheduler has to keep: one v_dot2c behind every second MFMA (left alone the whole chain is sunk into the
    // loop's latch block, behind the last MFMA)
#pragma unroll
    for (int i = 0; i < 8 * SPW; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < SPW; ++i) asm volatile("" : "+v"(ls_sum[i]), "+v"(ls_sq[i]));   // (the sums are "used" here: no sinking)
    __builtin_amdgcn_s_setprio(0);
  };
#define MOBI_RING_BARRIER()                   \
  do {                                        \
    __builtin_amdgcn_sched_barrier(0);        \
    __builtin_amdgcn_s_barrier();             \
    __builtin_amdgcn_sched_barrier(0);        \
  } while (0)

  issue_step(); issue_step(); issue_step();
  auto k_loop = [&](auto wnc) {                              // (wnc: the wave's column index as a constant -- LayerNorm fold only)
  if constexpr (WIDE && MOBI_RING_STAGGER) {
    // Two waves share a SIMD (w and w + 4).  In lockstep both read / request, then both queue on the one matrix
    // pipe; here waves 4-7 run HALF A STEP behind waves 0-3: every step is a LOAD phase (fragment reads, the requests
    // of step s + 3, the counted wait) and a MATRIX phase (40 MFMAs), a barrier at every phase boundary, so that one
    // wave of each SIMD multiplies while its partner loads.
    //   RAW  step s + 1 is first read in the phase after the one in which every wave waits for its own pieces of it
    //        (early waves: behind MATRIX(s); late waves: at the end of LOAD(s) -- the same phase).
    //   WAR  the requests of step s + 3 overwrite slot (s - 1) % 4, last read by the late waves two phases earlier.
    const bool late = wave_s >= NW / 2;
    wait_step();
    MOBI_STAMP_AT(1);
    if (late) MOBI_RING_BARRIER();
#if MOBI_STAMP == 4                                           // shader cycles per phase of a k-step, per wave (tools/phase_ring.py)
    unsigned long long wp_t[8], wp_acc[7] = {0, 0, 0, 0, 0, 0, 0};
#define MOBI_WP(i) wp_t[i] = __builtin_amdgcn_s_memtime()
#else
#define MOBI_WP(i) ((void)0)
#endif
#pragma clang loop unroll(disable)
    for (int s = ks_begin; s < ks_end; ++s) {
      MOBI_WP(0);
      MOBI_RING_BARRIER();
      MOBI_WP(1);
      frag_t xf[MT], wf[NT];
      read_frags(s, xf, wf);
      __builtin_amdgcn_sched_barrier(0);
      issue_step();
      __builtin_amdgcn_sched_barrier(0);
      MOBI_WP(2);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MOBI_WP(3);
      if (late) wait_step();
      MOBI_WP(4);
      MOBI_RING_BARRIER();
      MOBI_WP(5);
      if constexpr (lnf) multiply_stats(xf, wf, wnc); else multiply(xf, wf);
#if MOBI_STAMP == 4
      asm volatile("s_nop 0" ::: "memory");
#endif
      MOBI_WP(6);
      if (!late) wait_step();
#if MOBI_STAMP == 4
      MOBI_WP(7);
      if (s > ks_begin) for (int i = 0; i < 7; ++i) wp_acc[i] += wp_t[i + 1] - wp_t[i];
#endif
    }
#if MOBI_STAMP == 4
    if (g_phase && lane == 0) {
      unsigned long long* d = g_phase + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wave) * 8;
      for (int i = 0; i < 7; ++i) d[i] = wp_acc[i];
      d[7] = (unsigned long long)(ks_end - ks_begin - 1);
    }
#endif
#undef MOBI_WP
    if (!late) MOBI_RING_BARRIER();
  } else {
#if MOBI_STAMP == 4                                           // shader cycles per phase of a k-step, per wave (tools/stamp_ring.py)
    unsigned long long rp_t[6], rp_acc[5] = {0, 0, 0, 0, 0};
#define MOBI_RP(i) rp_t[i] = __builtin_amdgcn_s_memtime()
#else
#define MOBI_RP(i) ((void)0)
#endif
#pragma clang loop unroll(disable)
    for (int s = ks_begin; s < ks_end; ++s) {
      MOBI_RP(0);
      wait_step();
      MOBI_RP(1);
      MOBI_RING_BARRIER();
      MOBI_RP(2);
      frag_t xf[MT], wf[NT];
      read_frags(s, xf, wf);
      __builtin_amdgcn_sched_barrier(0);
      issue_step();                                          // step s + 3, behind the reads' latency
      __builtin_amdgcn_sched_barrier(0);
      MOBI_RP(3);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MOBI_RP(4);
      if constexpr (lnf) multiply_stats(xf, wf, wnc); else multiply(xf, wf);
#if MOBI_STAMP == 4
      asm volatile("s_nop 0" ::: "memory");
      MOBI_RP(5);
      if (s > ks_begin) for (int i = 0; i < 5; ++i) rp_acc[i] += rp_t[i + 1] - rp_t[i];
#endif
    }
#if MOBI_STAMP == 4
    if (g_phase && lane == 0) {
      unsigned long long* d = g_phase + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + wave) * 8;
      for (int i = 0; i < 5; ++i) d[i] = rp_acc[i];
      d[5] = (unsigned long long)(ks_end - ks_begin - 1);
    }
#endif
#undef MOBI_RP
  }
  };
  if constexpr (lnf) {
    const int wn_s = wave_s >> 1;
    if (wn_s == 0) k_loop(std::integral_constant<int, 0>{});
    else if (wn_s == 1 || NW == 4) k_loop(std::integral_constant<int, 1>{});
    else if (wn_s == 2) k_loop(std::integral_constant<int, NW == 8 ? 2 : 0>{});
    else k_loop(std::integral_constant<int, NW == 8 ? 3 : 0>{});
  } else {
    k_loop(std::integral_constant<int, 0>{});
  }
#undef MOBI_RING_BARRIER
  MOBI_STAMP_AT(2);
  if constexpr (lnf) {
    // LayerNorm fold: acc <- rstd (acc - mean s) [+ bias where the epilogue below does not add it: the 256 x 320 register epilogue,
    // whose bias normally starts the sums]
    ln_fold_acc<NT, MT, SPW>(a, acc, ls_sum, ls_sq, s_rowstat, lane, wm, wave_s >> 1, n0 + wn * WAVE_N,
                             WIDE && MT == 8 && a.ring_direct);
  }
  if constexpr (MT == 4 && !TR) {
    if (a.sm_direct) {                                       // wave-uniform
      // 128-pixel tiles, full: the wave tile is the ping-pong kernel's (64 pixels x NT * 16 channels), so are its register
      // epilogue and its slab stores; no LDS staging, no block barrier
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every request of the ring has landed before the wave ends
      const int nw0 = n0 + wn * WAVE_N, mw0 = m0 + wm * 64;
      if (a.split_ws) {
        float* wsp = a.split_ws + (long long)blockIdx.y * a.M * a.n_packed;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          float* rowp = wsp + (long long)(mw0 + mi * 16 + r16) * a.n_packed + nw0 + g4 * 4;
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) slab_store16(a.sync_mode == 1, rowp + ni * 16, __builtin_bit_cast(u32x4, acc[ni][mi]));
        }
        if (a.sync) {                                          // split-K finished inside the launch by the tile's last block
          splitk_arrive_and_finish<T, NW * 64>(a, tile_m * a.tiles_n + tile_n, m0, BM, n0, BN, &s_arrive);
        }
      } else {
        DirectEpiRegs<NT> q;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) { q.bias[ni] = u32x4{0u, 0u, 0u, 0u}; q.res[ni][0] = q.res[ni][1] = u32x4{0u, 0u, 0u, 0u}; }
        direct_epilogue_request<T, NT>(a, q, lane, group, nw0, mw0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.epilogue == MOBI_EPI_GEGLU) direct_epilogue<T, NT, true>(a, acc, q, lane, group, nw0, mw0);
        else direct_epilogue<T, NT, false>(a, acc, q, lane, group, nw0, mw0);
      }
      return;
    }
  }
  if constexpr (WIDE && MT == 8 && !TR) {
    if (a.ring_direct) {                                     // wave-uniform
      // register epilogue: no LDS staging, no block barrier (the waves leave at their own pace); the accumulator layout is the
      // ping-pong kernel's (channels x pixels), 64 pixels of the wave's 128 at a time.  Every request of the ring (also the
      // all-zero ones past the k range) has to have landed before the wave ends.
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (MT == 8) {
        if (a.epilogue == MOBI_EPI_GEGLU) ring_register_epilogue<T, NT, true>(a, acc, lane, group, n0 + wn * WAVE_N, m0 + wm * 128);
        else ring_register_epilogue<T, NT, false>(a, acc, lane, group, n0 + wn * WAVE_N, m0 + wm * 128);
      }
#if MOBI_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      MOBI_STAMP_AT(3);
      if (g_stamps && threadIdx.x == 0) {
        unsigned long long* d = g_stamps + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d[4] = ((unsigned long long)xcc << 32) | hw;
        d[5] = (unsigned long long)(ks_end - ks_begin);
        d[6] = 1;
      }
#endif
      return;
    }
  }
  // the ring becomes the epilogue's staging area: every request (also the all-zero ones) must have landed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* stage = reinterpret_cast<float*>(lds) + wave * (STAGE_BYTES / 4);
#pragma unroll
  for (int h = 0; h < MT / 4; ++h) {                         // the epilogue takes 64-pixel wave tiles
    f32x4 part[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[i][j] = acc[i][4 * h + j];
    igemm_epilogue<T, NT, TR, 32, MT == 4>(a, stage, part, lane, group, n0 + wn * WAVE_N, m0 + wm * MT * 16 + 64 * h);
  }
  if constexpr (!TR) {
    if (a.split_ws && a.sync) {                              // split-K finished inside the launch by the tile's last block
      splitk_arrive_and_finish<T, NW * 64>(a, tile_m * a.tiles_n + tile_n, m0, BM, n0, BN, &s_arrive);
    }
  }
#if MOBI_STAMP                                                // (eight-wave geometry: tools/stamp_ring.py)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MOBI_STAMP_AT(3);
  if (g_stamps && threadIdx.x == 0) {
    unsigned long long* d = g_stamps + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    d[4] = ((unsigned long long)xcc << 32) | hw;
    d[5] = (unsigned long long)(ks_end - ks_begin);
    d[6] = 1;
  }
#endif
}

// =========================================================================================================
// SMALL-M main loop with 64-deep k-steps (launches of the 128-pixel geometry whose channel counts are multiples of 64):
// the four-wave, 128 x (2 * WAVE_N) tile of igemm_ring_kernel, ONE block per CU, ring of four 64-deep k-slots of
// 128-byte LDS rows.  Why: a CU's request path takes a request by its row segments -- 8 rows x 128 B go in at 42 B per
// clock (four waves), 16 rows x 64 B (the 32-deep steps) at 21.5 (tools/probes/lds_dma_rate.hip) -- and the small
// launches are bound by exactly that: of ~1,700 cycles per 32-deep step 1,130 are spent ISSUING the step's 4-5 requests per
// wave, 357 in its 20 MFMAs (tools/phase_ring.py).  Here a step carries twice the MFMAs per barrier pair and its
// activation requests cost half as much per byte.
// LDS image of a slot: [row][8 x 16 B] (128-byte rows), chunk c of row r at position c ^ (r & 7) (conflict-free for
// ds_read_b128: the ping-pong kernel's rule); a 1-KiB request is 8 rows, lane l fills position l & 7 of row l >> 3.
// One raw s_barrier per k-step, three steps of prefetch, every wave issues R = 4 + NT requests per step (all-zero
// pieces past its k range), counted waits.  Epilogues: the ring kernel's (registers / register slab stores for full tiles,
// else LDS-staged through the drained ring).
// =========================================================================================================
template <typename T, int NT, bool TR, bool LNF = false>
__global__ __launch_bounds__(256, 1) void igemm_ring64_kernel(const IgemmArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int NW = 4, MT = 4, BM = 128, SL = 4;
  constexpr int APW = BM / 8 / NW;                           // 4 activation requests per wave and k-step
  constexpr int WAVE_N = NT * 16;
  constexpr int BN = 2 * WAVE_N;                             // NT weight requests per wave and k-step
  constexpr int SLOT = (BM + BN) * 128;
  constexpr int RING = SL * SLOT;
  constexpr int STAGE_BYTES = EpiGeom<32, TR, NT>::BYTES;
  constexpr int LDS_BYTES = RING > NW * STAGE_BYTES ? RING : NW * STAGE_BYTES;
  constexpr unsigned OOB = 0x80000000u;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  __shared__ int s_arrive;                                   // split-K finished inside the launch: the arrival order
  __shared__ float s_rowstat[LNF ? BM * 2 : 2];              // LayerNorm folded into the launch: {mean, rstd} of the block's rows
  static_assert(!(LNF && TR), "the LayerNorm fold has no transposed-output form");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wm = wave & 1, wn = wave >> 1;
  const int group = blockIdx.z;
  const int nblk = a.tiles_m * a.tiles_n;
  const int L = xcd_remap(blockIdx.x, nblk);
  MOBI_TILE_OF(L, tile_m, tile_n);
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const T* wgt = reinterpret_cast<const T*>(a.weight) + (long long)group * a.w_group_stride;
  const __amdgpu_buffer_rsrc_t rx0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src0), 0, a.src0_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx1 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.src1 ? a.src1 : a.src0), 0, a.src1 ? a.src1_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wgt), 0, a.w_bytes, 0x00020000);

  // ---- request side: this lane fills position (lane & 7) of row (lane >> 3) of each 8-row piece ---------------------
  const int rp = lane >> 3;
  const unsigned chunk16 = (unsigned)((lane & 7) ^ rp) * 16u;
  int x_base[APW], x_h[APW], x_w[APW];
  bool x_ok[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int m = m0 + 8 * (APW * wave + j) + rp;
    x_ok[j] = m < a.M;
    const int mm = x_ok[j] ? m : 0;
    const int img = mm / a.hw_out, rem = mm - img * a.hw_out;
    const int ho = rem / a.wout, wo = rem - ho * a.wout;
    x_base[j] = (group * a.imgs_per_group + img) * a.img_pix_stride;
    x_h[j] = ho * a.stride - a.pad_h;
    x_w[j] = wo * a.stride - a.pad_w;
  }
  unsigned w_off[NT];                                        // weight rows 8 * (wave + NW i) + rp of the tile
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int n = n0 + 8 * (wave + NW * i) + rp;
    w_off[i] = n < a.n_packed ? (unsigned)n * (unsigned)a.ktot * 2u + chunk16 : OOB;
  }
  const int hlog = a.hin << a.up, wlog = a.win << a.up;
  const int ks_all = a.ktot >> 6;                            // 64-deep steps; split-K: blockIdx.y owns [y nk_per, (y + 1) nk_per)
  const int ks_begin = blockIdx.y * a.nk_per;
  const int ks_end = min(ks_all, ks_begin + a.nk_per);
  int u_c, u_tap, u_ky, u_kx;
  {
    const long long c_first = (long long)ks_begin * 64;
    u_tap = (int)(c_first / a.C); u_c = (int)(c_first - (long long)u_tap * a.C);
    u_ky = u_tap / a.kw; u_kx = u_tap - u_ky * a.kw;
  }
  unsigned f_row[APW];
#pragma unroll
  for (int j = 0; j < APW; ++j) f_row[j] = OOB;
  int f_tap = -1, f_src = -1, f_ks = ks_begin;

  auto issue_step = [&]() {
    unsigned char* st = lds + (f_ks & (SL - 1)) * SLOT;
    const bool live = f_ks < ks_end;
    const int src = u_c >= a.c0 ? 1 : 0;
    if (live && (u_tap != f_tap || src != f_src)) {
      f_tap = u_tap; f_src = src;
      const unsigned cs2 = (unsigned)(src ? a.c1 : a.c0) * 2u;
#pragma unroll
      for (int j = 0; j < APW; ++j) {
        const int hi = x_h[j] + u_ky, wi = x_w[j] + u_kx;
        const bool ok = x_ok[j] && (unsigned)hi < (unsigned)hlog && (unsigned)wi < (unsigned)wlog;
        f_row[j] = ok ? (unsigned)(x_base[j] + (hi >> a.up) * a.win + (wi >> a.up)) * cs2 + chunk16 : OOB;
      }
    }
    const int soff = (src ? u_c - a.c0 : u_c) * 2;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      const unsigned vo = live ? f_row[j] : OOB;
      if (src) __builtin_amdgcn_raw_ptr_buffer_load_lds(rx1, (lds_ptr_t)(st + 8 * (APW * wave_s + j) * 128), 16, vo, soff, 0, 0);
      else     __builtin_amdgcn_raw_ptr_buffer_load_lds(rx0, (lds_ptr_t)(st + 8 * (APW * wave_s + j) * 128), 16, vo, soff, 0, 0);
    }
    const int kb = f_ks * 128;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const unsigned vo = live ? w_off[i] : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(st + BM * 128 + 8 * (wave_s + NW * i) * 128), 16, vo, kb, 0, 0);
    }
    ++f_ks;
    if (live) {
      u_c += 64;
      if (u_c >= a.C) { u_c = 0; ++u_tap; if (++u_kx == a.kw) { u_kx = 0; ++u_ky; } }
    }
  };
  // own requests of the two youngest steps may stay in flight: R = 4 + NT
  auto wait_step = [&]() {
    if (NT == 5) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  };

  f32x4 acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int SPW = MT / (NW / 2);                         // LayerNorm fold (ln_fold_acc): this wave's share of the A fragments
  constexpr bool lnf = LNF;
  float ls_sum[SPW], ls_sq[SPW];
#pragma unroll
  for (int i = 0; i < SPW; ++i) ls_sum[i] = ls_sq[i] = 0.f;

  const int r16 = lane & 15, g4 = lane >> 4;
  const unsigned char* xrd = lds + (wm * 64 + r16) * 128;
  const unsigned char* wrd = lds + BM * 128 + (wn * WAVE_N + r16) * 128;

  issue_step(); issue_step(); issue_step();
  auto k_loop = [&](auto wnc) {                              // (compiled once per wave column: LayerNorm fold, see igemm_ring_kernel)
  constexpr int WNC = decltype(wnc)::value;
#pragma clang loop unroll(disable)
  for (int s = ks_begin; s < ks_end; ++s) {
    wait_step();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const int so = (s & (SL - 1)) * SLOT;
    frag_t xf[2][MT], wf[2][NT];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int sw = ((ks * 4 + g4) ^ (r16 & 7)) << 4;
#pragma unroll
      for (int mi = 0; mi < MT; ++mi) xf[ks][mi] = __builtin_bit_cast(frag_t, ld16(xrd + so + mi * 16 * 128 + sw));
#pragma unroll
      for (int ni = 0; ni < NT; ++ni) wf[ks][ni] = __builtin_bit_cast(frag_t, ld16(wrd + so + ni * 16 * 128 + sw));
    }
    __builtin_amdgcn_sched_barrier(0);
    issue_step();                                            // step s + 3, behind the reads' latency
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int ni = 0; ni < NT; ++ni)
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
          acc[ni][mi] = TR ? mfma16(xf[ks][mi], wf[ks][ni], acc[ni][mi]) : mfma16(wf[ks][ni], xf[ks][mi], acc[ni][mi]);
          if constexpr (lnf) if (ni == 0 && mi < SPW) rowstat_acc(xf[ks][SPW * WNC + mi], ls_sum[mi], ls_sq[mi]);
        }
    if constexpr (lnf) {                                     // one v_dot2c behind every MFMA (see igemm_ring_kernel)
#pragma unroll
      for (int i = 0; i < 16 * SPW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < SPW; ++i) asm volatile("" : "+v"(ls_sum[i]), "+v"(ls_sq[i]));
    }
    __builtin_amdgcn_s_setprio(0);
  }
  };
  if constexpr (lnf) {
    if ((wave_s >> 1) == 0) k_loop(std::integral_constant<int, 0>{}); else k_loop(std::integral_constant<int, 1>{});
  } else {
    k_loop(std::integral_constant<int, 0>{});
  }
  if constexpr (lnf) ln_fold_acc<NT, MT, SPW>(a, acc, ls_sum, ls_sq, s_rowstat, lane, wm, wave_s >> 1, n0 + wn * WAVE_N, false);
  if constexpr (!TR) {
    if (a.sm_direct) {                                       // full tiles: registers -> memory (no staging, no block barrier)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int nw0 = n0 + wn * WAVE_N, mw0 = m0 + wm * 64;
      if (a.split_ws) {
        float* wsp = a.split_ws + (long long)blockIdx.y * a.M * a.n_packed;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          float* rowp = wsp + (long long)(mw0 + mi * 16 + r16) * a.n_packed + nw0 + g4 * 4;
#pragma unroll
          for (int ni = 0; ni < NT; ++ni) slab_store16(a.sync_mode == 1, rowp + ni * 16, __builtin_bit_cast(u32x4, acc[ni][mi]));
        }
        if (a.sync) {                                          // split-K finished inside the launch by the tile's last block
          splitk_arrive_and_finish<T, NW * 64>(a, tile_m * a.tiles_n + tile_n, m0, BM, n0, BN, &s_arrive);
        }
      } else {
        DirectEpiRegs<NT> q;
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) { q.bias[ni] = u32x4{0u, 0u, 0u, 0u}; q.res[ni][0] = q.res[ni][1] = u32x4{0u, 0u, 0u, 0u}; }
        direct_epilogue_request<T, NT>(a, q, lane, group, nw0, mw0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.epilogue == MOBI_EPI_GEGLU) direct_epilogue<T, NT, true>(a, acc, q, lane, group, nw0, mw0);
        else direct_epilogue<T, NT, false>(a, acc, q, lane, group, nw0, mw0);
      }
      return;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  float* stage = reinterpret_cast<float*>(lds) + wave * (STAGE_BYTES / 4);
  igemm_epilogue<T, NT, TR, 32, true>(a, stage, acc, lane, group, n0 + wn * WAVE_N, m0 + wm * 64);
  if constexpr (!TR) {
    if (a.split_ws && a.sync) {
      splitk_arrive_and_finish<T, NW * 64>(a, tile_m * a.tiles_n + tile_n, m0, BM, n0, BN, &s_arrive);
    }
  }
}

// split-K finish: sum the partial slabs, then the ordinary epilogue (bias, per-image vector, residual)
template <typename T>
__global__ __launch_bounds__(256) void igemm_splitk_reduce_kernel(const IgemmArgs a) {
  const int vpr = a.cout >> 3;
  const long long total = (long long)a.M * vpr;
  T* __restrict__ outT = reinterpret_cast<T*>(a.out);
  float* __restrict__ outF = reinterpret_cast<float*>(a.out);
  const T* __restrict__ resid = reinterpret_cast<const T*>(a.residual);
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int m = (int)(i / vpr);
    const int n = (int)(i - (long long)m * vpr) * 8;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    // four slabs requested at once, summed in ascending order (the same sums as one at a time: a launch with 8-16 splits and a
    // few thousand items per CU was a chain of 8-16 dependent L2 round trips)
    const long long slab = (long long)a.M * a.n_packed;
    const float* p0 = a.split_ws + (long long)m * a.n_packed + n;
    int s = 0;
    for (; s + 4 <= a.splits; s += 4) {
      f32x4 v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* p = p0 + (s + u) * slab;
        v[u][0] = *reinterpret_cast<const f32x4*>(p);
        v[u][1] = *reinterpret_cast<const f32x4*>(p + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] += v[u][0][j]; o[4 + j] += v[u][1][j]; }
    }
    for (; s < a.splits; ++s) {
      const float* p = p0 + s * slab;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] += v0[j]; o[4 + j] += v1[j]; }
    }
    const int img = m / a.hw_out, rem = m - img * a.hw_out;
    if (a.bias) {
      float bb[8];
      ld8f(a.bias + n, bb);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += bb[j];
    }
    if (a.rowvec) {
      float rv[8];
      ld8f(a.rowvec + (long long)img * a.rowvec_stride + n, rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += rv[j];
    }
    if (resid) {
      float rf[8];
      unpack8<T>(ld16(resid + (long long)img * a.res_img_stride + (long long)rem * a.cout + n), rf);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += rf[j];
    }
    const long long off = (long long)img * a.out_img_stride + (long long)rem * a.cout + n;
    if (a.out_mode == MOBI_OUT_ROWS_F32) {
      *reinterpret_cast<f32x4*>(outF + off) = f32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(outF + off + 4) = f32x4{o[4], o[5], o[6], o[7]};
    } else {
      st16(outT + off, pack8<T>(o));
    }
  }
}

// CUs of the current device (cached per device ordinal); 256 on MI355X.  MOBI_IGEMM_PERSIST_BLOCKS overrides the
// persistent grid size (tests: few blocks walk many output tiles).
static int compute_units() {
  if (tuning().persist_blocks > 0) return tuning().persist_blocks;
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (!cached[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

template <typename T>
static int launch_splitk_reduce(const IgemmArgs& a, hipStream_t st) {
  long long blocks = ((long long)a.M * (a.cout >> 3) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((igemm_splitk_reduce_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

template <typename T>
static int launch_igemm(const mobi_igemm_params* p, const IgemmArgs& a, int groups, hipStream_t st) {
  const bool tr = p->out_mode == MOBI_OUT_TRANSPOSED;
  const bool nt5 = (a.n_packed % 160) == 0;
  dim3 grid(a.tiles_m * a.tiles_n, a.splits, groups);
#define MOBI_IGEMM_LAUNCH(NT_, TR_, WM_, FAST_) \
  hipLaunchKernelGGL((igemm_kernel<T, NT_, TR_, WM_, FAST_>), grid, dim3(128 * WM_), 0, st, a)
#ifdef MOBI_DEV   // the register-staged kernel's fast-addressing instances: A/B partners of the ring kernel only
#define MOBI_IGEMM_BY_FAST(NT_, TR_, WM_) \
  do { if (a.fast) MOBI_IGEMM_LAUNCH(NT_, TR_, WM_, true); else MOBI_IGEMM_LAUNCH(NT_, TR_, WM_, false); } while (0)
#else             // shipped: the generic-gather instance is the fallback for operands beyond 2 GB / chunk-major k
#define MOBI_IGEMM_BY_FAST(NT_, TR_, WM_) MOBI_IGEMM_LAUNCH(NT_, TR_, WM_, false)
#endif
#define MOBI_IGEMM_BY_TR(NT_, WM_) \
  do { if (tr) MOBI_IGEMM_BY_FAST(NT_, true, WM_); else MOBI_IGEMM_BY_FAST(NT_, false, WM_); } while (0)
  if (a.wide == 2) {
#define MOBI_W2_LAUNCH(NT_, TR_) hipLaunchKernelGGL((igemm_ring_kernel<T, NT_, TR_, 8, 8>), grid, dim3(512), 0, st, a)
    if (a.ln_svec) {                                         // LayerNorm folded into the launch: instantiations of their own
      if (nt5) hipLaunchKernelGGL((igemm_ring_kernel<T, 5, false, 8, 8, true>), grid, dim3(512), 0, st, a);
      else     hipLaunchKernelGGL((igemm_ring_kernel<T, 4, false, 8, 8, true>), grid, dim3(512), 0, st, a);
    }
    else if (nt5) { if (tr) MOBI_W2_LAUNCH(5, true); else MOBI_W2_LAUNCH(5, false); }
    else     { if (tr) MOBI_W2_LAUNCH(4, true); else MOBI_W2_LAUNCH(4, false); }
#undef MOBI_W2_LAUNCH
  }
  else if (a.wide == 1) {
    if (nt5) hipLaunchKernelGGL((igemm_ring_kernel<T, 5, false, 8, 4>), grid, dim3(512), 0, st, a);
    else     hipLaunchKernelGGL((igemm_ring_kernel<T, 4, false, 8, 4>), grid, dim3(512), 0, st, a);
  }
  else if (a.wm == 4 && a.fast && a.glds && a.pp) {
    // persistent: one 156-KB-LDS block per CU walks the output tiles
    dim3 block(512);
    dim3 pgrid(grid.x < (unsigned)compute_units() ? grid.x : (unsigned)compute_units(), grid.y, grid.z);
    if (a.split_ws) {
      if (nt5) hipLaunchKernelGGL((igemm_pp_kernel<T, 5, false, true>), pgrid, block, 0, st, a);
      else     hipLaunchKernelGGL((igemm_pp_kernel<T, 4, false, true>), pgrid, block, 0, st, a);
    } else {
#define MOBI_PP_LAUNCH(NT_, G_) hipLaunchKernelGGL((igemm_pp_kernel<T, NT_, G_>), pgrid, block, 0, st, a)
      if (a.epilogue == MOBI_EPI_GEGLU) { if (nt5) MOBI_PP_LAUNCH(5, true); else MOBI_PP_LAUNCH(4, true); }
      else                              { if (nt5) MOBI_PP_LAUNCH(5, false); else MOBI_PP_LAUNCH(4, false); }
#undef MOBI_PP_LAUNCH
    }
  }
#ifdef MOBI_DEV
  else if (a.wm == 4 && a.fast && a.glds) {
    dim3 block(512);
    dim3 pgrid(grid.x < (unsigned)compute_units() ? grid.x : (unsigned)compute_units(), grid.y, grid.z);
    const int mode = tr ? 1 : (a.epi_direct ? (a.epilogue == MOBI_EPI_GEGLU ? 3 : 2) : 0);
#define MOBI_GLDS_LAUNCH(NT_, MODE_) hipLaunchKernelGGL((igemm_glds_kernel<T, NT_, MODE_>), pgrid, block, 0, st, a)
#define MOBI_GLDS_BY_MODE(NT_)                                                                     \
  do { switch (mode) { case 0: MOBI_GLDS_LAUNCH(NT_, 0); break; case 1: MOBI_GLDS_LAUNCH(NT_, 1); break; \
                       case 2: MOBI_GLDS_LAUNCH(NT_, 2); break; default: MOBI_GLDS_LAUNCH(NT_, 3); break; } } while (0)
    if (nt5) MOBI_GLDS_BY_MODE(5); else MOBI_GLDS_BY_MODE(4);
#undef MOBI_GLDS_BY_MODE
#undef MOBI_GLDS_LAUNCH
  }
#endif
  else if (a.wm == 4) { if (nt5) MOBI_IGEMM_BY_TR(5, 4); else MOBI_IGEMM_BY_TR(4, 4); }
  else if (a.sm && a.sm64) {
#define MOBI_SM64_LAUNCH(NT_, TR_) hipLaunchKernelGGL((igemm_ring64_kernel<T, NT_, TR_>), grid, dim3(256), 0, st, a)
    if (a.ln_svec) {
      if (nt5) hipLaunchKernelGGL((igemm_ring64_kernel<T, 5, false, true>), grid, dim3(256), 0, st, a);
      else     hipLaunchKernelGGL((igemm_ring64_kernel<T, 4, false, true>), grid, dim3(256), 0, st, a);
    }
    else if (nt5) { if (tr) MOBI_SM64_LAUNCH(5, true); else MOBI_SM64_LAUNCH(5, false); }
    else     { if (tr) MOBI_SM64_LAUNCH(4, true); else MOBI_SM64_LAUNCH(4, false); }
#undef MOBI_SM64_LAUNCH
  }
  else if (a.sm) {
#define MOBI_SM_LAUNCH(NT_, TR_) hipLaunchKernelGGL((igemm_ring_kernel<T, NT_, TR_, 4, 4>), grid, dim3(256), 0, st, a)
    if (a.ln_svec) {
      if (nt5) hipLaunchKernelGGL((igemm_ring_kernel<T, 5, false, 4, 4, true>), grid, dim3(256), 0, st, a);
      else     hipLaunchKernelGGL((igemm_ring_kernel<T, 4, false, 4, 4, true>), grid, dim3(256), 0, st, a);
    }
    else if (nt5) { if (tr) MOBI_SM_LAUNCH(5, true); else MOBI_SM_LAUNCH(5, false); }
    else     { if (tr) MOBI_SM_LAUNCH(4, true); else MOBI_SM_LAUNCH(4, false); }
#undef MOBI_SM_LAUNCH
  }
  else                { if (nt5) MOBI_IGEMM_BY_TR(5, 2); else MOBI_IGEMM_BY_TR(4, 2); }
#undef MOBI_IGEMM_BY_TR
#undef MOBI_IGEMM_BY_FAST
#undef MOBI_IGEMM_LAUNCH
  MOBI_CHECK_LAUNCH();
  // (with `sync` the tile's last block has summed the slabs; with defer_finish the consumer does -- mobi_groupnorm's src0_split --
  //  or a later mobi_igemm_finish)
  if (a.split_ws && !a.sync && !p->defer_finish) return launch_splitk_reduce<T>(a, st);
  return MOBI_OK;
}

// Small problems go to csrc/igemm_small.hip (coalesced loads, a wave-private LDS transposition, k split over the block's four
// waves, no slabs): returns its tile (32), or 0.  Rules (tools/small_lab.py, graph-timed on cold operands,
// profiles/r04_small_lab.txt):
//   1 x 1: up to 1.8 GFLOP of 2 M N K -- 1.0 GFLOP at K = 320, where a wave has a single batch and the LDS kernels' launch is a
//          10-step loop (8192 x 320 x 320: 16.6 against 12.6 us; 4096 x 320 x 320: 9.9 against 11.8);
//   3 x 3 (pad 1, stride 1): NOT routed (MOBI_IGEMM_SMALL_CONV_M output pixels, default 0) -- built for the 4 x 4 levels of
//          `mobi_nusc_256` (128 pixels, 30-60 MB of weights per launch) and measured SLOWER there: 36.1 against 24.7 us
//          (1280 -> 1280), 66 against 35 (2560 -> 1280): a wave walks 36-72 batches one HBM latency after the other, and 160
//          blocks of four waves are 2.5 waves per CU (profiles/r04_small_lab_conv.txt).  Kept for tests and the A/B.
// MOBI_IGEMM_SMALL_MFLOP overrides the 1 x 1 cap; MOBI_IGEMM_SMALL=0 never; 32: every eligible launch whatever its size (A/B, tests).
constexpr int SMALL_MFLOP_DEFAULT = 1800, SMALL_MFLOP_K320 = 1000, SMALL_CONV_M_DEFAULT = 0;
static int small_tile(const mobi_igemm_params* p) {
  if (tuning().small == 0) return 0;
  // a forced tile geometry (the A/B knobs of the LDS kernels) keeps the launch on them unless this kernel is forced too
  if (tuning().small < 0 && (tuning().wm == 2 || tuning().wm == 4 || tuning().wide >= 0 || tuning().sm == 0)) return 0;
  const bool conv3 = p->kh == 3 && p->kw == 3 && p->pad_h == 1 && p->pad_w == 1;
  if (!(p->kh == 1 && p->kw == 1) && !conv3) return 0;
  if (p->stride != 1 || p->upsample || p->groups != 1 || p->epilogue != MOBI_EPI_NONE || p->k_order != 0 || p->split_k > 1) return 0;
  if (p->out_mode != MOBI_OUT_ROWS && p->out_mode != MOBI_OUT_ROWS_F32) return 0;
  if (p->hout != p->hin || p->wout != p->win) return 0;
  const int ct = p->c0 + p->c1, N = p->n_packed;
  const long long K = (long long)ct * p->kh * p->kw;
  if (ct % 320 || (p->c1 > 0 && p->c0 % 80) || N != p->cout || N % 32) return 0;
  const long long hw = (long long)p->hin * p->win, M = (long long)p->batch * hw;
  const long long ips = p->src_img_stride ? p->src_img_stride / p->c0 : hw;
  const long long cmax = p->c0 > p->c1 ? p->c0 : p->c1;
  if ((((long long)p->batch - 1) * ips + hw) * cmax * 2 >= 0x7fffffffLL || (long long)N * K * 2 >= 0x7fffffffLL) return 0;
  if (tuning().small > 0) return 32;
  if (conv3) {
    const int cap_m = tuning().small_conv_m >= 0 ? tuning().small_conv_m : SMALL_CONV_M_DEFAULT;
    return M <= cap_m ? 32 : 0;
  }
  const int cap = tuning().small_mflop >= 0 ? tuning().small_mflop : (K == 320 ? SMALL_MFLOP_K320 : SMALL_MFLOP_DEFAULT);
  if (2.0 * (double)M * N * (double)K > 1e6 * cap) return 0;
  return 32;
}

// split-K plan: only when the tile grid cannot fill the chip and k is long
static int plan_splits(long long M, int n_packed, int ktot) {
  const int bn = (n_packed % 160) == 0 ? 160 : 128;
  const long long tiles = ((M + 127) / 128) * ((n_packed + bn - 1) / bn);
  const int nk = (ktot + 63) / 64;
  // round 5 (profiles/r05_split_sweep.txt, slabs + reduce launch timed together): 20 k-tiles never pay for a second launch
  // ([2048 | 1024 x 1280 x 1280] 22.2 / 20.7 us unsplit against 23.3 / 21.1 split in two; [8192 x 320 x 1280] 21.9 against 23.3)
  const bool r5 = tuning().split_round4 != 0;   // MOBI_IGEMM_SPLIT_ROUND4=0: round 4's plan (A/B)
  if (tiles >= 384 || nk < (r5 ? 24 : 16)) return 1;
  if (tiles >= 256 && (r5 ? nk <= 40 : nk < 40)) return 1;   // one full wave of blocks, short k ([4096 x 1280 x 2560]: 40.3 us unsplit, 44.1 in two)
  const long long target = tuning().split_target > 0 ? tuning().split_target : 512;      // MOBI_IGEMM_SPLIT_TARGET (sweeps)
  long long s = (target + tiles - 1) / tiles;
  const long long cap = tiles <= 16 ? 16 : 8;               // a handful of tiles (the 4x4 / 8x8 levels): measured best at 16
  if (s > cap) s = cap;
  if (s > nk / 8) s = nk / 8;
  // the reduce launch sums four slabs per round: 5 .. 7 splits buy a second round for little more parallelism ([1024 x 1280 x 2560]
  // 23.6 us at 4 against 29.5 at 5; [2048 x 640 x 2560] 23.0 against 28.3), and so do 8 on 64 tiles or more with at most 96 k-tiles
  // ([1024 x 1280 x 5120] 31.3 against 35.7, [2048 x 640 x 5760] 32.6 against 35.7; fewer tiles still want 8: [512 x 1280 x 5120])
  if (r5) {
    if (s > 4 && s < 8) s = 4;
    if (s == 8 && nk <= 96 && tiles >= 64) s = 4;
  }
  // one round of 128-pixel tiles and a very long k range per split (the 16 x 16 level's 1280 -> 1280 and 2560 -> 1280 3 x 3
  // convolutions: 90 / 180 k-tiles per split at s = 2): twice the splits -- 114.7 against 126.3 us and 198.8 against 225.3
  // (tools/sweep_split.py), -0.08 ms per step; MOBI_IGEMM_SPLIT_LONGK=0 keeps the old plan (A/B)
  if (tiles >= 128 && tiles <= 256 && s >= 2 && nk / s >= 80 && tuning().split_longk != 0) s *= 2;
  return s < 2 ? 1 : (int)s;
}

}  // namespace mobi

#if MOBI_STAMP
extern "C" int mobi_debug_set_stamps(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mobi::g_stamps), &buf, sizeof(void*)) == hipSuccess ? MOBI_OK : MOBI_ERR_LAUNCH;
}
#endif
#if MOBI_STAMP >= 2
extern "C" int mobi_debug_set_phases(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(mobi::g_phase), &buf, sizeof(void*)) == hipSuccess ? MOBI_OK : MOBI_ERR_LAUNCH;
}
#endif

extern "C" int mobi_igemm_plan_splits(const mobi_igemm_params* p) {
  if (!p || p->groups != 1 || p->epilogue != MOBI_EPI_NONE || p->out_mode == MOBI_OUT_TRANSPOSED || p->ln_svec) return 1;
  {
    mobi_igemm_params q = *p;
    q.split_k = 0;
    if (mobi::small_tile(&q)) return 1;
  }
  return mobi::plan_splits((long long)p->batch * p->hout * p->wout, p->n_packed, p->kh * p->kw * (p->c0 + p->c1));
}

extern "C" size_t mobi_igemm_workspace_bytes(const mobi_igemm_params* p, int32_t splits) {
  if (!p || splits <= 1) return 0;
  return (size_t)splits * p->batch * p->hout * p->wout * p->n_packed * sizeof(float);
}

// arrival counters of a split launch that finishes itself: one int per output tile of the SMALLEST tile geometry (128 pixels
// x 128 channels), whichever kernel runs it
extern "C" size_t mobi_igemm_sync_bytes(const mobi_igemm_params* p, int32_t splits) {
  if (!p || splits <= 1) return 0;
  const size_t m = (size_t)p->batch * p->hout * p->wout;
  return ((m + 127) / 128) * (((size_t)p->n_packed + 127) / 128) * sizeof(int32_t);
}

// argument checks + launch plan (tile height, addressing path, main-loop variant, split-K ranges, epilogue kind)
static int igemm_prepare(const mobi_igemm_params* p, mobi::IgemmArgs& a) {
  using namespace mobi;
  if (!p || !p->src0 || !p->weight || !p->out) return MOBI_ERR_ARG;
  if (p->dtype != MOBI_F16 && p->dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (p->c0 <= 0 || (p->c0 & 31) || p->c1 < 0 || (p->c1 & 31) || (p->c1 > 0 && !p->src1)) return MOBI_ERR_UNSUPPORTED;
  if (p->batch <= 0 || p->hin <= 0 || p->win <= 0 || p->hout <= 0 || p->wout <= 0) return MOBI_ERR_ARG;
  if (p->kh <= 0 || p->kw <= 0 || p->stride <= 0 || p->groups <= 0 || p->batch % p->groups) return MOBI_ERR_ARG;
  if (p->cout <= 0 || p->n_packed <= 0) return MOBI_ERR_ARG;
  if (p->upsample != 0 && p->upsample != 1) return MOBI_ERR_ARG;
  const bool geglu = p->epilogue == MOBI_EPI_GEGLU;
  if (p->epilogue != MOBI_EPI_NONE && !geglu) return MOBI_ERR_ARG;
  if (p->out_mode < 0 || p->out_mode > 2) return MOBI_ERR_ARG;
  if (p->out_mode != MOBI_OUT_TRANSPOSED && (p->cout & 7)) return MOBI_ERR_UNSUPPORTED;
  if (geglu && (p->out_mode != MOBI_OUT_ROWS || p->rowvec || p->residual)) return MOBI_ERR_UNSUPPORTED;
  if (!geglu && p->n_packed != p->cout) return MOBI_ERR_ARG;
  if (geglu && p->n_packed != 2 * p->cout) return MOBI_ERR_ARG;
  if (p->out_mode == MOBI_OUT_TRANSPOSED && (p->rowvec || p->residual)) return MOBI_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(p->bias) | reinterpret_cast<uintptr_t>(p->rowvec)) & 15) return MOBI_ERR_ALIGN;
  if (p->rowvec && (p->rowvec_stride & 3)) return MOBI_ERR_ALIGN;
  if ((reinterpret_cast<uintptr_t>(p->src0) | reinterpret_cast<uintptr_t>(p->src1) |
       reinterpret_cast<uintptr_t>(p->weight) | reinterpret_cast<uintptr_t>(p->out) |
       reinterpret_cast<uintptr_t>(p->residual)) & 15) return MOBI_ERR_ALIGN;

  // LayerNorm folded into the launch (ln_fold_acc): a 1 x 1 product of ONE source whose block sweeps all of k
  const bool lnf = p->ln_svec != nullptr;
  if (lnf) {
    if (p->kh != 1 || p->kw != 1 || p->c1 || p->stride != 1 || p->upsample || p->groups != 1 || p->split_k > 1 || p->rowvec ||
        p->residual || p->out_mode == MOBI_OUT_TRANSPOSED || p->scale != 1.0f || p->k_order != 0 || p->hout != p->hin || p->wout != p->win)
      return MOBI_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(p->ln_svec) & 15) return MOBI_ERR_ALIGN;
  }
  a.ln_svec = p->ln_svec; a.ln_eps = p->ln_eps;
  a.src0 = p->src0; a.src1 = p->src1;
  a.c0 = p->c0; a.c1 = p->c1; a.C = p->c0 + p->c1;
  a.hin = p->hin; a.win = p->win; a.up = p->upsample;
  a.hout = p->hout; a.wout = p->wout; a.hw_out = p->hout * p->wout;
  a.kh = p->kh; a.kw = p->kw; a.stride = p->stride; a.pad_h = p->pad_h; a.pad_w = p->pad_w;
  const long long hw_in = (long long)p->hin * p->win;
  if (p->src_img_stride && (p->c1 || p->src_img_stride % p->c0)) return MOBI_ERR_UNSUPPORTED;
  const long long ips = p->src_img_stride ? p->src_img_stride / p->c0 : hw_in;
  if (ips * p->batch > 0x7fffffffLL) return MOBI_ERR_UNSUPPORTED;
  a.img_pix_stride = (int)ips;
  a.weight = p->weight; a.w_group_stride = p->groups > 1 ? p->w_group_stride : 0;
  a.n_packed = p->n_packed; a.cout = p->cout;
  a.ktot = p->kh * p->kw * a.C;
  a.nk = (a.ktot + 63) / 64;
  a.imgs_per_group = p->batch / p->groups;
  a.M = a.imgs_per_group * a.hw_out;
  a.bias = p->bias; a.rowvec = p->rowvec; a.residual = p->residual;
  a.rowvec_stride = p->rowvec_stride ? p->rowvec_stride : p->cout;
  a.res_img_stride = p->res_img_stride ? p->res_img_stride : (long long)a.hw_out * p->cout;
  a.out = p->out;
  a.out_img_stride = p->out_img_stride ? p->out_img_stride : (long long)a.hw_out * p->cout;
  a.out_mode = p->out_mode; a.epilogue = p->epilogue; a.scale = p->scale;
  const int bn = (p->n_packed % 160) == 0 ? 160 : 128;
  // 256-pixel tiles (8 waves, one block per CU) when that still gives every CU a block; MOBI_IGEMM_WM overrides
  {
    const long long tiles256 = ((a.M + 255) / 256) * (long long)((p->n_packed + bn - 1) / bn);
    a.wm = tiles256 >= 256 ? 4 : 2;
    // split-K launches with long k ranges: 256-pixel tiles (ping-pong kernel) once tiles x splits fill the chip
    if (p->split_k > 1 && a.M % 256 == 0 && tiles256 * p->split_k >= 256 && a.nk / p->split_k >= 16) a.wm = 4;
    if (tuning().pp_split == 0 && p->split_k > 1 && tiles256 < 256) a.wm = 2;
    if (tuning().wm == 2 || tuning().wm == 4) a.wm = tuning().wm;
  }
  a.tiles_m = (a.M + 64 * a.wm - 1) / (64 * a.wm);
  bool ring_ok = false;
  {
    // FAST path: k-tiles never straddle a tap or a source, and every byte offset fits 31 bits
    const long long ext0 = (((long long)p->batch - 1) * ips + hw_in) * p->c0 * 2;
    const long long ext1 = p->c1 ? (((long long)p->batch - 1) * ips + hw_in) * p->c1 * 2 : 0;
    const long long wext = (long long)p->n_packed * a.ktot * 2;
    a.fast = (a.C % 64 == 0) && (p->c1 == 0 || p->c0 % 64 == 0) && ext0 < 0x7fffffffLL && ext1 < 0x7fffffffLL &&
             wext < 0x7fffffffLL;
    if (tuning().fast == 0) a.fast = 0;
    a.k_order = p->k_order;
    if (p->k_order != 0 && p->k_order != 1) return MOBI_ERR_ARG;
#ifndef MOBI_DEV
    if (p->k_order == 1) return MOBI_ERR_UNSUPPORTED;      // chunk-major k: development build only (measured slower)
#endif
    if (p->k_order == 1 && !a.fast) return MOBI_ERR_UNSUPPORTED;      // the generic gather walks k tap-major only
    a.glds = 1;
    if (tuning().glds == 0) a.glds = 0;
    // 128-pixel tiles: the LDS-DMA ring kernel whenever the 32-bit buffer offsets reach every operand byte and k runs
    // tap-major (k-steps are 32 deep there: channel counts are multiples of 32 by the ABI's own rule)
    ring_ok = p->k_order == 0 && ext0 < 0x7fffffffLL && ext1 < 0x7fffffffLL && wext < 0x7fffffffLL;
    a.sm = a.wm == 2 && ring_ok;
    if (tuning().sm == 0) a.sm = 0;
    a.src0_bytes = (int)(ext0 < 0x7fffffffLL ? ext0 : 0);
    a.src1_bytes = (int)(ext1 < 0x7fffffffLL ? ext1 : 0);
    a.w_bytes = (int)(wext < 0x7fffffffLL ? wext : 0);
  }
  a.tiles_n = (p->n_packed + bn - 1) / bn;
  if ((long long)a.tiles_m * a.tiles_n > 0x7fffffffLL) return MOBI_ERR_UNSUPPORTED;
  a.splits = 1; a.nk_per = a.nk; a.split_ws = nullptr;
  if (p->split_k > 1) {
    if (!p->ws || p->groups != 1 || geglu || p->out_mode == MOBI_OUT_TRANSPOSED || p->split_k > 64) return MOBI_ERR_UNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(p->ws) & 15) return MOBI_ERR_ALIGN;
    a.nk_per = (a.nk + p->split_k - 1) / p->split_k;
    a.splits = (a.nk + a.nk_per - 1) / a.nk_per;            // no empty k ranges: every slab that is summed is written
    a.split_ws = reinterpret_cast<float*>(p->ws);
  }
  if (p->defer_finish != 0 && p->defer_finish != 1) return MOBI_ERR_ARG;
  // deferred finish: the slabs are the launch's result -- T rows only (what a consumer reconstructs), never with `sync`
  if (p->defer_finish && (!a.split_ws || p->sync || p->out_mode != MOBI_OUT_ROWS)) return MOBI_ERR_UNSUPPORTED;
  // (taps-innermost k order only: it changes tap every k-tile; measured 1.5-3 % slower than the window re-derivation
  //  on tap-major k, where a tap lasts C/64 k-tiles -- tools/sweep_korder.py)
  a.lin_window = p->k_order == 1 && p->upsample == 0 && p->kh * p->kw <= 16;
  if (tuning().lin == 0) a.lin_window = 0;
  // register epilogue of the direct-to-LDS kernel: every tile full, row-major T output, no per-image vector
  // (a per-image vector takes the bias registers: never both, and every wave's 64 pixels inside one image)
  a.epi_direct = a.wm == 4 && a.fast && a.glds && p->out_mode == MOBI_OUT_ROWS && !a.split_ws &&
                 (!p->rowvec || (!p->bias && a.hw_out % 64 == 0)) && a.M % 256 == 0 && p->n_packed % bn == 0 &&
                 a.nk_per >= 3;
  if (tuning().epi_direct == 0) a.epi_direct = 0;
  // ping-pong kernel: window pixels linear in the tap (no upsampling, <= 16 taps), one k range
  auto log2_exact = [](int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; };
  a.hw_shift = log2_exact(a.hw_out); a.w_shift = log2_exact(a.wout);
  a.pp = a.epi_direct && a.splits == 1 && p->scale == 1.0f && a.hw_shift >= 6 && a.w_shift >= 0 &&
         a.hout < 32768 && a.wout < 32768;
  // split-K on the ping-pong kernel (fp32 slabs from the registers, finished by the reduce launch): long k, full tiles
  if (a.split_ws && a.wm == 4 && a.fast && a.glds && p->out_mode != MOBI_OUT_TRANSPOSED && a.M % 256 == 0 &&
      p->n_packed % bn == 0 && a.nk_per >= 3 && p->scale == 1.0f && a.hw_shift >= 6 && a.w_shift >= 0 &&
      a.hout < 32768 && a.wout < 32768)
    a.pp = 1;
  if (tuning().pp == 0) a.pp = 0;
  // 256 x 320 (256) tiles on the ring kernel
  // (measured, tools/ab_wide.py: 3x3 convolutions at 64x64 x 16: 1071-1245 vs 813-976 TFLOP/s on the ping-pong kernel;
  //  it needs about one block per CU: with 128 tiles -- the 32x32 level at N = 640 -- half the chip idles and it loses;
  //  the 128 x 320 tile (twice the blocks at the ping-pong tile's bytes per FLOP) measured 829-965 TFLOP/s there against
  //  982-1085 on the ping-pong kernel: kept for A/B (MOBI_IGEMM_WIDE128=1), not routed)
  a.wide = 0;
  if ((a.wm == 4 || tuning().wide > 0) && ring_ok && tuning().wide != 0) {
    const int bnw = (p->n_packed % 160) == 0 ? 320 : 256;
    const long long tn = (p->n_packed + bnw - 1) / bnw;
    auto fills = [&](long long blocks) {
      const long long rounds = (blocks + 255) / 256;
      return blocks >= 190 && blocks * 5 >= rounds * 256 * 4 - 64 * (rounds == 1);   // rounds about 80 % full on average
                                                                       // (192 blocks, 1280 -> 3840 at 16 x 16 x 16: 43.4 vs 51.7 us)
    };
    const long long t256 = ((a.M + 255) / 256) * tn * a.splits, t128 = ((a.M + 127) / 128) * tn * a.splits;
    int pick = tuning().wide > 0 ? tuning().wide : (fills(t256) ? 2 : (fills(t128) && tuning().wide128 == 1 ? 1 : 0));
    // a channel tile that is half padding or more (128 output channels in a 256-wide tile: the VAEs' 128-channel levels)
    // loses to the ping-pong kernel's 256 x 128 tile: 878 vs 1160 us on 128 -> 128 3x3 at 512 x 512 x 8 (tools/kbench.py conv)
    if (tuning().wide <= 0 && pick == 2 && p->n_packed * 2 <= bnw && a.pp) pick = 0;
#ifndef MOBI_DEV
    // launches of the 256-pixel geometry that the ping-pong kernel's register epilogue does not take (transposed / fp32
    // output, ragged tiles, bias AND per-image vector, per-image weights): the ring kernel's LDS-staged epilogue does
    // (a source of fewer than 64 channels into a wide layer -- the UNet's 9-channel input, padded to 32: a tap per k-step --
    //  stays on the register-staged kernel: 38 against 104 us on 9 -> 320 at 64 x 64 x 16)
    if (!pick && a.wm == 4 && !a.pp && (a.C >= 64 || p->out_mode != MOBI_OUT_ROWS || p->n_packed < 256)) pick = 2;
    if (lnf) pick = a.wm == 4 ? 2 : 0;                      // the LayerNorm fold lives in the 256 x 320 / 128 x 160 ring tiles
#endif
    if (pick) {
      a.wide = pick;
      a.wm = pick == 2 ? 4 : 2;
      a.tiles_m = (a.M + 64 * a.wm - 1) / (64 * a.wm);
      a.tiles_n = (int)tn;
      a.sm = 0;
    }
  }
  // order of the work list (tile_of): by channel tile when a 1 x 1 launch's weights outweigh the activations it reads --
  // every XCD's L2 then fetches its slice of the weights and all of the (smaller) activations instead of all of the weights
  // and its slice of the activations.  Measured (tools/ab_nmajor.sh): the 16 x 16 level's GEGLU projection (26 MB of weights,
  // 10.5 MB of activations) 219-223 against 233-237 us; 3 x 3 convolutions LOSE 3-6 % (143.7 vs 139.5 us at 1280 -> 1280,
  // 16 x 16: nine taps re-read the whole activation tensor through every L2), so they keep the pixel-major order.
  {
    const long long wbytes = (long long)p->n_packed * a.ktot * 2;
    const long long abytes = (long long)p->batch * p->hin * p->win * a.C * 2;
    a.n_major = p->groups == 1 && a.tiles_n >= 2 && p->kh * p->kw == 1 && wbytes > abytes && tuning().n_major != 0;
    if (tuning().n_major == 1 && p->groups == 1) a.n_major = 1;
    if (!a.wide && a.wm == 4 && a.fast && a.glds && a.pp) a.n_major = 0;      // the ping-pong kernel's launches
  }
  // 64-deep steps for the 128-pixel geometry: channel runs of 64 (a k-step never straddles a tap or a source), >= 3 steps
  // (one block per CU: only grids of at most one round of blocks -- with more, two co-resident blocks of the 32-deep ring win:
  //  36.2 vs 45.3 us on 320 -> 320 3x3 at 32 x 32 x 8 split 4, tools/ab_sm64.py)
  a.sm64 = a.sm && a.C % 64 == 0 && (p->c1 == 0 || p->c0 % 64 == 0) && a.nk_per >= 1 &&
           (long long)a.tiles_m * a.tiles_n * a.splits * p->groups <= (long long)compute_units() && tuning().sm64 != 0;
  if (tuning().sm64 == 1 && a.sm && a.C % 64 == 0 && (p->c1 == 0 || p->c0 % 64 == 0)) a.sm64 = 1;
  a.w_tiled = (p->weight_tiled && !a.sm64 && (a.sm || a.wide) &&
               (p->groups == 1 || p->w_group_stride == (long long)p->n_packed * a.ktot) && p->k_order == 0 && p->n_packed % 16 == 0 &&
               a.ktot % 32 == 0 && !(reinterpret_cast<uintptr_t>(p->weight_tiled) & 15) && tuning().w_tiled != 0)
                  ? p->weight_tiled : nullptr;
  // 128 x 160 ring tiles: register epilogue / slab stores when every tile is full (the staged epilogue keeps ragged tiles,
  // transposed / fp32 output, bias AND per-image vector)
  // (groups: every group's rows are whole tiles -- a.M is the rows of ONE group -- and the epilogue addresses rows, residual and
  //  per-image vector through the group's first image; the bias is shared by the groups)
  a.sm_direct = (a.sm || a.wide == 1) && (p->out_mode == MOBI_OUT_ROWS || a.split_ws) && p->out_mode != MOBI_OUT_TRANSPOSED &&
                a.M % 128 == 0 && p->n_packed % (a.wide == 1 ? 2 * bn : bn) == 0 &&
                (a.split_ws || !p->rowvec || (!p->bias && a.hw_out % 64 == 0)) && tuning().sm_direct != 0;
  {
    const int bnw = (p->n_packed % 160) == 0 ? 320 : 256;
    a.ring_direct = a.wide == 2 && p->out_mode == MOBI_OUT_ROWS && !a.split_ws && p->groups == 1 &&
                    (!p->rowvec || (!p->bias && a.hw_out % 128 == 0)) && a.M % 256 == 0 && p->n_packed % bnw == 0 &&
                    p->scale == 1.0f && !p->residual && tuning().ring_direct != 0;
    // (with a residual the staged epilogue is as fast or a microsecond faster: 28.1 vs 29.2 us on 320 -> 320 1x1 at 64 x 64 x 16,
    //  98.9 vs 100.2 on the 3x3; MOBI_IGEMM_RING_DIRECT=1 forces the register epilogue there too)
    if (tuning().ring_direct == 1 && p->residual && a.wide == 2 && p->out_mode == MOBI_OUT_ROWS && !a.split_ws && p->groups == 1 &&
        (!p->rowvec || (!p->bias && a.hw_out % 128 == 0)) && a.M % 256 == 0 && p->n_packed % bnw == 0 && p->scale == 1.0f)
      a.ring_direct = 1;
  }
  a.small = lnf ? 0 : small_tile(p);
  if (lnf && !(a.wide || a.sm)) return MOBI_ERR_UNSUPPORTED;  // (operands beyond the ring kernels' 2 GB: LayerNorm as a launch)
  // split-K finished inside the launch: the LDS-DMA kernels (ring tiles of every geometry; the ping-pong kernel when every
  // block owns exactly one output tile -- its epilogue is deferred into the next tile's loop otherwise); the register-staged
  // fallback keeps the reduce launch.  MOBI_IGEMM_FUSED_SPLIT=0: never (A/B)
  // At most FOUR splits (splitk_finish_tile).  MOBI_IGEMM_FUSED_SPLIT: 0 never; 1 device-coherent slab traffic; 2 (default) the
  // same-XCD form where the grid's x extent (the tile count: one tile per block) is a multiple of 8, the reduce launch elsewhere
  a.sync = nullptr; a.sync_mode = 0;
  if (a.split_ws && p->sync && !a.small && tuning().fused_split != 0 && a.splits <= 4 &&
      (long long)a.splits * a.M * a.n_packed * 4 < 0x7fffffffLL) {
    if (reinterpret_cast<uintptr_t>(p->sync) & 3) return MOBI_ERR_ALIGN;
    const long long tiles = (long long)a.tiles_m * a.tiles_n;
    const bool pp_launch = !a.wide && a.wm == 4 && a.fast && a.glds && a.pp;
    const bool ring_launch = a.wide || a.sm;
    const int mode = tuning().fused_split == 1 ? 1 : 2;
    if ((ring_launch || (pp_launch && tiles <= (long long)compute_units())) && (mode == 1 || tiles % 8 == 0)) {
      a.sync = reinterpret_cast<int*>(p->sync);
      a.sync_mode = mode;
    }
  }
  return MOBI_OK;
}

static int launch_small(const mobi_igemm_params* p, const mobi::IgemmArgs& a, hipStream_t st) {
  mobi::SmallGemmArgs s;
  s.src0 = a.src0; s.src1 = a.src1; s.c0 = a.c0; s.c1 = a.c1;
  s.taps = a.kh * a.kw; s.hin = a.hin; s.win = a.win;
  s.hw = a.hw_out; s.img_pix_stride = a.img_pix_stride;
  s.weight = a.weight;
  s.M = a.M; s.N = a.n_packed; s.K = a.ktot;
  s.bias = a.bias; s.rowvec = a.rowvec; s.rowvec_stride = a.rowvec_stride;
  s.residual = a.residual; s.res_img_stride = a.res_img_stride;
  s.out = a.out; s.out_img_stride = a.out_img_stride; s.out_f32 = a.out_mode == MOBI_OUT_ROWS_F32;
  s.scale = a.scale;
  s.tiles_m = (a.M + a.small - 1) / a.small; s.tiles_n = a.n_packed / a.small;
  const long long wbytes = (long long)a.n_packed * a.ktot * 2, abytes = (long long)a.M * a.ktot * 2;
  s.n_major = wbytes > abytes;
  return mobi::launch_small_gemm(s, p->dtype, st);
}

extern "C" int mobi_igemm(const mobi_igemm_params* p, void* stream) {
  using namespace mobi;
  IgemmArgs a;
  const int rc = igemm_prepare(p, a);
  if (rc != MOBI_OK) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (a.small) return launch_small(p, a, st);
  return p->dtype == MOBI_F16 ? launch_igemm<f16_t>(p, a, p->groups, st) : launch_igemm<bf16_t>(p, a, p->groups, st);
}

extern "C" int32_t mobi_igemm_slab_count(const mobi_igemm_params* p) {
  mobi::IgemmArgs a;
  const int rc = igemm_prepare(p, a);
  if (rc != MOBI_OK) return rc;
  return a.small ? 1 : a.splits;
}

extern "C" int mobi_igemm_finish(const mobi_igemm_params* p, void* stream) {
  using namespace mobi;
  IgemmArgs a;
  const int rc = igemm_prepare(p, a);
  if (rc != MOBI_OK) return rc;
  if (!a.split_ws || a.small || a.sync) return MOBI_ERR_ARG;       // nothing was deferred
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  return p->dtype == MOBI_F16 ? launch_splitk_reduce<f16_t>(a, st) : launch_splitk_reduce<bf16_t>(a, st);
}

extern "C" int mobi_igemm_kernel_variant(const mobi_igemm_params* p) {
  mobi::IgemmArgs a;
  const int rc = igemm_prepare(p, a);
  if (rc != MOBI_OK) return rc;
  if (a.small) return MOBI_IGEMM_SMALL;
  if (a.wide) return a.wide == 2 ? MOBI_IGEMM_RING_256 : MOBI_IGEMM_RING_128W;
  if (a.wm == 4 && a.fast && a.glds) return a.pp ? MOBI_IGEMM_PINGPONG : MOBI_IGEMM_DIRECT_LDS;
  if (a.wm == 4) return MOBI_IGEMM_STAGED_256;
  return a.sm ? MOBI_IGEMM_RING_128 : MOBI_IGEMM_STAGED_128;
}
