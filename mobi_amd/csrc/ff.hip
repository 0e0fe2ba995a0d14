// Fused GEGLU feed-forward for gfx950 (FeedForward / GEGLU of the transformer blocks, attention.py:38-65 of the reference):
//
//     out = (x W1v^T + b1v) * gelu_erf(x W1g^T + b1g) . W2^T + b2 (+ residual)
//
// as ONE kernel: the hidden activation [rows][4C] (168 MB written and read back per launch at 64 x 64 x 16, C = 320) never
// leaves the chip.  Structure = the attention kernel's with the softmax replaced by the gate: per wave 32 token rows,
//     H^T[unit][row]   = W1[unit][:] . X[row][:]          (A = W1 fragments from LDS, B = X held in registers, K = C)
//     O^T[ch][row]    += W2[ch][unit] . H^T[unit][row]    (A = W2 fragments from LDS, B = the H^T accumulator itself,
//                                                          converted to T in registers, in the accumulator's unit order)
// over chunks of 32 hidden units; the output accumulators O^T (C x 32 rows: C/2 registers) stay in registers.  MFMA 32x32x16,
// fp32 accumulate; H is rounded to the storage type between the two products exactly as the two-launch path stores it.
//
// Block = 4 waves (128 rows), ONE wave per SIMD (the kernel needs ~330 registers: X 80, O 160, H 32 + 8, fragments).
// Weights arrive as ready-made LDS images ("chunk images", packed once by mobi_ff_geglu_pack_index / ops.pack_ff_geglu):
// fragment f of a chunk is 1 KiB = 64 lanes x 16 B in the lane order of the MFMA A operand, so a fragment read is one
// conflict-free ds_read_b128 at a compile-time offset and a chunk is filled by 61 LDS-DMA requests of 1 KiB
// (buffer_load ... lds, no VGPR staging).  W1 images go through a ring of 2 slots, W2 images (+ the chunk's b1 values)
// through a ring of 3: step c multiplies W1(c) and, interleaved with the gate arithmetic of chunk c, W2(c-1); chunk c+1 is in
// flight meanwhile.  One barrier per chunk.
#include "common.h"

#ifndef MOBI_FF_DBG
#define MOBI_FF_DBG 0    // diagnosis only (wrong results): bit 0 = no GELU arithmetic, bit 1 = no LDS-DMA after the first two chunks,
#endif                  // bit 2 = no second product, bit 3 = no barrier, bit 4 = two chunks only
#ifndef MOBI_FF_D
#define MOBI_FF_D 6      // depth of the fragment queue
#endif

namespace mobi {

struct FfArgs {
  const void* x;
  const void* w1p;          // [chunks][2 * KS fragments][64 lanes][8 T]
  const void* w2p;          // [chunks][2 * MT fragments + 1][64 lanes][8 T]; the last KiB holds b1: 32 value + 32 gate floats
  const float* b2;
  const void* residual;
  void* out;
  long long rows;
  int chunks;
  const float* ln_gamma;    // LayerNorm over the C channels applied to x first (fp32 [C] each), or NULL
  const float* ln_beta;
  float ln_eps;
};

template <typename T, int C>
__global__ __launch_bounds__(256, 1) void ff_geglu_kernel(const FfArgs a) {
  typedef typename Vec8<T>::type frag_t;
  constexpr int KS = C / 16, MT = C / 32;
  constexpr int W1_SLOT = 2 * KS * 1024, W2_SLOT = (2 * MT + 1) * 1024;
  constexpr int NP1 = 2 * KS, NP = NP1 + 2 * MT + 1;              // DMA pieces of a chunk
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * W1_SLOT + 3 * W2_SLOT];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  unsigned char* ldsA = lds;
  unsigned char* ldsB = lds + 2 * W1_SLOT;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ql = lane & 31, half = lane >> 5;
  const long long row = (long long)blockIdx.x * 128 + wave * 32 + ql;
  const long long rowc = row < a.rows ? row : a.rows - 1;

  const T* __restrict__ xp = reinterpret_cast<const T*>(a.x) + rowc * C;
  frag_t xf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) xf[ks] = __builtin_bit_cast(frag_t, ld16(xp + ks * 16 + half * 8));
  if (a.ln_gamma) {
    // LayerNorm of the token rows in the registers they are multiplied from (norm3 of the transformer block: the launch
    // that would write LN(x) and the second read of it go away; `residual` is then x itself).  A row's channels lie in
    // lanes ql and ql + 32: mean, then the variance about the mean (layernorm_kernel's arithmetic), the result rounded
    // to the storage type as that kernel's output would be.
    float s1 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) s1 += (float)xf[ks][j];
    s1 += __shfl_xor(s1, 32, 64);
    const float mean = s1 * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)xf[ks][j] - mean; q += d * d; }
    q += __shfl_xor(q, 32, 64);
    const float rstd = rsqrtf(q * (1.0f / C) + a.ln_eps);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.ln_gamma + ks * 16 + half * 8);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(a.ln_gamma + ks * 16 + half * 8 + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(a.ln_beta + ks * 16 + half * 8);
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(a.ln_beta + ks * 16 + half * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        xf[ks][j] = (T)(((float)xf[ks][j] - mean) * rstd * g0[j] + h0[j]);
        xf[ks][4 + j] = (T)(((float)xf[ks][4 + j] - mean) * rstd * g1[j] + h1[j]);
      }
    }
  }

  const int nch = (MOBI_FF_DBG & 16) ? 2 : a.chunks;      // (bit 4: two chunks only -- what the prologue and the epilogue cost)
  const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w1p), 0, nch * W1_SLOT, 0x00020000);
  const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w2p), 0, nch * W2_SLOT, 0x00020000);
  const unsigned lane16 = (unsigned)lane * 16u;
  // the chunk's 61 pieces are dealt to the four waves round-robin: piece p = wave + 4 i (i < 10: W1 image, 10 <= i < 15: W2
  // image, i = 15: the b1 piece, wave 0 only); request i of a step is issued behind MFMA pair i of its first product
  constexpr int NREQ = (NP + 3) / 4;
  static_assert(4 * (NP1 / 4) == NP1 && NP - NP1 == 4 * (NREQ - 1 - NP1 / 4) + 1, "piece deal assumes C = 320");
  auto request_piece = [&](int c, int i) {
    const int p = wave + 4 * i;
    if (i < NP1 / 4) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r1, (lds_ptr_t)(ldsA + (c & 1) * W1_SLOT + p * 1024), 16, lane16,
                                               c * W1_SLOT + p * 1024, 0, 0);
    } else if (i < NREQ - 1 || wave == 0) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (lds_ptr_t)(ldsB + (c % 3) * W2_SLOT + (p - NP1) * 1024), 16, lane16,
                                               c * W2_SLOT + (p - NP1) * 1024, 0, 0);
    }
  };
#define MOBI_FF_BARRIER()                                           \
  do {                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                              \
    if (!(MOBI_FF_DBG & 8)) __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);                              \
  } while (0)

  f32x16 o[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[m][r] = 0.f;

  // gate arithmetic of a chunk, two units at a time
  auto gate2 = [&](const f32x16& hv, const f32x16& hg, int i) {
    const float h0 = hv[2 * i] * gelu_erf_f(hg[2 * i]), h1 = hv[2 * i + 1] * gelu_erf_f(hg[2 * i + 1]);
    typedef T T2 __attribute__((ext_vector_type(2)));
    T2 v;
    v[0] = (T)h0;
    v[1] = (T)h1;
    return __builtin_bit_cast(unsigned, v);
  };

  // One step = ONE unrolled MFMA stream with a rolling queue of operand fragments: fragment i + D is requested from LDS
  // right behind MFMA i (left to the compiler every MFMA waited for its own read: an LDS round trip per MFMA).  Step c:
  //   MFMAs 0 .. 2 KS - 1             first product of chunk c+1 (value / gate accumulators alternate, started from b1), with
  //                                   the gate arithmetic of chunk c (~6 vector instructions per gap) dealt into its gaps
  //   MFMAs 2 KS .. 2 KS + 2 MT - 1   second product of chunk c
  // and the LDS-DMA requests of chunk c+2, one behind every fourth MFMA or so.
  constexpr int D = MOBI_FF_D;
  auto step = [&](auto has_next, int c, bool request, const f32x16& hv, const f32x16& hg, f32x16& nv, f32x16& ng) {
    constexpr bool NEXT = decltype(has_next)::value;
    constexpr int N1 = NEXT ? 2 * KS : 0, N2 = 2 * MT, NF = N1 + N2;
    const unsigned char* a1 = ldsA + ((c + 1) & 1) * W1_SLOT + lane16;            // W1 image of chunk c+1
    const unsigned char* b2p = ldsB + (c % 3) * W2_SLOT + lane16;                 // W2 image of chunk c
    auto frag_addr = [&](int i) -> const unsigned char* {
      if (i < N1) return a1 + ((i & 1) * KS + (i >> 1)) * 1024;                   // value ks, gate ks, value ks+1, ...
      return b2p + (i - N1) * 1024;
    };
    frag_t fq[D];
#pragma unroll
    for (int i = 0; i < D && i < NF; ++i) fq[i] = __builtin_bit_cast(frag_t, ld16(frag_addr(i)));
    if (NEXT) {
      // b1: register r of lane-half h holds unit (r & 3) + 8 (r >> 2) + 4 h: four 16-byte broadcast reads per accumulator
      const float* bv = reinterpret_cast<const float*>(ldsB + ((c + 1) % 3) * W2_SLOT + 2 * MT * 1024);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(bv + 8 * q + 4 * half);
        const f32x4 g = *reinterpret_cast<const f32x4*>(bv + 32 + 8 * q + 4 * half);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nv[4 * q + j] = v[j];
          ng[4 * q + j] = g[j];
        }
      }
    }
    unsigned pw[8];
    float he[2] = {0.f, 0.f};
    if (!NEXT) {
#pragma unroll
      for (int u = 0; u < 8; ++u) pw[u] = gate2(hv, hg, u);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const frag_t f = fq[i % D];
      if (i < N1) {
        const int ks = i >> 1;
        if (i & 1) ng = mfma32(f, xf[ks], ng); else nv = mfma32(f, xf[ks], nv);
        // the gate of chunk c: element e (one erf-GELU, ~20 vector instructions) behind MFMA (5 e + 4) / 2, i.e. one per
        // 2.5 gaps; a pair is packed when its second element is done
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if ((5 * e + 4) / 2 == i) {
            he[e & 1] = (MOBI_FF_DBG & 1) ? hv[e] * hg[e] : hv[e] * gelu_erf_f(hg[e]);
            if (e & 1) {
              typedef T T2 __attribute__((ext_vector_type(2)));
              T2 v2;
              v2[0] = (T)he[0];
              v2[1] = (T)he[1];
              pw[e >> 1] = __builtin_bit_cast(unsigned, v2);
            }
          }
        }
      } else {
        const int k = i - N1, m = k >> 1;
        const u32x4 pu = (k & 1) ? u32x4{pw[4], pw[5], pw[6], pw[7]} : u32x4{pw[0], pw[1], pw[2], pw[3]};
        if (MOBI_FF_DBG & 4) o[m][k & 1] += (float)f[0] * (float)pu[0]; else o[m] = mfma32(f, __builtin_bit_cast(frag_t, pu), o[m]);
      }
      if (i + D < NF) fq[i % D] = __builtin_bit_cast(frag_t, ld16(frag_addr(i + D)));
      if (NEXT && i >= 1) {
        const int r = ((i - 1) * NREQ + NF - 1) / NF;                              // request r sits behind MFMA r NF / NREQ + 1:
        if (r < NREQ && (r * NF) / NREQ + 1 == i && request && !(MOBI_FF_DBG & 2)) request_piece(c + 2, r);   // all NREQ, evenly over the step
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

#pragma unroll
  for (int i = 0; i < NREQ; ++i) request_piece(0, i);
  if (nch > 1) {
#pragma unroll
    for (int i = 0; i < NREQ; ++i) request_piece(1, i);
  }
  MOBI_FF_BARRIER();
  f32x16 va, ga, vb, gb;
  {
    // first product of chunk 0 (nothing to overlap with yet)
    const unsigned char* a1 = ldsA + lane16;
    const float* bv = reinterpret_cast<const float*>(ldsB + 2 * MT * 1024);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(bv + 8 * q + 4 * half);
      const f32x4 g = *reinterpret_cast<const f32x4*>(bv + 32 + 8 * q + 4 * half);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        va[4 * q + j] = v[j];
        ga[4 * q + j] = g[j];
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      va = mfma32(__builtin_bit_cast(frag_t, ld16(a1 + ks * 1024)), xf[ks], va);
      ga = mfma32(__builtin_bit_cast(frag_t, ld16(a1 + (KS + ks) * 1024)), xf[ks], ga);
    }
  }
  MOBI_FF_BARRIER();          // step 0 refills the W1 slot of chunk 0: every wave must be done reading it
  int c = 0;
  for (; c + 2 < nch; c += 2) {
    step(std::true_type{}, c, true, va, ga, vb, gb);
    MOBI_FF_BARRIER();
    step(std::true_type{}, c + 1, c + 3 < nch, vb, gb, va, ga);
    MOBI_FF_BARRIER();
  }
  // here c = nch - 2 (even chunk count) or nch - 1, the accumulators of chunk c in (va, ga)
  if (c + 1 < nch) {
    step(std::true_type{}, c, false, va, ga, vb, gb);
    MOBI_FF_BARRIER();
    step(std::false_type{}, c + 1, false, vb, gb, va, ga);
  } else {
    step(std::false_type{}, c, false, va, ga, vb, gb);
  }

  // ---- epilogue: lane holds O^T[ch][row] for ch = 32 m + 8 g + 4 half + (0..3).  Stored from these registers a wave
  // instruction would write 64 pieces of 8 bytes at a 640-byte stride (and read the residual the same way): the wave's
  // 32 x C tile goes through LDS instead (fp32, + b2; half the channels at a time, 656-byte rows), comes back as whole
  // 16-byte output pieces in row order, takes the residual (one rounding, as in the two-launch path) and leaves as
  // 320-byte runs.  The ring is free by now; every wave only touches its own region, so no block barrier inside.
  MOBI_FF_BARRIER();
#undef MOBI_FF_BARRIER
  {
    constexpr int HC = C / 2, SROW = HC + 4;                       // channels per pass, floats per staged row
    float* stg = reinterpret_cast<float*>(lds) + wave * (32 * SROW);
    const long long row0 = (long long)blockIdx.x * 128 + wave * 32;
    T* outp = reinterpret_cast<T*>(a.out);
    const T* resp = reinterpret_cast<const T*>(a.residual);
#pragma unroll
    for (int hlf = 0; hlf < 2; ++hlf) {
#pragma unroll
      for (int mm = 0; mm < MT / 2; ++mm) {
        const int m = hlf * (MT / 2) + mm;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = m * 32 + g * 8 + half * 4;
          f32x4 v = {o[m][g * 4], o[m][g * 4 + 1], o[m][g * 4 + 2], o[m][g * 4 + 3]};
          if (a.b2) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.b2 + ch);
            v += b;
          }
          *reinterpret_cast<f32x4*>(stg + ql * SROW + (ch - hlf * HC)) = v;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      constexpr int CPR = HC / 8;                                  // 16-byte output pieces per row and pass
#pragma unroll
      for (int it = 0; it < (32 * CPR) / 64; ++it) {
        const int q = it * 64 + lane, r = q / CPR, cc = q - r * CPR;
        const f32x4 lo = *reinterpret_cast<const f32x4*>(stg + r * SROW + cc * 8);
        const f32x4 hi = *reinterpret_cast<const f32x4*>(stg + r * SROW + cc * 8 + 4);
        float f[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const long long rr = row0 + r;
        if (rr < a.rows) {
          const long long off = rr * C + hlf * HC + cc * 8;
          if (resp) {
            float rf[8];
            unpack8<T>(ld16(resp + off), rf);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] += rf[j];
          }
          st16(outp + off, pack8<T>(f));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the next pass overwrites the rows just read
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

}  // namespace mobi

extern "C" size_t mobi_ff_geglu_packed_bytes(int32_t c, int32_t hidden) {
  if (c <= 0 || hidden <= 0 || (c & 31) || (hidden & 31)) return 0;
  const size_t chunks = (size_t)hidden / 32;
  return chunks * ((size_t)(2 * (c / 16)) + (size_t)(2 * (c / 32) + 1)) * 1024;
}

extern "C" int mobi_ff_geglu(const mobi_ff_geglu_params* p, void* stream) {
  using namespace mobi;
  if (!p || !p->x || !p->w_packed || !p->out) return MOBI_ERR_ARG;
  if (p->dtype != MOBI_F16 && p->dtype != MOBI_BF16) return MOBI_ERR_ARG;
  if (p->rows <= 0 || p->hidden <= 0 || (p->hidden & 31)) return MOBI_ERR_ARG;
  if (p->c != 320) return MOBI_ERR_UNSUPPORTED;          // the output accumulators of one 32-row tile must fit the registers
  if (p->ln_gamma && (!p->ln_beta || ((reinterpret_cast<uintptr_t>(p->ln_gamma) | reinterpret_cast<uintptr_t>(p->ln_beta)) & 15)))
    return p->ln_beta ? MOBI_ERR_ALIGN : MOBI_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(p->x) | reinterpret_cast<uintptr_t>(p->w_packed) | reinterpret_cast<uintptr_t>(p->out) |
       reinterpret_cast<uintptr_t>(p->residual) | reinterpret_cast<uintptr_t>(p->b2)) & 15) return MOBI_ERR_ALIGN;
  const long long blocks = (p->rows + 127) / 128;
  if (blocks > 0x7fffffffLL) return MOBI_ERR_UNSUPPORTED;
  FfArgs a;
  a.x = p->x; a.b2 = p->b2; a.residual = p->residual; a.out = p->out; a.rows = p->rows; a.chunks = p->hidden / 32;
  a.ln_gamma = p->ln_gamma; a.ln_beta = p->ln_gamma ? p->ln_beta : nullptr; a.ln_eps = p->ln_eps;
  a.w1p = p->w_packed;
  a.w2p = reinterpret_cast<const unsigned char*>(p->w_packed) + (size_t)a.chunks * (2 * (p->c / 16)) * 1024;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p->dtype == MOBI_F16) hipLaunchKernelGGL((ff_geglu_kernel<f16_t, 320>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((ff_geglu_kernel<bf16_t, 320>), dim3((unsigned)blocks), dim3(256), 0, st, a);
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}
