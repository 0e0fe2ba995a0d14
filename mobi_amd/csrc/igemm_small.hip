// mobi_igemm, small 1 x 1 problems (the transformer blocks' linears at the 16 x 16 / 8 x 8 levels, most linears of
// `mobi_nusc_256`): a launch of the LDS-ring kernels is a latency chain there -- one 128 x 160 tile per CU, a block barrier,
// fragment reads and 4-5 LDS-DMA requests per 32-deep k-step (0.44 us per step), split-K slabs and a second launch to sum
// them: 14-22 us for 0.8-1.7 GFLOP.  This kernel has no block barrier in its main loop and no slabs:
//
//   * a block = four waves = ONE 32 x 32 output tile (hundreds of blocks even for 256 rows); the four waves split K between
//     them (wave w: k in [w K/4, (w+1) K/4)), each wave multiplies the whole tile over its quarter on MFMA 32x32x16 --
//     weights as the A operand, token rows as the B operand;
//   * batches of 80 k (five MFMA steps, 160 bytes of every row): ALL of a wave's batches (up to four: K <= 1280) are
//     requested at the start, each as five + five coalesced 16-byte loads per lane -- ten consecutive lanes read one
//     row's 160 bytes (loading the operands AS fragments, 32 rows x 2 pieces per instruction, holds the vector-memory
//     path to ~16 B per clock and CU: profiles/r04_small_lab_v1_fragment_loads.txt);
//   * a batch is transposed into fragment order through a WAVE-PRIVATE LDS image (64 rows x 176 bytes: written as loaded,
//     read back as ds_read_b128 fragments, conflict-free by the 16-byte pad): only the wave's own counters order it;
//   * the four partial tiles meet in LDS and are summed in the fixed order wave 0 + 1 + 2 + 3, then scale, bias, per-image
//     vector, residual, one rounding, 16-byte stores: no second launch, bit-reproducible.
//
// K % 320 == 0 (a wave's quarter is whole batches; the UNet's widths are 320 * 2^i), N % 32 == 0, operands within 2 GB; M may
// be ragged; one or two sources (channel concat, c0 % 80 == 0: a batch never straddles them).
// TAPS = 9: the same for 3 x 3 convolutions (pad 1, stride 1) with a handful of pixels -- the 4 x 4 / 8 x 8 levels of
// `mobi_nusc_256`, where a launch streams 30-60 MB of weights for 128 output pixels: k runs tap-major (k = tap * C + c, the
// packed weights' order), a batch lies inside one tap, the token row of a batch is the tap's neighbour pixel (a zero piece
// outside the image: a 9-bit mask per row).
#include "common.h"
#include "igemm_small.h"

namespace mobi {

namespace {

__device__ __forceinline__ void ld8fp(const float* p, float (&f)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
}

constexpr int SM_RP = 176;                 // bytes between rows of the staged image: 160 + 16
constexpr int SM_STAGE = 64 * SM_RP;       // a wave's image: 32 weight rows, then 32 token rows

template <typename T>
struct SmallBatch {
  typename Vec8<T>::type w[5], x[5];
};

// the wave's own LDS traffic only: its counter, and nothing moves across for the compiler
#define MOBI_SMALL_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// GENERAL = false: 1 x 1, one source -- a token row's byte offset is computed once (the hot instantiations: the generic
// addressing costs them 5-8 % per launch, measured).  GENERAL = true: two sources and / or TAPS = 9.
template <typename T, int NPRE, int TAPS, bool GENERAL>
__global__ __launch_bounds__(256, (NPRE == 4 || TAPS == 9) ? 2 : 3) void small_gemm_kernel(const SmallGemmArgs a) {
  static_assert(GENERAL || TAPS == 1, "the 3 x 3 form uses the general addressing");
  typedef typename Vec8<T>::type V;
  constexpr int TILE = 32, PITCH = TILE + 4;
  constexpr int RED_BYTES = 4 * TILE * PITCH * 4, LDS_BYTES = 4 * SM_STAGE > RED_BYTES ? 4 * SM_STAGE : RED_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tdiv = a.n_major ? a.tiles_m : a.tiles_n;
  const int tq = bid / tdiv, tr = bid - tq * tdiv;
  const int m0 = (a.n_major ? tr : tq) * TILE, n0 = (a.n_major ? tq : tr) * TILE;

  // piece p = 64 j + lane of an operand's 32 x 10 pieces: row p / 10, 16-byte piece p % 10
  int woff[5], xpix[5], c16[5], loff[5], tapmask[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int p = 64 * j + lane, row = p / 10, c = p - row * 10;
    loff[j] = row * SM_RP + c * 16;
    c16[j] = c * 16;
    woff[j] = ((n0 + row) * a.K + 8 * c) * 2;
    int m = m0 + row;
    m = m < a.M ? m : a.M - 1;                       // ragged last tile: a valid row, never stored
    const int img = m / a.hw, rem = m - img * a.hw;
    xpix[j] = GENERAL ? img * a.img_pix_stride + rem : ((img * a.img_pix_stride + rem) * a.K + 8 * c) * 2;   // !GENERAL: bytes
    tapmask[j] = 0x1ff;
    if (TAPS == 9) {
      const int y = rem / a.win, x = rem - y * a.win;
      int mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        mk |= (yy >= 0 && yy < a.hin && xx >= 0 && xx < a.win) ? (1 << t) : 0;
      }
      tapmask[j] = mk;
    }
  }
  const int nb = a.K / 320;                          // batches per wave
  const int kbase = wave * (a.K >> 2);
  const int ct = a.c0 + a.c1;
  const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.weight) + (long long)kbase * 2;
  SmallBatch<T> g[NPRE];
  auto request = [&](SmallBatch<T>& f, int b) {
    const unsigned char* wp = wsrc + b * 160;
    if constexpr (!GENERAL) {
      const unsigned char* xp = reinterpret_cast<const unsigned char*>(a.src0) + (long long)kbase * 2 + b * 160;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        f.w[j] = *reinterpret_cast<const V*>(wp + woff[j]);
        f.x[j] = *reinterpret_cast<const V*>(xp + xpix[j]);
      }
      return;
    }
    // wave-uniform: the batch's tap, source and channel offset
    const int k = kbase + b * 80;
    const int tap = TAPS == 1 ? 0 : k / ct;
    const int cc = k - tap * ct;
    const bool second = cc >= a.c0;
    const unsigned char* xs = reinterpret_cast<const unsigned char*>(second ? a.src1 : a.src0);
    const int cs = second ? a.c1 : a.c0, ccs = second ? cc - a.c0 : cc;
    const int dpix = TAPS == 1 ? 0 : (tap / 3 - 1) * a.win + (tap % 3 - 1);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      f.w[j] = *reinterpret_cast<const V*>(wp + woff[j]);
      const int off = ((xpix[j] + dpix) * cs + ccs) * 2 + c16[j];
      if (TAPS == 1 || ((tapmask[j] >> tap) & 1)) f.x[j] = *reinterpret_cast<const V*>(xs + off);
      else f.x[j] = __builtin_bit_cast(V, u32x4{0u, 0u, 0u, 0u});
    }
  };
#pragma unroll
  for (int u = 0; u < NPRE; ++u)
    if (u < nb) request(g[u], u);

  // the finishing item of this lane (rows [8 w, 8 w + 8) of the tile, pieces of 8 channels: lanes 0-31): its bias, per-image
  // vector and residual are requested NOW, behind the operand requests, so that the epilogue waits for nothing
  T* __restrict__ outT = reinterpret_cast<T*>(a.out);
  float* __restrict__ outF = reinterpret_cast<float*>(a.out);
  const T* __restrict__ resid = reinterpret_cast<const T*>(a.residual);
  const int erow = wave * 8 + (lane >> 2), em = m0 + erow, en = n0 + 8 * (lane & 3);
  const bool eact = lane < 32 && em < a.M;
  const int eimg = eact ? em / a.hw : 0, erem = em - eimg * a.hw;
  float ebias[8], evec[8];
  u32x4 eres = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
  for (int j = 0; j < 8; ++j) { ebias[j] = 0.f; evec[j] = 0.f; }
  if (eact) {
    if (a.bias) ld8fp(a.bias + en, ebias);
    if (a.rowvec) ld8fp(a.rowvec + (long long)eimg * a.rowvec_stride + en, evec);
    if (resid) eres = ld16(resid + (long long)eimg * a.res_img_stride + (long long)erem * a.N + en);
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  unsigned char* st = lds + wave * SM_STAGE;
  const unsigned char* fr = st + i * SM_RP + 16 * h;      // fragment (step s): + 32 s; token rows: + 32 rows
  for (int b0 = 0; b0 < nb; b0 += NPRE) {
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      const int b = b0 + u;
      if (b < nb) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          *reinterpret_cast<V*>(st + loff[j]) = g[u].w[j];
          *reinterpret_cast<V*>(st + 32 * SM_RP + loff[j]) = g[u].x[j];
        }
        if (b + NPRE < nb) request(g[u], b + NPRE);
        MOBI_SMALL_LDS_FENCE();
        V wf[5], xf[5];
#pragma unroll
        for (int s = 0; s < 5; ++s) {
          wf[s] = *reinterpret_cast<const V*>(fr + 32 * s);
          xf[s] = *reinterpret_cast<const V*>(fr + 32 * SM_RP + 32 * s);
        }
        MOBI_SMALL_LDS_FENCE();                          // every fragment is in registers: the image may be overwritten
#pragma unroll
        for (int s = 0; s < 5; ++s) acc = mfma32(wf[s], xf[s], acc);
      }
    }
  }

  // partial tiles -> LDS: red[wave][m][n], lane = token row i, four consecutive n per accumulator quad
  __syncthreads();                                       // the images of all four waves are dead
  float* red = reinterpret_cast<float*>(lds);
  float* mine = red + wave * TILE * PITCH;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    *reinterpret_cast<f32x4*>(mine + i * PITCH + 8 * q + 4 * h) = f32x4{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
  __syncthreads();

  // wave w finishes rows [8 w, 8 w + 8): 32 items of 8 channels
  if (eact) {
    const int piece = lane & 3;
    float o[8];
    ld8fp(red + erow * PITCH + 8 * piece, o);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      float p[8];
      ld8fp(red + (w * TILE + erow) * PITCH + 8 * piece, p);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += p[j];
    }
    if (a.scale != 1.0f) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] *= a.scale;
    }
    float rf[8];
    unpack8<T>(eres, rf);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = ((o[j] + ebias[j]) + evec[j]) + rf[j];
    const long long off = (long long)eimg * a.out_img_stride + (long long)erem * a.N + en;
    if (a.out_f32) {
      *reinterpret_cast<f32x4*>(outF + off) = f32x4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(outF + off + 4) = f32x4{o[4], o[5], o[6], o[7]};
    } else {
      st16(outT + off, pack8<T>(o));
    }
  }
}
#undef MOBI_SMALL_LDS_FENCE

}  // namespace

int launch_small_gemm(const SmallGemmArgs& a, int dtype, hipStream_t st) {
  const dim3 grid((unsigned)(a.tiles_m * a.tiles_n)), block(256);
  const int nb = a.K / 320;
#define MOBI_SMALL_LAUNCH(T_, NPRE_, TAPS_, GEN_) \
  hipLaunchKernelGGL((small_gemm_kernel<T_, NPRE_, TAPS_, GEN_>), grid, block, 0, st, a)
#define MOBI_SMALL_BY_NB(T_)                                                                              \
  do {                                                                                                    \
    if (a.taps == 9) MOBI_SMALL_LAUNCH(T_, 2, 9, true);    /* (a ring of four batches spills beside the tap masks) */ \
    else if (a.c1 > 0) MOBI_SMALL_LAUNCH(T_, 4, 1, true);                                                 \
    else if (nb == 1) MOBI_SMALL_LAUNCH(T_, 1, 1, false);                                                 \
    else if (nb == 2) MOBI_SMALL_LAUNCH(T_, 2, 1, false);                                                 \
    else MOBI_SMALL_LAUNCH(T_, 4, 1, false);                                                              \
  } while (0)
  if (dtype == MOBI_F16) MOBI_SMALL_BY_NB(f16_t); else MOBI_SMALL_BY_NB(bf16_t);
#undef MOBI_SMALL_BY_NB
#undef MOBI_SMALL_LAUNCH
  MOBI_CHECK_LAUNCH();
  return MOBI_OK;
}

}  // namespace mobi
