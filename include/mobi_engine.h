/*
 * mobi_engine.h -- C ABI of the MI355X (gfx950) denoising engine for MObI's
 * camera+lidar sampling path.
 *
 * The reference has no FFI on this path: every FLOP runs inside PyTorch ops
 * called from Python (boundary B2).  Each entry point below
 * therefore names the reference *call site(s)* whose arithmetic it replaces
 * (paths relative to the reference repo root).  INTEGRATION.md shows the
 * ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every tensor is a caller-allocated DEVICE buffer; the library never
 *     allocates, frees or retains memory.  Scratch space is passed in `ws`
 *     with the size given by the matching *_workspace_bytes query.
 *   - asynchronous on `stream` (a hipStream_t passed as void*), no internal
 *     synchronisation, graph-capture safe.
 *   - returns MOBI_OK (0) or a negative error code; never throws.
 *   - activations are channels-last: [image][y][x][channel], channel
 *     contiguous, in a 16-bit storage type selected by `dtype`
 *     (MOBI_F16 | MOBI_BF16).  All accumulation / statistics are fp32.
 *   - "T" below means that 16-bit storage type.
 */
#ifndef MOBI_ENGINE_H_
#define MOBI_ENGINE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever a parameter struct OR a function signature changes: a caller built against another version must
 * not call into this library -- check mobi_abi_version() == MOBI_ABI_VERSION next to the mobi_struct_size checks.
 *   1  first layout (later also mobi_igemm_params.weight_tiled, mobi_ddim_step_params.coef_dev)
 *   2  mobi_attention_params.q_log2_scaled; mobi_two_key_adapter_params.ln_out / ln_gamma / ln_beta / ln_eps;
 *      mobi_ff_geglu_params (struct id 14); mobi_two_key_adapter_fuses_ln(channels, total_rows)
 *   3  mobi_row_chain_params / mobi_chain_op (struct ids 15, 16), mobi_row_chain*; the backward entry points
 *      (mobi_layernorm_bwd_params 17, mobi_attention_bwd_params 18)
 *   4  mobi_groupnorm_bwd takes a workspace (mobi_groupnorm_bwd_workspace_floats); mobi_igemm_kernel_variant may answer
 *      MOBI_IGEMM_SMALL; mobi_tile_weights
 *   5  mobi_igemm_params.sync + mobi_igemm_sync_bytes (split-K finished inside the launch); mobi_igemm_params.groups may be
 *      any divisor of batch; mobi_igemm_params.ln_svec / ln_eps (LayerNorm folded into the consuming launch);
 *      mobi_groupnorm_params.src_f32 / out_mode (0..3); mobi_split_f32
 *   6  mobi_igemm_params.defer_finish + mobi_igemm_slab_count / mobi_igemm_finish; mobi_split_source (struct id 19) and
 *      mobi_groupnorm_params.src0_split + mobi_groupnorm_takes_split: a split-K launch's partial sums are summed by the
 *      GroupNorm that consumes them instead of by a reduce launch of their own */
#define MOBI_ABI_VERSION 6

enum { MOBI_OK = 0, MOBI_ERR_ARG = -1, MOBI_ERR_UNSUPPORTED = -2, MOBI_ERR_LAUNCH = -3, MOBI_ERR_ALIGN = -4 };
enum { MOBI_F16 = 0, MOBI_BF16 = 1 };

int mobi_abi_version(void);
const char* mobi_error_string(int code);
/* sizeof() of a parameter struct as the library was compiled; bindings compare it with
 * their own layout before the first call.  id: 0 igemm, 1 groupnorm, 2 layernorm,
 * 3 attention, 4 ctx_attention, 5 skinny_linear, 6 conv_small_cin, 7 conv_small_cout,
 * 8 ddim_step, 9 two_key_adapter, 10 range_paste, 11 lidar_metrics, 12 range_prepare, 13 image_prepare, 14 ff_geglu,
 * 15 row_chain, 16 chain_op, 17 layernorm_bwd, 18 attention_bwd, 19 split_source.  Returns 0 for an unknown id. */
size_t mobi_struct_size(int id);
/* Development hook: the library reads its MOBI_* A/B environment variables once, at the first launch
 * (mobi_amd/csrc/tuning.h lists them); this re-reads them.  Not needed by a product caller. */
int mobi_tuning_reload(void);
/* Bit 0: the library was built with -DMOBI_DEV, i.e. it also carries the A/B partner kernels (lockstep direct-to-LDS
 * igemm, fast-addressing register-staged igemm, software-pipelined attention) that the shipped build leaves out. */
int mobi_build_info(void);

/* ---------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer on the matrix cores.
 *   out[m][n] = epilogue( scale * sum_k A[m][k] * W[n][k] )
 * where row m is an output pixel (image, y, x) and k runs over
 * (tap, channel) of the gathered input window.  A 1x1 convolution and an
 * nn.Linear over tokens are the kh = kw = 1 case.
 * Replaces: nn.Conv2d / nn.Linear calls of ResBlock (openaimodel.py:255-275),
 * Downsample (:158-160), Upsample incl. the F.interpolate(nearest, x2) before
 * it (:109-119), torch.cat([h, hs.pop()], 1) feeding a ResBlock (:893-894),
 * SpatialTransformer proj_in/proj_out (attention.py:306,311), every Linear of
 * CrossAttention / FeedForward / GEGLU (attention.py:38-65,162-194), and the
 * VAE's Conv2d layers incl. the asymmetric-pad stride-2 Downsample
 * (model.py:47,66,72-79,92,102) and AttnBlock q/k/v/proj_out + both bmm
 * (model.py:178-202).
 * ------------------------------------------------------------------------- */
enum { MOBI_EPI_NONE = 0, MOBI_EPI_GEGLU = 1 };
enum { MOBI_OUT_ROWS = 0,        /* T   [m][cout]                               */
       MOBI_OUT_TRANSPOSED = 1,  /* T   [image][cout][hout*wout]                */
       MOBI_OUT_ROWS_F32 = 2 };  /* f32 [m][cout]                               */

typedef struct mobi_igemm_params {
  const void* src0;      /* T, channels-last, c0 channels                              */
  const void* src1;      /* T or NULL: second source concatenated after src0's channels */
  int32_t c0, c1;        /* channel counts (c1 = 0 without src1); each % 32 == 0        */
  int32_t batch;         /* images                                                      */
  int32_t hin, win;      /* stored spatial size of the sources                          */
  int32_t upsample;      /* 1: the logical input is the nearest-neighbour x2 of sources */
  int32_t hout, wout;    /* output spatial size                                         */
  int32_t kh, kw, stride, pad_h, pad_w; /* pad = top / left zero padding; reads past the
                                           bottom / right edge are zero as well          */
  int64_t src_img_stride;  /* elements between images of src0/src1; 0 = dense           */
  const void* weight;    /* T [groups][n_packed][kh*kw*(c0+c1)], k contiguous, k = tap*C + c */
  int32_t groups;        /* 1: one weight matrix; a divisor g of batch: image i is multiplied by matrix i / (batch / g)
                            (== batch: one matrix per image; 2: the camera images' and the lidar images' projections of a
                            [camera images ; lidar images] batch as ONE launch, ldm/modules/attention.py:245-263)         */
  int64_t w_group_stride;/* elements between the per-group weight matrices              */
  int32_t n_packed;      /* rows of W; GEGLU: n_packed = 2 * cout, every 16 rows = 8 value rows
                            then the 8 gate rows of the same 8 outputs (cout % 8 == 0)    */
  int32_t cout;          /* logical output columns                                       */
  const float* bias;     /* [cout] or NULL (GEGLU: [n_packed], packed like the rows)     */
  const float* rowvec;   /* f32 [batch][cout] or NULL: per-image additive vector         */
  int32_t rowvec_stride; /* elements between images of rowvec; 0 = cout                  */
  const void* residual;  /* T [m][cout] or NULL                                          */
  int64_t res_img_stride;/* elements between images of residual; 0 = dense               */
  void* out;
  int64_t out_img_stride;/* elements between images of out; 0 = dense                    */
  int32_t out_mode;      /* MOBI_OUT_*                                                   */
  int32_t epilogue;      /* MOBI_EPI_*                                                   */
  float scale;           /* multiplies the accumulator before bias (1.0f normally)       */
  int32_t dtype;
  int32_t split_k;       /* <= 1: single pass.  > 1: k is cut into split_k ranges that run as
                            separate workgroups (fills the chip when m is small, e.g. the 8x8 /
                            16x16 UNet levels); fp32 partial slabs go to `ws` and a second launch
                            sums them and applies the epilogue.  Not with GEGLU / transposed /
                            per-image weights.                                              */
  void* ws;              /* mobi_igemm_workspace_bytes(p, split_k) bytes, 16-byte aligned  */
  int32_t k_order;       /* layout of W's k axis.  0: k = tap*C + c.  1: k = (c/64)*taps*64 + tap*64
                            + c%64 -- the taps of one 64-channel slice are consecutive k-tiles, so the
                            9 reads of a 3x3 window's pixels hit L2 instead of being re-fetched per
                            tap.  Needs C % 64 == 0 (and c0 % 64 == 0 with two sources).        */
  const void* weight_tiled; /* optional second copy of W (k_order 0, groups 1, n_packed % 16 == 0, k % 32 == 0) as
                            [n_packed / 16][k / 32] blocks of 1 KiB: block (p, s) is the LDS image of rows 16 p .. 16 p + 15,
                            k 32 s .. 32 s + 31 -- row r at bytes 64 r, its four 16-byte chunks c at slot c ^ P[(r >> 2) & 3],
                            P = {0, 2, 3, 1}.  The LDS-DMA main loops then fetch one contiguous KiB per request instead of
                            sixteen 64-byte row segments (a CU's request path takes ~60 against ~25 B per clock).  NULL: W only. */
  void* sync;            /* split_k > 1, optional: mobi_igemm_sync_bytes(p, split_k) bytes of int32 arrival counters, ZERO
                            before the launch and zero again after it (one buffer can serve every launch of a stream).
                            With it the launch finishes its own split: the workgroup that arrives LAST at an output tile sums
                            the tile's slabs (same order, same sums as the reduce launch: bit-identical) and applies the
                            epilogue -- no second launch.  NULL (or a kernel variant that cannot): the reduce launch.        */
  const float* ln_svec;  /* optional, f32 [n_packed]: LayerNorm FOLDED into this launch (1 x 1, one source, groups 1, no split, no
                            rowvec / residual, scale 1): `weight` holds W diag(gamma) (rounded to T), ln_svec its row sums, `bias`
                            W beta + b; the launch computes the statistics of its own input rows and returns
                            rstd (x W'^T - mean ln_svec) + bias == Linear(LayerNorm(x)) (ldm/modules/attention.py:234, :264:
                            `attn1(norm1(x))`, `ff(norm3(x))`) without a normalised copy of x.  Runs on the LDS-DMA ring kernels. */
  float ln_eps;
  int32_t defer_finish;  /* split_k > 1, no `sync`, out_mode MOBI_OUT_ROWS: 1 = launch the k ranges only.  `ws` then holds
                            mobi_igemm_slab_count(p) slabs f32 [rows][n_packed] of partial sums and `out` is NOT written: the
                            caller hands them to the consumer (mobi_groupnorm_params.src0_split: a ResBlock's GroupNorm behind
                            its 3 x 3 convolution, openaimodel.py:255-275, or the next block's) or calls mobi_igemm_finish. */
} mobi_igemm_params;

int mobi_igemm(const mobi_igemm_params* p, void* stream);
/* The number of slabs a launch of these parameters writes (k ranges are never empty: it can be less than split_k); 1: no
 * split.  Negative = the error mobi_igemm would return. */
int32_t mobi_igemm_slab_count(const mobi_igemm_params* p);
/* The second half of a split-K launch on its own (same parameters as the mobi_igemm call that ran with defer_finish = 1):
 * sums the slabs in ascending order, adds bias / per-image vector / residual, writes `out`. */
int mobi_igemm_finish(const mobi_igemm_params* p, void* stream);
/* Library heuristic for split_k (1 = do not split) and the workspace it needs. */
int mobi_igemm_plan_splits(const mobi_igemm_params* p);
/* Which main loop mobi_igemm would run for these parameters (no launch; measurement / tests):
 * the persistent direct-to-LDS kernels (256-pixel tiles; ping-pong schedule where the register epilogue applies),
 * or the register-staged kernel with 256- or 128-pixel tiles.  Negative = the error mobi_igemm would return. */
enum { MOBI_IGEMM_STAGED_128 = 0, MOBI_IGEMM_STAGED_256 = 1, MOBI_IGEMM_DIRECT_LDS = 2, MOBI_IGEMM_PINGPONG = 3,
       MOBI_IGEMM_RING_128 = 4, MOBI_IGEMM_RING_256 = 5, MOBI_IGEMM_RING_128W = 6, MOBI_IGEMM_SMALL = 7 };
int mobi_igemm_kernel_variant(const mobi_igemm_params* p);
size_t mobi_igemm_workspace_bytes(const mobi_igemm_params* p, int32_t splits);
size_t mobi_igemm_sync_bytes(const mobi_igemm_params* p, int32_t splits);

/* ---------------------------------------------------------------------------
 * GroupNorm (32 groups) with optional fused SiLU over one or two
 * channel-concatenated sources.  Replaces GroupNorm32 + nn.SiLU
 * (ldm/modules/diffusionmodules/util.py:199-216, openaimodel.py:211-236,
 * 832-836), Normalize (attention.py:77-78, model.py:38-39) + nonlinearity
 * (model.py:33-35), and the torch.cat in front of an output ResBlock.
 * ------------------------------------------------------------------------- */
/* A tensor T [batch][hw][channels] given as the UNFINISHED partial sums of the split-K mobi_igemm that produces it
 * (mobi_igemm_params.defer_finish = 1).  Element (row, c) = T( sum_s slabs[s][row][c] (s ascending) + bias[c] + rowvec[image][c]
 * + residual[row][c] ), in that order: bit for bit what mobi_igemm_finish writes. */
typedef struct mobi_split_source {
  const float* slabs;      /* f32 [count][batch * hw][row_stride] (the producer's `ws`) */
  int32_t count;           /* mobi_igemm_slab_count of the producer, 2 .. 64 */
  int32_t row_stride;      /* the producer's n_packed (>= channels) */
  const float* bias;       /* f32 [channels] or NULL */
  const float* rowvec;     /* f32 [batch][channels] or NULL */
  int32_t rowvec_stride;   /* elements between images of rowvec; 0 = channels */
  const void* residual;    /* T [batch][hw][channels] or NULL */
  int64_t res_img_stride;  /* elements between images of residual; 0 = dense */
  void* finished;          /* T [batch][hw][channels] or NULL: the finished tensor is ALSO written here (a ResBlock's output is
                              the next GroupNorm's input and a residual / skip connection, openaimodel.py:275, 893-894) */
} mobi_split_source;

typedef struct mobi_groupnorm_params {
  const void* src0;
  const void* src1;        /* or NULL */
  int32_t c0, c1;
  int32_t batch, hw;
  const float* gamma;      /* [c0+c1] */
  const float* beta;
  float eps;
  int32_t silu;            /* 1: y = y * sigmoid(y) */
  void* out;               /* T [batch][hw][c0+c1] */
  void* ws;                /* mobi_groupnorm_workspace_bytes(batch, hw) bytes */
  int32_t dtype;
  int32_t src_f32;         /* 1: src0 is f32 [batch][hw][c0] (no src1): the VAE decoder's fp32 streams */
  int32_t out_mode;        /* 0: T [batch][hw][C].  1: T [batch][hw][2 C] = hi | lo, hi = T(y), lo = T(y - hi): a convolution
                              whose weights are duplicated along its input channels then multiplies y to ~22 bits (the lidar
                              decoder's tail of the fp16 parity configuration, model.py:612-623).  2: f32 [batch][hw][C].
                              3: T [batch][hw][3 C] = hi | lo | hi, for weights [W ; W ; W - T(W)]: the weights' own rounding is
                              corrected as well (the decoders of the fp16 parity configuration, model.py:559-630) */
  void* sync;              /* optional: int32 [batch] arrival counters, ZERO before the launch and zero again after it (one buffer
                              serves every launch of a stream).  With them the library may run the one-launch form whose workgroups
                              own pixel chunks and meet through memory (large groups: the 64 x 64 level); NULL: never. */
  const mobi_split_source* src0_split; /* optional (then src0 may be NULL; T sources, out_mode 0): src0 as the partial sums of the
                              launch that produces it -- this launch sums them while it loads its rows, so the producer needs
                              no reduce launch.  Only where mobi_groupnorm_takes_split says so. */
} mobi_groupnorm_params;

size_t mobi_groupnorm_workspace_bytes(int32_t batch, int32_t hw);
/* 1: mobi_groupnorm accepts src0_split for these sizes (the one-launch register form holds the tensor); 0: finish the
 * producer with mobi_igemm_finish first. */
int mobi_groupnorm_takes_split(int32_t c0, int32_t c1, int32_t batch, int32_t hw);
int mobi_groupnorm(const mobi_groupnorm_params* p, void* stream);

/* LayerNorm over the channel axis of token rows (nn.LayerNorm, eps 1e-5,
 * attention.py:213-223,232-256).  `images` blocks of `rows_per_image` rows; the
 * image strides allow the camera / lidar halves of the interleaved batch
 * (x[::2], x[1::2], attention.py:246-247) to be addressed without a copy. */
typedef struct mobi_layernorm_params {
  const void* src; void* out;
  int32_t images, rows_per_image, channels;
  int64_t src_img_stride, out_img_stride;   /* elements; 0 = dense */
  const float* gamma; const float* beta; float eps;
  int32_t dtype;
} mobi_layernorm_params;
int mobi_layernorm(const mobi_layernorm_params* p, void* stream);

/* ---------------------------------------------------------------------------
 * Fused attention  out = softmax(q k^T * scale) v, one launch for all images
 * and heads (CrossAttention.forward, attention.py:171-194: rearrange, einsum,
 * softmax, einsum, rearrange).  v is consumed TRANSPOSED ([channel][token]) as
 * written by mobi_igemm with MOBI_OUT_TRANSPOSED.  Image strides let the
 * cross-modal calls read the partner modality in place (attention.py:245-261).
 * dh % 8 == 0 and dh <= 160.
 * ------------------------------------------------------------------------- */
typedef struct mobi_attention_params {
  const void* q;  int64_t q_img_stride;  int32_t q_row_stride;    /* T [image][tq][...], head h at column h*dh */
  const void* k;  int64_t k_img_stride;  int32_t k_row_stride;    /* T [image][tk][...]                         */
  const void* vt; int64_t vt_img_stride; int32_t vt_row_stride;   /* v_layout 0: T [image][heads*dh][tk...] (V^T)
                                                                     v_layout 1: T [image][tk][...] (V rows, head
                                                                     h at column h*dh, like k)                  */
  void* out;      int64_t out_img_stride; int32_t out_row_stride; /* T [image][tq][heads*dh]                    */
  int32_t images, heads, dh, tq, tk;
  float scale;
  int32_t dtype;
  int32_t v_layout;      /* 0: vt holds V transposed; 1: vt holds V row-major (q|k|v stacked projections)       */
  int32_t q_log2_scaled; /* 1: q already carries scale * log2(e) (folded into the to_q weights when they were
                            packed, attention.py:162,178): `scale` is then ignored and the kernels exponentiate
                            q.k as a power of two directly; 0: the kernels scale q (one more rounding of q to T
                            in the V row-major kernels)                                                        */
} mobi_attention_params;
int mobi_attention(const mobi_attention_params* p, void* stream);

/* ---------------------------------------------------------------------------
 * Fused GEGLU feed-forward (FeedForward / GEGLU, attention.py:38-65):
 *   out = ((x' W1v^T + b1v) * gelu_erf(x' W1g^T + b1g)) W2^T + b2 (+ residual),  x' = x or LayerNorm(x)
 * in one launch; the hidden activation [rows][hidden] is never written.  c = 320
 * (the 64 x 64 level of the UNet), hidden % 32 == 0.
 * w_packed: mobi_ff_geglu_packed_bytes(c, hidden) bytes of "chunk images" (one
 * chunk = 32 hidden units), every fragment 1 KiB = 64 lanes x 8 T in the lane order
 * of the MFMA A operand (lane l: row l & 31, column group l >> 5):
 *   first   [hidden/32][2 c/16 fragments]: fragment t c/16 + ks (t = 0 value rows, 1 gate
 *           rows of GEGLU.proj.weight [2 hidden][c]): element j = W1[t hidden + 32 chunk +
 *           (l & 31)][16 ks + 8 (l >> 5) + j];
 *   then    [hidden/32][2 c/32 fragments + 1]: fragment 2 m + s: element j =
 *           W2[32 m + (l & 31)][32 chunk + 16 s + 8 (j >> 2) + 4 (l >> 5) + (j & 3)]
 *           (net[2].weight [c][hidden]; the unit order of the first product's accumulator),
 *           the last KiB = the chunk's 32 value and 32 gate entries of GEGLU.proj.bias as fp32.
 * ------------------------------------------------------------------------- */
typedef struct mobi_ff_geglu_params {
  const void* x;            /* T [rows][c], dense */
  int64_t rows;
  int32_t c, hidden;
  const void* w_packed;
  const float* b2;          /* [c] or NULL */
  const void* residual;     /* T [rows][c] or NULL (may alias out) */
  void* out;                /* T [rows][c] */
  int32_t dtype;
  /* Optional LayerNorm of x over the c channels, applied in the kernel before the first product (norm3 of
   * BasicTransformerBlock, attention.py:265: ff(norm3(x)) + x; `residual` is then x): f32 [c] each, eps.  NULL: none. */
  const float* ln_gamma; const float* ln_beta; float ln_eps;
} mobi_ff_geglu_params;
size_t mobi_ff_geglu_packed_bytes(int32_t c, int32_t hidden);
int mobi_ff_geglu(const mobi_ff_geglu_params* p, void* stream);

/* ---------------------------------------------------------------------------
 * Row-resident chain of C -> C linear layers over token rows (C = 320: the
 * 64 x 64 level of the UNet; `mobi_nusc_256`'s 32 x 32 level), one launch.
 * Replaces the launches BETWEEN the attention kernels of a transformer block
 * (BasicTransformerBlock._forward, attention.py:230-266 of the reference), whose
 * [rows][C] results otherwise make an HBM round trip each:
 *   after attn1:   x = to_out(a) + attn2 vector + x            (:234-235)
 *                  x = x + cond_adapter(...)                   (:237-243, the two-key fold of mobi_two_key_adapter)
 *                  q = cross_modal_attn_*.to_q(LayerNorm(x)), k | v = to_k | to_v(x of the partner)   (:249-261)
 *   after the camera's cross-modal attention:  x_cam = x_cam + connector(to_out(a));  k | v of the lidar's attention
 *   before attn1:  t = proj_in(GroupNorm(x)) (:306-307), q | k | v = attn1.to_*(LayerNorm(t))        (:234)
 * A wave keeps 32 token rows in registers as MFMA 32x32x16 B fragments (lane l: row l & 31, channels
 * 16 ks + 8 (l >> 5) + 0..7 of fragment ks) and runs a PROGRAM of operations on them; weights stream through an
 * LDS ring as 20-KiB chunk images.  A block = 4 waves = 128 rows of one image: rows_per_image % 128 == 0.
 *
 * Weight image of one product (mobi_row_chain_weight_bytes() = 200 KiB for a [320][320] matrix W, out = W x):
 *   [chunk c < 10][fragment f < 20][lane l < 64][8 T],  f = 10 kk + m  (kk < 2, m < 10):
 *   element j = W[32 m + tau(l & 31)][16 (2 c + kk) + 8 (l >> 5) + j],  tau(i) = i with bits 2 and 3 swapped
 *   (so that the accumulator registers a lane receives are the channels of the fragments it holds).
 *
 * A wave's registers hold the row state `s` (the operand of every product) and, until it is consumed, a residual `r`.
 * Operations (code):
 *   MOBI_CH_LOAD_S    s <- rows of p0 (T; image index img / img_div, strides in elements)
 *   MOBI_CH_LOAD_R    r <- rows of p0
 *   MOBI_CH_AFFINE_S  s <- s * bias[img][ch] + svec[img][ch]   (f32 [images][C] each: GroupNorm folded to scale / shift)
 *   MOBI_CH_ROWSTATS  (rs, cs) <- (rstd, -rstd * mean) of s over the C channels (eps)
 *   MOBI_CH_PRODUCT   v = W s (p0 = weight image, p1 = the NEXT product's image or NULL: prefetched);
 *                     flags FOLD: v = rs * v + cs * svec[ch] (LayerNorm folded into W: W' = W diag(gamma),
 *                                 svec = row sums of the ROUNDED W', bias = W beta);
 *                     v += bias[(img / bias_img_div) * bias_img_stride + ch] (f32, required: zeros where the layer has none);
 *                     RESID: v += r (r is consumed);   TO_S: s <- round(v);   STORE: dst rows <- round(v)
 *   MOBI_CH_ADAPTER   s <- two-key adapter of s (the tables of mobi_two_key_adapter_params as images, ad_image); flags STORE
 *                     and dst required: the result rows are written
 *   MOBI_CH_STORE_S   dst rows <- s
 * At most 4 products per program.
 * nprog = 2: even images run prog[0], odd images prog[1] (camera / lidar rows of the interleaved batch).
 * ------------------------------------------------------------------------- */
enum { MOBI_CH_LOAD_S = 0, MOBI_CH_LOAD_R = 1, MOBI_CH_AFFINE_S = 2, MOBI_CH_ROWSTATS = 3, MOBI_CH_PRODUCT = 4,
       MOBI_CH_ADAPTER = 5, MOBI_CH_STORE_S = 6 };
enum { MOBI_CH_FOLD = 1, MOBI_CH_RESID = 2, MOBI_CH_TO_S = 4, MOBI_CH_STORE = 8 };
#define MOBI_CHAIN_MAX_OPS 10
typedef struct mobi_chain_op {
  int32_t code, flags;
  const void* p0;                 /* LOAD_*: T tensor;  PRODUCT: weight image */
  const void* p1;                 /* PRODUCT: the next product's weight image, or NULL */
  const float* bias;              /* PRODUCT: f32 bias;  AFFINE_S: f32 scale [images][C] */
  const float* svec;              /* PRODUCT + FOLD: f32 [C];  AFFINE_S: f32 shift [images][C] */
  void* dst;                      /* T rows (PRODUCT + STORE, ADAPTER + STORE, STORE_S) */
  int64_t img_stride, row_stride;           /* of p0 (LOAD_*), elements */
  int64_t dst_img_stride, dst_row_stride;   /* of dst, elements */
  int64_t bias_img_stride;                  /* elements; 0: one vector for all images */
  int32_t img_div, dst_img_div, bias_img_div;   /* image index = img / div (0 is read as 1) */
  float eps;                                /* ROWSTATS */
} mobi_chain_op;
typedef struct mobi_row_chain_params {
  int32_t dtype, channels;        /* channels == 320 */
  int32_t images, rows_per_image; /* rows_per_image % 128 == 0 */
  int32_t nprog;                  /* 1 or 2 */
  int32_t nops[2];
  mobi_chain_op prog[2][MOBI_CHAIN_MAX_OPS];
  /* MOBI_CH_ADAPTER: the tables of mobi_two_key_adapter_params as LDS images, mobi_row_chain_adapter_image_bytes() per
   * image, written once per conditioning by mobi_row_chain_adapter_image (NULL: the programs have no ADAPTER); the
   * LayerNorm eps of the adapter's statistics. */
  const void* ad_image; float ad_eps;
} mobi_row_chain_params;
/* a, u: f32 [images][heads][C]; c: f32 [images][heads]; b: f32 [images][C] (mobi_two_key_adapter_params' tables; a_sum is
 * re-derived from the split table) -> out: images x mobi_row_chain_adapter_image_bytes(C) bytes. */
size_t mobi_row_chain_adapter_image_bytes(int32_t channels);
int mobi_row_chain_adapter_image(const float* a, const float* c, const float* u, const float* b, int32_t images,
                                 int32_t heads, int32_t channels, int32_t dtype, void* out, void* stream);
/* GroupNorm (32 groups, eps) of x (T [images][hw][channels] dense) as two per-image vectors for MOBI_CH_AFFINE_S:
 * scale = rstd * gamma, shift = beta - mean * rstd * gamma (f32 [images][channels] each): SpatialTransformer.norm in front of
 * proj_in (attention.py:306 of the reference) when the chain applies it to the rows it holds.  channels % 64 == 0. */
int mobi_groupnorm_scale_shift(const void* x, const float* gamma, const float* beta, float eps, float* scale, float* shift,
                               int32_t images, int32_t hw, int32_t channels, int32_t dtype, void* stream);
size_t mobi_row_chain_weight_bytes(int32_t channels);
int mobi_row_chain_supported(int32_t channels, int32_t rows_per_image);
int mobi_row_chain(const mobi_row_chain_params* p, void* stream);

/* The VAE decoder's fp32 residual trunk (Decoder.forward / ResnetBlock / AttnBlock, model.py:121-141, 178-202, 587-630 of the
 * reference run in fp32): trunk (f32, n elements, updated in place) += inc (T), x16 = T(trunk) -- the 16-bit copy the next
 * GroupNorm / convolution reads.  inc == NULL: only the conversion.  n % 8 == 0. */
int mobi_trunk_add(float* trunk, const void* inc, void* x16, int64_t n, int32_t dtype, void* stream);
/* The operand form of mobi_groupnorm's out_mode 1 / 3 for an fp32 tensor no GroupNorm stands in front of (the inputs of the
 * decoders' upsampling convolutions and 1 x 1 shortcuts, model.py:60-80, 136-139 of the reference): x f32 [rows][channels] ->
 * out T [rows][parts * channels] = hi | lo (parts 2) or hi | lo | hi (parts 3), hi = T(x), lo = T(x - hi).  channels % 8 == 0. */
int mobi_split_f32(const float* x, void* out, int64_t rows, int32_t channels, int32_t parts, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Backward pass of the transformer block (FIRST SLICE: the training step of the adapter
 * parameters, ldm/models/diffusion/ddpm.py:356-370, 1616-1669 of the reference: `cond_adapter*` / `cross_modal*` of
 * every BasicTransformerBlock, ldm/modules/attention.py:197-266).  What torch.autograd does for the reference:
 *   linear layers     dx = dy W  and  dW = dy^T x  are mobi_igemm launches (the weight read as [in][out]; both operands of
 *                     the weight gradient transposed by mobi_transpose, the token axis as k, MOBI_OUT_ROWS_F32), the bias
 *                     gradient is mobi_colsum;
 *   nn.LayerNorm      mobi_layernorm_bwd;   GEGLU (attention.py:38-46)  mobi_geglu_fwd on the un-fused projection / _bwd;
 *   attention         mobi_attention_bwd (softmax(q k^T scale) v per head, attention.py:171-194).
 * Correct and bit-reproducible (fixed-order reductions), not tuned: fp32 vector arithmetic on LDS tiles.
 * ------------------------------------------------------------------------- */
int mobi_transpose(const void* src, int64_t src_row_stride, void* out, int32_t rows, int32_t cols, int32_t dtype,
                   void* stream);
/* w: T [n][k] (n % 16 == 0, k % 32 == 0) -> out: the same values as mobi_igemm_params.weight_tiled wants them ([n / 16][k / 32]
 * blocks of 1 KiB, layout above).  A load-time step of the sampling path; in training it runs after every optimizer update. */
int mobi_tile_weights(const void* w, void* out, int32_t n, int32_t k, void* stream);                          /* T [rows][cols] (row stride in elements) -> T [cols][rows] */
int32_t mobi_backward_partial_blocks(int64_t rows);        /* blocks of per-block partial sums the two calls below use */
/* out[c] = sum_rows dy[row][c]; partial: f32 [mobi_backward_partial_blocks(rows) + 64][cols] scratch. */
int mobi_colsum(const void* dy, int64_t row_stride, int64_t rows, int32_t cols, int32_t dtype, float* partial, float* out,
                void* stream);
typedef struct mobi_layernorm_bwd_params {
  const void* x; const void* dy;            /* T [rows][channels]; row strides in elements (0 = dense) */
  int64_t x_row_stride, dy_row_stride;
  const float* gamma; float eps;
  const void* dx_add;                       /* T [rows][channels] dense or NULL: added to dx (the residual branch's gradient) */
  void* dx;                                 /* T [rows][channels] dense */
  float* partial;                           /* f32 [mobi_backward_partial_blocks(rows) + 64][2][channels] scratch */
  float* dgamma_dbeta;                      /* f32 [2][channels]: d gamma, then d beta */
  int64_t rows; int32_t channels; int32_t dtype;
} mobi_layernorm_bwd_params;
int mobi_layernorm_bwd(const mobi_layernorm_bwd_params* p, void* stream);
/* pre: T [rows][2 inner] = [value | gate] (GEGLU.proj's output, un-fused); h = value * gelu_erf(gate): T [rows][inner]. */
int mobi_geglu_fwd(const void* pre, void* h, int64_t rows, int32_t inner, int32_t dtype, void* stream);
int mobi_geglu_bwd(const void* pre, const void* dh, void* dpre, int64_t rows, int32_t inner, int32_t dtype, void* stream);
/* GroupNorm (32 groups) (+ SiLU) backward, data gradient only (the UNet's GroupNorm parameters are frozen):
 * x, dy, dx (and dx_add, or NULL): T [image][hw][channels] dense.  Replaces autograd through GroupNorm32 + SiLU
 * (util.py:199-216, openaimodel.py:211-236) and Normalize (attention.py:77-78). */
size_t mobi_groupnorm_bwd_workspace_floats(int32_t images, int32_t hw, int32_t channels);
/* ws: mobi_groupnorm_bwd_workspace_floats() floats of scratch -> three coalesced passes (statistics, gradient sums, result), every
 * reduction in a fixed order; ws == NULL: one block per (image, group), four strided passes (any shape; the A/B partner). */
int mobi_groupnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, float eps, int32_t silu,
                       const void* dx_add, void* dx, int32_t images, int32_t hw, int32_t channels, int32_t dtype, float* ws,
                       void* stream);
/* out[image][y][x][c] = sum of the 2 x 2 pixels src[image][2y..][2x..][c]: the backward of F.interpolate(nearest, x2)
 * (openaimodel.py:116).  src: T [image][2h][2w][channels]. */
int mobi_sumpool2(const void* src, void* out, int32_t images, int32_t h, int32_t w, int32_t channels, int32_t dtype, void* stream);
/* out = a + b over n elements of T (two gradients meeting at a fork: a skip connection's consumers). */
int mobi_add(const void* a, const void* b, void* out, int64_t n, int32_t dtype, void* stream);
/* dx = dy * silu'(z), fp32 (the bbox embedder's MLP, ldm/modules/encoders/modules.py:77-83 of the reference). */
int mobi_silu_bwd_f32(const float* z, const float* dy, float* dx, int64_t n, void* stream);
/* One AdamW update of fp32 master parameters in place (torch.optim.AdamW, ddpm.py:1649 of the reference): decoupled weight
 * decay, bias-corrected moments; step counts from 1. */
int mobi_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                    float beta2, float eps, float weight_decay, int32_t step, void* stream);
typedef struct mobi_attention_bwd_params {
  const void* q; int64_t q_img_stride, q_row_stride;       /* T [image][tq][>= heads*dh], strides in elements */
  const void* k; int64_t k_img_stride, k_row_stride;       /* T [image][tk][..] */
  const void* v; int64_t v_img_stride, v_row_stride;
  const void* o; int64_t o_img_stride, o_row_stride;       /* the forward result (read only by the A/B form of the row term, D = do . o;
                                                              the default takes D = sum_j P dP from the pass's own P and dP) */
  const void* dout; int64_t dout_img_stride, dout_row_stride;
  void* dq; void* dk; void* dv;                            /* T [image][t][heads*dh] dense */
  float* lse; float* dvec;                                 /* f32 [image][heads][tq] scratch each */
  int32_t images, heads, dh, tq, tk;                       /* dh <= 160 */
  float scale; int32_t dtype;
  int32_t force_vector;                                    /* 1: the fp32 vector-ALU passes (any shape; A/B and tests); 0: the
                                                              matrix-core passes where they apply (dh % 8 == 0, 16-padded width
                                                              16 / 32 / 48 / 64 / 80 / 160, 16-byte aligned rows), else vector */
} mobi_attention_bwd_params;
int mobi_attention_bwd(const mobi_attention_bwd_params* p, void* stream);

/* Attention against a handful of context tokens (tk <= 8): the bbox adapter
 * (attention.py:237-243, tk = 2).  k, v are fp32 [image][tk][heads*dh]. */
typedef struct mobi_ctx_attention_params {
  const void* q; void* out;                   /* T [image][tq][heads*dh] */
  const float* k; const float* v;
  int32_t images, heads, dh, tq, tk;
  float scale;
  int32_t dtype;
} mobi_ctx_attention_params;
int mobi_ctx_attention(const mobi_ctx_attention_params* p, void* stream);

/* The bbox adapter of a transformer block as ONE pass over the tokens (attention.py:237-243 of the reference:
 *   x = x + connector(to_out(softmax(to_q(norm(x)) k^T * scale) v)),  k, v = to_k / to_v of TWO context tokens).
 * With two keys the softmax is a sigmoid of the score difference, and every matrix that touches the tokens can be
 * folded into per-image vectors (exact algebra, done once per context by the caller):
 *   score difference of head h:  scale * to_q(LN(x))_h . (k1 - k2)_h = rstd * (x . a_h - mean * sum(a_h)) + c_h
 *       a_h = gamma (.) (Wq_h^T (k1 - k2)_h) * scale,   c_h = beta . (Wq_h^T (k1 - k2)_h) * scale
 *   update:  x + b + sum_h sigmoid(.)_h * u_h,   u_h = W_h (v1 - v2)_h,   b = W v2 + bias   (W = connector o to_out)
 * mean / rstd are the LayerNorm statistics of the token (eps).  heads <= 8, channels % 8 == 0, <= 1536.
 * out may alias x (each token is read before it is written). */
typedef struct mobi_two_key_adapter_params {
  const void* x; void* out;                /* T [image][token][channels]                                       */
  int64_t x_img_stride, out_img_stride;    /* elements between images; 0 = dense                               */
  const float* a;                          /* f32 [image][heads][channels]                                     */
  const float* a_sum;                      /* f32 [image][heads]                                               */
  const float* c;                          /* f32 [image][heads]                                               */
  const float* u;                          /* f32 [image][heads][channels]                                     */
  const float* b;                          /* f32 [image][channels]                                            */
  int32_t images, rows_per_image, channels, heads;
  float eps;
  int32_t dtype;
  /* Optional second result: the LayerNorm of the RESULT rows as the cross-modal step reads it next (attention.py:
   * cross_modal_norm_camera on the even images, cross_modal_norm_lidar on the odd ones).  Image i goes to ln_out[i & 1]
   * at image index i >> 1 (T [images / 2][token][channels], dense) with ln_gamma / ln_beta [i & 1] (f32 [channels]).
   * ln_out[0] == NULL: none.  Served where mobi_two_key_adapter_fuses_ln(channels, images * rows_per_image) says so: every
   * launch of the default routing (MOBI_ERR_UNSUPPORTED otherwise: the caller then runs mobi_layernorm on the two halves). */
  void* ln_out[2];
  const float* ln_gamma[2];
  const float* ln_beta[2];
  float ln_eps;
} mobi_two_key_adapter_params;
int mobi_two_key_adapter(const mobi_two_key_adapter_params* p, void* stream);
int mobi_two_key_adapter_fuses_ln(int32_t channels, int64_t total_rows);   /* 1: this build / tuning writes ln_out for that width and images * rows_per_image */

/* Row softmax fp32 -> T (AttnBlock, model.py:189-190). */
int mobi_softmax_rows(const float* src, void* out, int64_t rows, int32_t cols, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------
 * Small dense layers on fp32 vectors (m <= 16 rows per call):
 *   out[m][n] = post( sum_k W[n][k] * pre(x[m][k]) + b[n] )
 * time_embed MLP and every ResBlock emb_layers (openaimodel.py:627-631,
 * 219-225, 874-875), attn2's to_v/to_out on the reference token
 * (attention.py:235), the bbox adapter's to_k/to_v.
 * ------------------------------------------------------------------------- */
enum { MOBI_ACT_NONE = 0, MOBI_ACT_SILU = 1, MOBI_ACT_GELU = 2 /* erf form, post only: the token mapper's MLP, xf.py:47-57 */ };
typedef struct mobi_skinny_linear_params {
  const float* x; int32_t m, k; int32_t x_row_stride;   /* elements */
  const void* weight;      /* T [n][k] */
  const float* bias;       /* [n] or NULL */
  float* out; int32_t n; int32_t out_row_stride;
  int32_t pre_act, post_act;
  int32_t dtype;
} mobi_skinny_linear_params;
int mobi_skinny_linear(const mobi_skinny_linear_params* p, void* stream);

/* out[m][n] = sum_k x[m][k] * w[n][k] (+ bias[n]), everything fp32, fp32 FMA chains (k ascending): the per-run folds of
 * the conditioning tokens into per-image vectors (BasicTransformerBlock, attention.py:237-243 of the reference: the two
 * bbox tokens' keys / values pushed through to_q^T and connector o to_out once per sampling run).  Row strides in
 * elements; any m, n, k. */
int mobi_linear_f32(const float* x, const float* w, const float* bias, float* out, int32_t m, int32_t n, int32_t k,
                    int32_t x_stride, int32_t w_stride, int32_t out_stride, void* stream);

/* LayerNorm over the last axis of fp32 rows (row strides in elements): the single-token mapper and its final_ln
 * (ldm/modules/encoders/xf.py:78-101, modules.py:153-168 of the reference). */
int mobi_layernorm_rows_f32(const float* x, const float* gamma, const float* beta, float* out, int32_t rows, int32_t cols,
                            int32_t x_stride, int32_t out_stride, float eps, void* stream);

/* out = x * sigmoid(1.702 x) elementwise on n T values (n % 8 == 0): the activation of the CLIP vision tower's MLP
 * (transformers' "quick_gelu"; FrozenCLIPImageEmbedder, ldm/modules/encoders/modules.py:142-180 -- SURVEY.md 8(f) row 1). */
int mobi_quick_gelu(const void* src, void* out, int64_t n, int32_t dtype, void* stream);

/* Sinusoidal timestep embedding (util.py:151-171); `freqs` is the fp32 table
 * exp(-ln(1e4) * i / half) computed by the host exactly as the reference does. */
int mobi_timestep_embedding(const int64_t* t, const float* freqs, float* out, int32_t n, int32_t half,
                            void* stream);

/* ---------------------------------------------------------------------------
 * Direct convolutions for the thin ends of the networks.
 * small_cin: fp32 NCHW sources (up to 3, concatenated on channels: the
 *   torch.cat([x, inpaint_image, inpaint_mask], 1) of ddim.py:170 feeding
 *   input_blocks.0; the VAE encoders' conv_in / conv_in_lidar,
 *   model.py:384-401; post_quant_conv, autoencoder.py:70) -> T channels-last
 *   or fp32 NCHW.  Total input channels <= 16.
 * small_cout: T channels-last -> fp32 NCHW with cout <= 8 (UNet `out.2`,
 *   openaimodel.py:832-836; VAE conv_out / conv_out_lidar / encoder conv_out,
 *   model.py:447-452,580-585), optional clamp (ddpm.py:1476,1504).
 * ------------------------------------------------------------------------- */
typedef struct mobi_conv_small_cin_params {
  const float* src[3]; int32_t c[3];      /* unused sources: NULL / 0 */
  int32_t batch, h, w;
  int32_t kh, kw, pad_h, pad_w;
  const float* weight;     /* f32 [cout][cin_total*kh*kw], k = (c*kh + ky)*kw + kx (OIHW flattened) */
  const float* bias;
  int32_t cout;
  void* out; int32_t out_f32_nchw;         /* 0: T channels-last, 1: fp32 NCHW */
  int32_t dtype;
} mobi_conv_small_cin_params;
int mobi_conv_small_cin(const mobi_conv_small_cin_params* p, void* stream);

typedef struct mobi_conv_small_cout_params {
  const void* src; int32_t cin;            /* T channels-last; cin % 8 == 0 */
  int32_t batch, h, w;
  int32_t kh, kw, pad_h, pad_w;
  const void* weight;      /* T [cout][kh*kw*cin], k = tap*cin + c */
  const float* bias;
  int32_t cout;
  float* out;              /* fp32 NCHW */
  int32_t clamp; float clamp_lo, clamp_hi;
  float in_scale;          /* unused (1.0f) */
  int32_t dtype;
} mobi_conv_small_cout_params;
int mobi_conv_small_cout(const mobi_conv_small_cout_params* p, void* stream);

/* ---------------------------------------------------------------------------
 * Sampler arithmetic on the fp32 latent state (NCHW, n elements flat).
 * ------------------------------------------------------------------------- */
/* p_sample_ddim, ddim.py:180-213 / get_x_prev_and_pred_x0, plms.py:199-214:
 *   e      = e_uncond + cfg_scale * (e_cond - e_uncond)   (e_uncond NULL: e = e_cond)
 *   pred   = (x - sqrt_one_minus_at * e) / sqrt(a_t)
 *   x_prev = sqrt(a_prev) * pred + sqrt(1 - a_prev - sigma^2) * e + sigma * noise * temperature
 * `e_out` (or NULL) receives e (PLMS keeps it in old_eps).
 * `coef_dev` (or NULL): DEVICE pointer to four floats {a_t, a_prev, sigma_t, sqrt_one_minus_at} that replace the
 * by-value fields -- a denoising step captured in a HIP graph reads its per-step coefficients from a buffer the
 * host refreshes between replays; same fp32 arithmetic, bit-identical results.
 * sigma_t != 0 with noise == NULL is MOBI_ERR_ARG (the reference always draws the noise, ddim.py:209); with
 * `coef_dev` the noise pointer is mandatory unless the caller promises eta == 0 by passing sigma_t == 0 by value. */
typedef struct mobi_ddim_step_params {
  const float* x; const float* e_cond; const float* e_uncond; const float* noise;
  float* x_prev; float* pred_x0; float* e_out;
  int64_t n;
  float cfg_scale, a_t, a_prev, sigma_t, sqrt_one_minus_at, temperature;
  const float* coef_dev;
} mobi_ddim_step_params;
int mobi_ddim_step(const mobi_ddim_step_params* p, void* stream);

/* out = c0*e0 + c1*e1 + c2*e2 + c3*e3, the Adams-Bashforth mixes of plms.py:219-233
 * (null pointers are skipped). */
int mobi_lincomb4(float* out, const float* e0, const float* e1, const float* e2, const float* e3,
                  float c0, float c1, float c2, float c3, int64_t n, void* stream);

/* Harness post-processing of the decoded range view (first slice): what
 * LatentDiffusion.log_data does to the clamped lidar sample on the host (ddpm.py:1527-1543 of the reference):
 *   depth:     inverse_depth_normalization (ldm/data/utils.py:560-580) per sample with that sample's
 *              (min_d, max_d):  [-alpha, alpha] -> [min_d, max_d],  [-1, -alpha) -> [-1, min_d),  (alpha, 1] -> (max_d, 1]
 *   intensity: clamp(-0.5 * log(1 - (x + 1) / 2) - 1, -1, 1)
 * sample: f32 [batch][2][hw] (channel 0 depth, 1 intensity, already clamped to [-1, 1]); depth_out / int_out: f32
 * [batch][hw] (either may be NULL).  The scalar combinations are passed as the fp32 values the reference's
 * Python-scalar arithmetic produces (2*alpha, alpha-1, 1-alpha computed in double, then cast), so the depth branch
 * is bit-identical to the reference's torch-CPU result; the logarithm is not (tolerance 1e-6 absolute).
 * object_norm = 0 copies the depth channel, int_norm = 0 the intensity channel. */
int mobi_range_denorm(const float* sample, const float* min_d, const float* max_d, float alpha, float two_alpha,
                      float alpha_m1, float one_m_alpha, int32_t object_norm, int32_t int_norm, float* depth_out,
                      float* int_out, int32_t batch, int32_t hw, void* stream);

/* DDPM.q_sample (ddpm.py:284-287) with extract_into_tensor (util.py:96-99): per image b,
 *   out[b] = sqrt_ac[t[b]] * x0[b] + sqrt_1m_ac[t[b]] * noise[b]
 * t: int64 [batch] on the device (no host read-back), tables: f32 [table_len]; un-fused mul / add as torch does. */
int mobi_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_ac,
                  const float* sqrt_1m_ac, float* out, int32_t batch, int32_t per_image, int32_t table_len,
                  void* stream);

/* ---------------------------------------------------------------------------
 * Harness post-processing on the device.  All buffers are device memory.
 * ------------------------------------------------------------------------- */
/* Range view of one batch: postprocess_range_depth_int (ldm/data/utils.py:471-505) =
 * LidarConverter.undo_default_transforms per sample (ldm/data/lidar_converter.py:436-485: avg-pool resize of the
 * (hc, wc) sample to (h0, width_crop[b]) -- F.avg_pool2d with kernel (hc / h0, wc / width_crop[b]), sequential fp32
 * window sums -- pasted over columns [crop_left[b] % w0, +width_crop[b]) of the original sweep, wrapping around),
 * then, when `planes` is given, the harness's paste of scripts/inference_test_bench.py:567-610:
 *   points = range2pcd(un-cropped depth, pitch, yaw)      (lidar_converter.py:122-172: (d + 1) / 2 * depth_max,
 *            valid iff depth_min < depth < depth_max, x = cos(yaw) cos(pitch) d, y = -sin(yaw) cos(pitch) d, z = sin(pitch) d)
 *   pred_mask = points_in_bbox_corners(points, box)        (ldm/data/box_np_ops.py:453-471, :736-771: a point is inside
 *            iff x nx + y ny + z nz + d < 0 for all six surfaces; planes = f32 [batch][6][4] = (nx, ny, nz, d), computed
 *            on the host from the 8 corners as surface_equ_3d does, :712-732)
 *   final = where(pred_mask | gt_mask, un-cropped sample, original)
 * sample_int / int_orig / int_unc / int_final may all be NULL (depth only); any output may be NULL. */
typedef struct mobi_range_paste_params {
  const float* sample_depth; const float* sample_int;       /* [batch][hc][wc] */
  const float* depth_orig; const float* int_orig;           /* [batch][h0][w0] */
  const float* pitch; const float* yaw;                     /* [batch][h0][w0] (needed with planes) */
  const uint8_t* gt_mask;                                   /* [batch][h0][w0] or NULL */
  const float* planes;                                      /* [batch][6][4] or NULL: un-crop only */
  const int32_t* crop_left; const int32_t* width_crop;      /* [batch] */
  float* depth_unc; float* int_unc; float* depth_final; float* int_final;   /* [batch][h0][w0] */
  uint8_t* pred_mask;                                       /* [batch][h0][w0] */
  int32_t batch, hc, wc, h0, w0;
  float depth_min, depth_max;                               /* LidarConverter.depth_interval = (1.4, 54) */
} mobi_range_paste_params;
int mobi_range_paste(const mobi_range_paste_params* p, void* stream);

/* Per-sample lidar error scores of LatentDiffusion.log_data (ddpm.py:1545-1590) for one (pred, gt) pair, without a
 * host read-back per score: pred / gt are avg-pooled and the two masks max-pooled to (pool_h, width_crop[b])
 * (pool_resize, lidar_converter.py:8-19); over the cells whose pooled mask == 1,
 *   out[b][region][0] = sqrt(mean((pred - gt)^2)),  [1] = lower median of |pred - gt| (torch.median),  [2] = cell count
 * region 0 = instance mask, 1 = box mask; an empty region gives NaN scores (the reference drops / propagates NaN).
 * pred / gt / inst_mask / box_mask: f32 [batch][h][w]; max_width = max over b of width_crop[b] (sizes the sort space). */
typedef struct mobi_lidar_metrics_params {
  const float* pred; const float* gt; const float* inst_mask; const float* box_mask;
  const int32_t* width_crop;
  float* out;                                               /* [batch][2][3] */
  int32_t batch, h, w, pool_h, max_width;
} mobi_lidar_metrics_params;
int mobi_lidar_metrics(const mobi_lidar_metrics_params* p, void* stream);

/* Dataset side: the object's range view for a whole batch in one launch (ldm/data/nuscenes.py:418-470 over
 * lidar_converter.py:387-434 tile / bbox_crop / resize, data/utils.py:537-557 depth_normalization):
 *   view[b][r][c] = sweep[b][nn(r: height <- h0)][(crop_left[b] + nn(c: width <- width_crop[b])) mod w0]   (nn = cv2's
 *   INTER_NEAREST index), depth through the object normalisation (alpha, [min_depth, max_depth] per sample) when
 *   object_norm, intensity ((i / 255) - 0.5) * 2 and, when int_norm, clamp(2 (1 - exp(-2 (x + 1))) - 1, -1, 1);
 *   range_data = [depth, intensity]; range_data_inpaint = range_data * edit_mask; inst_out = view of inst_orig.
 * Needs width_crop[b] <= width and h0 <= height (the enlarging case of `resize`; else MOBI_ERR_UNSUPPORTED). */
typedef struct mobi_range_prepare_params {
  const float* depth_orig; const float* int_orig; const float* inst_orig;      /* [batch][h0][w0]; inst may be NULL */
  const int32_t* crop_left; const int32_t* width_crop;                         /* [batch] (columns of the 3-sweep tiling) */
  const float* min_depth; const float* max_depth;                              /* [batch] (object_norm) */
  const float* edit_mask;                                                      /* [batch][1][height][width], 1 = keep */
  float* range_data; float* range_data_inpaint;                                /* [batch][2][height][width] */
  float* inst_out;                                                             /* [batch][1][height][width] or NULL */
  int32_t batch, h0, w0, height, width;
  float alpha;
  int32_t object_norm, int_norm;
} mobi_range_prepare_params;
int mobi_range_prepare(const mobi_range_prepare_params* p, void* stream);

/* Edit masks of a batch of projected boxes (ldm/data/utils.py:146-198): corners_xy int32 [batch][8][2] pixel coordinates
 * (already truncated, as before cv2.fillPoly); out f32 [batch][H][W] = 0 inside / on the outline of one of the six faces,
 * 1 elsewhere (may be NULL); stats int32 [batch][5] = {edit pixels, min x, max x, min y, max y}, pre-set by the caller to
 * {0, W, -1, H, -1} (may be NULL).  cv2's rasteriser is restated (centre inside the convex face or within half a pixel
 * of its outline); single outline pixels can differ from OpenCV's. */
int mobi_box_mask(const int32_t* corners_xy, float* out, int32_t* stats, int32_t batch, int32_t H, int32_t W, void* stream);

/* Dataset side, the camera view of a whole batch in one launch (ldm/data/nuscenes.py:495-594): frames u8 [batch][H][W][3]
 * RGB -> ((u8 / 255) - 0.5) / 0.5, edit mask from corners_xy (as mobi_box_mask; flipped where invert[b]), crop
 * (left, top, crop_w, crop_h) per sample, bilinear resize to (height, width) with align_corners = False and no
 * antialias (torchvision 0.11 tensor Resize): gt f32 [batch][3][height][width], mask f32 [batch][1][height][width],
 * inpaint = gt * mask. */
typedef struct mobi_image_prepare_params {
  const uint8_t* frames; const int32_t* corners_xy; const int32_t* invert; const int32_t* crop;
  float* gt; float* inpaint; float* mask;
  int32_t batch, H, W, height, width;
} mobi_image_prepare_params;
int mobi_image_prepare(const mobi_image_prepare_params* p, void* stream);

/* Camera paste-back of one sample (scripts/inference_test_bench.py:478-510):
 *   mobi_paste_patch    F.interpolate(patch, (crop_h, crop_w), bilinear, align_corners=False), (((x + 1) / 2) * 255)
 *                       -> uint8, RGB planes -> BGR bytes, written into frame[top : top + crop_h, left : left + crop_w]
 *                       (frame: u8 [H][W][3], zero-filled by the caller = np.zeros_like(image))
 *   mobi_gaussian_blur  cv2.GaussianBlur(mask, (ksize, ksize), sigma) as a separable pass pair with BORDER_REFLECT_101;
 *                       kern: f32 [ksize] taps (the host computes cv2.getGaussianKernel's formula); tmp: f32 [H][W]
 *   mobi_blend_frame    image_recon = m * uint8(image) + (1 - m) * image_pred, f32 [H][W][3] BGR; image: f32 [3][H][W]
 *                       RGB in [-1, 1] as the batch holds it */
int mobi_paste_patch(const float* patch, int32_t hs, int32_t ws, uint8_t* frame, int32_t H, int32_t W, int32_t top,
                     int32_t left, int32_t crop_h, int32_t crop_w, void* stream);
int mobi_gaussian_blur(const float* src, float* tmp, float* dst, int32_t H, int32_t W, const float* kern, int32_t ksize,
                       void* stream);
int mobi_blend_frame(const float* mask_blur, const float* image, const uint8_t* pred, float* out, int32_t H, int32_t W,
                     void* stream);

/* ddim.py:145-148 with q_sample (ddpm.py:284-287):
 *   img = (sa[t]*x0 + s1ma[t]*noise) * mask + (1 - mask) * img
 * mask is [batch][1][hw] broadcast over `channels`. */
int mobi_mask_blend(float* img, const float* x0, const float* noise, const float* mask,
                    float sqrt_ac_t, float sqrt_1m_ac_t, int32_t batch, int32_t channels, int32_t hw,
                    void* stream);

/* DiagonalGaussianDistribution.sample with supplied noise and the latent scale
 * (distributions.py:25-37, ddpm.py:601-608): moments f32 NCHW [b][2c][hw] ->
 * z f32 [b][c][hw], written at channel offset `out_c_off` of an NCHW tensor with
 * `out_c_total` channels (the torch.cat of ddpm.py:1021). */
int mobi_posterior_sample(const float* moments, const float* noise, float* out, int32_t batch, int32_t c,
                          int32_t hw, int32_t out_c_total, int32_t out_c_off, float scale, void* stream);

/* F.interpolate(mode="nearest") on fp32 NCHW (ddpm.py:1020): index-only. */
int mobi_nearest_resize(const float* src, float* out, int32_t planes, int32_t hin, int32_t win,
                        int32_t hout, int32_t wout, int32_t out_plane_stride, void* stream);

/* torch.cat of up to three fp32 NCHW sources on the channel axis (ddim.py:170; a VAE's image
 * input) -> T channels-last zero-padded to c_pad channels (c_pad % 32 == 0 feeds mobi_igemm,
 * whose matching weights are zero-padded on the input-channel axis). */
int mobi_pack_nchw_sources(const float* s0, const float* s1, const float* s2, int32_t c0, int32_t c1, int32_t c2,
                           int32_t batch, int32_t hw, int32_t c_pad, void* out, int32_t dtype, void* stream);

/* Layout converters at the API boundary (tests, module-level calls). */
int mobi_nchw_f32_to_nhwc(const float* src, void* out, int32_t batch, int32_t c, int32_t hw, int32_t dtype,
                          void* stream);
int mobi_nhwc_to_nchw_f32(const void* src, float* out, int32_t batch, int32_t c, int32_t hw, int32_t dtype,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOBI_ENGINE_H_ */
