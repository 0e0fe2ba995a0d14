// Development / A-B knobs of the engine.  The MOBI_* environment variables are read ONCE, at the first launch
// (never per launch: a denoising step is ~600 launches), into this table; `mobi_tuning_reload()` re-reads them
// (tests and the tools/ A-B scripts flip a variable and call it).  -1 = unset (the library's own choice).
#pragma once
namespace mobi {
struct Tuning {
  int persist_blocks;   // MOBI_IGEMM_PERSIST_BLOCKS  persistent grid size (tests: few blocks walk many tiles)
  int pp_split;         // MOBI_IGEMM_PP_SPLIT        0: split-K launches stay on the register-staged kernel
  int wm;               // MOBI_IGEMM_WM              2 | 4: force the 128- / 256-pixel block
  int fast;             // MOBI_IGEMM_FAST            0: generic gather addressing
  int glds;             // MOBI_IGEMM_GLDS            0: register-staged loads instead of direct-to-LDS
  int lin;              // MOBI_IGEMM_LIN             0: no linear window stepping (chunk-major k order)
  int epi_direct;       // MOBI_IGEMM_EPI_DIRECT      0: LDS-staged epilogue
  int pp;               // MOBI_IGEMM_PP              0: lockstep schedule instead of ping-pong
  int sm;               // MOBI_IGEMM_SM              0: 128-pixel tiles on the register-staged kernel (A/B)
  int wide;             // MOBI_IGEMM_WIDE            0: never the eight-wave ring kernel; 2 / 1: always its 256 / 128 x 320 tiles (A/B)
  int wide128;          // MOBI_IGEMM_WIDE128         1: route the 128 x 320 tiles where they fill the chip (A/B; measured slower)
  int ring_direct;      // MOBI_IGEMM_RING_DIRECT     0: 256 x 320 ring tiles always through the LDS-staged epilogue (A/B)
  int sm_direct;        // MOBI_IGEMM_SM_DIRECT       0: 128 x 160 ring tiles always through the LDS-staged epilogue (A/B)
  int w_tiled;          // MOBI_IGEMM_WTILED          0: ring kernels fetch weights as row segments even when request images are given (A/B)
  int n_major;          // MOBI_IGEMM_N_MAJOR         0: igemm work lists always walk the channel tiles of a pixel tile first (A/B); 1: never
  int sm64;             // MOBI_IGEMM_SM64            0: 128-pixel tiles always on the 32-deep-step ring kernel (A/B)
  int split_target;     // MOBI_IGEMM_SPLIT_TARGET    blocks the split-K plan aims at (default 512; sweeps)
  int fused_split;      // MOBI_IGEMM_FUSED_SPLIT     0: split-K launches always finish in igemm_splitk_reduce_kernel, also with `sync` given (A/B)
  int split_longk;      // MOBI_IGEMM_SPLIT_LONGK     0: no extra doubling of the splits on one-round launches with >= 80 k-tiles per split (A/B)
  int split_round4;     // MOBI_IGEMM_SPLIT_ROUND4    0: the split plan of round 4 for 5 .. 8 splits (A/B); default: 4 where the reduce's rounds say so
  int attn_bwd_exact_d; // MOBI_ATTN_BWD_EXACT_D      0: the softmax backward's row term from the STORED output (do . o; A/B); default: sum_j P dP
  int small;            // MOBI_IGEMM_SMALL           0: never the small-problem kernel (igemm_small.hip); 32: every eligible launch (A/B)
  int small_mflop;      // MOBI_IGEMM_SMALL_MFLOP     largest 2 M N K (MFLOP) of a 1 x 1 launch routed to it (sweeps)
  int small_conv_m;     // MOBI_IGEMM_SMALL_CONV_M    most output pixels of a 3 x 3 launch routed to it (sweeps)
  int tka_mfma;         // MOBI_TKA_MFMA              0: two-key adapter on the vector-ALU kernel; 2 / 1: the LDS-tile kernel to C = 320 / 640 (A/B)
  int attn_nw;          // MOBI_ATTN_NW               4 | 8: waves per attention block
  int attn_sp;          // MOBI_ATTN_SP               1: software-pipelined attention kernel (dh 33..48)
  int tka_rows;         // MOBI_TKA_ROWS              rows per block of the register two-key adapter kernel (sweeps)
  int attn_nw8_blocks;  // MOBI_ATTN_NW8_BLOCKS       8-wave attention blocks from this many blocks on (default 256; 1024 = rounds 1-3; A/B)
  int attn_h16;         // MOBI_ATTN_H16              1: dh = 40 attention runs the P.V of its last (3/4 padded) channel block on MFMA 16x16x32 (A/B: measured slower)
  int attn_xcd;         // MOBI_ATTN_XCD              0: attention workgroups in hardware order (A/B of the XCD-aware map)
  int cout_mfma;        // MOBI_COUT_MFMA             0: few-output-channel convolutions on the one-wave-per-pixel kernel (A/B)
  int skinny_mfma;      // MOBI_SKINNY_MFMA           0: fp32-row linears on the vector-ALU kernel (A/B)
  int gn_coop;          // MOBI_GN_COOP               1: GroupNorm as pixel chunks meeting through memory wherever the geometry fits; 0: never (A/B)
  int gn_fused;         // MOBI_GN_FUSED              0: two-launch GroupNorm; 1: one launch, slab in LDS, where it fits (A/B)
  int gn_split_pw4;     // MOBI_GN_SPLIT_PW4          1: split-K slabs also into the 8-byte-piece geometry of the register GroupNorm (A/B; default: 16-byte pieces only)
  int attn_v3;          // MOBI_ATTN_V3               0: V row-major launches stay on attention_kernel (A/B); development build: 2 / 3 = the
                        //                            software-pipelined variants of attention_rows_kernel
};
const Tuning& tuning();
}  // namespace mobi
