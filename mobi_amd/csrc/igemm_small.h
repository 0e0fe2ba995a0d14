// The small-problem form of mobi_igemm's 1 x 1 case (csrc/igemm_small.hip): coalesced loads, a wave-private LDS transposition,
// k split over the four waves of a block, one fixed-order sum through LDS.  Launched by igemm.hip's plan (`small_tile`).
#pragma once
#include <hip/hip_runtime.h>

namespace mobi {

struct SmallGemmArgs {
  const void* src0; const void* src1;   // T [image][pixel][c0 | c1]: channels [0, c0) of a pixel from src0, [c0, c0 + c1) from src1
  int c0, c1;
  int taps;                             // 1: 1 x 1;  9: 3 x 3, pad 1, stride 1 (k = tap * (c0 + c1) + channel)
  int hin, win;                         // image height / width (hw = hin * win)
  int hw;                               // pixel rows per image
  int img_pix_stride;                   // pixels between images of the sources
  const void* weight;                   // T [N][K], K = taps * (c0 + c1)
  int M, N, K;
  const float* bias; const float* rowvec; int rowvec_stride;
  const void* residual; long long res_img_stride;
  void* out; long long out_img_stride;  // elements between images of the output / residual (rows are N wide)
  int out_f32;
  float scale;
  int tiles_m, tiles_n;
  int n_major;                          // consecutive blocks walk the row tiles of one column tile (weights outweigh activations)
};

// 32 x 32 output tiles (tiles_m, tiles_n count those).  dtype: MOBI_F16 | MOBI_BF16.
int launch_small_gemm(const SmallGemmArgs& a, int dtype, hipStream_t st);

}  // namespace mobi
