// Probe: how fast does a CU accept `buffer_load_dwordx4 ... lds` requests (1 KiB per wave instruction), by the shape of the
// global side of a request, source resident in L2:
//   rows128   8 rows x 128 B   (the ping-pong kernel's k-tile rows: 64-deep k)
//   rows64   16 rows x  64 B   (the ring kernels' 32-deep k-steps)
//   rows32   32 rows x  32 B
//   linear    1 KiB contiguous
// W waves per CU (one block per CU), every wave issues N requests back to back, waits once; cycles per request and
// bytes per clock and CU.   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_dma_rate.hip -o /tmp/lds_dma_rate && /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int ROWB>
__global__ __launch_bounds__(512) void k(const unsigned char* p, unsigned long long* out, int bytes, int row_stride, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(p), 0, bytes, 0x00020000);
  constexpr int LPR = ROWB / 16;                             // lanes per row
  const int row = lane / LPR, col = (lane % LPR) * 16;
  // each wave walks its own rows; block b starts at a different place; the whole footprint stays inside `bytes` (L2-sized)
  unsigned base = (unsigned)(((blockIdx.x * 8 + wave) * 1024 * 64) % (bytes / 2)) + row * row_stride + col;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + wave * 4096 + (i & 3) * 1024), 16, base, 0, 0, 0);
    base += (64 / LPR) * row_stride;                         // next group of rows
    if (base + 64 * row_stride > (unsigned)bytes) base -= bytes / 2;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
}

int main() {
  const int bytes = 64 << 20, blocks = 256, n = 256;
  unsigned char* p; unsigned long long* o;
  hipMalloc(&p, bytes); hipMemset(p, 1, bytes); hipMalloc(&o, blocks * 8 * 2 * sizeof(unsigned long long));
  unsigned long long* h = (unsigned long long*)malloc(blocks * 8 * 2 * sizeof(unsigned long long));
  struct { const char* name; int rowb; int stride; } pats[] = {{"rows128 (stride 2560)", 128, 2560}, {"rows64  (stride 2560)", 64, 2560},
      {"rows64  (stride 640)", 64, 640}, {"rows32  (stride 2560)", 32, 2560}, {"linear 1 KiB", 128, 128}, {"rows64 packed (stride 64)", 64, 64}};
  for (int w : {4, 8}) {
    for (auto& pt : pats) {
      for (int rep = 0; rep < 2; ++rep) {
        if (pt.rowb == 128) hipLaunchKernelGGL((k<128>), dim3(blocks), dim3(64 * w), 32768, 0, p, o, bytes, pt.stride, n);
        else if (pt.rowb == 64) hipLaunchKernelGGL((k<64>), dim3(blocks), dim3(64 * w), 32768, 0, p, o, bytes, pt.stride, n);
        else hipLaunchKernelGGL((k<32>), dim3(blocks), dim3(64 * w), 32768, 0, p, o, bytes, pt.stride, n);
        hipDeviceSynchronize();
      }
      hipMemcpy(h, o, blocks * 8 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double issue = 0, total = 0; int cnt = 0;
      for (int b = 0; b < blocks; ++b) for (int ww = 0; ww < w; ++ww) { issue += h[(b * 8 + ww) * 2]; total += h[(b * 8 + ww) * 2 + 1]; ++cnt; }
      issue /= cnt; total /= cnt;
      printf("%d waves/CU  %-28s issue %7.1f cycles per request and wave, all landed after %8.0f cycles: %5.1f B per clock and CU\n",
             w, pt.name, issue / n, total, (double)w * n * 1024 / total);
    }
  }
  return 0;
}
