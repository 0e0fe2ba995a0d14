// Probe: does an out-of-range buffer_load_dwordx4 ... lds write zeros into LDS, or leave the old bytes?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/oob_lds.hip -o /tmp/oob_lds && /tmp/oob_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const unsigned* p, unsigned* o, int nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4 * 64 * 2];
  for (int i = threadIdx.x; i < 4 * 64 * 2; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p), 0, nbytes, 0x00020000);
  const unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;      // odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, off, 0, 0, 0);
  // second instruction, soffset carries part of the address, odd lanes out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + 256), 16, off, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) o[i] = lds[i];
}
int main() {
  unsigned *p, *o, h[512], src[1024];
  for (int i = 0; i < 1024; ++i) src[i] = 0x1000 + i;
  hipMalloc(&p, sizeof(src)); hipMalloc(&o, sizeof(h));
  hipMemcpy(p, src, sizeof(src), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, (int)sizeof(src));
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  printf("lane0 (in range): %x %x %x %x\n", h[0], h[1], h[2], h[3]);
  printf("lane1 (OOB):      %x %x %x %x\n", h[4], h[5], h[6], h[7]);
  printf("2nd, lane0 (soffset 16): %x %x %x %x\n", h[256], h[257], h[258], h[259]);
  printf("2nd, lane1 (OOB):        %x %x %x %x\n", h[260], h[261], h[262], h[263]);
  int zeros = 0, stale = 0;
  for (int l = 1; l < 64; l += 2) for (int j = 0; j < 4; ++j) { zeros += h[l * 4 + j] == 0; stale += h[l * 4 + j] == 0xdeadbeefu; }
  printf("OOB dwords: %d zero, %d stale of %d\n", zeros, stale, 32 * 4);
  return 0;
}
