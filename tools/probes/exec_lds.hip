// Probe: does an EXEC-masked lane of buffer_load_dwordx4 ... lds leave its 16 bytes of LDS untouched?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/exec_lds.hip -o /tmp/exec_lds && /tmp/exec_lds
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const unsigned* p, unsigned* o, int nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4 * 64];
  for (int i = threadIdx.x; i < 4 * 64; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p), 0, nbytes, 0x00020000);
  if ((threadIdx.x % 7) < 5)                               // lanes 5, 6 of every 7 masked off
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, threadIdx.x * 16u, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) o[i] = lds[i];
}
int main() {
  unsigned *p, *o, h[256], src[1024];
  for (int i = 0; i < 1024; ++i) src[i] = 0x1000 + i;
  hipMalloc(&p, sizeof(src)); hipMalloc(&o, sizeof(h));
  hipMemcpy(p, src, sizeof(src), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, p, o, (int)sizeof(src));
  hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  int ok_active = 0, stale_masked = 0, n_active = 0, n_masked = 0;
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
    if ((l % 7) < 5) { ++n_active; ok_active += h[l * 4 + j] == 0x1000u + l * 4 + j; }
    else { ++n_masked; stale_masked += h[l * 4 + j] == 0xdeadbeefu; }
  }
  printf("active lanes: %d of %d dwords landed at lane * 16; masked lanes: %d of %d dwords untouched\n", ok_active, n_active,
         stale_masked, n_masked);
  printf("lane 5: %x %x %x %x   lane 6: %x   lane 7: %x\n", h[20], h[21], h[22], h[23], h[24], h[28]);
  return 0;
}
