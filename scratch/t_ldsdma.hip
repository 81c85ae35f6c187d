#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
using i32x4 = int __attribute__((ext_vector_type(4)));
__global__ void k(const int* src, int* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) int lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) lds[i] = -7;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, nbytes, 0x00020000);
  // lane l loads 16 B from byte offset: even lanes in range, odd lanes OOB
  unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = lds[i];
}
int main() {
  int *s, *o; hipMalloc(&s, 4096); hipMalloc(&o, 4096);
  int h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1;
  hipMemcpy(s, h, 4096, hipMemcpyHostToDevice);
  k<<<1, 64>>>(s, o, 4096);
  int r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 6; ++l) printf("lane %d: %d %d %d %d\n", l, r[l*4], r[l*4+1], r[l*4+2], r[l*4+3]);
  return 0;
}
