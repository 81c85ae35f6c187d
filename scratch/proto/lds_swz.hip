// LDS bank-conflict microbenchmark for the GEMM fragment reads: 64-byte rows (32 bf16), ds_read_b128, lane -> (row l15,
// 16-byte chunk l4), slot = chunk ^ f(row).  Which f is conflict-free for this access pattern?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(int* out, long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[64 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int i = threadIdx.x; i < 64 * 1024 / 4; i += 512) ((int*)smem)[i] = i;
  __syncthreads();
  int f;
  if (MODE == 0) f = (l15 >> 2) & 3;
  else if (MODE == 1) f = (l15 >> 1) & 3;
  else if (MODE == 2) f = l15 & 3;
  else if (MODE == 3) f = ((l15 >> 1) & 3) ^ ((l15 >> 3) & 1);
  else f = 0;
  const uint32_t base = (uint32_t)(wave * 128 + l15) * 64u + (uint32_t)((l4 ^ f) << 4);
  i32x4 acc = {0, 0, 0, 0};
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
    i32x4 v0, v1, v2, v3, v4, v5, v6, v7;
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                 "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
                 "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3), "=v"(v4), "=v"(v5), "=v"(v6), "=v"(v7) : "v"(base) : "memory");
    acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  long long c1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = c1 - c0;
  out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  int* out; long long* cyc;
  hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 4000;
  for (int mode = 0; mode < 5; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, out, cyc, iters); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, out, cyc, iters); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, out, cyc, iters); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 0, 0, out, cyc, iters); break;
        default: hipLaunchKernelGGL(k<4>, dim3(256), dim3(512), 0, 0, out, cyc, iters); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double bytes = 256.0 * 8 * iters * 8 * 1024;     // per launch: 256 CUs x 8 waves x iters x 8 reads x 1 KiB
    printf("mode %d: %.3f ms, %.1f B/clk/CU at 2.4 GHz, wave0 cycles/read %.1f\n", mode, ms, bytes / 256 / (ms * 1e-3) / 2.4e9,
           (double)c / (iters * 8.0));
  }
  return 0;
}
