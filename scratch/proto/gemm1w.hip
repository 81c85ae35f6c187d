// Prototype (NOT part of the product): NT GEMM C[M][N] = A[M][K] . W[N][K]^T, bf16, one wave per SIMD.
// Workgroup = 4 waves, tile 256 x 256, wave tile 128 x 128 (64 accumulators of v_mfma_f32_16x16x32_bf16 = 256 registers,
// the kernel owns the whole 512-register file), LDS-DMA ring of 4 stages x 32 KiB, fragments double-buffered in registers:
// the reads of K-step k+1 are issued in front of the 64 MFMAs of K-step k.  LDS reads per K-step: 4 x 16 KiB = 64 KiB
// (gemm_big's 8 waves of 128 x 64: 96 KiB).  Question: does the lower LDS traffic beat the lost ping-pong overlap?
#include "../../diffews_amd/csrc/common.h"
using namespace dfw;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

extern "C" __global__ __launch_bounds__(256, 1) void gemm1w_kernel(const char* A, const char* W, char* C, int M, int N, int K,
                                                                    uint32_t a_bytes, uint32_t w_bytes) {
  constexpr int S = 4, RB = 64, HALF = 256 * RB, STAGE = 2 * HALF;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, l4 = lane >> 4;
  const uint32_t lds0 = lds_addr(smem);
  const int ntn = N / 256;
  const int m0 = ((int)blockIdx.x / ntn) * 256, n0 = ((int)blockIdx.x % ntn) * 256;
  const u32x4 ra = make_srd(A, a_bytes), rw = make_srd(W, w_bytes);
  const int nk = K / 32;
  // loader: 32 instructions per stage (16 A + 16 W), 8 per wave; instruction g covers rows 16 g .. 16 g + 15 of A or W
  const int kc = (lane & 3) ^ ((lane >> 4) & 3);
  uint32_t a_off[4], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = 16 * (wave + 4 * i) + (lane >> 2);
    a_off[i] = (uint32_t)(((size_t)(m0 + r) * K + kc * 8) * 2);
    w_off[i] = (uint32_t)(((size_t)(n0 + r) * K + kc * 8) * 2);
  }
  int ld_k = 0;
  auto issue = [&]() __attribute__((always_inline)) {
    const uint32_t dst = lds0 + (uint32_t)(ld_k & (S - 1)) * STAGE;
    const uint32_t ko = (uint32_t)ld_k * 64u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(ra, a_off[i] + ko, dst + (uint32_t)(wave + 4 * i) * 1024u);
      dma16(rw, w_off[i] + ko, dst + HALF + (uint32_t)(wave + 4 * i) * 1024u);
    }
    ++ld_k;
  };
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t ra6 = (uint32_t)(wm * 128 + l15) * RB + (uint32_t)((l4 ^ ((l15 >> 2) & 3)) << 4);
  const uint32_t rw6 = (uint32_t)(wn * 128 + l15) * RB + (uint32_t)((l4 ^ ((l15 >> 2) & 3)) << 4);
  // A fragments single-buffered and refilled in place (row block i is re-read for K-step k+1 right after its 8 MFMAs of
  // K-step k), W fragments double-buffered: 32 + 64 registers beside the 256 accumulators.
  bf16x8 fa[8], fw[2][8];
  auto read_w = [&](int buf, int slot) __attribute__((always_inline)) {
    const char* bw = smem + slot * STAGE + HALF + rw6;
#pragma unroll
    for (int j = 0; j < 8; ++j) fw[buf][j] = __builtin_bit_cast(bf16x8, *(const i32x4*)(bw + j * 16 * RB));
  };
  auto read_a = [&](int i, int slot) __attribute__((always_inline)) {
    fa[i] = __builtin_bit_cast(bf16x8, *(const i32x4*)(smem + slot * STAGE + ra6 + i * 16 * RB));
  };
  auto step = [&](int cur, int nslot) __attribute__((always_inline)) {
    read_w(cur ^ 1, nslot);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[cur][j], fa[i], acc[i][j], 0, 0, 0);
      read_a(i, nslot);
    }
  };
  // Branch-free steady state: DMA issues past the last stage read beyond the buffers (zeros into slots nobody reads
  // any more), so every wait is vmcnt(16) = two younger stages in flight; nk is even.
#pragma unroll
  for (int i = 0; i < S - 1; ++i) issue();
  wait_vm<16>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_w(0, 0);
#pragma unroll
  for (int i = 0; i < 8; ++i) read_a(i, 0);
  auto sync_next = [&]() __attribute__((always_inline)) {     // before K-step k: stage k+1 complete, slot of k-1 free
    wait_vm<16>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
  };
  for (int k = 0; k < nk; k += 2) {
    sync_next();
    step(0, (k + 1) & (S - 1));
    sync_next();
    step(1, (k + 2) & (S - 1));
  }
  wait_vm<0>();
  // plain epilogue: D[i][j]: row (A side, m) = lane & 15 within block i ... mfma(fw, fa): D[n16 rows][m16 cols]:
  // lane holds col = lane & 15 (m), rows 4 * (lane >> 4) + e (n)
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wm * 128 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int n = n0 + wn * 128 + j * 16 + 4 * l4;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      *(i32x2*)(C + ((size_t)m * N + n) * 2) = pack4<__bf16>(v);
    }
  }
}

extern "C" int proto_gemm1w(const void* A, const void* W, void* C, int M, int N, int K, void* stream) {
  if (M % 256 || N % 256 || K % 64) return -1;   // nk even
  const size_t lds = 4 * 2 * 256 * 64;
  (void)hipFuncSetAttribute((const void*)gemm1w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(gemm1w_kernel, dim3((M / 256) * (N / 256)), dim3(256), lds, (hipStream_t)stream, (const char*)A, (const char*)W,
                     (char*)C, M, N, K, (uint32_t)((size_t)M * K * 2), (uint32_t)((size_t)N * K * 2));
  return (int)hipGetLastError();
}
