// Prototype: 256 x 256 x 64 NT GEMM on the guide's 8-phase schedule (cdna_hip_programming.md section 5, 'The 256^2 8-phase
// template'), written from its description: 8 waves (2 M x 4 N), wave tile 128 x 64, v_mfma_f32_16x16x32, 128-byte LDS rows
// (full cache lines per DMA lane group), LDS-DMA half-tiles (128 rows x 64 k = 16 KiB = 2 wave-instructions per wave), two
// buffers per half-tile (128 KiB), four 16-MFMA phases per K-tile, wave groups (wr = 0 / 1) staggered by one barrier, counted
// vmcnt once per K-tile.  Variants are template switches so that one process can A/B them.
//
// C[m][n] = sum_k A[m][k] W[n][k]; M, N multiples of 256, K a multiple of 64, K / 64 >= 2.
//
// Per K-tile t (buffer t & 1), per wave:   quadrant (s, u): rows 64 s .. +63 of the wave's 128, columns 32 u .. +31 of its 64
//   P0  read A[s0] (8) B[u0] (4) | issue A0(t+1) | lgkm0 | bar | 16 MFMA (s0,u0) | bar
//   P1  read B[u1] (4)           | issue A1(t+1) | lgkm0 | bar | 16 MFMA (s0,u1) | bar
//   P2  read A[s1] (8)           | issue B0(t+2) | lgkm0 | bar | 16 MFMA (s1,u1) | bar
//   P3                           | issue B1(t+2) | vmcnt(4) | bar | 16 MFMA (s1,u0) | bar
// RAW: the wait of P3(t) (every wave, before its first barrier of the phase) leaves only B0(t+2), B1(t+2) outstanding, so
// A(t+1) and B(t+1) have landed; the first reader is P0(t+1), one barrier later for either group.
// WAR: the reads of a phase have RETURNED (lgkmcnt(0)) before the wave's first barrier of that phase; B(t) is last read in P1(t)
// and refilled from P2(t) on, A(t) last read in P2(t) and refilled from P0(t+1) on.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

using f32x4 = float __attribute__((ext_vector_type(4)));
using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x2 = int __attribute__((ext_vector_type(2)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));
using u32x4 = unsigned __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 make_srd(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void dma16(u32x4 srd, uint32_t voff, uint32_t lds_byte) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_byte)), "v"(voff), "s"(srd) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ bf16x8 as8(i32x4 r) { return __builtin_bit_cast(bf16x8, r); }

// VAR bit 0: no stagger (both groups in phase)      bit 1: lgkmcnt(0) after the barrier instead of before it
// VAR bit 2: 2-D blocked tile -> XCD mapping (4 x 8 tiles per XCD) instead of a contiguous row-major range
// VAR bit 3: no s_setprio around the MFMA clusters
// VAR bit 4: ABLATION no MFMAs   bit 5: ABLATION no DMA issue in the loop   bit 6: ABLATION no fragment reads   (results wrong)
// VAR bit 7: two 32-MFMA phases per K-tile instead of four 16-MFMA phases (half the barriers)
template <int VAR>
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(const __bf16* __restrict__ A, const __bf16* __restrict__ W, __bf16* __restrict__ C,
                                                        int M, int N, int K) {
  constexpr bool NOSTAG = VAR & 1, LGKM_AFTER = VAR & 2, XCD2D = VAR & 4, NOPRIO = VAR & 8;
  constexpr bool NOMFMA = VAR & 16, NODMA = VAR & 32, NOREAD = VAR & 64, COARSE = VAR & 128;
  constexpr int HT = 16384;                              // half-tile bytes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // slot(X, buf) = X * 32768 + buf * 16384,  X = 0: A rows 0-127, 1: A rows 128-255, 2: W rows 0-127, 3: W rows 128-255
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l15 = lane & 15, l4 = lane >> 4;
  const uint32_t lds0 = lds_addr(smem);

  const int ntn = N >> 8, ntm = M >> 8;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3, per = (ntm * ntn) >> 3;
    if constexpr (XCD2D) {
      // XCD x owns a block of tiles; needs ntm % 4 == 0 and ntn % 8 == 0 (or falls back)
      if ((ntm & 3) == 0 && (ntn & 7) == 0 && per % 32 == 0) {
        const int blk = xcd * (per / 32) + j / 32, jj = j & 31;          // 4 x 8 blocks
        const int bpr = ntn >> 3;
        tile = ((blk / bpr) * 4 + (jj >> 3)) * ntn + (blk % bpr) * 8 + (jj & 7);
      } else tile = xcd * per + j;
    } else tile = xcd * per + j;
  }
  const int tm = tile / ntn, tn = tile - tm * ntn;
  const int m0 = tm << 8, n0 = tn << 8;
  const int nkt = K >> 6;

  const u32x4 ra = make_srd(A, (uint32_t)((size_t)M * K * 2));
  const u32x4 rw = make_srd(W, (uint32_t)((size_t)N * K * 2));

  // ---- loader: wave-instruction j (0, 1) of this wave covers rows (j * 8 + wave) * 8 .. + 8 of a half-tile; lane -> row
  // lane >> 3, LDS slot lane & 7, source chunk slot ^ f(row), f(row) = (row >> 1) & 7 = ((wave & 1) * 4 + (lane >> 4)) & 7
  const int lrow = lane >> 3;
  const int kc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  uint32_t a_off[2][2], w_off[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = h * 128 + (j * 8 + wave) * 8 + lrow;
      a_off[h][j] = (uint32_t)(((size_t)(m0 + r) * K + kc * 8) * 2);
      w_off[h][j] = (uint32_t)(((size_t)(n0 + r) * K + kc * 8) * 2);
    }
  auto issue_a = [&](int h, int t, int buf) __attribute__((always_inline)) {
    if constexpr (NODMA) { if (t >= 2) return; }
    const uint32_t dst = lds0 + (uint32_t)(h * 32768 + buf * HT) + (uint32_t)wave * 1024u, ko = (uint32_t)t * 128u;
    dma16(ra, a_off[h][0] + ko, dst);
    dma16(ra, a_off[h][1] + ko, dst + 8192);
  };
  auto issue_w = [&](int h, int t, int buf) __attribute__((always_inline)) {
    if constexpr (NODMA) { if (t >= 2) return; }
    const uint32_t dst = lds0 + (uint32_t)((2 + h) * 32768 + buf * HT) + (uint32_t)wave * 1024u, ko = (uint32_t)t * 128u;
    dma16(rw, w_off[h][0] + ko, dst);
    dma16(rw, w_off[h][1] + ko, dst + 8192);
  };

  // ---- fragment read bases: row l15, chunk (l4 + 4 kh) ^ f, f = l15 >> 1; kh = 1 flips bit 6 of the byte offset
  const uint32_t fb = (uint32_t)(l15 * 128 + ((l4 ^ (l15 >> 1)) << 4));
  const uint32_t abase0 = (uint32_t)(wr * 32768) + fb, abase1 = abase0 ^ 64u;
  const uint32_t wbase0 = (uint32_t)((2 + (wc >> 1)) * 32768 + (wc & 1) * 64 * 128) + fb, wbase1 = wbase0 ^ 64u;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[4][2], fw0[2][2], fw1[2][2];

  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lgkm0 = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto read_a = [&](int s, int buf) __attribute__((always_inline)) {
    if constexpr (NOREAD) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(fa[i][0])); asm volatile("" : "+v"(fa[i][1])); }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[i][0] = as8(*(const i32x4*)(smem + abase0 + buf * HT + (64 * s + 16 * i) * 128));
      fa[i][1] = as8(*(const i32x4*)(smem + abase1 + buf * HT + (64 * s + 16 * i) * 128));
    }
  };
  auto read_w = [&](bf16x8 (&fw)[2][2], int u, int buf) __attribute__((always_inline)) {
    if constexpr (NOREAD) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { asm volatile("" : "+v"(fw[j][0])); asm volatile("" : "+v"(fw[j][1])); }
      return;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fw[j][0] = as8(*(const i32x4*)(smem + wbase0 + buf * HT + (32 * u + 16 * j) * 128));
      fw[j][1] = as8(*(const i32x4*)(smem + wbase1 + buf * HT + (32 * u + 16 * j) * 128));
    }
  };
  auto mfmas = [&](int s, int u, const bf16x8 (&fw)[2][2]) __attribute__((always_inline)) {
    if constexpr (NOMFMA) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(fa[i][kh]));
#pragma unroll
        for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(fw[j][kh]));
      }
      return;
    }
    if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * s + i][2 * u + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[j][kh], fa[i][kh], acc[4 * s + i][2 * u + j], 0, 0, 0);
    if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(0);
  };
  auto ktile = [&](int t, auto BUF) __attribute__((always_inline)) {
    constexpr int buf = decltype(BUF)::value;
    if constexpr (COARSE) {
      // PA: A[s0], B[u0], B[u1] | A0(t+1), A1(t+1) | 32 MFMA      PB: A[s1] | B0(t+2), B1(t+2), vmcnt(4) | 32 MFMA
      read_a(0, buf);
      read_w(fw0, 0, buf);
      read_w(fw1, 1, buf);
      issue_a(0, t + 1, buf ^ 1);
      issue_a(1, t + 1, buf ^ 1);
      if constexpr (!LGKM_AFTER) lgkm0();
      bar();
      if constexpr (LGKM_AFTER) lgkm0();
      mfmas(0, 0, fw0);
      mfmas(0, 1, fw1);
      bar();
      read_a(1, buf);
      issue_w(0, t + 2, buf);
      issue_w(1, t + 2, buf);
      wait_vm<4>();
      if constexpr (!LGKM_AFTER) lgkm0();
      bar();
      if constexpr (LGKM_AFTER) lgkm0();
      mfmas(1, 1, fw1);
      mfmas(1, 0, fw0);
      bar();
      return;
    }
    // P0
    read_a(0, buf);
    read_w(fw0, 0, buf);
    issue_a(0, t + 1, buf ^ 1);
    if constexpr (!LGKM_AFTER) lgkm0();
    bar();
    if constexpr (LGKM_AFTER) lgkm0();
    mfmas(0, 0, fw0);
    bar();
    // P1
    read_w(fw1, 1, buf);
    issue_a(1, t + 1, buf ^ 1);
    if constexpr (!LGKM_AFTER) lgkm0();
    bar();
    if constexpr (LGKM_AFTER) lgkm0();
    mfmas(0, 1, fw1);
    bar();
    // P2
    read_a(1, buf);
    issue_w(0, t + 2, buf);
    if constexpr (!LGKM_AFTER) lgkm0();
    bar();
    if constexpr (LGKM_AFTER) lgkm0();
    mfmas(1, 1, fw1);
    bar();
    // P3
    issue_w(1, t + 2, buf);
    wait_vm<4>();
    bar();
    mfmas(1, 0, fw0);
    bar();
  };

  // ---- prologue: W(0), A(0), W(1) in stream order; K-tile 0 landed = all but the 4 youngest instructions
  issue_w(0, 0, 0); issue_w(1, 0, 0);
  issue_a(0, 0, 0); issue_a(1, 0, 0);
  issue_w(0, 1, 1); issue_w(1, 1, 1);
  wait_vm<4>();
  bar();
  if constexpr (!NOSTAG) {
    if (wr == 1) bar();
  }
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
#pragma unroll 1
  for (int t = 0; t < nkt; t += 2) {
    ktile(t, I0{});
    ktile(t + 1, I1{});
  }
  if constexpr (!NOSTAG) {
    if (wr == 0) bar();
  }
  wait_vm<0>();    // the tail issues beyond K read neighbouring rows / zeros into slots nobody reads: drain before exit

  // ---- epilogue: lane (l15, l4) of block (i, j) owns row i * 16 + l15, columns j * 16 + 4 * l4 .. + 3
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + l15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + 4 * l4;
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (__bf16)acc[i][j][e];
      *(i32x2*)((char*)C + ((size_t)m * N + n) * 2) = __builtin_bit_cast(i32x2, v);
    }
  }
}

template <int VAR> static int launch(const void* A, const void* W, void* C, int M, int N, int K, hipStream_t st) {
  if ((M & 255) || (N & 255) || (K & 127) || ((M >> 8) * (N >> 8)) % 8) return -1;
  auto kfn = gemm8p_kernel<VAR>;
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipLaunchKernelGGL(kfn, dim3((M >> 8) * (N >> 8)), dim3(512), 131072, st, (const __bf16*)A, (const __bf16*)W, (__bf16*)C, M, N, K);
  return (int)hipGetLastError();
}

extern "C" int proto_gemm8p(const void* A, const void* W, void* C, int M, int N, int K, int var, void* st) {
  hipStream_t s = (hipStream_t)st;
  switch (var) {
    case 0: return launch<0>(A, W, C, M, N, K, s);
    case 1: return launch<1>(A, W, C, M, N, K, s);
    case 2: return launch<2>(A, W, C, M, N, K, s);
    case 4: return launch<4>(A, W, C, M, N, K, s);
    case 8: return launch<8>(A, W, C, M, N, K, s);
    case 6: return launch<6>(A, W, C, M, N, K, s);
    case 16: return launch<16>(A, W, C, M, N, K, s);
    case 32: return launch<32>(A, W, C, M, N, K, s);
    case 64: return launch<64>(A, W, C, M, N, K, s);
    case 48: return launch<48>(A, W, C, M, N, K, s);       // no MFMA, no DMA: reads + barriers
    case 80: return launch<80>(A, W, C, M, N, K, s);       // no MFMA, no reads: DMA + barriers
    case 96: return launch<96>(A, W, C, M, N, K, s);       // no DMA, no reads: MFMA + barriers
    case 112: return launch<112>(A, W, C, M, N, K, s);     // barriers + loop only
    case 128: return launch<128>(A, W, C, M, N, K, s);
    case 130: return launch<130>(A, W, C, M, N, K, s);
  }
  return -2;
}
