// 256-row implicit-GEMM kernel for the large conv / linear shapes (VAE and the 64x64 UNet level).
//
// Why a second kernel: with 128x128x64 tiles every CU has to pull 32 KB through its 64 B/clk
// L2->LDS path per 2.1 MFLOP -- exactly the MFMA rate, so that kernel tops out near 1/3 of peak.
// Here a workgroup of 8 waves owns a 256 x BN (BN = 256 | 128) tile: 128 | 87 FLOP per staged byte.
//
// Structure (one workgroup per CU, 2 waves per SIMD):
//   * K-step = 32; LDS is a ring of S = 4 stages of [(256 + BN) rows][32 k] (64-byte rows,
//     16-byte chunks XOR-swizzled by ((row>>2)&3): conflict-free ds_read_b128 fragments);
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds), issued from inline asm so that hipcc does
//     not drain it: three stages stay in flight ACROSS the barriers, retired by a counted
//     s_waitcnt vmcnt(N) (N = DMA instructions of the two younger stages), one barrier per K-step;
//   * the ring runs continuously across the tiles of the persistent workgroup (the next tile's
//     first stages are in flight during the epilogue);
//   * conv tiles are 16x16 output-pixel patches (halo reuse in L2); out-of-range taps, ragged M and
//     padded N are zero-filled by the buffer bounds check.
// Wave tile: (256/WGM) x 64, v_mfma_f32_32x32x16, D[n][m] orientation (lane owns 4 contiguous
// output channels), same epilogue as gemm.hip.
#include "gemm_common.h"

namespace dfw {

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

// One LDS-DMA wave-instruction: 64 lanes x 16 B -> LDS [m0 .. m0 + 1 KiB).  M0 is written in the
// same statement that uses it (hipcc does not preserve it around asm); s_nop covers the
// SALU-write-M0 -> LDS-DMA hazard.  Invisible to hipcc's waitcnt bookkeeping by design.
__device__ __forceinline__ void dma16(u32x4 srd, uint32_t voff, uint32_t lds_byte) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(lds_byte), "v"(voff), "s"(srd) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

template <typename T, int BN, bool CONV>
__global__ __launch_bounds__(512, 1) void gemm_big_kernel(const GemmP p) {
  constexpr int BM = 256, S = 4;
  constexpr int WGN = BN / 64, WGM = 8 / WGN;   // wave grid: 2x4 (BN=256) or 4x2 (BN=128)
  constexpr int WTM = BM / WGM;                 // 128 or 64
  constexpr int MB = WTM / 32, NB = 2;
  constexpr int STAGE = (BM + BN) * 64;         // bytes
  constexpr int SA = 2, SW = BN / 128;          // DMA wave-instructions per stage per wave
  constexpr int DPS = SA + SW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = lane & 31, lh = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- persistent tile walk (same XCD-contiguous order as gemm.hip)
  const int ntiles = p.ntm * p.ntn;
  const int nxb = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int Q = (ntiles + 7) >> 3;
  const int t_end = min(ntiles, (xcd + 1) * Q);
  const int tile0 = xcd * Q + (blockIdx.x >> 3);
  if (tile0 >= t_end) return;
  const int my_tiles = (t_end - tile0 + nxb - 1) / nxb;
  const int nks = p.K >> 5;            // K-steps of 32
  const int cpt = p.Cin >> 5;          // K-steps per tap

  const int z = blockIdx.y;
  const char* Ab = p.A;
  const char* Wb = p.W;
  char* Cb = p.C;
  if (p.batch > 1) {
    Ab += (size_t)z * p.strideA * sizeof(T);
    Wb += (size_t)z * p.strideW * sizeof(T);
    Cb += (size_t)z * p.strideC * (p.out_mode == DFW_OUT_T ? sizeof(T) : sizeof(float));
  }
  const u32x4 ra = make_srd(Ab, p.a_bytes);
  const u32x4 rw = make_srd(Wb, p.w_bytes);

  auto tile_coords = [&](int t) -> TileC {
    TileC c;
    const int tn = t % p.ntn, tm = t / p.ntn;
    c.m0 = tm * BM;
    c.n0 = tn * BN;
    c.img = 0; c.oy0 = 0; c.ox0 = 0;
    if constexpr (CONV) {
      c.img = tm / p.tpi;
      const int t2 = tm - c.img * p.tpi, tyi = t2 / p.tpr, txi = t2 - tyi * p.tpr;
      c.oy0 = tyi << 4;
      c.ox0 = txi << 4;
    }
    return c;
  };
  // conv tiles are always 16x16 pixel patches here (host-checked), so no divisions per row
  auto row_to_m = [&](const TileC& c, int r, int& oy, int& ox, int& img) -> int {
    if constexpr (CONV) {
      oy = c.oy0 + (r >> 4);
      ox = c.ox0 + (r & 15);
      img = c.img;
      return (img * p.Ho + oy) * p.Wo + ox;
    } else {
      return c.m0 + r;
    }
  };

  // ---- loader: wave-instruction i of a wave covers tile rows (i*8 + wave)*16 .. +16,
  // lane -> row + (lane>>2), LDS slot lane&3, source chunk = slot ^ ((row>>2)&3) = slot ^ ((lane>>4)&3)
  const int lrow = lane >> 2;
  const int kc = (lane & 3) ^ ((lane >> 4) & 3);
  uint32_t a_off[SA];
  int a_iy0[SA], a_ix0[SA];
  uint32_t a_pix[SA];
  uint32_t w_off[SW];
  int tap = 0, cc = 0;
  const unsigned limH = p.ups ? 2 * p.Hi : p.Hi, limW = p.ups ? 2 * p.Wi : p.Wi;
  const int ush = p.ups ? 1 : 0;
  auto setup_loader = [&](const TileC& c) {
#pragma unroll
    for (int i = 0; i < SA; ++i) {
      int oy = 0, ox = 0, img = 0;
      const int m = row_to_m(c, (i * 8 + wave) * 16 + lrow, oy, ox, img);
      if constexpr (!CONV) {
        a_off[i] = m < p.M ? (uint32_t)(((size_t)m * p.lda + kc * 8) * sizeof(T)) : kOOB;
      } else {
        if (m < p.M) {
          a_iy0[i] = oy * p.stride - p.pad;
          a_ix0[i] = ox * p.stride - p.pad;
          a_pix[i] = (uint32_t)img * (uint32_t)(p.Hi * p.Wi);
        } else {
          a_iy0[i] = -(1 << 20);
          a_ix0[i] = 0;
          a_pix[i] = 0;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < SW; ++i) {
      const int n = c.n0 + (i * 8 + wave) * 16 + lrow;
      w_off[i] = n < p.N ? (uint32_t)(((size_t)n * p.K + kc * 8) * sizeof(T)) : kOOB;
    }
    tap = 0;
    cc = 0;
  };
  auto issue = [&](int ks, int slot) {
    const uint32_t dst = lds0 + (uint32_t)slot * STAGE + (uint32_t)wave * 1024u;
    if constexpr (!CONV) {
#pragma unroll
      for (int i = 0; i < SA; ++i) dma16(ra, a_off[i] + (uint32_t)ks * 64u, dst + i * 8192);
    } else {
      const int ky = tap / 3, kx = tap - ky * 3;
      const uint32_t coff = (uint32_t)(cc * 32 + kc * 8);
#pragma unroll
      for (int i = 0; i < SA; ++i) {
        int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        const bool ok = (unsigned)iy < limH && (unsigned)ix < limW;
        iy >>= ush;
        ix >>= ush;
        const uint32_t off = ((a_pix[i] + (uint32_t)(iy * p.Wi + ix)) * (uint32_t)p.lda + coff) * (uint32_t)sizeof(T);
        dma16(ra, ok ? off : kOOB, dst + i * 8192);
      }
      if (++cc == cpt) { cc = 0; ++tap; }
    }
#pragma unroll
    for (int i = 0; i < SW; ++i) dma16(rw, w_off[i] + (uint32_t)ks * 64u, dst + BM * 64 + i * 8192);
  };
  // ---- fragment read addresses within a stage (k-substep s: ^ (s<<5))
  uint32_t lds_ra[MB], lds_rw[NB];
  const int sw4 = (lr >> 2) & 3;
#pragma unroll
  for (int i = 0; i < MB; ++i) lds_ra[i] = (uint32_t)(wm * WTM + i * 32 + lr) * 64u + (uint32_t)((lh ^ sw4) << 4);
#pragma unroll
  for (int j = 0; j < NB; ++j) lds_rw[j] = (uint32_t)(BM + wn * 64 + j * 32 + lr) * 64u + (uint32_t)((lh ^ sw4) << 4);

  f32x16 acc[MB][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto epilogue = [&](const TileC& c) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      int oy_ = 0, ox_ = 0, img_ = 0;
      const int m = row_to_m(c, wm * WTM + i * 32 + lr, oy_, ox_, img_);
      if (m >= p.M) continue;
      if constexpr (!CONV) img_ = p.rowbias ? m / p.rows_per_img : 0;
      if (p.geglu) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int na = c.n0 + wn * 64 + 8 * g + 4 * lh;
          if (na >= p.N) continue;
          const int no = ((c.n0 + wn * 64) >> 1) + 8 * g + 4 * lh;
          f32x4 ba = *(const f32x4*)(p.bias + na), bg = *(const f32x4*)(p.bias + na + 32);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (acc[i][0][4 * g + e] + ba[e]) * gelu_erf(acc[i][1][4 * g + e] + bg[e]);
          *(i32x2*)(Cb + ((size_t)m * p.ldc + no) * sizeof(T)) = pack4<T>(v);
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = c.n0 + wn * 64 + j * 32 + 8 * g + 4 * lh;
          if (n >= p.N) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          if (p.bias) {
            const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
          }
          if (p.rowbias) {
            const f32x4 b = *(const f32x4*)(p.rowbias + (size_t)img_ * p.ldrb + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
          }
          if (p.residual) {
            float r[4];
            unpack4<T>(*(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T)), r);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
          *(i32x2*)(Cb + ((size_t)m * p.ldc + n) * sizeof(T)) = pack4<T>(v);
        }
    }
  };

  auto compute = [&](int slot) {
    const char* buf = smem + slot * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      typename Tr<T>::v8 fa[MB], fw[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) fa[i] = as_v8<T>(*(const i32x4*)(buf + (lds_ra[i] ^ (s << 5))));
#pragma unroll
      for (int j = 0; j < NB; ++j) fw[j] = as_v8<T>(*(const i32x4*)(buf + (lds_rw[j] ^ (s << 5))));
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = Tr<T>::mfma(fw[j], fa[i], acc[i][j]);
    }
  };
  auto retire_and_sync = [&](int younger) {
    // this wave's DMA of the stage about to be read is done once at most the `younger` stages'
    // instructions are outstanding; the barrier then publishes every wave's pieces and frees
    // the slot read one step ago
    if (younger >= 2) wait_vm<2 * DPS>();
    else if (younger == 1) wait_vm<DPS>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };

  // ---- pipeline: ring slot of (tile-local) K-step k is (base + k) & 3; three stages in flight.
  // nks >= 4 (host-checked).  The hot loop is branch-free; the last three steps of a tile stage the
  // first three steps of the next tile, so the ring never drains between tiles.
  TileC ct = tile_coords(tile0);
  setup_loader(ct);
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  int base = 0;  // ring slot of this tile's step 0
  for (int ti = 0; ti < my_tiles; ++ti) {
    const bool has_next = ti + 1 < my_tiles;
    zero_acc();
    for (int k = 0; k < nks - 3; ++k) {
      wait_vm<2 * DPS>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(k + 3, (base + k + 3) & 3);
      compute((base + k) & 3);
    }
    if (has_next) setup_loader(tile_coords(tile0 + (ti + 1) * nxb));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int k = nks - 3 + j;
      retire_and_sync(has_next ? 2 : 2 - j);
      if (has_next) issue(j, (base + k + 3) & 3);
      compute((base + k) & 3);
    }
    epilogue(ct);
    base = (base + nks) & 3;
    if (has_next) ct = tile_coords(tile0 + (ti + 1) * nxb);
  }
}

template <typename T, int BN>
static int launch_big(const GemmP& p, hipStream_t st) {
  constexpr int BM = 256;
  GemmP q = p;
  q.ntm = (p.M + BM - 1) / BM;
  q.ntn = (p.N + BN - 1) / BN;
  q.tw = 0; q.tw_log2 = 0; q.tpr = 0; q.tpi = 0;
  if (p.taps == 9) {
    q.tw = 16; q.tw_log2 = 4;
    q.tpr = p.Wo / 16;
    q.tpi = q.tpr * (p.Ho / 16);
  }
  const size_t lds = 4 * (BM + BN) * 64;
  const int zdim = p.batch > 1 ? p.batch : 1;
  int nwg = q.ntm * q.ntn;
  if (nwg > 256) nwg = 256;
  nwg = (nwg + 7) & ~7;
  dim3 grid(nwg, zdim);
  static bool attr_set[2] = {false, false};
  if (p.taps == 1) {
    auto kfn = gemm_big_kernel<T, BN, false>;
    if (!attr_set[0]) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set[0] = true; }
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, q);
  } else {
    auto kfn = gemm_big_kernel<T, BN, true>;
    if (!attr_set[1]) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set[1] = true; }
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, q);
  }
  DFW_CHECK_LAUNCH();
  return 0;
}

// Eligible: no split-K, N a multiple of 128 (wave tiles are 64 wide, GEGLU pairs stay inside one),
// 16-byte-aligned fp32 epilogue vectors, and enough 256-row tiles to occupy the chip.
bool gemm_big_eligible(const GemmP& p, int& bn) {
  static const char* off = getenv("DFW_GEMM_NOBIG");
  if (off) return false;
  if (p.splitk > 1 || (p.N % 128) != 0 || (p.K % 32) != 0 || p.K < 128) return false;
  if (p.out_mode != DFW_OUT_T || p.act != DFW_ACT_NONE) return false;   // slim epilogue: NHWC storage dtype
  if (p.taps == 9 && (p.Wo % 16 != 0 || p.Ho % 16 != 0)) return false;  // 16x16 pixel tiles
  bn = (p.N % 256) == 0 ? 256 : 128;
  const long long tiles = (long long)((p.M + 255) / 256) * (p.N / bn) * (p.batch > 1 ? p.batch : 1);
  return tiles >= 192;
}

int launch_gemm_big(const GemmP& p, hipStream_t st) {
  int bn = 0;
  if (!gemm_big_eligible(p, bn)) return DFW_ESHAPE;
  const bool bf = p.dtype_bf16 != 0;
  if (bn == 256) return bf ? launch_big<__bf16, 256>(p, st) : launch_big<_Float16, 256>(p, st);
  return bf ? launch_big<__bf16, 128>(p, st) : launch_big<_Float16, 128>(p, st);
}

}  // namespace dfw
