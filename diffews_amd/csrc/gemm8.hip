// 256 x 256 x 64 implicit GEMM on the eight-phase schedule (cdna_hip_programming.md section 5, 'The 256^2 8-phase template'),
// round 4: the successor of gemm_big_kernel<.,256,256,32> for Linear layers and for the conv3x3 shapes that gather their A
// operand per tap (stride 2, fused nearest-2x upsampling; the stride-1 convs run conv_patch.hip).
//
// What changed against gemm_big.hip (same output tile, same 128 x 64 wave tile, same v_mfma_f32_16x16x32, same epilogue):
//   * K-tile 64: LDS rows are 128 bytes = whole cache lines per LDS-DMA lane group (8 rows x 128 B per wave-instruction),
//     16-byte chunks XOR-swizzled by (row >> 1) & 7 on the DMA's SOURCE address (the LDS image is written linearly);
//   * staging in HALF-TILES (128 rows x 64 k = 16 KiB = two wave-instructions per wave), two buffers per half-tile
//     (A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255: 128 KiB), consumed progressively: the W half-tiles of
//     K-tile t are in registers after the first phase and are refilled with K-tile t + 2 in the second, the A half-tiles
//     after the second and refilled with t + 1... one counted s_waitcnt vmcnt per K-tile;
//   * EVERY wave issues its DMA in its fragment-read phase, never between a barrier and its MFMAs; the two wave groups
//     (wave >> 2: the two row halves of the tile) run ONE instruction stream, staggered by one barrier;
//   * the epilogue's 32 KiB staging area is its own LDS region (160 KiB in all), so the next tile's stages are in flight
//     through the epilogue without any aliasing rule.
// Measured on the plain-GEMM prototype of this loop (scratch/proto/gemm8p.hip, uniform random bf16, same box, interleaved):
// 4096^3 1.40-1.44 PFLOP/s against gemm_big 1.27 and hipBLASLt 1.48; 32768 x 512 x 4608 1.23-1.31 against 1.06 / 1.40.
//
// Per K-tile g (buffer g & 1), per wave; quadrant (s, u) = rows 64 s .. + 63 of the wave's 128, columns 32 u .. + 31 of its 64:
//   PA  read A[s0] (8 x b128), W[u0] (4), W[u1] (4) | issue A0(g+1), A1(g+1)           | lgkmcnt(0) | bar | 32 MFMA | bar
//   PB  read A[s1] (8)                              | issue W0(g+2), W1(g+2), vmcnt(4) | lgkmcnt(0) | bar | 32 MFMA | bar
// RAW: the wait of PB(g) leaves only W(g+2) outstanding, so A(g+1) and W(g+1) have landed for every wave before the barrier
// that precedes PA(g+1) of either group.  WAR: a phase's reads have returned before the wave's first barrier of the phase;
// W(g) is last read in PA(g) and refilled in PB(g); A(g) is last read in PB(g) and refilled in PA(g+1).
// The K-tile stream runs on across the tiles of the persistent workgroup (two cursors: A one K-tile ahead, W two).
#include "gemm_common.h"
#include <type_traits>

namespace dfw {

// BN = 256: wave grid 2 (M) x 4 (N), wave tile 128 x 64 (the scheme above).  BN = 128: wave grid 4 x 2, wave tile 64 x 64, ONE W
// half-tile per K-tile, 16-MFMA phases; the wave groups (wave >> 2) still own the A half-tiles 0 / 1.
template <typename T, bool CONV, int BN>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(const GemmP p) {
  constexpr int BM = 256, HT = 16384;
  // BN = 160 (Linear only; the N = 320 layers of the UNet's 64^2 level: 128 x 2 = 256 tiles on the lock-step batch): wave grid 4 x 2,
  // wave tile 64 x 80 (five 16-column blocks), ONE 160-row W stage per K-tile (three wave-instructions per wave, the third of waves
  // 4-7 zero-fills 32 pad rows) in two 24 KiB buffers behind the A half-tile slots, 160-byte staging rows (conv_patch8.hip's tile)
  constexpr bool N160 = BN == 160;
  static_assert(!(N160 && CONV), "the 256 x 160 tile is instantiated for Linear shapes only");
  constexpr int WH = BN / 128;                          // W half-tiles per K-tile
  constexpr int WGN = N160 ? 2 : BN / 64, WGM = 8 / WGN;   // wave grid
  constexpr int NBW = N160 ? 5 : 4;                     // 16-column blocks of a wave tile
  constexpr int WTM = BM / WGM, MB6 = WTM / 16, MH = MB6 / 2;   // wave tile rows, its 16-row blocks, blocks per quadrant
  constexpr int WDMA = N160 ? 3 : 2 * WH;               // W wave-instructions per wave per K-tile
  constexpr int WB160 = 65536, WS160 = 24576;           // BN = 160: W buffers at WB160 + buf * WS160
  constexpr int STG = N160 ? WB160 + 2 * WS160 : (2 + WH) * 32768;   // epilogue staging behind the stage slots
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2;                             // wave group = A half-tile
  const int wm = wave / WGN, wc = wave % WGN;           // position in the wave grid (wm / (WGM / 2) == wr)
  const uint32_t lds0 = lds_addr(smem);

  // ---- persistent tile walk (gemm_big.hip's XCD-contiguous order)
  const int ntiles = p.ntm * p.ntn;
  const int nxb = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int Q = (ntiles + 7) >> 3;
  const int t_end = min(ntiles, (xcd + 1) * Q);
  const int tile0 = xcd * Q + (blockIdx.x >> 3);
  if (tile0 >= t_end) return;
  const int my_tiles = (t_end - tile0 + nxb - 1) / nxb;
  const int nkt = p.K >> 6;            // K-tiles per tile
  const int cpt = p.Cin >> 6;          // 64-channel chunks per tap

  const int z = blockIdx.y;
  const char* Ab = p.A;
  const char* Wb = p.W;
  char* Cb = p.C;
  if (p.batch > 1) {
    Ab += (size_t)z * p.strideA * sizeof(T);
    Wb += (size_t)z * p.strideW * sizeof(T);
    Cb += (size_t)z * p.strideC * sizeof(T);
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
  auto row_to_m = [&](const TileC& c, int r) -> int {
    if constexpr (CONV) return (c.img * p.Ho + c.oy0 + (r >> 4)) * p.Wo + c.ox0 + (r & 15);
    else return c.m0 + r;
  };

  // ---- loaders.  Wave-instruction j (0, 1) of this wave covers rows (j * 8 + wave) * 8 .. + 8 of a half-tile: lane -> row
  // lane >> 3, LDS slot lane & 7, source chunk slot ^ f(row) with f(row) = (row >> 1) & 7 = ((wave & 1) * 4 + (lane >> 4)) & 7.
  const int lrow = lane >> 3;
  const int kc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  const unsigned limH = p.ups ? 2 * p.Hi : p.Hi, limW = p.ups ? 2 * p.Wi : p.Wi;
  const int ush = p.ups ? 1 : 0;
  // A cursor
  int a_kt = 0, a_ti = 0;
  bool a_live = true;
  uint32_t a_v0 = 0;         // linear: byte offset of row (m0 + wave * 8 + lrow), chunk kc
  int a_m = 0, a_ix0 = 0;    // linear: that row's index; conv: input x of pixel column (ox0 + (wave & 1) * 8 + lrow), tap kx = 0
  int a_iy0 = 0;             // conv (wave-uniform): input y of pixel row oy0 + (wave >> 1), tap ky = 0
  uint32_t a_pix = 0;        // conv (wave-uniform): img * Hi * Wi
  auto a_setup = [&](const TileC& c) {
    if constexpr (!CONV) {
      a_m = c.m0 + wave * 8 + lrow;
      a_v0 = (uint32_t)(((size_t)a_m * p.lda + kc * 8) * sizeof(T));
    } else {
      a_iy0 = (c.oy0 + (wave >> 1)) * p.stride - p.pad;
      a_ix0 = (c.ox0 + (wave & 1) * 8 + lrow) * p.stride - p.pad;
      a_pix = (uint32_t)c.img * (uint32_t)(p.Hi * p.Wi);
    }
  };
  auto a_advance = [&]() {
    if (++a_kt == nkt) {
      a_kt = 0;
      if (++a_ti < my_tiles) a_setup(tile_coords(tile0 + a_ti * nxb));
      else a_live = false;
    }
  };
  auto issue_a = [&](int buf) __attribute__((always_inline)) {   // both halves of the cursor's K-tile -> buffer buf
    int ky = 0, kx = 0, cc = 0;
    if constexpr (CONV) {
      cc = a_kt / 9;                       // chunk-major K walk: the nine taps of one 64-channel chunk back to back
      const int tap = a_kt - cc * 9;
      ky = tap / 3;
      kx = tap - ky * 3;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint32_t dst = lds0 + (uint32_t)(h * 32768 + buf * HT) + (uint32_t)wave * 1024u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        uint32_t off;
        if constexpr (!CONV) {
          const int dr = h * 128 + j * 64;
          off = (a_m + dr < p.M && a_live) ? a_v0 + (uint32_t)dr * (uint32_t)p.lda * (uint32_t)sizeof(T) + (uint32_t)a_kt * 128u : kOOB;
        } else {
          int iy = a_iy0 + (h * 8 + j * 4) * p.stride + ky, ix = a_ix0 + kx;
          const bool ok = (unsigned)iy < limH && (unsigned)ix < limW && a_live;
          iy >>= ush;
          ix >>= ush;
          off = ok ? ((a_pix + (uint32_t)(iy * p.Wi + ix)) * (uint32_t)p.lda + (uint32_t)(cc * 64 + kc * 8)) * (uint32_t)sizeof(T) : kOOB;
        }
        dma16(ra, off, dst + j * 8192);
      }
    }
    a_advance();
  };
  // W cursor (BN = 256: N % 256 == 0, every row of a W tile exists; BN = 128: rows beyond N are zero-filled)
  int w_kt = 0, w_ti = 0;
  bool w_live = true;
  uint32_t w_v0 = 0;
  int w_n = 0;
  auto w_setup = [&](const TileC& c) {
    w_n = c.n0 + wave * 8 + lrow;
    w_v0 = (uint32_t)(((size_t)w_n * p.K + kc * 8) * sizeof(T));
  };
  auto issue_w = [&](int buf) __attribute__((always_inline)) {
    uint32_t koff = (uint32_t)w_kt * 128u;
    if constexpr (CONV) {
      const int cc = w_kt / 9, tap = w_kt - cc * 9;
      koff = (uint32_t)(tap * p.Cin + cc * 64) * (uint32_t)sizeof(T);
    }
    if constexpr (N160) {
      const uint32_t dst = lds0 + (uint32_t)(WB160 + buf * WS160) + (uint32_t)wave * 1024u;
#pragma unroll
      for (int j = 0; j < 3; ++j) {      // rows j * 64 + wave * 8 + lrow of the 160 (j = 2: waves 0-3; the others write zeros)
        const uint32_t off = w_v0 + (uint32_t)(j * 64) * (uint32_t)p.K * (uint32_t)sizeof(T) + koff;
        const bool ok = w_live && (j < 2 || wave < 4);
        dma16(rw, ok ? off : kOOB, dst + j * 8192);
      }
    }
#pragma unroll
    for (int h = 0; h < (N160 ? 0 : WH); ++h) {
      const uint32_t dst = lds0 + (uint32_t)((2 + h) * 32768 + buf * HT) + (uint32_t)wave * 1024u;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t off = w_v0 + (uint32_t)(h * 128 + j * 64) * (uint32_t)p.K * (uint32_t)sizeof(T) + koff;
        bool ok = w_live;
        if constexpr (BN == 128) ok = ok && (w_n + j * 64 < p.N);
        dma16(rw, ok ? off : kOOB, dst + j * 8192);
      }
    }
    if (++w_kt == nkt) {
      w_kt = 0;
      if (++w_ti < my_tiles) {
        if (p.ntn > 1) w_setup(tile_coords(tile0 + w_ti * nxb));
      } else w_live = false;
    }
  };

  // ---- fragment read bases (buffer 0; toggled by ^ HT per K-tile): row l15, chunk (l4 + 4 kh) ^ f, f = l15 >> 1
  uint32_t ab0, ab1, wb0, wb1;
  {
    const int l15 = lane & 15, l4 = lane >> 4;
    const uint32_t fb = (uint32_t)(l15 * 128 + ((l4 ^ (l15 >> 1)) << 4));
    ab0 = (uint32_t)(wr * 32768 + (wm % (WGM / 2)) * WTM * 128) + fb;
    ab1 = ab0 ^ 64u;
    if constexpr (N160) wb0 = (uint32_t)(WB160 + wc * 80 * 128) + fb;   // rows wc * 80 .. + 79 (80 / 2 = 0 mod 8: the same swizzle term)
    else wb0 = (uint32_t)((2 + (wc >> 1)) * 32768 + (wc & 1) * 64 * 128) + fb;
    wb1 = wb0 ^ 64u;
  }
  const uint32_t wbase160 = wb0;

  f32x4 acc6[MB6][NBW];
  typename Tr<T>::v8 fa[MH][2], fw0[2][2], fw1[2][2], fw4[2];     // fw4: the fifth column block of the 80-wide wave tile
  auto zero6 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MB6; ++i)
#pragma unroll
      for (int j = 0; j < NBW; ++j) acc6[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
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
  auto read_a = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < MH; ++i) {
      fa[i][0] = as_v8<T>(*(const i32x4*)(smem + ab0 + (16 * MH * s + 16 * i) * 128));
      fa[i][1] = as_v8<T>(*(const i32x4*)(smem + ab1 + (16 * MH * s + 16 * i) * 128));
    }
  };
  auto read_w = [&](typename Tr<T>::v8 (&fw)[2][2], int u) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fw[j][0] = as_v8<T>(*(const i32x4*)(smem + wb0 + (32 * u + 16 * j) * 128));
      fw[j][1] = as_v8<T>(*(const i32x4*)(smem + wb1 + (32 * u + 16 * j) * 128));
    }
  };
  auto read_w4 = [&]() __attribute__((always_inline)) {
    fw4[0] = as_v8<T>(*(const i32x4*)(smem + wb0 + 64 * 128));
    fw4[1] = as_v8<T>(*(const i32x4*)(smem + wb1 + 64 * 128));
  };
  auto mfmas4 = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < MH; ++i) acc6[MH * s + i][NBW - 1] = Tr<T>::mfma16(fw4[kh], fa[i][kh], acc6[MH * s + i][NBW - 1]);
  };
  auto mfmas = [&](int s, int u, const typename Tr<T>::v8 (&fw)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc6[MH * s + i][2 * u + j] = Tr<T>::mfma16(fw[j][kh], fa[i][kh], acc6[MH * s + i][2 * u + j]);
  };

  // ---- staged epilogue: gemm_big.hip's 16x16x32 epilogue (32 tile rows per round through the wave's 4 KiB: row r,
  // 16-byte chunk c at ((c ^ (r & 7)) << 4) of a 128-byte line; statistics of the STORED values; whole-line stores)
  auto epilogue6 = [&](const TileC& c, char* stg, int tile_id) {
    constexpr int NB6 = 4;
    if (c.n0 + wc * 64 >= p.N) return;      // ragged N (a multiple of 64 on 128-wide tiles): this wave multiplied zero-filled W rows
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15 = lane_e & 15, l4 = lane_e >> 4, lane = lane_e;
    if constexpr (N160) {
      // 64 x 80 wave tile: 32 rows per round through 160-byte staging rows (5 KiB per wave), stored as ten 16-byte chunks per row
#pragma unroll
      for (int i = 0; i < MB6 / 2; ++i) {
        f32x4 add[2][NBW];
        i32x2 res[2][NBW];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = c.m0 + wm * WTM + i * 32 + h * 16 + l15;
          const int img_ = (p.rowbias && m < p.M) ? m / p.rows_per_img : 0;
#pragma unroll
          for (int j = 0; j < NBW; ++j) {
            const int n = c.n0 + wc * 80 + j * 16 + 4 * l4;
            f32x4 bb = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bb = *(const f32x4*)(p.bias + n);
            if (p.rowbias && m < p.M) {
              const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)img_ * p.ldrb + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) bb[e] += r[e];
            }
            add[h][j] = bb;
            res[h][j] = i32x2{0, 0};
            if (p.residual && m < p.M) res[h][j] = *(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T));
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < NBW; ++j) {
            float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.residual) unpack4<T>(res[h][j], r);
            const float osc = (c.n0 + wc * 80 + j * 16) < p.cs_n ? p.cs : p.out_scale;     // cs_n is a multiple of 64: whole 16-column blocks
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc6[2 * i + h][j][e] + add[h][j][e] + r[e]) * osc;
            *(i32x2*)(stg + (h * 16 + l15) * 160 + j * 32 + l4 * 8) = pack4<T>(v);
          }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const int idx = lane + 64 * t;               // 320 chunks of the round
          const int r = (idx * 6554) >> 16, ch = idx - r * 10;
          const int m2 = c.m0 + wm * WTM + i * 32 + r;
          const i32x4 val = *(const i32x4*)(stg + r * 160 + ch * 16);
          if (m2 < p.M)
            *(i32x4*)(Cb + ((size_t)m2 * p.ldc + c.n0 + wc * 80 + ch * 8) * sizeof(T)) = val;
        }
      }
      return;
    }
    float gs0 = 0.f, gs1 = 0.f, gq0 = 0.f, gq1 = 0.f;
#pragma unroll
    for (int i = 0; i < MB6 / 2; ++i) {
      if (p.geglu) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = row_to_m(c, wm * WTM + i * 32 + h * 16 + l15);
          if (m >= p.M) continue;
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) {     // channel blocks 0,1 = value, 2,3 = gate (packing.pack_geglu)
            const int na = c.n0 + wc * 64 + jb * 16 + 4 * l4;
            const int no = ((c.n0 + wc * 64) >> 1) + jb * 16 + 4 * l4;
            const f32x4 ba = *(const f32x4*)(p.bias + na), bg = *(const f32x4*)(p.bias + na + 32);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc6[2 * i + h][jb][e] + ba[e]) * gelu_erf(acc6[2 * i + h][jb + 2][e] + bg[e]);
            *(i32x2*)(Cb + ((size_t)m * p.ldc + no) * sizeof(T)) = pack4<T>(v);
          }
        }
        continue;
      }
      f32x4 add[2][NB6];
      i32x2 res[2][NB6];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = row_to_m(c, wm * WTM + i * 32 + h * 16 + l15);
        int img_ = c.img;
        if constexpr (!CONV) img_ = (p.rowbias && m < p.M) ? m / p.rows_per_img : 0;
#pragma unroll
        for (int j = 0; j < NB6; ++j) {
          const int n = c.n0 + wc * 64 + j * 16 + 4 * l4;
          f32x4 bb = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) bb = *(const f32x4*)(p.bias + n);
          if (p.rowbias && m < p.M) {
            const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)img_ * p.ldrb + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] += r[e];
          }
          add[h][j] = bb;
          res[h][j] = i32x2{0, 0};
          if (p.residual && m < p.M) res[h][j] = *(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T));
        }
      }
      float osc6[NB6];
#pragma unroll
      for (int j = 0; j < NB6; ++j) osc6[j] = (c.n0 + wc * 64 + j * 16 + 4 * l4) < p.cs_n ? p.cs : p.out_scale;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < NB6; ++j) {
          float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.residual) unpack4<T>(res[h][j], r);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (acc6[2 * i + h][j][e] + add[h][j][e] + r[e]) * osc6[j];
          const int row = h * 16 + l15, quad = j * 4 + l4;   // 8-byte piece `quad` of the staged 128-byte row
          *(i32x2*)(stg + row * 128 + (((quad >> 1) ^ (row & 7)) << 4) + (quad & 1) * 8) = pack4<T>(v);
        }
      if (p.gn_partial) {
        const int cp = lane & 31, rh = lane >> 5;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const int r = rh * 16 + t;
          const uint32_t w2 = *(const uint32_t*)(stg + r * 128 + (((cp >> 2) ^ (r & 7)) << 4) + (cp & 3) * 4);
          typename Tr<T>::v4 pr = __builtin_bit_cast(typename Tr<T>::v4, i32x2{(int)w2, 0});
          const float a0 = (float)pr[0], a1 = (float)pr[1];
          gs0 += a0; gq0 += a0 * a0;
          gs1 += a1; gq1 += a1 * a1;
        }
      }
      const int c16 = lane & 7;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int r = (lane >> 3) + 8 * t;
        const int m2 = row_to_m(c, wm * WTM + i * 32 + r);
        const i32x4 val = *(const i32x4*)(stg + r * 128 + ((c16 ^ (r & 7)) << 4));
        if (m2 < p.M)
          *(i32x4*)(Cb + ((size_t)m2 * p.ldc + c.n0 + wc * 64 + c16 * 8) * sizeof(T)) = val;
      }
    }
    if (p.gn_partial) {
      float s2 = gs0 + gs1, q2 = gq0 + gq1;
      s2 += __shfl_xor(s2, 32, 64);
      q2 += __shfl_xor(q2, 32, 64);
      const int cpg = p.N / p.gn_groups, ppg = cpg >> 1;
      for (int o = 1; o < ppg; o <<= 1) {
        s2 += __shfl_xor(s2, o, 64);
        q2 += __shfl_xor(q2, o, 64);
      }
      const int cp = lane & 31;
      if (lane < 32 && (cp & (ppg - 1)) == 0) {
        const int tm = tile_id / p.ntn;
        const int chunk = (tm - c.img * p.tpi) * WGM + wm;
        const int grp = (c.n0 + wc * 64 + 2 * cp) / cpg;
        float* o2 = p.gn_partial + (((size_t)c.img * p.gn_chunks + chunk) * p.gn_groups + grp) * 2;
        o2[0] = s2;
        o2[1] = q2;
      }
    }
  };

  // ---- prologue: W(0), A(0), W(1) in stream order; K-tile 0 has landed once only W(1)'s four instructions are outstanding
  TileC ct = tile_coords(tile0);
  a_setup(ct);
  w_setup(ct);
  issue_w(0);
  issue_a(0);
  issue_w(1);
  wait_vm<WDMA>();
  bar();
  if (wr == 1) bar();                       // the stagger: wave group 1 runs one barrier behind group 0
  zero6();
  int kt = 0, ti = 0, buf = 0;
  const int total = my_tiles * nkt;
#pragma unroll 1
  for (int g = 0; g < total; ++g) {
    // PA
    read_a(0);
    read_w(fw0, 0);
    read_w(fw1, 1);
    if constexpr (N160) read_w4();
    issue_a(buf ^ 1);
    lgkm0();
    bar();
    __builtin_amdgcn_s_setprio(1);
    mfmas(0, 0, fw0);
    mfmas(0, 1, fw1);
    if constexpr (N160) mfmas4(0);
    __builtin_amdgcn_s_setprio(0);
    bar();
    // PB
    read_a(1);
    issue_w(buf);
    wait_vm<WDMA>();
    lgkm0();
    bar();
    __builtin_amdgcn_s_setprio(1);
    if constexpr (N160) mfmas4(1);
    mfmas(1, 1, fw1);
    mfmas(1, 0, fw0);
    __builtin_amdgcn_s_setprio(0);
    const bool last = kt + 1 == nkt;
    // group 0 stores its half of the tile behind the phase's closing barrier, group 1 (one barrier behind) in front of it:
    // the two epilogues then run side by side instead of one after the other
    if (!(last && wr == 1)) bar();
    buf ^= 1;
    ab0 ^= (uint32_t)HT; ab1 ^= (uint32_t)HT;
    if constexpr (N160) { wb0 = wbase160 + (uint32_t)(buf * WS160); wb1 = wb0 ^ 64u; }
    else { wb0 ^= (uint32_t)HT; wb1 ^= (uint32_t)HT; }
    if (last) {
      epilogue6(ct, smem + STG + wave * (N160 ? 5120 : 4096), tile0 + ti * nxb);
      zero6();
      kt = 0;
      ++ti;
      if (ti < my_tiles) ct = tile_coords(tile0 + ti * nxb);
      if (wr == 1) bar();
    } else {
      ++kt;
    }
  }
  if (wr == 0) bar();
  wait_vm<0>();                              // the cursors' tail issues (zero-filled, into slots nobody reads) drain before exit
}

template <typename T, int BN>
static int launch8(const GemmP& p, hipStream_t st) {
  GemmP q = p;
  q.ntm = (p.M + 255) / 256;
  q.ntn = (p.N + BN - 1) / BN;
  q.gn_chunks = p.gn_partial ? gemm_big_gn_chunks(p) : 0;
  if (q.gn_chunks == 0) q.gn_partial = nullptr;
  q.tw = 0; q.tw_log2 = 0; q.tpr = 0; q.tpi = 0;
  if (p.taps == 9) {
    q.tw = 16; q.tw_log2 = 4;
    q.tpr = p.Wo / 16;
    q.tpi = q.tpr * (p.Ho / 16);
  }
  constexpr size_t lds = BN == 160 ? (size_t)(65536 + 2 * 24576 + 8 * 5120) : (size_t)(2 + BN / 128) * 32768 + 32768;
  const int zdim = p.batch > 1 ? p.batch : 1;
  int nwg = q.ntm * q.ntn;
  if (nwg > 256) nwg = 256;
  nwg = (nwg + 7) & ~7;
  dim3 grid(nwg, zdim);
  if (p.taps == 1) {
    auto kfn = gemm8_kernel<T, false, BN>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, q);
  } else if constexpr (BN != 160) {
    auto kfn = gemm8_kernel<T, true, BN>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, q);
  } else {
    return DFW_ESHAPE;
  }
  DFW_CHECK_LAUNCH();
  return 0;
}

// Shapes of gemm_big's 256 x 256 configuration with a 64-deep K walk: storage-dtype output, N % 256 == 0, K % 64 == 0 (convs:
// Cin % 64 == 0), at least four K-tiles.
bool gemm8_eligible(const GemmP& p, int bn) {
  if (!cfg().k8 || p.out_mode != DFW_OUT_T || p.res_f32) return false;
  if ((p.K % 64) != 0 || (p.Cin % 64) != 0 || p.K / 64 < 2) return false;
  if (bn == 256) return (p.N % 256) == 0;
  return bn == 128 && (p.N % 64) == 0 && cfg().k8 >= 2;
}

// The 256 x 160 tile: Linear shapes with N a multiple of 160 but not of 128 (the UNet's 320-channel 64^2 level: to_out, proj_in /
// proj_out, FF2; N = 960: the fused QKV projection with its column scale) and enough rows that its tiles fill the chip; bias / row bias /
// residual / column-scale epilogue (no GEGLU, no activation).
bool gemm8_n160_eligible(const GemmP& p) {
  if (!cfg().k8 || !cfg().big_kernels || p.taps != 1 || p.out_mode != DFW_OUT_T || p.res_f32 || p.geglu || p.act != DFW_ACT_NONE) return false;
  if (p.batch > 1 || p.splitk > 1 || (p.N % 160) != 0 || (p.N % 128) == 0 || (p.K % 64) != 0 || p.K / 64 < 2) return false;
  if ((p.ldc % 8) != 0 || ((uintptr_t)p.C & 15)) return false;      // the epilogue stores 16-byte chunks of output rows
  return (long long)((p.M + 255) / 256) * (p.N / 160) >= cfg().big_min_tiles;
}

int launch_gemm8(const GemmP& p, hipStream_t st, int bn) {
  if (bn == 160) return p.dtype_bf16 ? launch8<__bf16, 160>(p, st) : launch8<_Float16, 160>(p, st);
  if (bn == 256) return p.dtype_bf16 ? launch8<__bf16, 256>(p, st) : launch8<_Float16, 256>(p, st);
  return p.dtype_bf16 ? launch8<__bf16, 128>(p, st) : launch8<_Float16, 128>(p, st);
}

}  // namespace dfw
