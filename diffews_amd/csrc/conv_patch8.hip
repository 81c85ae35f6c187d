// conv3x3 (stride 1, pad 1) with an LDS-resident input PATCH (conv_patch.hip's idea) on gemm8.hip's eight-phase schedule:
// K-tile = one tap of one 64-channel chunk (128-byte LDS rows = whole cache lines per LDS-DMA lane group), the W operand
// staged in half-tiles with two buffers each and refilled two K-tiles ahead, the two wave groups on ONE instruction stream
// staggered by one barrier, one counted s_waitcnt vmcnt per K-tile, never a DMA between a barrier and the MFMAs behind it.
//
// A workgroup owns a 16 x 16 output-pixel tile x BN channels.  Per 64-channel chunk it holds the 18 x 18 input patch
// (324 pixels x 128 B, out-of-image pixels zero-filled by the buffer bounds check) in one of two patch buffers; the nine
// taps take their MFMA A fragments from it at shifted pixel addresses, so per chunk the CU ingests 41 KB of A instead of
// the 9 x 32 KB gemm8_kernel<conv> gathers.  The next chunk's patch arrives in six pieces per wave, one per tap in the
// PA phases of taps 1..6 (one LDS-DMA instruction beside the tap's W half-tiles' four).
//
// LDS (160 KiB): patch 0 | patch 1 (48 KiB each: 6 pieces x 8 waves x 1 KiB) | W half-tile slots (BN / 128 x 2 x 16 KiB; the
// 256 x 160 tile: two 24 KiB stages of 160 + 32 pad rows at 0x18000 / 0x20000, 152 KiB in all).
// The epilogue's 32 KiB (256 x 160 tile: 40 KiB) staging area is the patch buffer the tile has just RETIRED (its last reader's ds_reads returned
// before the barrier in front of the tile's last 32 MFMAs; the next writer of that buffer is the patch of the NEXT tile's
// second chunk, issued from that tile's tap 1 on, four barriers behind both groups' epilogues).
//
// Patch pixel q = qy * 18 + qx lives at q * 128; its 16-byte chunk c sits in slot c ^ (q & 6): for ANY 16 consecutive
// pixels (any tap offset) the four lane groups of a ds_read_b128 -- {0-3, 12-15, 20-27} ... mix two l4 values -- hit 16
// distinct slots of the 256-byte bank row (exhaustive check over all start pixels, both K halves; the gemm8 form
// (q >> 1) & 7 is 2-way for odd pixel pairs).  A DMA piece covers 8 pixels whose first index is a multiple of 8, so the
// source-side swizzle is lane-constant: chunk (lane & 7) ^ ((lane >> 3) & 6).
//
// Per K-tile g (W buffer g & 1), per wave; quadrant (s, u) = pixel rows MH s .. + MH - 1 of the wave's, columns 32 u .. + 31 of its 64:
//   PA  read A[s0] from the patch at tap (ky, kx), W[u0], W[u1] | patch piece (taps 1..6) | lgkmcnt(0) | bar | MFMA | bar
//   PB  read A[s1]                          | issue W(g+2) -> buffer g & 1, vmcnt(2 WH) | lgkmcnt(0) | bar | MFMA | bar
// RAW: PB(g)'s wait leaves only W(g+2) outstanding -- W(g+1) and the piece of PA(g) have landed for every wave before the
// barriers that precede PA(g+1); the sixth piece is retired at tap 6, two taps before the next chunk reads the patch.
// WAR: W(g) is last read in PA(g) (returned before the reader's first barrier of the phase) and refilled in PB(g); a patch
// buffer is last read in PB of its chunk's tap 8 and refilled from tap 1 of the next chunk on.
#include "gemm_common.h"
#include <type_traits>

namespace dfw {

// GroupNorm + SiLU of the INPUT inside this kernel (each arriving patch normalised once, in LDS, by the wave that fetched it; bit-equal
// to the separate pass; scratch/conv_patch8_gnin_fused.patch.txt) was built and measured in round 4: the conv gets 3-33 % slower and
// the inference step 1.15 ms slower than with dfw_groupnorm's own pass (profiles/r04_gnin_fusion.txt) -- not in the library.
template <typename T, int BN>
__global__ __launch_bounds__(512, 2) void conv_patch8_kernel(const GemmP p) {
  constexpr int BM = 256, PATCH = 49152;
  constexpr bool N160 = BN == 160;                      // 256 x 160 tile (the N = 320 UNet layers): wave grid 4 x 2, wave tile 64 x 80
  constexpr int WH = BN / 128;                          // W half-tiles per K-tile (BN = 160: one 160-row stage, see issue_w)
  constexpr int WGN = N160 ? 2 : BN / 64, WGM = 8 / WGN;   // wave grid
  constexpr int NBW = N160 ? 5 : 4;                     // 16-column blocks of a wave tile
  constexpr int WTM = BM / WGM, MB6 = WTM / 16, MH = MB6 / 2;   // wave tile rows, its pixel rows, pixel rows per quadrant
  constexpr int WB = 2 * PATCH;                         // W slots: WB + h * 32768 + buf * HT
  // BN = 160: two 24 KiB W buffers (160 rows + 32 rows the third wave-instruction of waves 4-7 zero-fills) at 0x18000 and 0x20000:
  // the read bases toggle by ^ 0x38000
  constexpr uint32_t HT = N160 ? 0x38000u : 16384u;     // XOR that toggles a W read base between the two buffers
  constexpr int WDMA = N160 ? 3 : 2 * WH;               // W wave-instructions per wave per K-tile
  constexpr int PW = 18, PPIX = 18 * 18;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2;                             // wave group (the stagger)
  const int wm = wave / WGN, wc = wave % WGN;           // position in the wave grid
  const uint32_t lds0 = lds_addr(smem);

  // ---- persistent tile walk (XCD-contiguous order)
  const int ntiles = p.ntm * p.ntn;
  const int nxb = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int Q = (ntiles + 7) >> 3;
  const int t_end = min(ntiles, (xcd + 1) * Q);
  const int tile0 = xcd * Q + (blockIdx.x >> 3);
  if (tile0 >= t_end) return;
  const int my_tiles = (t_end - tile0 + nxb - 1) / nxb;
  const int cpt = p.Cin >> 6;          // 64-channel chunks
  const int nkt = cpt * 9;             // K-tiles per tile

  char* Cb = p.C;
  const u32x4 ra = make_srd(p.A, p.a_bytes);
  const u32x4 rw = make_srd(p.W, p.w_bytes);

  auto tile_coords = [&](int t) -> TileC {
    TileC c;
    const int tn = t % p.ntn, tm = t / p.ntn;
    c.m0 = tm * BM;
    c.n0 = tn * BN;
    c.img = tm / p.tpi;
    const int t2 = tm - c.img * p.tpi, tyi = t2 / p.tpr, txi = t2 - tyi * p.tpr;
    c.oy0 = tyi << 4;
    c.ox0 = txi << 4;
    return c;
  };
  auto row_to_m = [&](const TileC& c, int r) -> int { return (c.img * p.Ho + c.oy0 + (r >> 4)) * p.Wo + c.ox0 + (r & 15); };

  // ---- patch cursor: the chunk whose patch is being fetched (one ahead of the chunk being computed)
  const int kc = (lane & 7) ^ ((lane >> 3) & 6);
  int nx_img = 0, nx_oy0 = 0, nx_ox0 = 0, nx_c = 0, nx_ti = 0;
  bool nx_live = true;
  auto nx_set = [&](const TileC& c) { nx_img = c.img; nx_oy0 = c.oy0; nx_ox0 = c.ox0; };
  auto nx_advance = [&]() {
    if (++nx_c == cpt) {
      nx_c = 0;
      if (++nx_ti < my_tiles) nx_set(tile_coords(tile0 + nx_ti * nxb));
      else nx_live = false;
    }
  };
  auto issue_piece = [&](int i, int pbuf) __attribute__((always_inline)) {   // piece i (0..5) of this wave -> patch buffer pbuf
    const int pi = wave + 8 * i;
    const int q = 8 * pi + (lane >> 3);
    const int qy = (q * 3641) >> 16, qx = q - qy * PW;        // q / 18 for q < 512
    const int iy = nx_oy0 - 1 + qy, ix = nx_ox0 - 1 + qx;
    const bool ok = q < PPIX && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && nx_live;
    const uint32_t off = ok ? (uint32_t)((((size_t)nx_img * p.Hi + iy) * p.Wi + ix) * p.lda + nx_c * 64 + kc * 8) * (uint32_t)sizeof(T) : kOOB;
    dma16(ra, off, lds0 + (uint32_t)(pbuf * PATCH) + (uint32_t)pi * 1024u);
  };

  // ---- W cursor (two K-tiles ahead).  Wave-instruction j (0, 1) of this wave covers rows (j * 8 + wave) * 8 .. + 8 of a
  // half-tile: lane -> row lane >> 3, LDS slot lane & 7, source chunk slot ^ ((row >> 1) & 7) (gemm8.hip's W image)
  const int lrow = lane >> 3;
  const int wkc = (lane & 7) ^ ((((wave & 1) << 2) + (lane >> 4)) & 7);
  int w_kt = 0, w_ti = 0;
  bool w_live = true;
  uint32_t w_v0 = 0;
  int w_n = 0;
  auto w_setup = [&](const TileC& c) {
    w_n = c.n0 + wave * 8 + lrow;
    w_v0 = (uint32_t)(((size_t)w_n * p.K + wkc * 8) * sizeof(T));
  };
  auto issue_w = [&](int buf) __attribute__((always_inline)) {
    const int cc = w_kt / 9, tap = w_kt - cc * 9;
    const uint32_t koff = (uint32_t)(tap * p.Cin + cc * 64) * (uint32_t)sizeof(T);
    if constexpr (N160) {
      const uint32_t dst = lds0 + (buf ? 0x20000u : 0x18000u) + (uint32_t)wave * 1024u;
#pragma unroll
      for (int j = 0; j < 3; ++j) {      // rows j * 64 + wave * 8 + lrow of the 160 (j = 2: waves 0-3; the others write zeros)
        const uint32_t off = w_v0 + (uint32_t)(j * 64) * (uint32_t)p.K * (uint32_t)sizeof(T) + koff;
        const bool ok = w_live && (j < 2 || wave < 4);
        dma16(rw, ok ? off : kOOB, dst + j * 8192);
      }
    }
#pragma unroll
    for (int h = 0; h < (N160 ? 0 : WH); ++h) {
      const uint32_t dst = lds0 + (uint32_t)(WB + h * 32768 + buf * HT) + (uint32_t)wave * 1024u;
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

  // ---- fragment read bases.  W as in gemm8.hip (buffer 0; toggled by ^ HT per K-tile).  Patch: pixel of (tap (ky, kx), pixel
  // row r of the wave's) is q = (wm * MB6 + r + ky) * 18 + kx + l15; 18 (wm * MB6) is a multiple of 8 and 18 r = 2 r (mod 8), so
  // q & 6 = (kx + l15 + 2 ((r + ky) & 3)) & 6: per kx four lane-constant bases (one per (r + ky) & 3), the row term
  // (r + ky) * 18 * 128 an immediate of the ds_read; the second K half is the base ^ 64.
  uint32_t wb0, wb1;
  uint32_t ab[3][4];
  {
    const int l15 = lane & 15, l4 = lane >> 4;
    const uint32_t fb = (uint32_t)(l15 * 128 + ((l4 ^ (l15 >> 1)) << 4));
    if constexpr (N160) wb0 = (uint32_t)(WB + wc * 80 * 128) + fb;      // rows wc * 80 .. + 79 (80 / 2 = 0 mod 8: same swizzle term)
    else wb0 = (uint32_t)(WB + (wc >> 1) * 32768 + (wc & 1) * 64 * 128) + fb;
    wb1 = wb0 ^ 64u;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int u = kx + l15;
        ab[kx][rr] = (uint32_t)((wm * MB6 * PW + u) * 128 + ((l4 ^ ((u + 2 * rr) & 6)) << 4));
      }
  }

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
  auto read_a = [&](int s, int tap) __attribute__((always_inline)) {    // tap: compile-time (the tap loop is unrolled)
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int i = 0; i < MH; ++i) {
      const int r = MH * s + i + ky;
      const uint32_t a0 = ab[kx][r & 3];
      fa[i][0] = as_v8<T>(*(const i32x4*)(smem + a0 + r * PW * 128));
      fa[i][1] = as_v8<T>(*(const i32x4*)(smem + (a0 ^ 64u) + r * PW * 128));
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

  // ---- staged epilogue (gemm8.hip's: 32 tile rows per round through the wave's 4 KiB; row r, 16-byte chunk c at
  // ((c ^ (r & 7)) << 4) of a 128-byte line; statistics of the STORED values; whole-line stores)
  auto epilogue6 = [&](const TileC& c, char* stg, int tile_id) {
    constexpr int NB6 = 4;
    if (c.n0 + wc * 64 >= p.N) return;      // ragged N (a multiple of 64 on 128-wide tiles): this wave multiplied zero-filled W rows
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15 = lane_e & 15, l4 = lane_e >> 4, lane = lane_e;
    if constexpr (N160) {
      // 64 x 80 wave tile: 32 rows per round through 160-byte staging rows (5 KiB per wave, `stg` is this wave's), stored as ten
      // 16-byte chunks per row = one contiguous 160-byte run of the output row; no fused statistics (10 channels per group)
#pragma unroll
      for (int i = 0; i < MB6 / 2; ++i) {
        f32x4 add[2][NBW];
        i32x2 res[2][NBW];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int m = row_to_m(c, wm * WTM + i * 32 + h * 16 + l15);
#pragma unroll
          for (int j = 0; j < NBW; ++j) {
            const int n = c.n0 + wc * 80 + j * 16 + 4 * l4;
            f32x4 bb = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bb = *(const f32x4*)(p.bias + n);
            if (p.rowbias) {
              const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)c.img * p.ldrb + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) bb[e] += r[e];
            }
            add[h][j] = bb;
            res[h][j] = i32x2{0, 0};
            if (p.residual) res[h][j] = *(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T));
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < NBW; ++j) {
            float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.residual) unpack4<T>(res[h][j], r);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc6[2 * i + h][j][e] + add[h][j][e] + r[e]) * p.out_scale;
            *(i32x2*)(stg + (h * 16 + l15) * 160 + j * 32 + l4 * 8) = pack4<T>(v);
          }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const int idx = lane + 64 * t;               // 320 chunks of the round
          const int r = (idx * 6554) >> 16, ch = idx - r * 10;
          const int m2 = row_to_m(c, wm * WTM + i * 32 + r);
          const i32x4 val = *(const i32x4*)(stg + r * 160 + ch * 16);
          *(i32x4*)(Cb + ((size_t)m2 * p.ldc + c.n0 + wc * 80 + ch * 8) * sizeof(T)) = val;
        }
      }
      return;
    }
    float gs0 = 0.f, gs1 = 0.f, gq0 = 0.f, gq1 = 0.f;
#pragma unroll
    for (int i = 0; i < MB6 / 2; ++i) {
      f32x4 add[2][NB6];
      i32x2 res[2][NB6];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = row_to_m(c, wm * WTM + i * 32 + h * 16 + l15);
#pragma unroll
        for (int j = 0; j < NB6; ++j) {
          const int n = c.n0 + wc * 64 + j * 16 + 4 * l4;
          f32x4 bb = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) bb = *(const f32x4*)(p.bias + n);
          if (p.rowbias) {
            const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)c.img * p.ldrb + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] += r[e];
          }
          add[h][j] = bb;
          res[h][j] = i32x2{0, 0};
          if (p.residual) res[h][j] = *(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T));
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < NB6; ++j) {
          float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.residual) unpack4<T>(res[h][j], r);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (acc6[2 * i + h][j][e] + add[h][j][e] + r[e]) * p.out_scale;
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

  // ---- prologue: patch of chunk 0 -> buffer 0, W(0), W(1) in stream order; chunk 0's patch and W(0) have landed once only
  // W(1)'s instructions are outstanding
  TileC ct = tile_coords(tile0);
  nx_set(ct);
  w_setup(ct);
#pragma unroll
  for (int i = 0; i < 6; ++i) issue_piece(i, 0);
  nx_advance();
  issue_w(0);
  issue_w(1);
  wait_vm<WDMA>();
  bar();
  if (wr == 1) bar();                       // the stagger: wave group 1 runs one barrier behind group 0
  zero6();
  int pb = 0;                               // patch buffer of the chunk being computed
  int wbuf = 0;                             // W buffer of the K-tile being computed (nine K-tiles per chunk: not the tap's parity)
#pragma unroll 1
  for (int ti = 0; ti < my_tiles; ++ti) {
#pragma unroll 1
    for (int cc = 0; cc < cpt; ++cc) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // PA
        read_a(0, tap);
        read_w(fw0, 0);
        read_w(fw1, 1);
        if constexpr (N160) read_w4();
        if (tap >= 1 && tap <= 6) issue_piece(tap - 1, pb ^ 1);
        lgkm0();
        bar();
        __builtin_amdgcn_s_setprio(1);
        mfmas(0, 0, fw0);
        mfmas(0, 1, fw1);
        if constexpr (N160) mfmas4(0);
        __builtin_amdgcn_s_setprio(0);
        bar();
        // PB
        read_a(1, tap);
        issue_w(wbuf);                      // W(g+2) -> the buffer this K-tile's W came from
        wait_vm<WDMA>();
        lgkm0();
        bar();
        __builtin_amdgcn_s_setprio(1);
        if constexpr (N160) mfmas4(1);
        mfmas(1, 1, fw1);
        mfmas(1, 0, fw0);
        __builtin_amdgcn_s_setprio(0);
        const bool last = tap == 8 && cc + 1 == cpt;
        // group 0 stores its half of the tile behind the phase's closing barrier, group 1 (one barrier behind) in front of it
        if (!(last && wr == 1)) bar();
        wbuf ^= 1;
        wb0 ^= (uint32_t)HT; wb1 ^= (uint32_t)HT;
        if (tap == 6) nx_advance();
        if (tap == 8) {
          // this chunk's patch buffer is retired: flip the fragment bases to the other one
          const uint32_t d = pb ? (uint32_t)(-PATCH) : (uint32_t)PATCH;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) ab[kx][rr] += d;
          if (last) {
            epilogue6(ct, smem + pb * PATCH + wave * (N160 ? 5120 : 4096), tile0 + ti * nxb);
            zero6();
            if (ti + 1 < my_tiles) ct = tile_coords(tile0 + (ti + 1) * nxb);
            if (wr == 1) bar();
          }
          pb ^= 1;
        }
      }
    }
  }
  if (wr == 0) bar();
  wait_vm<0>();                              // the cursors' tail issues (zero-filled, into slots nobody reads) drain before exit
}

template <typename T, int BN>
static int launch_patch8(const GemmP& p, hipStream_t st, int gn_chunks) {
  GemmP q = p;
  q.ntm = p.M / 256;
  q.ntn = (p.N + BN - 1) / BN;
  q.tw = 16; q.tw_log2 = 4;
  q.tpr = p.Wo / 16;
  q.tpi = q.tpr * (p.Ho / 16);
  q.gn_chunks = p.gn_partial ? gn_chunks : 0;
  if (q.gn_chunks == 0) q.gn_partial = nullptr;
  constexpr size_t lds = BN == 160 ? (size_t)0x26000 : 2 * 49152 + (size_t)(BN / 128) * 32768;
  int nwg = q.ntm * q.ntn;
  if (nwg > 256) nwg = 256;
  nwg = (nwg + 7) & ~7;
  auto kfn = conv_patch8_kernel<T, BN>;
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), lds, st, q);
  DFW_CHECK_LAUNCH();
  return 0;
}

// Shapes: conv_patch.hip's (stride-1 / pad-1 conv3x3, whole 16 x 16 pixel tiles, Cin % 64 == 0, storage-dtype NHWC output without
// activation, enough tiles) on the 16-bit path: N % 256 == 0 -> 256 x 256 tiles, or 256 x 128 tiles where that is what fills the
// chip (dfw_config.conv_patch >= 3); N % 64 == 0 otherwise -> 256 x 128 tiles, the last tile column ragged by 64
// (dfw_config.conv_patch >= 4: measured slower than conv_patch_kernel<512,128> on the N = 128 layers, not the default).
// The fp32-output layers of the fp32 residual stream stay on conv_patch_kernel<.,.,F32O>: an F32O instantiation of this kernel
// (direct fp32 stores from the accumulators) was built and measured +0.85 ms on the parity-mode step (50.6 vs 49.7 ms).
bool conv_patch8_eligible(const GemmP& p, int& bn) {
  const int mode = cfg().conv_patch;
  if (mode < 3) return false;
  if (p.taps != 9 || p.stride != 1 || p.pad != 1 || p.ups || p.splitk > 1 || p.batch > 1) return false;
  if (p.Hi != p.Ho || p.Wi != p.Wo || (p.Wo % 16) != 0 || (p.Ho % 16) != 0 || (p.Cin % 64) != 0 || (p.M % 256) != 0) return false;
  if (p.out_mode != DFW_OUT_T || p.act != DFW_ACT_NONE || p.geglu || p.cs_n > 0 || p.res_f32) return false;
  if (p.rows_per_img != p.Ho * p.Wo) return false;
  if ((p.ldc % 8) != 0 || ((uintptr_t)p.C & 15)) return false;      // the epilogues store 16-byte chunks of output rows
  const long long mt = p.M / 256;
  if ((p.N % 160) == 0 && (p.N % 128) != 0 && mt * (p.N / 160) >= cfg().big_min_tiles) {
    bn = 160;       // N = 320 (the UNet's 64^2 level on the lock-step batch: 128 x 2 = 256 tiles, one per CU)
    return true;
  }
  if ((p.N % 256) == 0) {
    // too few 256 x 256 tiles for the chip (the VAE's 64^2 level on 4 images: 128): the 256 x 128 tile doubles the count
    bn = (mt * (p.N / 256) >= cfg().big_min_tiles) ? 256 : 128;
  } else if ((p.N % 64) == 0 && mode >= 4) bn = 128;
  else return false;
  return mt * ((p.N + bn - 1) / bn) >= cfg().big_min_tiles;
}

int conv_patch8_gn_chunks(const GemmP& p) {
  int bn = 0;
  if (p.gn_groups <= 0 || !conv_patch8_eligible(p, bn) || bn == 160 || p.N % p.gn_groups) return 0;
  const int cpg = p.N / p.gn_groups;
  if (cpg < 4 || cpg > 64 || (cpg & (cpg - 1))) return 0;
  return (p.Wo / 16) * (p.Ho / 16) * (8 / (bn / 64));
}

int launch_conv_patch8(const GemmP& p, hipStream_t st) {
  int bn = 0;
  if (!conv_patch8_eligible(p, bn)) return DFW_ESHAPE;
  const int chunks = conv_patch8_gn_chunks(p);
  const bool bf = p.dtype_bf16 != 0;
  if (bn == 160) return bf ? launch_patch8<__bf16, 160>(p, st, chunks) : launch_patch8<_Float16, 160>(p, st, chunks);
  if (bn == 256) return bf ? launch_patch8<__bf16, 256>(p, st, chunks) : launch_patch8<_Float16, 256>(p, st, chunks);
  return bf ? launch_patch8<__bf16, 128>(p, st, chunks) : launch_patch8<_Float16, 128>(p, st, chunks);
}

}  // namespace dfw
