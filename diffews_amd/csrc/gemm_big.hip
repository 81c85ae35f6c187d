// 256-row implicit-GEMM kernel for the large conv / linear shapes (VAE and the 64x64 UNet level).
//
// Why a second kernel: with 128x128x64 tiles every CU has to pull 32 KB through its 64 B/clk
// L2->LDS path per 2.1 MFLOP -- exactly the MFMA rate, so that kernel tops out near 1/3 of peak.
// Here a workgroup of 8 waves owns a 256 x BN (BN = 256 | 128) tile: 128 | 87 FLOP per staged byte.
//
// Structure (one workgroup per CU, 2 waves per SIMD):
//   * K-step = 32; LDS is a ring of S = 4 stages of [(256 + BN) rows][32 k] (64-byte rows,
//     16-byte chunks XOR-swizzled by ((row>>1)&3): conflict-free ds_read_b128 fragments -- measured: the
//     ((row>>2)&3) swizzle of rounds 1-2 had a 2-way conflict on every fragment read, SQ_LDS_BANK_CONFLICT = 50 % of
//     SQ_LDS_IDX_ACTIVE, scratch/proto/lds_swz.hip: 116 vs 221 B/clk/CU);
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
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

namespace dfw {

// PP (ping-pong): the two waves of every SIMD (wave w and w + 4) run half a K-step apart, two barriers
// per K-step: while waves 0-3 issue their 16 MFMAs from registers, waves 4-7 read the stage's 12
// fragments from LDS, and vice versa -- the MFMA pipe always has exactly one wave feeding it and the
// LDS reads are never in its way.  With all eight waves in the same phase (PP = false) a K-step costs
// LDS phase + MFMA phase instead of their maximum (0.93 -> see DESIGN.md for the measured gain).
//   waves 0-3:        reads(k) | B2k | issue(k+3) mfma(k) wait | B2k+1 | reads(k+1) ...
//   waves 4-7:                 | B2k | issue(k+3) reads(k) wait | B2k+1 | mfma(k)   | B2k+2 ...
// RAW: every wave's counted wait for stage k+1 sits before B2k+1, its first reader starts after it.
// WAR: stage k+3 overwrites the slot of stage k-1, whose last reads (waves 4-7, between B2k-2 and
// B2k-1) were retired by the lgkmcnt wait in front of their MFMAs, i.e. before B2k.
// M16 (with PP): v_mfma_f32_16x16x32 instead of 32x32x16 -- same cycles per FLOP, same LDS image and
// fragment read count (8 + 4 ds_read_b128 per K-step), but on random data the chip holds a higher clock
// on this shape (MI355X_MICROARCH 'DVFS give-back' item 7: ~1.12-1.15x); the kernels are power-limited, so
// that is where the remaining headroom was.  Lane (l&15, l>>4) of block (i, j) owns pixel i*16 + (l&15),
// channels j*16 + 4*(l>>4) + 0..3.
// F32O (16x16x32 ping-pong configurations only): fp32 NHWC output and an fp32 (or storage-dtype) residual -- the fp32
// residual stream (dfw_gemm_args.residual_f32 / DFW_OUT_F32).  The accumulators are stored straight from registers
// (a lane owns 4 consecutive channels of a pixel: 16-byte stores, four lanes per 64-byte segment); its own
// instantiation, so the 16-bit default path's staged epilogue compiles exactly as before.
template <typename T, int BM, int BN, int BK, int S, int OCC, bool CONV, bool PP, bool M16 = false, bool F32O = false>
__global__ __launch_bounds__(512, 2 * OCC) void gemm_big_kernel(const GemmP p) {
  // S ring stages (S-1 in flight); OCC workgroups per CU (2 * OCC waves per SIMD)
  constexpr int CH = BK / 8, RB = BK * 2;       // 16-byte chunks per row, bytes per row
  constexpr int RPI = 1024 / RB;                // tile rows per DMA wave-instruction
  constexpr int WGN = BN / 64, WGM = 8 / WGN;   // wave grid: 2x4 (BN=256) or 4x2 (BN=128)
  constexpr int WTM = BM / WGM;                 // 128 or 64
  constexpr int MB = WTM / 32, NB = 2;
  constexpr int STAGE = (BM + BN) * RB;         // bytes
  constexpr int SA = BM / RPI / 8, SW = BN / RPI / 8;   // DMA wave-instructions per stage per wave
  constexpr int DPS = SA + SW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = lane & 31, lh = lane >> 5;
  const uint32_t lds0 = lds_addr(smem);

  // ---- persistent tile walk (same XCD-contiguous order as gemm.hip)
  const int ntiles = p.ntm * p.ntn;
  const int nxb = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int Q = (ntiles + 7) >> 3;
  const int t_end = min(ntiles, (xcd + 1) * Q);
  const int tile0 = xcd * Q + (blockIdx.x >> 3);
  if (tile0 >= t_end) return;
  const int my_tiles = (t_end - tile0 + nxb - 1) / nxb;
  const int nks = p.K / BK;            // K-steps
  const int cpt = p.Cin / BK;          // K-steps per tap

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
      c.oy0 = tyi * (BM / 16);
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

  // ---- loader: wave-instruction i of a wave covers tile rows (i*8 + wave)*RPI .. +RPI (1 KiB of
  // LDS, written linearly: lane -> row + lane/CH, 16-byte slot lane%CH).  The bank swizzle lives
  // on the SOURCE side: slot j of a row holds global chunk j ^ swz(row), with
  //   BK = 32 (64-B rows):  swz = (row>>1)&3      BK = 64 (128-B rows, full cache lines): swz = (row>>1)&7
  // which depends only on the lane (and the wave's parity), not on i.
  const int lrow = lane / CH;
  const int kc = BK == 32 ? ((lane & 3) ^ ((lrow >> 1) & 3))
                          : ((lane & 7) ^ ((((wave & 1) << 2) + (lrow >> 1)) & 7));
  uint32_t a_off[SA];
  int a_iy0[SA], a_ix0[SA];
  uint32_t a_pix[SA];
  uint32_t w_off[SW];
  const unsigned limH = p.ups ? 2 * p.Hi : p.Hi, limW = p.ups ? 2 * p.Wi : p.Wi;
  const int ush = p.ups ? 1 : 0;
  auto setup_loader = [&](const TileC& c) {
#pragma unroll
    for (int i = 0; i < SA; ++i) {
      int oy = 0, ox = 0, img = 0;
      const int m = row_to_m(c, (i * 8 + wave) * RPI + lrow, oy, ox, img);
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
      const int n = c.n0 + (i * 8 + wave) * RPI + lrow;
      w_off[i] = n < p.N ? (uint32_t)(((size_t)n * p.K + kc * 8) * sizeof(T)) : kOOB;
    }
  };
  auto issue = [&](int ks, int slot) {
    const uint32_t dst = lds0 + (uint32_t)slot * STAGE + (uint32_t)wave * 1024u;
    // conv K walk: step ks of the tile -> (tap, channel chunk); derived from ks (scalar arithmetic), not
    // carried as loop state (state captured by the two loop instantiations ended up in scratch, and its
    // reload put an s_waitcnt vmcnt(0) -- a full drain of the DMA ring -- into every K-step)
    int tap = 0, cc = 0;
    if constexpr (CONV) {
      if (p.conv_chunk_major) { cc = ks / 9; tap = ks - cc * 9; }
      else { tap = ks / cpt; cc = ks - tap * cpt; }
    }
    if constexpr (!CONV) {
#pragma unroll
      for (int i = 0; i < SA; ++i) dma16(ra, a_off[i] + (uint32_t)ks * RB, dst + i * 8192);
    } else {
      const int ky = tap / 3, kx = tap - ky * 3;
      const uint32_t coff = (uint32_t)(cc * BK + kc * 8);
#pragma unroll
      for (int i = 0; i < SA; ++i) {
        int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        const bool ok = (unsigned)iy < limH && (unsigned)ix < limW;
        iy >>= ush;
        ix >>= ush;
        const uint32_t off = ((a_pix[i] + (uint32_t)(iy * p.Wi + ix)) * (uint32_t)p.lda + coff) * (uint32_t)sizeof(T);
        dma16(ra, ok ? off : kOOB, dst + i * 8192);
      }
    }
    uint32_t koff = (uint32_t)ks * RB;
    if constexpr (CONV) {
      // K walk of a conv tile.  Channel-chunk-major (the nine taps of one 32-channel chunk back to back):
      // the shifted re-reads of the input patch come one K-step after the first touch and hit the XCD's
      // L2, where the tap-major walk (all chunks of a tap, then the next tap) re-read each byte Cin/32
      // steps later, after 32 workgroups had pushed ~4 MB through that 4 MB L2.  Fewer bytes from beyond
      // L2 is also the largest clock lever the guide lists for a power-limited MFMA loop (rule 28).
      koff = (uint32_t)(tap * p.Cin + cc * BK) * (uint32_t)sizeof(T);
    }
#pragma unroll
    for (int i = 0; i < SW; ++i) dma16(rw, w_off[i] + koff, dst + BM * RB + i * 8192);
  };
  // ---- fragment read addresses within a stage (k-substep s: ^ (s<<5))
  uint32_t lds_ra[MB], lds_rw[NB];
  const int sw4 = BK == 32 ? ((lr >> 1) & 3) : ((lr >> 1) & 7);
#pragma unroll
  for (int i = 0; i < MB; ++i) lds_ra[i] = (uint32_t)(wm * WTM + i * 32 + lr) * RB + (uint32_t)((lh ^ sw4) << 4);
#pragma unroll
  for (int j = 0; j < NB; ++j) lds_rw[j] = (uint32_t)(BM + wn * 64 + j * 32 + lr) * RB + (uint32_t)((lh ^ sw4) << 4);

  f32x16 acc[MB][NB];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  // Epilogue.  Non-GEGLU tiles are stored through a 4 KiB per-wave LDS staging area (the ring slot
  // that was read last and is free until the next barrier): the accumulator layout gives each lane
  // 4 channels of one row (8-byte pieces of 32 different rows per store instruction); after the
  // round trip a lane holds 16 contiguous bytes and a store instruction covers 8 rows x 128 B of
  // whole cache lines -- half the store instructions, no partial-line writes.
  constexpr bool kStage = STAGE >= 8 * 4096;
  auto epilogue = [&](const TileC& c, char* stg, int tile_id) {
    // ragged N (a multiple of 64, e.g. the UNet's 320 channels on 128-wide tiles): a wave tile is 64 columns, so it
    // lies entirely inside or entirely outside the matrix -- outside waves (they multiplied zero-filled W rows) skip
    if (c.n0 + wn * 64 >= p.N) return;
    float gs0 = 0.f, gs1 = 0.f, gq0 = 0.f, gq1 = 0.f;   // fused GroupNorm sums of this lane's channel pair
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      int oy_ = 0, ox_ = 0, img_ = 0;
      const int m = row_to_m(c, wm * WTM + i * 32 + lr, oy_, ox_, img_);
      if (m >= p.M && (!kStage || p.geglu)) continue;
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
      if constexpr (!kStage) {
        if (m < p.M) epi_block<T, NB, true>(p, Cb, m, img_, c.n0 + wn * 64 + 4 * lh, acc[i]);
        continue;
      }
      // all loads of this row block first (a load issued after a store waits for that store)
      f32x4 add[NB][4];
      i32x2 res[NB][4];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = c.n0 + wn * 64 + j * 32 + 8 * g + 4 * lh;
          f32x4 bb = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) bb = *(const f32x4*)(p.bias + n);
          if (p.rowbias && m < p.M) {
            const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)img_ * p.ldrb + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] += r[e];
          }
          add[j][g] = bb;
          res[j][g] = i32x2{0, 0};
          if (p.residual && m < p.M) res[j][g] = *(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T));
        }
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.residual) unpack4<T>(res[j][g], r);
          const float osc = (c.n0 + wn * 64 + j * 32 + 8 * g + 4 * lh) < p.cs_n ? p.cs : p.out_scale;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * g + e] + add[j][g][e] + r[e]) * osc;
          const int c16 = j * 4 + g;  // 16-byte chunk of the wave tile's 128-byte row
          *(i32x2*)(stg + lr * 128 + ((c16 ^ (lr & 7)) << 4) + lh * 8) = pack4<T>(v);
        }
      if constexpr (kStage) {
        // same wave wrote and reads this region: program order + the compiler's lgkmcnt wait suffice
        if (p.gn_partial) {
          // lane -> channel pair cp of the wave's 64 channels, half rh of the 32 staged rows
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
          int oy2 = 0, ox2 = 0, img2 = 0;
          const int m2 = row_to_m(c, wm * WTM + i * 32 + r, oy2, ox2, img2);
          const i32x4 val = *(const i32x4*)(stg + r * 128 + ((c16 ^ (r & 7)) << 4));
          if (m2 < p.M)
            *(i32x4*)(Cb + ((size_t)m2 * p.ldc + c.n0 + wn * 64 + c16 * 8) * sizeof(T)) = val;
        }
      }
    }
    if constexpr (kStage) {
      if (p.gn_partial) {
        // rows: other half-wave; channels: pairs -> groups (cpg/2 adjacent lanes); fixed order => deterministic
        float s = gs0 + gs1, q = gq0 + gq1;
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        const int cpg = p.N / p.gn_groups, ppg = cpg >> 1;
        for (int o = 1; o < ppg; o <<= 1) {
          s += __shfl_xor(s, o, 64);
          q += __shfl_xor(q, o, 64);
        }
        const int cp = lane & 31;
        if (lane < 32 && (cp & (ppg - 1)) == 0) {
          const int tm = tile_id / p.ntn;
          const int chunk = (tm - c.img * p.tpi) * WGM + wm;
          const int grp = (c.n0 + wn * 64 + 2 * cp) / cpg;
          float* o2 = p.gn_partial + (((size_t)c.img * p.gn_chunks + chunk) * p.gn_groups + grp) * 2;
          o2[0] = s;
          o2[1] = q;
        }
      }
    }
  };

  auto compute = [&](int slot) {
    const char* buf = smem + slot * STAGE;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
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
    __builtin_amdgcn_s_setprio(0);
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
  auto nxt = [](int slot) { return slot + 1 == S ? 0 : slot + 1; };

  // ---- pipeline: S-1 stages in flight; rs = slot read this step, ws = slot written this step.
  // nks >= S (host-checked).  The hot loop is branch-free; the last S-1 steps of a tile stage the
  // first S-1 steps of the next tile, so the ring never drains between tiles.
  TileC ct = tile_coords(tile0);
  setup_loader(ct);
  int rs = 0, ws = 0;
#pragma unroll
  for (int i = 0; i < S - 1; ++i) { issue(i, ws); ws = nxt(ws); }
  if constexpr (PP && M16) {
    static_assert(OCC == 1 && BK == 32 && kStage, "16x16x32 path: ping-pong configurations with the staged epilogue");
    constexpr int MB6 = WTM / 16, NB6 = 4;            // 16-row / 16-channel blocks of the 128 x 64 wave tile
    f32x4 acc6[MB6][NB6];
    typename Tr<T>::v8 fa6[MB6], fw6[NB6];
    const int l15 = lane & 15, l4 = lane >> 4;
    // block i / j is 16 rows = 1 KiB further on and the swizzle ((row >> 1) & 3) only sees row & 15:
    // one base address per operand, the blocks are immediates of the ds_read
    const uint32_t sw6 = (uint32_t)((l4 ^ ((l15 >> 1) & 3)) << 4);
    const uint32_t ra6 = (uint32_t)(wm * WTM + l15) * RB + sw6;
    const uint32_t rw6 = (uint32_t)(BM + wn * 64 + l15) * RB + sw6;
    auto reads = [&](int slot) __attribute__((always_inline)) {
      const char* ba = smem + slot * STAGE + ra6;
      const char* bw = smem + slot * STAGE + rw6;
#pragma unroll
      for (int j = 0; j < NB6; ++j) fw6[j] = as_v8<T>(*(const i32x4*)(bw + j * 16 * RB));
#pragma unroll
      for (int i = 0; i < MB6; ++i) fa6[i] = as_v8<T>(*(const i32x4*)(ba + i * 16 * RB));
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < MB6; ++i)
#pragma unroll
        for (int j = 0; j < NB6; ++j) acc6[i][j] = Tr<T>::mfma16(fw6[j], fa6[i], acc6[i][j]);
      __builtin_amdgcn_s_setprio(0);
    };
    auto zero6 = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < MB6; ++i)
#pragma unroll
        for (int j = 0; j < NB6; ++j) acc6[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // staged epilogue, 32 tile rows (two 16-row blocks) per round through the wave's 4 KiB; same LDS
    // image as the 32x32 path (row r: 16-byte chunk c at ((c ^ (r & 7)) << 4)), so the statistics and the
    // whole-line stores below are shared logic
    auto epilogue6 = [&](const TileC& c, char* stg, int tile_id) {
      // lane-derived indices are recomputed here, behind an opaque copy of the lane id: hoisted to kernel
      // entry (LICM) they live across the K loop and spill at this kernel's 256-register budget
      if (c.n0 + wn * 64 >= p.N) return;      // ragged N: see epilogue()
      int lane_e = lane;
      asm volatile("" : "+v"(lane_e));
      const int l15 = lane_e & 15, l4 = lane_e >> 4, lane = lane_e;
      if constexpr (F32O) {
        GnRegSums gsum;
        gsum.clear();
#pragma unroll
        for (int i = 0; i < MB6; ++i) {
          int oy_ = 0, ox_ = 0, img_ = 0;
          const int m = row_to_m(c, wm * WTM + i * 16 + l15, oy_, ox_, img_);
          if (m >= p.M) continue;
          if constexpr (!CONV) img_ = p.rowbias ? m / p.rows_per_img : 0;
          f32x4 add[NB6];
#pragma unroll
          for (int j = 0; j < NB6; ++j) {          // every load of the row block before its first store
            const int n = c.n0 + wn * 64 + j * 16 + 4 * l4;
            f32x4 bb = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bb = *(const f32x4*)(p.bias + n);
            if (p.rowbias) {
              const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)img_ * p.ldrb + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) bb[e] += r[e];
            }
            if (p.residual) {
              if (p.res_f32) {
                const f32x4 r = *(const f32x4*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(float));
#pragma unroll
                for (int e = 0; e < 4; ++e) bb[e] += r[e];
              } else {
                float r[4];
                unpack4<T>(*(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T)), r);
#pragma unroll
                for (int e = 0; e < 4; ++e) bb[e] += r[e];
              }
            }
            add[j] = bb;
          }
#pragma unroll
          for (int j = 0; j < NB6; ++j) {
            const int n = c.n0 + wn * 64 + j * 16 + 4 * l4;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc6[i][j][e] + add[j][e]) * p.out_scale;
            *(f32x4*)(Cb + ((size_t)m * p.ldc + n) * sizeof(float)) = v;
            gsum.add(j, v);
          }
        }
        if (p.gn_partial) {     // conv tiles are whole (every row stored): statistics of the fp32 values just written
          const int tm = tile_id / p.ntn;
          const int chunk = (tm - c.img * p.tpi) * WGM + wm;
          gsum.store(p, p.gn_partial + ((size_t)c.img * p.gn_chunks + chunk) * p.gn_groups * 2, c.n0 + wn * 64, lane);
        }
        return;
      }
      float gs0 = 0.f, gs1 = 0.f, gq0 = 0.f, gq1 = 0.f;
#pragma unroll
      for (int i = 0; i < MB6 / 2; ++i) {
        if (p.geglu) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            int oy_ = 0, ox_ = 0, img_ = 0;
            const int m = row_to_m(c, wm * WTM + i * 32 + h * 16 + l15, oy_, ox_, img_);
            if (m >= p.M) continue;
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {     // channel blocks 0,1 = value, 2,3 = gate (packing.pack_geglu)
              const int na = c.n0 + wn * 64 + jb * 16 + 4 * l4;
              if (na >= p.N) continue;
              const int no = ((c.n0 + wn * 64) >> 1) + jb * 16 + 4 * l4;
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
          int oy_ = 0, ox_ = 0, img_ = 0;
          const int m = row_to_m(c, wm * WTM + i * 32 + h * 16 + l15, oy_, ox_, img_);
          if constexpr (!CONV) img_ = (p.rowbias && m < p.M) ? m / p.rows_per_img : 0;
#pragma unroll
          for (int j = 0; j < NB6; ++j) {
            const int n = c.n0 + wn * 64 + j * 16 + 4 * l4;
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
        for (int j = 0; j < NB6; ++j) osc6[j] = (c.n0 + wn * 64 + j * 16 + 4 * l4) < p.cs_n ? p.cs : p.out_scale;
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
          int oy2 = 0, ox2 = 0, img2 = 0;
          const int m2 = row_to_m(c, wm * WTM + i * 32 + r, oy2, ox2, img2);
          const i32x4 val = *(const i32x4*)(stg + r * 128 + ((c16 ^ (r & 7)) << 4));
          if (m2 < p.M)
            *(i32x4*)(Cb + ((size_t)m2 * p.ldc + c.n0 + wn * 64 + c16 * 8) * sizeof(T)) = val;
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
          const int grp = (c.n0 + wn * 64 + 2 * cp) / cpg;
          float* o2 = p.gn_partial + (((size_t)c.img * p.gn_chunks + chunk) * p.gn_groups + grp) * 2;
          o2[0] = s2;
          o2[1] = q2;
        }
      }
    };
    auto bar = [&]() __attribute__((always_inline)) {
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    bool has_next = false;
    int ti_next = 0;
    auto kloop = [&](auto G1) __attribute__((always_inline)) {
      constexpr bool g1 = decltype(G1)::value;
#pragma unroll 1
      for (int k = 0; k < nks - (S - 1); ++k) {
        if constexpr (!g1) reads(rs);
        bar();
        issue(k + S - 1, ws);
        ws = nxt(ws);
        if constexpr (g1) reads(rs);
        else mfmas();
        wait_vm<(S - 2) * DPS>();
        bar();
        if constexpr (g1) mfmas();
        rs = nxt(rs);
      }
      if (has_next) setup_loader(tile_coords(tile0 + (ti_next) * nxb));
#pragma unroll 1
      for (int j = 0; j < S - 1; ++j) {
        if constexpr (!g1) reads(rs);
        bar();
        if (has_next) { issue(j, ws); ws = nxt(ws); }
        if constexpr (g1) reads(rs);
        else mfmas();
        const int younger = has_next ? S - 2 : S - 3 - j;
        if (younger >= 2) wait_vm<2 * DPS>();
        else if (younger == 1) wait_vm<DPS>();
        else wait_vm<0>();
        bar();
        if constexpr (g1) mfmas();
        rs = nxt(rs);
      }
    };
    wait_vm<(S - 2) * DPS>();
    bar();
    for (int ti = 0; ti < my_tiles; ++ti) {
      has_next = ti + 1 < my_tiles;
      ti_next = ti + 1;
      zero6();
      if (wave >= 4) kloop(std::true_type{});
      else kloop(std::false_type{});
      bar();
      char* stg = smem + (rs == 0 ? S - 1 : rs - 1) * STAGE + wave * 4096;
      epilogue6(ct, stg, tile0 + ti * nxb);
      if (has_next) ct = tile_coords(tile0 + (ti + 1) * nxb);
      else ws = rs;
    }
    return;
  }
  static_assert(!PP || M16, "the ping-pong schedule exists in its 16x16x32 form only (the 32x32x16 one measured 2 % slower)");
  for (int ti = 0; ti < my_tiles; ++ti) {
    const bool has_next = ti + 1 < my_tiles;
    zero_acc();
    for (int k = 0; k < nks - (S - 1); ++k) {
      wait_vm<(S - 2) * DPS>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      issue(k + S - 1, ws);
      ws = nxt(ws);
      compute(rs);
      rs = nxt(rs);
    }
    if (has_next) setup_loader(tile_coords(tile0 + (ti + 1) * nxb));
#pragma unroll
    for (int j = 0; j < S - 1; ++j) {
      retire_and_sync(has_next ? S - 2 : S - 2 - j);
      if (has_next) { issue(j, ws); ws = nxt(ws); }
      compute(rs);
      rs = nxt(rs);
    }
    char* stg = nullptr;
    if constexpr (kStage) {
      __builtin_amdgcn_s_barrier();   // every wave is done reading the last stage: its slot is free
      asm volatile("" ::: "memory");
      stg = smem + (rs == 0 ? S - 1 : rs - 1) * STAGE + wave * 4096;
    }
    epilogue(ct, stg, tile0 + ti * nxb);
    if (has_next) ct = tile_coords(tile0 + (ti + 1) * nxb);
    else ws = rs;
  }
}

template <typename T, int BM, int BN, int BK, int S, int OCC, bool PP = false, bool M16 = false, bool F32O = false>
static int launch_big(const GemmP& p, hipStream_t st) {
  GemmP q = p;
  q.ntm = (p.M + BM - 1) / BM;
  q.ntn = (p.N + BN - 1) / BN;
  q.gn_chunks = p.gn_partial ? gemm_big_gn_chunks(p) : 0;
  if (q.gn_chunks == 0) q.gn_partial = nullptr;
  q.tw = 0; q.tw_log2 = 0; q.tpr = 0; q.tpi = 0;
  if (p.taps == 9) {
    q.tw = 16; q.tw_log2 = 4;
    q.tpr = p.Wo / 16;
    q.tpi = q.tpr * (p.Ho / (BM / 16));
  }
  const size_t lds = (size_t)S * (BM + BN) * BK * 2;
  const int zdim = p.batch > 1 ? p.batch : 1;
  int nwg = q.ntm * q.ntn;
  if (nwg > 256 * OCC) nwg = 256 * OCC;
  nwg = (nwg + 7) & ~7;
  dim3 grid(nwg, zdim);
  // (the attribute is set on every launch: a host-side call of a few hundred ns, and no function-local static state)
  if (p.taps == 1) {
    auto kfn = gemm_big_kernel<T, BM, BN, BK, S, OCC, false, PP, M16, F32O>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, q);
  } else {
    auto kfn = gemm_big_kernel<T, BM, BN, BK, S, OCC, true, PP, M16, F32O>;
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, q);
  }
  DFW_CHECK_LAUNCH();
  return 0;
}

// Eligible: no split-K, N a multiple of 128 (wave tiles are 64 wide, GEGLU pairs stay inside one),
// storage-dtype NHWC output, conv maps divisible into (BM/16) x 16 pixel patches, and enough tiles to
// occupy the chip.  Configurations (BM x BN x BK): 256x256x32 (4-stage ring), 512x128x32,
// 256x128x64 (3-stage ring, 128-byte rows = full cache lines per DMA lane group), 256x128x32.
struct BigCfg { int bm, bn, bk, occ; };
static bool big_cfg_ok(const GemmP& p, const BigCfg& c) {
  // the 128x128 configuration stores through epi_block (per-quad column guards): ragged N is fine
  // 128-wide tiles also take N % 64 == 0 (whole 64-column wave tiles; the last column tile is half empty)
  if (c.bm == 128 ? (p.N % 8) != 0 : (c.bn == 128 ? (p.N % 64) != 0 : (p.N % c.bn) != 0)) return false;
  if ((p.K % c.bk) != 0 || (p.Cin % c.bk) != 0) return false;
  if (p.K / c.bk < 4) return false;
  if (p.taps == 9 && (p.Wo % 16 != 0 || p.Ho % (c.bm / 16) != 0)) return false;
  const long long z = p.batch > 1 ? p.batch : 1;
  return (long long)((p.M + c.bm - 1) / c.bm) * ((p.N + c.bn - 1) / c.bn) * z >= (c.occ == 2 ? 256 : cfg().big_min_tiles);
}

bool gemm_big_eligible(const GemmP& p, int& bm, int& bn, int& bk) {
  if (!cfg().big_kernels) return false;
  if (p.splitk > 1 || (p.N % 8) != 0) return false;
  const bool f32o = p.out_mode == DFW_OUT_F32;      // fp32 residual stream: the two ping-pong configurations only
  if ((p.out_mode != DFW_OUT_T && !f32o) || p.act != DFW_ACT_NONE || (p.res_f32 && !f32o)) return false;
  if (f32o) {
    if (p.geglu || p.cs_n > 0 || (p.N % 128) != 0 || (p.ldc % 4) != 0) return false;
    const BigCfg c = (p.N % 256) == 0 ? BigCfg{256, 256, 32, 1} : BigCfg{512, 128, 32, 1};
    if (!big_cfg_ok(p, c)) return false;
    bm = c.bm; bn = c.bn; bk = c.bk;
    return true;
  }
  if (cfg().big_bm) {      // sweeps: a forced configuration where it fits
    const BigCfg c = {cfg().big_bm, cfg().big_bn, cfg().big_bk, 1};
    const bool known = (c.bm == 256 && c.bn == 256 && c.bk == 32) || (c.bm == 512 && c.bn == 128 && c.bk == 32) ||
                       (c.bm == 256 && c.bn == 128 && (c.bk == 32 || c.bk == 64));
    if (known && big_cfg_ok(p, c)) {
      bm = c.bm; bn = c.bn; bk = c.bk;
      return true;
    }
  }
  static const BigCfg wide[] = {{256, 256, 32, 1}, {256, 128, 64, 1}, {256, 128, 32, 1}};
  static const BigCfg narrow[] = {{512, 128, 32, 1}, {256, 128, 64, 1}, {256, 128, 32, 1}};
  // N % 64 == 0 shapes (the UNet's 320 / 960 columns) on the 128-wide tiles: measured neutral on the whole (44.01 vs 43.98 ms
  // per step: the half-empty last column tile costs what the bigger tile gains), so gemm.hip's cost-model tiles keep them --
  // except where the half-empty tile is a small share of the columns (<= 10 %: N = 960 of the 64x64-level fused QKV and its
  // data gradient -- 38.7 vs 51.4 us on 32768 x 960 x 320, scratch/sweep_big.py; N = 320 wastes 20 % and stays).
  const bool rag_ok = ((p.N + 127) / 128 * 128 - p.N) * 10 <= p.N;
  if ((p.N % 128) == 0 || ((p.N % 64) == 0 && p.N > 128 && !p.geglu && rag_ok)) {
    const BigCfg* list = (p.N % 256) == 0 ? wide : narrow;
    // Short-K linears whose 256 x 256 tile count quantises badly over the 256 CUs (2048 x 10240 x 1280 GEGLU: 320 tiles =
    // 1.25 rounds; 49152 x 512 x 512: 1.5 rounds) run 12-14 % faster on 256 x 128 x 64 (scratch/sweep_big.py); long-K
    // convs keep the wide tile whatever the round count (49152 x 512 x 4608: 230 vs 251 us).
    if (list == wide && p.taps == 1 && p.K <= 1536 && big_cfg_ok(p, wide[0]) && big_cfg_ok(p, wide[1])) {
      const long long z = p.batch > 1 ? p.batch : 1;
      const long long t0 = (long long)((p.M + 255) / 256) * (p.N / 256) * z, t1 = (long long)((p.M + 255) / 256) * (p.N / 128) * z;
      const double e0 = (double)t0 / (double)((t0 + 255) / 256 * 256), e1 = (double)t1 / (double)((t1 + 255) / 256 * 256);
      if (e0 < 0.8 && e1 > e0 + 0.1) {
        bm = wide[1].bm; bn = wide[1].bn; bk = wide[1].bk;
        return true;
      }
    }
    if (list == narrow && cfg().k8 >= 3 && big_cfg_ok(p, narrow[1]) && gemm8_eligible(p, 128)) {
      bm = narrow[1].bm; bn = narrow[1].bn; bk = narrow[1].bk;
      return true;
    }
    for (int i = 0; i < 3; ++i)
      if (big_cfg_ok(p, list[i])) {
        bm = list[i].bm; bn = list[i].bn; bk = list[i].bk;
        return true;
      }
  }
  return false;
}

int gemm_big_gn_chunks(const GemmP& p) {
  int bm = 0, bn = 0, bk = 0;
  if (p.gn_groups <= 0 || p.taps != 9 || p.geglu || (p.out_mode != DFW_OUT_T && p.out_mode != DFW_OUT_F32) ||
      !gemm_big_eligible(p, bm, bn, bk)) return 0;
  if ((size_t)(bm + bn) * bk * 2 < 8 * 4096) return 0;            // staged epilogue needs a 32 KiB slot
  if (p.N % p.gn_groups) return 0;
  const int cpg = p.N / p.gn_groups;
  if (cpg < 4 || cpg > 64 || (cpg & (cpg - 1))) return 0;         // groups must tile the 64-channel wave tiles
  const int wgm = 8 / (bn / 64);
  return (p.Wo / 16) * (p.Ho / (bm / 16)) * wgm;
}

int launch_gemm_big(const GemmP& p, hipStream_t st) {
  int bm = 0, bn = 0, bk = 0;
  if (!gemm_big_eligible(p, bm, bn, bk)) return DFW_ESHAPE;
  const bool bf = p.dtype_bf16 != 0;
  // 256 x 256 and 512 x 128: ping-pong schedule with v_mfma_f32_16x16x32 (A/B history in DESIGN.md section 3: +2 % each
  // against the in-phase schedule and against 32x32x16 in the same schedule; those instantiations are gone)
  if (p.out_mode == DFW_OUT_F32) {
    if (bm == 256) return bf ? launch_big<__bf16, 256, 256, 32, 4, 1, true, true, true>(p, st) : launch_big<_Float16, 256, 256, 32, 4, 1, true, true, true>(p, st);
    return bf ? launch_big<__bf16, 512, 128, 32, 4, 1, true, true, true>(p, st) : launch_big<_Float16, 512, 128, 32, 4, 1, true, true, true>(p, st);
  }
  if (bm == 256 && bn == 256) {
    if (gemm8_eligible(p, 256)) return launch_gemm8(p, st, 256);      // round 4: the 64-deep K-tile kernel (gemm8.hip)
    return bf ? launch_big<__bf16, 256, 256, 32, 4, 1, true, true>(p, st) : launch_big<_Float16, 256, 256, 32, 4, 1, true, true>(p, st);
  }
  if (bm == 512)
    return bf ? launch_big<__bf16, 512, 128, 32, 4, 1, true, true>(p, st) : launch_big<_Float16, 512, 128, 32, 4, 1, true, true>(p, st);
  if (bm == 256 && bn == 128 && gemm8_eligible(p, 128)) return launch_gemm8(p, st, 128);   // dfw_config.k8 >= 2
  if (bk == 64) return bf ? launch_big<__bf16, 256, 128, 64, 3, 1>(p, st) : launch_big<_Float16, 256, 128, 64, 3, 1>(p, st);
  return bf ? launch_big<__bf16, 256, 128, 32, 4, 1>(p, st) : launch_big<_Float16, 256, 128, 32, 4, 1>(p, st);
}

}  // namespace dfw
