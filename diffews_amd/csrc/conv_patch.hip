// conv3x3 (stride 1, pad 1) as an implicit GEMM whose A operand is an LDS-resident input PATCH, on the schedule of
// gemm_big.hip's fastest path (8 waves, ping-pong wave groups, v_mfma_f32_16x16x32, staged whole-line epilogue with
// fused GroupNorm partial sums, persistent XCD-contiguous tile walk, counted-vmcnt LDS-DMA ring).
//
// gemm_big.hip stages a fresh [BM pixels][32 ch] A tile for every tap: 9 x BM x 64 B per 32-channel chunk, which for
// the N = 128 VAE layers (512 x 128 tiles: 40 KB through the CU's LDS-DMA path per 4.2 MFLOP K-step) is what the
// tile time is made of.  conv_halo.hip showed the patch idea but on a 256-row tile with a scattered-store epilogue.
// Here a workgroup owns a (BM/16) x 16 output-pixel tile -- 512 x 128 or 256 x 256 outputs -- and DMA's, once per
// 32-channel chunk, the (BM/16 + 2) x 18 input patch (39 KB / 20 KB, out-of-image pixels zero-filled by the buffer
// bounds check); the nine taps read their MFMA fragments from it at shifted pixel addresses.  Only the W stage
// ([BN][32 k] = 8 / 16 KB) streams through the 4-slot ring.  Per chunk the CU ingests 39 + 72 KB instead of 360 KB.
//
// K walk: chunk-major, tap-minor (step s of a tile -> chunk s / 9, tap s % 9), W columns tap * Cin + chunk * 32.
// LDS: W ring 4 x BN x 64 B | patch 0 | patch 1 | 32 KiB epilogue staging (4 KiB per wave).
// Patch pixel rows are 64 B with the 16-byte chunk swizzle slot = chunk ^ ((pixel >> 1) & 3): any 8 consecutive patch
// pixels hit 8 distinct 16-byte slots of a 128-byte bank row (measured conflict-free for ds_read_b128, scratch/proto/lds_swz.hip;
// the ((pixel >> 2) & 3) form used before cost a 2-way conflict on every fragment read).  [old text:] any 16 consecutive patch
// pixels hit 16 distinct 16-byte slots of a 256-byte bank row, so the 16x16x32 A-fragment reads are conflict-free at every
// tap offset.  The DMA writes LDS linearly; the swizzle sits on the source address.
//
// DMA accounting (per wave, in issue order): at step k the wave issues W(k+3) [SW instructions] and, on a chunk's first tap,
// the NEXT chunk's patch [PPW instructions, every wave the same number: the tail instructions are fully out of range and
// write zeros].  Before the barrier that publishes W(k+1) the wave waits until only the instructions younger than W(k+1)
// are outstanding: 2 SW, plus PPW while the patch issued on tap 0 is still younger than W(k+1) (taps 0, 1, 2).  From tap 3
// on, the in-order counter retires the patch together with W(k+1), six taps before its first reader.
#include "gemm_common.h"
#include <type_traits>
#include <stdlib.h>

namespace dfw {

// F32O: fp32 NHWC output + fp32 (or storage-dtype) residual, stored straight from the accumulators -- the fp32 residual
// stream (see gemm_big.hip); its own instantiation, the 16-bit path's staged epilogue is unchanged.
template <typename T, int BM, int BN, bool F32O = false>
__global__ __launch_bounds__(512, 2) void conv_patch_kernel(const GemmP p) {
  constexpr int S = 4, RB = 64;
  constexpr int WGN = BN / 64, WGM = 8 / WGN, WTM = BM / WGM;      // wave tile WTM x 64 (128 x 64 in both configurations)
  constexpr int MB6 = WTM / 16, NB6 = 4;
  constexpr int PH = BM / 16, PW = 18, PPIX = (PH + 2) * PW;        // patch: (PH + 2) x 18 pixels
  constexpr int PPW = ((PPIX + 15) / 16 + 7) / 8;                   // patch DMA instructions per wave (16 pixels each)
  constexpr int PATCH = PPW * 8 * 1024;
  constexpr int WSTAGE = BN * RB;
  constexpr int SW = BN / 16 / 8;                                   // W DMA instructions per wave per stage
  static_assert(WTM == 128 && MB6 == 8, "wave tile is 8 pixel rows x 16 pixels x 64 channels");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const pbase = smem + S * WSTAGE;
  char* const stgbase = pbase + 2 * PATCH;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int l15 = lane & 15, l4 = lane >> 4;
  const uint32_t lds0 = lds_addr(smem);

  const int ntiles = p.ntm * p.ntn;
  const int nxb = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int Q = (ntiles + 7) >> 3;
  const int t_end = min(ntiles, (xcd + 1) * Q);
  const int tile0 = xcd * Q + (blockIdx.x >> 3);
  if (tile0 >= t_end) return;
  const int my_tiles = (t_end - tile0 + nxb - 1) / nxb;
  const int cpt = p.Cin >> 5;                       // 32-channel chunks
  const int nks = cpt * 9;                          // K-steps per tile
  const int total = my_tiles * nks;                 // <= (tiles per workgroup) x 9 x Cin / 32: far below 2^31

  char* Cb = p.C;
  const u32x4 ra = make_srd(p.A, p.a_bytes);
  // blocked weight copy ([tap][Cin/32][N][32], dfw_gemm_args.W_blocked): a W stage is one contiguous N x 64 B block
  const bool wblk = p.Wblk != nullptr;
  const u32x4 rw = make_srd(wblk ? p.Wblk : p.W, p.w_bytes);

  auto tile_coords = [&](int t) -> TileC {
    TileC c;
    const int tn = t % p.ntn, tm = t / p.ntn;
    c.m0 = tm * BM;
    c.n0 = tn * BN;
    c.img = tm / p.tpi;
    const int t2 = tm - c.img * p.tpi, tyi = t2 / p.tpr, txi = t2 - tyi * p.tpr;
    c.oy0 = tyi * PH;
    c.ox0 = txi << 4;
    return c;
  };
  auto row_to_m = [&](const TileC& c, int r) -> int { return (c.img * p.Ho + c.oy0 + (r >> 4)) * p.Wo + c.ox0 + (r & 15); };

  // ---- loaders.  lane -> (pixel or W row) lane >> 2 of the instruction's 16, LDS slot lane & 3, source chunk slot ^ swizzle
  const int kc = (lane & 3) ^ ((lane >> 3) & 3);
  // patch source offsets are recomputed at every issue (once per chunk: ~10 VALU per instruction) rather than kept in
  // PPW registers through the K loop
  int pl_img = 0, pl_oy0 = 0, pl_ox0 = 0;
  auto patch_off = [&](int i) __attribute__((always_inline)) -> uint32_t {
    const int q = 16 * (wave + 8 * i) + (lane >> 2);
    const int qy = q / PW, qx = q - qy * PW;
    const int iy = pl_oy0 - 1 + qy, ix = pl_ox0 - 1 + qx;
    const bool ok = q < PPIX && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
    return ok ? (uint32_t)((((size_t)pl_img * p.Hi + iy) * p.Wi + ix) * p.lda + kc * 8) * (uint32_t)sizeof(T) : kOOB;
  };
  auto setup_patch = [&](const TileC& c) {
    pl_img = c.img; pl_oy0 = c.oy0; pl_ox0 = c.ox0;
  };
  uint32_t w_off[SW];
  auto setup_w = [&](const TileC& c) {
#pragma unroll
    for (int i = 0; i < SW; ++i) {
      const int n = c.n0 + (i * 8 + wave) * 16 + (lane >> 2);
      w_off[i] = n < p.N ? (uint32_t)(((size_t)n * (wblk ? 32 : p.K) + kc * 8) * sizeof(T)) : kOOB;
    }
  };
  // flattened streams: W step index wl (0 .. total), patch index pl (0 .. my_tiles * cpt)
  int wl = 0;
  int wl_tap = 0, wl_c = 0, wl_tile = tile0, wl_slot = 0;
  auto issue_w = [&]() {     // caller guarantees wl < total
    const uint32_t dst = lds0 + (uint32_t)wl_slot * WSTAGE + (uint32_t)wave * 1024u;
    wl_slot = wl_slot + 1 == S ? 0 : wl_slot + 1;
    const uint32_t koff = wblk ? (uint32_t)(wl_tap * cpt + wl_c) * (uint32_t)p.N * 64u
                               : (uint32_t)(wl_tap * p.Cin + wl_c * 32) * (uint32_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < SW; ++i) dma16(rw, w_off[i] == kOOB ? kOOB : w_off[i] + koff, dst + i * 8192);
    ++wl;
    if (++wl_tap == 9) {
      wl_tap = 0;
      if (++wl_c == cpt) {
        wl_c = 0;
        wl_tile += nxb;
        if (p.ntn > 1 && wl < total) setup_w(tile_coords(wl_tile));
      }
    }
  };
  int pl = 0;
  const int pl_total = my_tiles * cpt;
  int pl_c = 0, pl_tile = tile0;
  auto issue_patch = [&]() {  // caller guarantees pl < pl_total
    if (pl_c == cpt) {
      pl_c = 0;
      pl_tile += nxb;
      setup_patch(tile_coords(pl_tile));
    }
    const uint32_t dst = lds0 + S * WSTAGE + (uint32_t)(pl & 1) * PATCH;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const uint32_t o = patch_off(i);
      dma16(ra, o == kOOB ? kOOB : o + (uint32_t)pl_c * 64u, dst + (uint32_t)(wave + 8 * i) * 1024u);
    }
    ++pl_c;
    ++pl;
  };

  // ---- fragments
  f32x4 acc6[MB6][NB6];
  typename Tr<T>::v8 fa6[MB6], fw6[NB6];
  const uint32_t rw6 = (uint32_t)(wn * 64 + l15) * RB + (uint32_t)((l4 ^ ((l15 >> 1) & 3)) << 4);
  const int q00 = (wm * (WTM / 16)) * PW + l15;          // patch pixel of this lane for pixel-row block 0, tap (0, 0)
  // A-fragment addresses without per-tap arithmetic (it was 45 vector instructions per tap in the fragment-read half of the
  // ping-pong, i.e. in the half that has to finish inside the partner's 32 MFMAs).  Pixel of (tap (ky, kx), row block i):
  // q = (q00 + kx) + 18 s with s = ky + i; 18 s is even, so q >> 1 = ((q00 + kx) >> 1) + 9 s and the swizzle term
  // (q >> 1) & 3 = (h_kx + s) & 3 with h_kx = ((q00 + kx) >> 1) & 3: per kx four lane-constant bases (one per s & 3), and
  // s * 18 * 64 bytes is an immediate of the ds_read once the nine taps are unrolled.
  // The twelve bases are rebuilt once per 32-channel chunk (36 vector instructions per nine taps) behind an opaque copy of the
  // lane id: kept as loop constants they would cost twelve more registers than this kernel has.
  uint32_t ab[3][4];
  auto set_patch_buf = [&](int pbuf) __attribute__((always_inline)) {
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    const int l15o = lane_o & 15, l4o = lane_o >> 4;
    const uint32_t o = (uint32_t)(S * WSTAGE + pbuf * PATCH);
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int qk = (wm * (WTM / 16)) * PW + l15o + kx, h = (qk >> 1) & 3;
#pragma unroll
      for (int r = 0; r < 4; ++r) ab[kx][r] = (uint32_t)(qk * RB + ((l4o ^ ((h + r) & 3)) << 4)) + o;
    }
  };
  auto reads = [&](int slot, int tap) __attribute__((always_inline)) {    // tap: compile-time (the tap loop is unrolled)
    const char* bw = smem + slot * WSTAGE + rw6;
#pragma unroll
    for (int j = 0; j < NB6; ++j) fw6[j] = as_v8<T>(*(const i32x4*)(bw + j * 16 * RB));
    const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
    for (int i = 0; i < MB6; ++i) {
      const int sidx = ky + i;
      fa6[i] = as_v8<T>(*(const i32x4*)(smem + ab[kx][sidx & 3] + sidx * PW * RB));
    }
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
  // staged epilogue: the image of gemm_big.hip's 16x16x32 path (row r of a 32-row round: 16-byte chunk c at
  // ((c ^ (r & 7)) << 4) of a 128-byte line), statistics of the STORED values, whole-line stores
  auto epilogue6 = [&](const TileC& c, char* stg, int tile_id) {
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l15 = lane_e & 15, l4 = lane_e >> 4, lane = lane_e;
    if constexpr (F32O) {
      GnRegSums gsum;
      gsum.clear();
#pragma unroll
      for (int i = 0; i < MB6; ++i) {
        const int m = row_to_m(c, wm * WTM + i * 16 + l15);
        f32x4 add[NB6];
#pragma unroll
        for (int j = 0; j < NB6; ++j) {            // every load of the row block before its first store
          const int n = c.n0 + wn * 64 + j * 16 + 4 * l4;
          f32x4 bb = {0.f, 0.f, 0.f, 0.f};
          if (p.bias) bb = *(const f32x4*)(p.bias + n);
          if (p.rowbias) {
            const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)c.img * p.ldrb + n);
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
      if (p.gn_partial) {       // statistics of the fp32 values just stored (the next GroupNorm skips its own pass)
        const int tm = tile_id / p.ntn;
        const int chunk = (tm - c.img * p.tpi) * WGM + wm;
        gsum.store(p, p.gn_partial + ((size_t)c.img * p.gn_chunks + chunk) * p.gn_groups * 2, c.n0 + wn * 64, lane);
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
          const int n = c.n0 + wn * 64 + j * 16 + 4 * l4;
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
          const int row = h * 16 + l15, quad = j * 4 + l4;
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
  // wait until only the instructions younger than W(k+1) are outstanding: `wy` W stages (0..2) and `py` patches (0/1)
  auto wait_for = [&](int wy, int py) __attribute__((always_inline)) {
    if (py) {
      if (wy >= 2) wait_vm<2 * SW + PPW>();
      else if (wy == 1) wait_vm<SW + PPW>();
      else wait_vm<PPW>();
    } else {
      if (wy >= 2) wait_vm<2 * SW>();
      else if (wy == 1) wait_vm<SW>();
      else wait_vm<0>();
    }
  };

  // ---- prologue: patch 0, W(0..2); everything of step 0 landed before the first barrier
  TileC ct = tile_coords(tile0);
  setup_patch(ct);
  setup_w(ct);
  issue_patch();
#pragma unroll
  for (int i = 0; i < S - 1; ++i)
    if (wl < total) issue_w();
  {
    const int wy = wl - 1;                            // W stages younger than W(0)
    wait_for(wy >= 2 ? 2 : wy, 0);                // (the patch is OLDER than every W: retired with W(0))
  }
  bar();
  // One tile's K walk, as two separate instruction streams (wave groups half a K-step apart); the epilogue stays
  // common code below so that it is inlined once and the accumulators never leave the registers.
  int rs = 0, patch_age = 3;                           // ring slot of step k; taps since the last patch issue
  int k = 0, pc = 0;                                   // global step; patch index of the chunk being computed
  auto ktile = [&](auto G1) __attribute__((always_inline)) {
    constexpr bool g1 = decltype(G1)::value;
#pragma unroll 1
    for (int cc = 0; cc < cpt; ++cc) {
      set_patch_buf(pc & 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // Wave group 0 issues its DMA in its fragment-read half (before the barrier that starts its MFMAs) instead of between
        // that barrier and the MFMAs: same position relative to its waits, half a K-step earlier in time (the slot it fills
        // was last read before the previous tap's second barrier), and its 32 MFMAs start right behind the barrier
        // (-2..6 % per launch; issuing before the reads instead of after them is 1 % worse).
        if constexpr (!g1) {
          reads(rs, tap);
          if (wl < total) issue_w();
          if (tap == 0 && pl < pl_total) { issue_patch(); patch_age = 0; }
        }
        bar();
        if constexpr (g1) {
          if (wl < total) issue_w();
          if (tap == 0 && pl < pl_total) { issue_patch(); patch_age = 0; }
          reads(rs, tap);
        } else {
          mfmas();
        }
        {
          const int wy = wl - (k + 2);                 // W stages younger than W(k+1)
          wait_for(wy >= 2 ? 2 : (wy < 0 ? 0 : wy), patch_age <= 2 ? 1 : 0);
        }
        if (patch_age < 3) ++patch_age;
        // group 1's fragment reads of this stage have RETURNED before the barrier behind which group 0 may refill the slot
        // (they are needed right after it anyway; the DMA could not land that fast, but the order is now by construction)
        if constexpr (g1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        if constexpr (g1) mfmas();
        rs = rs + 1 == S ? 0 : rs + 1;
        ++k;
      }
      ++pc;
    }
  };
  for (int ti = 0; ti < my_tiles; ++ti) {
    zero6();
    if (wave >= 4) ktile(std::true_type{});
    else ktile(std::false_type{});
    epilogue6(ct, stgbase + wave * 4096, tile0 + ti * nxb);   // per-wave staging: no barrier needed around it
    if (ti + 1 < my_tiles) ct = tile_coords(tile0 + (ti + 1) * nxb);
  }
}

template <typename T, int BM, int BN, bool F32O = false>
static int launch_patch(const GemmP& p, hipStream_t st, int gn_chunks) {
  GemmP q = p;
  q.ntm = p.M / BM;
  q.ntn = p.N / BN;
  q.tw = 16; q.tw_log2 = 4;
  q.tpr = p.Wo / 16;
  q.tpi = q.tpr * (p.Ho / (BM / 16));
  q.gn_chunks = p.gn_partial ? gn_chunks : 0;
  if (q.gn_chunks == 0) q.gn_partial = nullptr;
  constexpr int PPIX = (BM / 16 + 2) * 18, PPW = ((PPIX + 15) / 16 + 7) / 8;
  const size_t lds = 4 * (size_t)BN * 64 + 2 * (size_t)PPW * 8 * 1024 + 32 * 1024;
  int nwg = q.ntm * q.ntn;
  if (nwg > 256) nwg = 256;
  nwg = (nwg + 7) & ~7;
  auto kfn = conv_patch_kernel<T, BM, BN, F32O>;
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), lds, st, q);
  DFW_CHECK_LAUNCH();
  return 0;
}

// Shapes: stride-1 / pad-1 conv3x3 with whole (BM/16) x 16 pixel tiles, N a multiple of the tile width, storage-dtype
// NHWC output, enough tiles to occupy the chip.  N % 256 == 0 -> 256 x 256, else N % 128 == 0 -> 512 x 128.
bool conv_patch_eligible(const GemmP& p, int& bm, int& bn) {
  const int mode = cfg().conv_patch;
  if (mode == 0) return false;
  if (p.taps != 9 || p.stride != 1 || p.pad != 1 || p.ups || p.splitk > 1 || p.batch > 1) return false;
  if (p.Hi != p.Ho || p.Wi != p.Wo || (p.Wo % 16) != 0 || (p.Cin % 64) != 0) return false;
  const bool f32o = p.out_mode == DFW_OUT_F32;        // fp32 residual stream
  if ((p.out_mode != DFW_OUT_T && !f32o) || p.act != DFW_ACT_NONE || p.geglu || p.cs_n > 0 || (p.res_f32 && !f32o)) return false;
  if (f32o && (p.ldc % 4) != 0) return false;
  if (p.rows_per_img != p.Ho * p.Wo) return false;
  // dfw_config.conv_patch: 1 = the 512 x 128 tile for the N = 128 layers only (+8..11 % over gemm_big there), 2 (default since the
  // fragment addresses became per-chunk bases + immediates) also the N % 256 == 0 layers on the 256 x 256 tile: +7..12 % per
  // kernel against gemm_big (vae256 0.997 -> 0.914 ms, vae128 0.791 -> 0.739, 128 -> 256 @256^2 0.469 -> 0.417) and 42.0 -> 41.3 ms
  // on the inference step; with the 45-instruction address block per tap it had been 1.5..2 % ahead per kernel and neutral on
  // the step.  0 disables the kernel.
  const bool only128 = mode != 2;
  if ((p.N % 256) == 0 && !only128) { bm = 256; bn = 256; }
  else if ((p.N % 128) == 0 && (p.N % 256) != 0) { bm = 512; bn = 128; }
  else return false;
  if ((p.Ho % (bm / 16)) != 0 || p.M % bm != 0) return false;
  return (long long)(p.M / bm) * (p.N / bn) >= 192;
}

int conv_patch_gn_chunks(const GemmP& p) {
  int bm = 0, bn = 0;
  if (p.gn_groups <= 0 || (p.out_mode != DFW_OUT_T && p.out_mode != DFW_OUT_F32) || !conv_patch_eligible(p, bm, bn) || p.N % p.gn_groups) return 0;
  const int cpg = p.N / p.gn_groups;
  if (cpg < 4 || cpg > 64 || (cpg & (cpg - 1))) return 0;
  return (p.Wo / 16) * (p.Ho / (bm / 16)) * (8 / (bn / 64));
}

int launch_conv_patch(const GemmP& p, hipStream_t st) {
  int bm = 0, bn = 0;
  if (!conv_patch_eligible(p, bm, bn)) return DFW_ESHAPE;
  const int chunks = conv_patch_gn_chunks(p);
  const bool bf = p.dtype_bf16 != 0;
  if (p.out_mode == DFW_OUT_F32) {
    if (bm == 512) return bf ? launch_patch<__bf16, 512, 128, true>(p, st, chunks) : launch_patch<_Float16, 512, 128, true>(p, st, chunks);
    return bf ? launch_patch<__bf16, 256, 256, true>(p, st, chunks) : launch_patch<_Float16, 256, 256, true>(p, st, chunks);
  }
  if (bm == 512) return bf ? launch_patch<__bf16, 512, 128>(p, st, chunks) : launch_patch<_Float16, 512, 128>(p, st, chunks);
  return bf ? launch_patch<__bf16, 256, 256>(p, st, chunks) : launch_patch<_Float16, 256, 256>(p, st, chunks);
}

}  // namespace dfw
