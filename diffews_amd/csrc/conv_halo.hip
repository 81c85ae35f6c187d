// 3x3 / stride-1 / pad-1 convolution as an implicit GEMM whose A operand is served from an
// LDS-resident HALO PATCH instead of nine shifted global reads.
//
// gemm_big.hip stages, for every tap, a fresh [256 pixels][32 ch] A tile: 9 x 16 KiB per 32-channel
// chunk through the CU's ~30 GB/s LDS-DMA path, which (measured: replacing those fetches by
// zero-fill lifts the kernel from 0.70-1.08 to 1.0-1.45 PFLOP/s) is what bounds the big VAE convs.
// Here a workgroup owns a 16x16 output-pixel tile and DMA's, once per 32-channel chunk, the
// 18x18 input patch around it (20.25 KiB, out-of-image pixels zero-filled by the buffer bounds
// check); the nine taps then read their MFMA fragments from that patch at shifted pixel
// addresses.  A-side DMA traffic drops 7.1x; the W tiles ([BN][32] per tap) keep streaming through
// a 4-stage ring exactly as in gemm_big.hip (counted vmcnt, one barrier per K-step).
//
// K order is chunk-major, tap-minor: step s -> chunk s/9, tap s%9, W columns (tap*Cin + chunk*32).
// LDS: W ring 6 x BN x 64 B (two K-steps per barrier, four stages in flight), patch 2 x 21 KiB (double-buffered: the next chunk's -- or next tile's --
// patch lands while the current chunk's nine taps run).  Pixel rows are 64 B with the 16-byte chunk
// swizzle slot = chunk ^ ((pixel>>2)&3): any 16 consecutive patch pixels hit 16 distinct 16-byte
// slots, so the fragment reads stay conflict-free at every tap offset.
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace dfw {

int conv_halo_gn_chunks(const GemmP& p);

//
// GNIN: the conv's input is GroupNorm(+SiLU) of the tensor in HBM.  The per-(image, channel)
// affine (scale, shift) table written by gn_finalize_kernel is DMA'd to LDS with each tile's first
// patch, and every landed patch is normalised IN PLACE once (two barrier intervals after its DMA
// was issued, two before its first tap reads it) -- 1/9 of the VALU work a per-tap transform would
// cost, and the normalised activation never exists in HBM.  Out-of-image halo pixels stay zero
// (the conv pads the NORMALISED tensor).  The arithmetic is gn_apply_kernel's, so fused and
// unfused results are bit-identical.
template <typename T, int BN, bool GNIN>
__global__ __launch_bounds__(512, 1) void conv_halo_kernel(const GemmP p) {
  constexpr int BM = 256, S = 6, PW = 18;
  constexpr int WGN = BN / 64, WGM = 8 / WGN, WTM = BM / WGM;
  constexpr int MB = WTM / 32, NB = 2;
  constexpr int WSTAGE = BN * 64;             // bytes of one W stage
  constexpr int PATCH = 21 * 1024;            // 336 pixel rows x 64 B (324 used)
  constexpr int SW = BN / 128;                // W DMA wave-instructions per stage per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [W ring][patch 0][patch 1][coef]
  char* const pbase = smem + S * WSTAGE;
  char* const coefbase = pbase + 2 * PATCH;   // GNIN: [Cin][2] floats of the tile's image

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int lr = lane & 31, lh = lane >> 5;
  const uint32_t lds0 = lds_addr(smem);

  const int ntiles = p.ntm * p.ntn;
  const int nxb = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int Q = (ntiles + 7) >> 3;
  const int t_end = min(ntiles, (xcd + 1) * Q);
  const int tile0 = xcd * Q + (blockIdx.x >> 3);
  if (tile0 >= t_end) return;
  const int my_tiles = (t_end - tile0 + nxb - 1) / nxb;
  const int cpt = p.Cin >> 5;                 // 32-channel chunks
  const int nsteps = cpt * 9;                 // K-steps per tile
  const long long total = (long long)my_tiles * nsteps;

  char* Cb = p.C;
  const u32x4 ra = make_srd(p.A, p.a_bytes);
  const u32x4 rw = make_srd(p.W, p.w_bytes);
  const u32x4 rc = make_srd(GNIN ? (const void*)p.gn_coef : (const void*)p.W,
                            (uint32_t)(p.M / (p.Ho * p.Wo)) * (uint32_t)p.Cin * 8u);

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

  // ---- patch loader: wave-instruction j (0..20) covers patch pixels 16j .. 16j+15; wave w issues
  // j = w, w+8, w+16 (< 21).  lane -> pixel 16j + (lane>>2), slot lane&3, source chunk = slot ^ ((pixel>>1)&3)
  const int kc = (lane & 3) ^ ((lane >> 3) & 3);
  uint32_t pp_off[3];
  int pl_tile = tile0, pl_c = 0;              // next (tile, chunk) whose patch will be issued
  long long pl_issued = 0;                    // patches issued so far (parity = LDS buffer)
  const long long pl_total = (long long)my_tiles * cpt;
  auto setup_patch = [&](const TileC& c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int q = 16 * (wave + 8 * i) + (lane >> 2);
      const int qy = q / PW, qx = q - qy * PW;
      const int iy = c.oy0 - 1 + qy, ix = c.ox0 - 1 + qx;
      const bool ok = q < PW * PW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
      pp_off[i] = ok ? (uint32_t)((((size_t)c.img * p.Hi + iy) * p.Wi + ix) * p.lda + kc * 8) * (uint32_t)sizeof(T) : kOOB;
    }
  };
  TileC pl_ct, tf_ct;                         // loader's tile; tile of the patch awaiting its transform
  int tf_cnt = 0, tf_c = 0, tf_buf = 0;
  auto issue_patch = [&]() {
    if (pl_issued >= pl_total) return;
    if (pl_c == cpt) {
      pl_c = 0;
      pl_tile += nxb;
      pl_ct = tile_coords(pl_tile);
      setup_patch(pl_ct);
    }
    const uint32_t dst = lds0 + S * WSTAGE + (uint32_t)(pl_issued & 1) * PATCH;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int j = wave + 8 * i;
      if (j < 21) dma16(ra, pp_off[i] == kOOB ? kOOB : pp_off[i] + (uint32_t)pl_c * 64u, dst + (uint32_t)j * 1024u);
    }
    if (GNIN) {
      if (pl_c == 0 && wave * 128 < p.Cin) {  // (scale, shift) table of this tile's image, 1 KiB per wave
        const uint32_t off = (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
        dma16(rc, off < (uint32_t)p.Cin * 8u ? (uint32_t)pl_ct.img * (uint32_t)p.Cin * 8u + off : kOOB,
              lds0 + S * WSTAGE + 2 * PATCH + (uint32_t)wave * 1024u);
      }
      tf_cnt = 2; tf_c = pl_c; tf_buf = (int)(pl_issued & 1); tf_ct = pl_ct;
    }
    ++pl_c;
    ++pl_issued;
  };
  // in-place GroupNorm(+SiLU) of a landed patch: thread -> 16-byte slots tid, tid+512, tid+1024
  int tq[3], tqy[3], tqx[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    tq[u] = (tid + u * 512) >> 2;
    tqy[u] = tq[u] / PW;
    tqx[u] = tq[u] - tqy[u] * PW;
  }
  auto transform = [&]() {
    char* const pb = pbase + tf_buf * PATCH;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int q = tq[u], sl = tid & 3;
      const int iy = tf_ct.oy0 - 1 + tqy[u], ix = tf_ct.ox0 - 1 + tqx[u];
      if (q < PW * PW && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi) {
        const int ch = tf_c * 32 + ((sl ^ ((q >> 1) & 3)) << 3);
        i32x4* const ptr = (i32x4*)(pb + q * 64 + sl * 16);
        const f32x4* const cf = (const f32x4*)(coefbase + ch * 8);
        float f[8];
        unpack8<T>(*ptr, f);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const f32x4 c2 = cf[h];
          f[2 * h] = f[2 * h] * c2[0] + c2[1];
          f[2 * h + 1] = f[2 * h + 1] * c2[2] + c2[3];
        }
        if (p.gn_silu) {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = silu_f(f[i]);
        }
        *ptr = pack8<T>(f);
      }
    }
  };

  // ---- W loader: 3 steps ahead; stage = BN rows x 64 B; wave-instruction i covers rows (i*8+wave)*16..+16
  uint32_t w_off[SW];
  int wl_tile = tile0, wl_c = 0, wl_t = 0, wl_slot = 0;
  long long wl_issued = 0;
  auto setup_w = [&](const TileC& c) {
#pragma unroll
    for (int i = 0; i < SW; ++i) {
      const int n = c.n0 + (i * 8 + wave) * 16 + (lane >> 2);
      w_off[i] = n < p.N ? (uint32_t)(((size_t)n * p.K + kc * 8) * sizeof(T)) : kOOB;
    }
  };
  auto issue_w = [&]() {
    if (wl_issued >= total) return;
    if (wl_c == cpt) {
      wl_c = 0;
      wl_tile += nxb;
      if (p.ntn > 1) setup_w(tile_coords(wl_tile));
    }
    const uint32_t dst = lds0 + (uint32_t)wl_slot * WSTAGE + (uint32_t)wave * 1024u;
    wl_slot = wl_slot + 1 == S ? 0 : wl_slot + 1;
    const uint32_t koff = (uint32_t)(wl_t * p.Cin + wl_c * 32) * (uint32_t)sizeof(T);
#pragma unroll
    for (int i = 0; i < SW; ++i) dma16(rw, w_off[i] == kOOB ? kOOB : w_off[i] + koff, dst + i * 8192);
    if (++wl_t == 9) { wl_t = 0; ++wl_c; }
    ++wl_issued;
  };

  // ---- fragment addresses
  int q0[MB];                                  // patch pixel of this lane's row for tap (0,0)
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int r = wm * WTM + i * 32 + lr;
    q0[i] = (r >> 4) * PW + (r & 15);
  }
  uint32_t lds_rw[NB];
  const int sw4 = (lr >> 1) & 3;
#pragma unroll
  for (int j = 0; j < NB; ++j) lds_rw[j] = (uint32_t)(wn * 64 + j * 32 + lr) * 64u + (uint32_t)((lh ^ sw4) << 4);

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
    // fused GroupNorm statistics of the output: per-lane (sum, sum of squares) of the STORED values
    // for each 4-channel quad (j, g), folded over the wave's rows below
    float gs[NB][4], gq[NB][4];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) { gs[j][g] = 0.f; gq[j][g] = 0.f; }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int r = wm * WTM + i * 32 + lr;
      const int m = (c.img * p.Ho + c.oy0 + (r >> 4)) * p.Wo + c.ox0 + (r & 15);
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = c.n0 + wn * 64 + j * 32 + 8 * g + 4 * lh;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          if (p.bias) {
            const f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
          }
          if (p.rowbias) {
            const f32x4 b = *(const f32x4*)(p.rowbias + (size_t)c.img * p.ldrb + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += b[e];
          }
          if (p.residual) {
            float rr[4];
            unpack4<T>(*(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T)), rr);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += rr[e];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
          const i32x2 pk = pack4<T>(v);
          *(i32x2*)(Cb + ((size_t)m * p.ldc + n) * sizeof(T)) = pk;
          if (p.gn_partial) {
            float r[4];
            unpack4<T>(pk, r);
            gs[j][g] += (r[0] + r[1]) + (r[2] + r[3]);
            gq[j][g] += (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
          }
        }
    }
    if (p.gn_partial) {
      // rows: butterfly over the 32 lanes of a half; channels: quads -> groups of cpg = 4/8/16/32
      const int cpg = p.N / p.gn_groups;
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int o = 1; o < 32; o <<= 1) {
            gs[j][g] += __shfl_xor(gs[j][g], o, 64);
            gq[j][g] += __shfl_xor(gq[j][g], o, 64);
          }
          if (cpg >= 8) {
            gs[j][g] += __shfl_xor(gs[j][g], 32, 64);
            gq[j][g] += __shfl_xor(gq[j][g], 32, 64);
          }
        }
      const int t2 = (c.oy0 >> 4) * p.tpr + (c.ox0 >> 4);
      float* const out = p.gn_partial + ((size_t)c.img * p.gn_chunks + (size_t)t2 * WGM + wm) * p.gn_groups * 2;
      const int nw = c.n0 + wn * 64;
      if (cpg == 4) {
        if (lr == 0) {
#pragma unroll
          for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *(float2*)(out + ((nw + j * 32 + 8 * g + 4 * lh) >> 2) * 2) = make_float2(gs[j][g], gq[j][g]);
        }
      } else if (lane == 0) {
        const int u = cpg >> 3;                 // 8-channel units per group: 1, 2 or 4
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            if (g % u) continue;
            float a = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (k >= g && k < g + u) { a += gs[j][k]; a2 += gq[j][k]; }
            *(float2*)(out + ((nw + j * 32 + 8 * g) / cpg) * 2) = make_float2(a, a2);
          }
      }
    }
  };

  // ---- pipeline over the flattened (tile, chunk, tap) stream, TWO K-steps per barrier: the
  // barrier + counted wait + DMA issue + first-LDS-read latency cost ~0.5 us per interval, which is
  // as long as 16 MFMAs per wave; 32 MFMAs per interval halve that share.  nsteps is even (Cin is a
  // multiple of 64), so a pair never straddles a tile; it may straddle a chunk (two patches live).
  auto compute_step = [&](const char* wbuf, const char* pbuf, int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int tapoff = ky * PW + kx;
    uint32_t pa[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int q = q0[i] + tapoff;
      pa[i] = (uint32_t)q * 64u + (uint32_t)((lh ^ ((q >> 1) & 3)) << 4);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      typename Tr<T>::v8 fa[MB], fw[NB];
#pragma unroll
      for (int i = 0; i < MB; ++i) fa[i] = as_v8<T>(*(const i32x4*)(pbuf + (pa[i] ^ (s << 5))));
#pragma unroll
      for (int j = 0; j < NB; ++j) fw[j] = as_v8<T>(*(const i32x4*)(wbuf + (lds_rw[j] ^ (s << 5))));
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = Tr<T>::mfma(fw[j], fa[i], acc[i][j]);
    }
  };
  TileC ct = tile_coords(tile0);
  pl_ct = ct;
  setup_patch(ct);
  setup_w(ct);
  issue_patch();                               // patch 0
  issue_w();
  issue_w();
  issue_w();
  issue_w();
  zero_acc();
  if (GNIN) {                                  // patch 0 is normalised before the first barrier of the loop
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    transform();
    tf_cnt = 0;
  }
  const bool tf_early = ((wave >> 2) & 1) == 0;  // the two waves of a SIMD transform at different times
  int c_c = 0, c_t = 0, ctile = tile0;         // compute-side chunk / tap / tile
  long long cpatch = 0;                        // global index of the patch step g reads
  int rs = 0;                                  // ring slot of step g
  for (long long g = 0; g < total; g += 2) {
    const long long younger = wl_issued - g - 2;
    if (younger >= 2) wait_vm<2 * SW>();
    else if (younger == 1) wait_vm<SW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    bool do_tf = false;
    if (GNIN) do_tf = tf_cnt != 0 && --tf_cnt == 0;   // the patch issued two intervals ago has landed
    issue_w();                                 // stages g+4, g+5 into the slots read one pair ago
    issue_w();
    const bool straddle = c_t == 8;            // step g+1 starts the next chunk
    if (!straddle && pl_issued == cpatch + 1) issue_patch();
    const int rs1 = rs + 1 == S ? 0 : rs + 1;
    if (GNIN && do_tf && tf_early) transform();
    __builtin_amdgcn_s_setprio(1);
    compute_step(smem + rs * WSTAGE, pbase + (int)(cpatch & 1) * PATCH, c_t);
    const int t1 = straddle ? 0 : c_t + 1;
    const long long p1 = straddle ? cpatch + 1 : cpatch;
    if (GNIN && do_tf && !tf_early) {
      __builtin_amdgcn_s_setprio(0);
      transform();
      __builtin_amdgcn_s_setprio(1);
    }
    compute_step(smem + rs1 * WSTAGE, pbase + (int)(p1 & 1) * PATCH, t1);
    __builtin_amdgcn_s_setprio(0);
    rs = rs1 + 1 == S ? 0 : rs1 + 1;
    // advance (chunk, tap) by two steps
    cpatch = p1;
    c_t = t1 + 1;
    if (straddle) ++c_c;
    if (c_t == 9) { c_t = 0; ++cpatch; ++c_c; }
    if (c_c == cpt) {
      c_c = 0;
      epilogue(ct);
      zero_acc();
      ctile += nxb;
      if (g + 2 < total) ct = tile_coords(ctile);
    }
  }
}

template <typename T, int BN, bool GNIN>
static int launch_halo(const GemmP& p, hipStream_t st) {
  GemmP q = p;
  q.ntm = p.M / 256;
  q.ntn = p.N / BN;
  q.tw = 16; q.tw_log2 = 4;
  q.tpr = p.Wo / 16;
  q.tpi = q.tpr * (p.Ho / 16);
  q.gn_chunks = p.gn_partial ? conv_halo_gn_chunks(p) : 0;
  if (q.gn_chunks == 0) q.gn_partial = nullptr;
  const size_t lds = 6 * (size_t)BN * 64 + 2 * 21 * 1024 + (GNIN ? (size_t)p.Cin * 8 : 0);
  int nwg = q.ntm * q.ntn;
  if (nwg > 256) nwg = 256;
  nwg = (nwg + 7) & ~7;
  auto kfn = conv_halo_kernel<T, BN, GNIN>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; }
  hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), lds, st, q);
  DFW_CHECK_LAUNCH();
  return 0;
}

static bool halo_shape_ok(const GemmP& p, int& bn) {
  if (p.taps != 9 || p.stride != 1 || p.pad != 1 || p.ups) return false;
  if (p.Hi != p.Ho || p.Wi != p.Wo || (p.Ho % 16) != 0 || (p.Wo % 16) != 0) return false;
  if (p.splitk > 1 || p.batch > 1 || (p.N % 128) != 0 || (p.Cin % 64) != 0) return false;
  if (p.out_mode != DFW_OUT_T || p.act != DFW_ACT_NONE || p.geglu || p.cs_n > 0) return false;
  if (p.M % 256 != 0 || p.rows_per_img != p.Ho * p.Wo) return false;
  bn = (p.N % 256) == 0 ? 256 : 128;
  return (long long)(p.M / 256) * (p.N / bn) >= 192;
}

// GroupNorm-on-input fusion: which shapes the kernel supports.  Whether to USE it is the caller's
// policy; measured on MI355X (scratch/bench_gnconv.py, norm + conv per call) it is not a win yet:
// 256 -> 256 channels gains 4-9 % in isolation but the whole episode is 1 % slower with it, 128-channel
// outputs lose 8 % (the BN=128 tile has one LDS fragment read per MFMA and is LDS-bound before the
// transform is added), 128 -> 256 loses 30 % (36-step tiles: per-tile costs dominate), and at 512
// input channels the saved HBM pass is only ~6 % of the conv.  The transform's exp + rcp per element
// (2 quarter-rate ops x 1.27 halo factor) is what it has to get rid of -- see DESIGN.md.
bool conv_halo_gn_input_ok(const GemmP& p) {
  int bn = 0;
  return halo_shape_ok(p, bn) && p.Cin <= 1024;
}

bool conv_halo_eligible(const GemmP& p, int& bn) {
  // Without a fused input norm this kernel ties gemm_big.hip (+-3 % measured on MI355X: both sit at
  // the power-limited MFMA rate of their 256-row structure), so the more general kernel stays the
  // default and DFW_CONV_HALO=1 opts in; a call that carries gn_in_coef always runs here.
  static const char* on = getenv("DFW_CONV_HALO");
  if (p.gn_coef) return conv_halo_gn_input_ok(p) && halo_shape_ok(p, bn);
  return on && halo_shape_ok(p, bn);
}

int conv_halo_gn_chunks(const GemmP& p) {
  int bn = 0;
  if (p.gn_groups <= 0 || !conv_halo_eligible(p, bn) || p.N % p.gn_groups) return 0;
  const int cpg = p.N / p.gn_groups;
  if (cpg != 4 && cpg != 8 && cpg != 16 && cpg != 32) return 0;
  return (p.Wo / 16) * (p.Ho / 16) * (8 / (bn / 64));
}

int launch_conv_halo(const GemmP& p, hipStream_t st) {
  int bn = 0;
  if (!conv_halo_eligible(p, bn)) return DFW_ESHAPE;
  const bool bf = p.dtype_bf16 != 0;
  if (p.gn_coef) {
    if (bn == 256) return bf ? launch_halo<__bf16, 256, true>(p, st) : launch_halo<_Float16, 256, true>(p, st);
    return bf ? launch_halo<__bf16, 128, true>(p, st) : launch_halo<_Float16, 128, true>(p, st);
  }
  if (bn == 256) return bf ? launch_halo<__bf16, 256, false>(p, st) : launch_halo<_Float16, 256, false>(p, st);
  return bf ? launch_halo<__bf16, 128, false>(p, st) : launch_halo<_Float16, 128, false>(p, st);
}

}  // namespace dfw
