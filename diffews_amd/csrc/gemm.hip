// Implicit-GEMM on MFMA for gfx950: Linear / conv1x1 / conv3x3 (stride 1|2, fused nearest-2x
// upsample) on NHWC activations, with bias / time-embedding / residual / SiLU / GEGLU epilogues.
//
// Tile: BM x BN x 64, 256 threads = 4 waves in a 2x2 grid, each wave (BM/2)x(BN/2) built from
// v_mfma_f32_32x32x16 blocks.  The MFMA is issued as D[n][m] = W_frag x A_frag so that every lane
// ends up owning 4 *contiguous output channels* of one output row -> 8-byte vector stores and
// vector bias / residual loads in the epilogue.
//
// LDS: double-buffered [BM+BN rows][64 k] storage-dtype tiles (128-byte rows), 16-byte chunks
// XOR-swizzled by ((row>>1)&7) which makes both the ds_write_b128 (8 lanes = one row) and the
// ds_read_b128 fragment reads (32 rows, same chunk) bank-conflict free (bank = (addr/4)%64).
// Global->LDS staging goes through registers (issue-early / write-late): the loads for K-step
// k+1 are issued before the MFMAs of step k and written to the other buffer after them, one
// barrier per K-step.  All global loads are bounds-checked buffer loads, so conv halos, ragged
// M tiles and N padding read as zero with no divergent branches.
#include "gemm_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace dfw {

// RF32: the residual is fp32 (dfw_gemm_args.residual_f32, the fp32 residual stream) -- its own instantiation so that
// the 16-bit default path's epilogue compiles exactly as before.
template <typename T, int BM, int BN, bool CONV, bool RF32 = false>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmP p) {
  constexpr int WTM = BM / 2, WTN = BN / 2, MB = WTM / 32, NB = WTN / 32;
  constexpr int SA = BM / 32, SW = BN / 32;
  constexpr int BUF = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  // ---- persistent tile walk.  Workgroup ids round-robin over the 8 XCDs, so XCD x owns the
  // contiguous tile range [x*Q, (x+1)*Q) and its resident workgroups sweep it together
  // (neighbouring tiles share A halos / W columns in that XCD's L2).  gridDim.x % 8 == 0.
  const int ntiles = p.ntm * p.ntn;
  const int nxb = gridDim.x >> 3, xcd = blockIdx.x & 7;
  const int Q = (ntiles + 7) >> 3;
  const int t_end = min(ntiles, (xcd + 1) * Q);
  int tile = xcd * Q + (blockIdx.x >> 3);

  const int z = blockIdx.y;
  int ks0 = 0, ks1 = p.nk;
  const char* Ab = p.A;
  const char* Wb = p.W;
  char* Cb = p.C;
  if (p.batch > 1) {
    Ab += (size_t)z * p.strideA * sizeof(T);
    Wb += (size_t)z * p.strideW * sizeof(T);
    Cb += (size_t)z * p.strideC * (p.out_mode == DFW_OUT_T ? sizeof(T) : sizeof(float));
  } else if (p.splitk > 1) {
    ks0 = (int)((long long)z * p.nk / p.splitk);
    ks1 = (int)((long long)(z + 1) * p.nk / p.splitk);
  }
  if (tile >= t_end || ks0 >= ks1) return;
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(Wb, p.w_bytes);

  // tile id -> coordinates.  Conv tiles are 2-D patches of tw x (BM/tw) output pixels so that the
  // 9 taps of a tile re-read an L2-resident halo instead of 9 shifted copies of a pixel row.
  auto tile_coords = [&](int t) -> TileC {
    TileC c;
    const int tn = t % p.ntn, tm = t / p.ntn;
    c.m0 = tm * BM;
    c.n0 = tn * BN;
    c.img = 0; c.oy0 = 0; c.ox0 = 0;
    if (CONV && p.tw) {
      c.img = tm / p.tpi;
      const int t2 = tm - c.img * p.tpi, tyi = t2 / p.tpr, txi = t2 - tyi * p.tpr;
      c.oy0 = tyi * (BM >> p.tw_log2);
      c.ox0 = txi << p.tw_log2;
    }
    return c;
  };
  // tile row r (0..BM) -> output row m (+ pixel coordinates for conv)
  auto row_to_m = [&](const TileC& c, int r, int& oy, int& ox, int& img) -> int {
    if (CONV && p.tw) {
      oy = c.oy0 + (r >> p.tw_log2);
      ox = c.ox0 + (r & (p.tw - 1));
      img = c.img;
      return (img * p.Ho + oy) * p.Wo + ox;
    }
    const int m = c.m0 + r;
    if constexpr (CONV) {
      img = m / p.rows_per_img;
      const int rem = m - img * p.rows_per_img;
      oy = rem / p.Wo;
      ox = rem - oy * p.Wo;
    }
    return m;
  };

  // ---- loader state: 16-byte chunk kc of tile rows (tid>>3) + 32*i, for the tile being LOADED
  // Staging is LDS-DMA (buffer_load ... lds): a wave-instruction writes 1 KiB = 8 tile rows
  // linearly (lane -> row lane>>3, 16-byte slot lane&7), so the bank swizzle is applied on the
  // SOURCE side: the thread that fills slot j of a row fetches global chunk j ^ ((row>>1)&7).
  const int srow = tid >> 3;
  const int kc = (tid & 7) ^ ((srow >> 1) & 7);
  uint32_t a_off[SA];       // linear: byte offset of the row (+chunk), or OOB
  int a_iy0[SA], a_ix0[SA]; // conv
  uint32_t a_pix[SA];
  uint32_t w_off[SW];
  const unsigned limH = p.ups ? 2 * p.Hi : p.Hi, limW = p.ups ? 2 * p.Wi : p.Wi;
  const int ush = p.ups ? 1 : 0;
  auto setup_loader = [&](const TileC& c) {
#pragma unroll
    for (int i = 0; i < SA; ++i) {
      int oy = 0, ox = 0, img = 0;
      const int m = row_to_m(c, srow + 32 * i, oy, ox, img);
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
      const int n = c.n0 + srow + 32 * i;
      w_off[i] = n < p.N ? (uint32_t)(((size_t)n * p.K + kc * 8) * sizeof(T)) : kOOB;
    }
  };

  // fragment read addresses (k-substep s: ^ (s<<5)); the LDS image is [row][slot] with
  // slot = chunk ^ ((row>>1)&7): conflict-free ds_read_b128 for 32 rows x one chunk
  uint32_t lds_ra[MB], lds_rw[NB];
  const uint32_t wave_lds = (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int row = wm * WTM + i * 32 + lr;
    lds_ra[i] = row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = BM + wn * WTN + i * 32 + lr;
    lds_rw[i] = row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);
  }

  using lds_ptr_t = __attribute__((address_space(3))) void*;
  auto dma16 = [&](__amdgpu_buffer_rsrc_t r, uint32_t off, char* dst) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, (int)off, 0, 0, 0);
  };
  // issue the LDS-DMA of K-step ks into buffer `buf` (8 rows x 128 B per wave-instruction)
  auto issue_loads = [&](int ks, char* buf) {
    char* dst = buf + wave_lds;
    // conv K walk: (tap, 64-channel chunk) derived from the K-step index -- carried as loop state they
    // were captured by reference and lived in scratch memory (a scratch load per K-step)
    int tap = 0, cc = 0;
    if constexpr (CONV) {
      if (p.conv_chunk_major) { cc = ks / 9; tap = ks - cc * 9; }
      else { tap = ks / p.cpt; cc = ks - tap * p.cpt; }
    }
    if constexpr (!CONV) {
#pragma unroll
      for (int i = 0; i < SA; ++i) dma16(ra, a_off[i] + (uint32_t)ks * 128u, dst + i * 4096);
    } else {
      const int ky = tap / 3, kx = tap - ky * 3;
      const uint32_t coff = (uint32_t)(cc * 64 + kc * 8);
#pragma unroll
      for (int i = 0; i < SA; ++i) {
        int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
        const bool ok = (unsigned)iy < limH && (unsigned)ix < limW;
        iy >>= ush;
        ix >>= ush;
        const uint32_t off = ((a_pix[i] + (uint32_t)(iy * p.Wi + ix)) * (uint32_t)p.lda + coff) * (uint32_t)sizeof(T);
        dma16(ra, ok ? off : kOOB, dst + i * 4096);   // out-of-range lanes write zeros (halo / padding)
      }
    }
    uint32_t koff = (uint32_t)ks * 128u;
    if constexpr (CONV) {
      // channel-chunk-major K walk (nine taps of a 64-channel chunk back to back): the shifted re-reads
      // of the input patch hit L2 one step after the first touch (see gemm_big.hip); split-K ranges are
      // contiguous runs of this walk
      koff = (uint32_t)(tap * p.Cin + cc * 64) * (uint32_t)sizeof(T);
    }
#pragma unroll
    for (int i = 0; i < SW; ++i) dma16(rw, w_off[i] + koff, dst + BM * 128 + i * 4096);
  };
  auto dma_wait_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  // ---- epilogue of one tile: lane owns output row r = .. + lr, channels 8g + 4*lh + (0..3)
  f32x16 acc[MB][NB];
  auto epilogue = [&](const TileC& c) {
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      int oy_, ox_, img_;
      const int m = row_to_m(c, wm * WTM + i * 32 + lr, oy_, ox_, img_);
      if (m >= p.M) continue;
      if constexpr (RF32) {   // host guarantees: no split-K, N % 4 == 0, no GEGLU
        const int img = p.rowbias ? (CONV && p.tw ? img_ : m / p.rows_per_img) : 0;
        epi_block_rf32<T, NB>(p, Cb, m, img, c.n0 + wn * WTN + 4 * lh, acc[i]);
        continue;
      }
      if (p.geglu) {
        if constexpr (NB >= 2) {
#pragma unroll
          for (int jp = 0; jp < NB / 2; ++jp)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int na = c.n0 + wn * WTN + (2 * jp) * 32 + 8 * g + 4 * lh;
              if (na >= p.N) continue;
              const int no = ((c.n0 + wn * WTN) >> 1) + jp * 32 + 8 * g + 4 * lh;
              f32x4 ba = *(const f32x4*)(p.bias + na), bg = *(const f32x4*)(p.bias + na + 32);
              float v[4];
#pragma unroll
              for (int e = 0; e < 4; ++e)
                v[e] = (acc[i][2 * jp][4 * g + e] + ba[e]) * gelu_erf(acc[i][2 * jp + 1][4 * g + e] + bg[e]);
              *(i32x2*)(Cb + ((size_t)m * p.ldc + no) * sizeof(T)) = pack4<T>(v);
            }
        }
        continue;
      }
      if (p.splitk <= 1 && (p.N & 3) == 0) {
        const int img = p.rowbias ? (CONV && p.tw ? img_ : m / p.rows_per_img) : 0;
        epi_block<T, NB>(p, Cb, m, img, c.n0 + wn * WTN + 4 * lh, acc[i]);
        continue;
      }
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = c.n0 + wn * WTN + j * 32 + 8 * g + 4 * lh;
          if (n >= p.N) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          if (p.splitk > 1) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *(f32x4*)(p.partial + ((size_t)z * p.M + m) * p.N + n) = o;
          } else {
            epilogue4<T>(p, Cb, m, n, v);
          }
        }
    }
  };

  // ---- software pipeline, continuous across tiles: the loads of the NEXT K-step (or of the next
  // tile's first K-step) are in flight while the current step's MFMAs run; one barrier per step.
  TileC ct = tile_coords(tile);
  setup_loader(ct);
  issue_loads(ks0, smem);
  dma_wait_barrier();
  int cur = 0;
  while (true) {
    const int next_tile = tile + nxb;
    const bool has_next = next_tile < t_end;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](const char* buf) {
      __builtin_amdgcn_s_setprio(1);   // the co-resident workgroup's loads / epilogue yield to this MFMA run
#pragma unroll
      for (int s = 0; s < 4; ++s) {
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
    for (int ks = ks0; ks + 1 < ks1; ++ks) {
      issue_loads(ks + 1, smem + (cur ^ 1) * BUF);   // lands in the other buffer during the MFMAs
      compute(smem + cur * BUF);
      dma_wait_barrier();
      cur ^= 1;
    }
    // last K-step of the tile (peeled): prefetch the next tile's first step, then the epilogue's
    // stores go out while that DMA is still landing
    if (has_next) {
      setup_loader(tile_coords(next_tile));
      issue_loads(ks0, smem + (cur ^ 1) * BUF);
    }
    compute(smem + cur * BUF);
    epilogue(ct);
    dma_wait_barrier();
    cur ^= 1;
    if (!has_next) break;
    tile = next_tile;
    ct = tile_coords(tile);
  }
}

// split-K second pass: sum the fp32 slabs in split order (deterministic) and run the epilogue.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmP p) {
  const int nq = p.N >> 2;
  const long long total = (long long)p.M * nq;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int m = (int)(e / nq), n = (int)(e - (long long)m * nq) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int zz = 0; zz < p.splitk; ++zz) {
      f32x4 t = *(const f32x4*)(p.partial + ((size_t)zz * p.M + m) * p.N + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += t[i];
    }
    if (p.res_f32) epilogue4<T, true>(p, p.C, m, n, v);
    else epilogue4<T>(p, p.C, m, n, v);
  }
}

template <typename T, int BM, int BN>
static int launch_tile(const GemmP& p, hipStream_t st) {
  GemmP q = p;
  q.ntm = (p.M + BM - 1) / BM;
  q.ntn = (p.N + BN - 1) / BN;
  q.tw = 0; q.tw_log2 = 0; q.tpr = 0; q.tpi = 0;
  if (p.taps == 9) {
    const int tw = 16, th = BM / tw;
    if (p.Wo % tw == 0 && p.Ho % th == 0) {
      q.tw = tw; q.tw_log2 = 4;
      q.tpr = p.Wo / tw;
      q.tpi = q.tpr * (p.Ho / th);
    }
  }
  // persistent workgroups: enough to fill 256 CUs at the tile's LDS-limited residency, multiple of 8
  const size_t lds = 2 * (BM + BN) * 128;
  const int per_cu = (int)((160 * 1024) / lds) < 4 ? (int)((160 * 1024) / lds) : 4;
  const int zdim = p.batch > 1 ? p.batch : (p.splitk > 1 ? p.splitk : 1);
  int nwg = q.ntm * q.ntn;
  int cap = 256 * per_cu / (zdim < 4 ? zdim : 4);
  if (cap < 256) cap = 256;
  if (nwg > cap) nwg = cap;
  nwg = (nwg + 7) & ~7;
  dim3 grid(nwg, zdim);
  const bool rf32 = p.res_f32 && p.splitk <= 1 && (p.N & 3) == 0 && !p.geglu;   // (split-K: the reduce pass adds the residual)
  if (p.taps == 1) {
    if (rf32) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false, true>), grid, dim3(256), lds, st, q);
    else hipLaunchKernelGGL((gemm_kernel<T, BM, BN, false>), grid, dim3(256), lds, st, q);
  } else {
    if (rf32) hipLaunchKernelGGL((gemm_kernel<T, BM, BN, true, true>), grid, dim3(256), lds, st, q);
    else hipLaunchKernelGGL((gemm_kernel<T, BM, BN, true>), grid, dim3(256), lds, st, q);
  }
  DFW_CHECK_LAUNCH();
  if (p.splitk > 1) {
    const long long total = (long long)p.M * (p.N >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, q);
    DFW_CHECK_LAUNCH();
  }
  return 0;
}

// Tile + split-K plan from a small cost model fitted on MI355X measurements (scratch/bench_small.py):
//   R   = K-steps per microsecond a CU sustains with its resident workgroups of that tile,
//   L   = K-step latency of a workgroup that is alone on its CU (the critical path when the grid
//         is smaller than the chip),
//   split-K adds the fp32 slab round trip and a second launch.
// The weight-bound 8x8 / 16x16 UNet levels pick 128-wide tiles with split-K 4..8, the token GEMMs of
// the 64x64 level pick narrow tiles, the big VAE convs 128x128 without split.
struct TileCost { int bm, bn; float R, L, E; };
static const TileCost kTiles[3] = {{128, 128, 1.67f, 0.70f, 0.f}, {128, 64, 2.8f, 0.50f, 0.f}, {64, 64, 3.7f, 0.40f, 0.f}};
static const TileCost kLinTiles[3] = {{128, 128, 5.29f, 0.312f, 1.94f}, {128, 64, 8.52f, 0.182f, 0.13f}, {64, 64, 12.9f, 0.145f, 0.f}};

static void plan_gemm(GemmP& p, int& bm, int& bn) {
  const int fbm = cfg().gemm_bm, fbn = cfg().gemm_bn;    // sweeps only: a forced tile instead of the cost model
  const bool fixed_sk = p.splitk >= 1;
  if (p.geglu) { bm = 128; bn = 128; p.splitk = 1; return; }
  const bool can_split = p.batch <= 1 && (p.N % 4) == 0;
  float best = 1e30f;
  int best_t = 0, best_sk = 1;
  for (int t = 0; t < 3; ++t) {
    const TileCost& tc = kTiles[t];
    if (fbm && (fbm != tc.bm || fbn != tc.bn)) continue;
    const double tiles = (double)((p.M + tc.bm - 1) / tc.bm) * ((p.N + tc.bn - 1) / tc.bn) * (p.batch > 1 ? p.batch : 1);
    for (int sk = 1; sk <= 16; sk *= 2) {
      if (!fixed_sk && sk > 1 && (!can_split || p.nk / sk < 8)) continue;
      const int use_sk = fixed_sk ? p.splitk : sk;
      const double blocks = tiles * use_sk, ksteps = (double)p.nk / use_sk;
      double t_us;
      if (p.taps == 1) {
        // Linear layers (short K: 10..160 K-steps): occupancy-quantised form, re-fitted on the lock-step batch of 8 latents
        // (scratch/sweep_plan.py, 3 tiles x 5 split factors x 15 shapes): the busiest CU runs ceil(blocks / 256) workgroups
        // at min(throughput R, n / latency L) K-steps per us and pays a per-tile prologue / epilogue E.  The older form
        // below mis-ranked the 128 x 128 tile on these shapes by up to 54 % (8192 x 640 x 640: 25.2 vs 17.1 us).
        const TileCost& tl = kLinTiles[t];
        const double ncu = (double)(long long)((blocks + 255.0) / 256.0);
        const double rate = tl.R < ncu / tl.L ? tl.R : ncu / tl.L;
        t_us = ncu * (ksteps / rate + tl.E) + 3.0;
        if (use_sk > 1) t_us += 2.0 * use_sk * (double)p.M * p.N * 4.0 / 3.0e6;
      } else {
        const double busy = blocks < 256.0 ? blocks : 256.0;
        t_us = blocks * ksteps / busy / tc.R;
        const double crit = ksteps * tc.L;
        if (crit > t_us) t_us = crit;
        t_us += 3.0;
        if (use_sk > 1) t_us += 3.0 + 2.0 * use_sk * (double)p.M * p.N * 4.0 / 3.0e6;
      }
      if (t_us < best) { best = (float)t_us; best_t = t; best_sk = use_sk; }
      if (fixed_sk) break;
    }
  }
  bm = kTiles[best_t].bm;
  bn = kTiles[best_t].bn;
  p.splitk = best_sk;
  if (p.splitk > p.nk) p.splitk = p.nk;
  if (p.splitk < 1) p.splitk = 1;
}

template <typename T>
static int launch_gemm(const GemmP& p, hipStream_t st) {
  int bm = p.plan_bm, bn = p.plan_bn;
  if (bm == 128 && bn == 128) return launch_tile<T, 128, 128>(p, st);
  if (bm == 128) return launch_tile<T, 128, 64>(p, st);
  return launch_tile<T, 64, 64>(p, st);
}

}  // namespace dfw

using namespace dfw;

static int fill_params(const dfw_gemm_args* a, GemmP& p, int& esz) {
  if (!a || !a->A || !a->W || !a->C) return DFW_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  esz = 2;
  if (a->taps != 1 && a->taps != 9) return DFW_ESHAPE;
  if (a->Cin <= 0 || a->Cin % 64 != 0 || a->K != a->taps * a->Cin) return DFW_ESHAPE;
  if (a->lda % 8 != 0) return DFW_ESHAPE;
  if (a->out_mode == DFW_OUT_T && (a->N % 4 == 0) && (a->ldc % 4 != 0)) return DFW_ESHAPE;
  if (a->residual && (a->N % 4 == 0) && (a->ldr % 4 != 0)) return DFW_ESHAPE;
  if (a->residual_f32 && (!a->residual || a->N % 4 != 0 || a->ldr % 4 != 0 || ((uintptr_t)a->residual & 15) || a->act != DFW_ACT_NONE ||
                          a->colscale_n > 0 || a->out_mode == DFW_OUT_NCHW_F32 || a->geglu))
    return DFW_ESHAPE;     // (GEGLU epilogues exist in the 16-bit-residual instantiations only: they would read the fp32 residual as 16-bit data)
  if (a->rowbias && a->ld_rowbias > 0 && (a->ld_rowbias % 4 != 0 || a->ld_rowbias < a->N)) return DFW_ESHAPE;
  if (a->a_elems <= 0 || a->w_elems <= 0) return DFW_EINVAL;
  if (a->a_elems * esz >= (1ll << 31) || a->w_elems * esz >= (1ll << 31)) return DFW_ERANGE;
  if (a->w_elems < (int64_t)a->N * a->K) return DFW_EINVAL;
  const int batch = a->batch > 1 ? a->batch : 1;
  const int splitk = a->splitk >= 1 ? a->splitk : 0;  // 0 = let plan_gemm choose
  if (batch > 1 && splitk > 1) return DFW_ESHAPE;
  if (a->geglu && (splitk > 1 || batch > 1 || a->N % 128 != 0 || !a->bias || a->out_mode != DFW_OUT_T ||
                   a->residual || a->rowbias))
    return DFW_ESHAPE;
  if (splitk > 1 && (a->N % 4 != 0)) return DFW_ESHAPE;
  if (a->taps == 9) {
    if (a->Hi <= 0 || a->Wi <= 0 || a->Ho <= 0 || a->Wo <= 0 || a->stride <= 0) return DFW_EINVAL;
    if (a->M % (a->Ho * a->Wo) != 0) return DFW_ESHAPE;
    const int64_t need = (int64_t)(a->M / (a->Ho * a->Wo)) * a->Hi * a->Wi * a->lda;
    if (a->a_elems < need - (a->lda - a->Cin)) return DFW_EINVAL;
  } else {
    if (a->a_elems < (int64_t)(a->M - 1) * a->lda + a->K) return DFW_EINVAL;
  }
  const int rpi = a->rows_per_img > 0 ? a->rows_per_img : (a->taps == 9 ? a->Ho * a->Wo : a->M);
  if ((a->rowbias || a->out_mode == DFW_OUT_NCHW_F32) && rpi <= 0) return DFW_EINVAL;
  p.A = (const char*)a->A; p.W = (const char*)a->W; p.C = (char*)a->C;
  p.Wblk = a->taps == 9 ? (const char*)a->W_blocked : nullptr;
  p.bias = a->bias; p.rowbias = a->rowbias; p.residual = (const char*)a->residual;
  p.partial = (float*)a->workspace;
  p.a_bytes = (uint32_t)(a->a_elems * esz);
  p.w_bytes = (uint32_t)(a->w_elems * esz);
  p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldc = a->ldc; p.ldr = a->ldr;
  p.ldrb = a->ld_rowbias > 0 ? a->ld_rowbias : a->N;
  p.taps = a->taps; p.Cin = a->Cin; p.Hi = a->Hi; p.Wi = a->Wi; p.Ho = a->Ho; p.Wo = a->Wo;
  p.stride = a->stride; p.pad = a->pad; p.ups = a->ups; p.rows_per_img = rpi;
  if (a->colscale_n < 0 || (a->colscale_n % 64) != 0 || a->colscale_n > a->N || (a->colscale_n > 0 && a->geglu)) return DFW_ESHAPE;
  p.cs = a->colscale; p.cs_n = a->colscale_n;
  p.out_scale = a->out_scale; p.act = a->act; p.geglu = a->geglu; p.out_mode = a->out_mode;
  p.splitk = splitk; p.batch = batch;
  p.res_f32 = a->residual_f32 ? 1 : 0;
  p.strideA = a->strideA; p.strideW = a->strideW; p.strideC = a->strideC;
  p.nk = a->K / 64; p.cpt = a->Cin / 64; p.ntn = 0; p.ntm = 0;
  p.dtype_bf16 = a->dtype == DFW_BF16;
  p.gn_partial = a->gn_partial; p.gn_groups = a->gn_groups; p.gn_chunks = 0;
  p.conv_chunk_major = a->taps == 9 ? 1 : 0;   // channel-chunk-major K walk (the tap-major one measured 4.5-6 % slower)
  if (p.splitk > p.nk) p.splitk = p.nk;
  plan_gemm(p, p.plan_bm, p.plan_bn);
  return 0;
}

extern "C" int dfw_gemm_kernel_name(const dfw_gemm_args* a, char* buf, size_t n) {
  GemmP p;
  int esz;
  int rc = fill_params(a, p, esz);
  if (rc) return rc;
  if (!buf || n == 0) return DFW_EINVAL;
  int big_bm = 0, big_bn = 0, big_bk = 0;
  if (gemm8_n160_eligible(p)) {
    snprintf(buf, n, "gemm8_kernel<%s,256,160,64,lin>", a->dtype == DFW_BF16 ? "bf16" : "f16");
    return 0;
  }
  {
    int pbm = 0, pbn = 0;
    if (conv_patch8_eligible(p, pbn)) {
      snprintf(buf, n, "conv_patch8_kernel<%s,256,%d>", a->dtype == DFW_BF16 ? "bf16" : "f16", pbn);
      return 0;
    }
    if (conv_patch_eligible(p, pbm, pbn)) {
      snprintf(buf, n, "conv_patch_kernel<%s,%d,%d>", a->dtype == DFW_BF16 ? "bf16" : "f16", pbm, pbn);
      return 0;
    }
  }
  if (gemm_big_eligible(p, big_bm, big_bn, big_bk)) {
    if (big_bm == 256 && (big_bn == 256 || big_bn == 128) && gemm8_eligible(p, big_bn)) {
      snprintf(buf, n, "gemm8_kernel<%s,256,%d,64,%s>", a->dtype == DFW_BF16 ? "bf16" : "f16", big_bn, a->taps == 9 ? "conv" : "lin");
      return 0;
    }
    snprintf(buf, n, "gemm_big_kernel<%s,%d,%d,%d,%s>", a->dtype == DFW_BF16 ? "bf16" : "f16", big_bm, big_bn,
             big_bk, a->taps == 9 ? "conv" : "lin");
    return 0;
  }
  snprintf(buf, n, "gemm_kernel<%s,%d,%d,%s>%s", a->dtype == DFW_BF16 ? "bf16" : "f16", p.plan_bm, p.plan_bn,
           a->taps == 9 ? "conv" : "lin", p.splitk > 1 ? "+splitk" : "");
  return 0;
}

extern "C" int32_t dfw_gemm_gn_chunks(const dfw_gemm_args* a) {
  GemmP p;
  int esz;
  if (fill_params(a, p, esz)) return 0;
  int pbm = 0, pbn = 0;
  if (conv_patch8_eligible(p, pbn)) return conv_patch8_gn_chunks(p);
  if (conv_patch_eligible(p, pbm, pbn)) return conv_patch_gn_chunks(p);
  return gemm_big_gn_chunks(p);
}

extern "C" size_t dfw_gemm_workspace_bytes(const dfw_gemm_args* a) {
  GemmP p;
  int esz;
  if (fill_params(a, p, esz) || p.splitk <= 1) return 0;
  return (size_t)p.splitk * (size_t)p.M * (size_t)p.N * sizeof(float);
}

extern "C" int dfw_gemm(const dfw_gemm_args* a, dfw_stream_t stream) {
  GemmP p;
  int esz;
  int rc = fill_params(a, p, esz);
  if (rc) return rc;
  if (p.splitk > 1) {
    if (!a->workspace || a->workspace_bytes < (size_t)p.splitk * p.M * p.N * sizeof(float)) return DFW_EWORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  int big_bm = 0, big_bn = 0, big_bk = 0;
  if (gemm8_n160_eligible(p)) return launch_gemm8(p, st, 160);
  {
    int pbm = 0, pbn = 0;
    if (conv_patch8_eligible(p, pbn)) return launch_conv_patch8(p, st);
    if (conv_patch_eligible(p, pbm, pbn)) return launch_conv_patch(p, st);
  }
  if (gemm_big_eligible(p, big_bm, big_bn, big_bk)) return launch_gemm_big(p, st);
  return a->dtype == DFW_BF16 ? launch_gemm<__bf16>(p, st) : launch_gemm<_Float16>(p, st);
}
