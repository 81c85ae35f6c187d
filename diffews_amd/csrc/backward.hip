// Backward kernels of the training step (train_icl_multitask_nocrop_nearest_nshot_v3.py:1374-1396):
// weight gradients (a TN GEMM on MFMA with both operands read transposed out of row-major LDS tiles), bias /
// time-projection gradients (column sums), GroupNorm / LayerNorm / GEGLU backward, the small data-movement
// pieces of the conv backward (zero-stuffing for stride 2, 2x2 pooling for the fused upsample), the loss
// gradient, and the optimizer step.  Data gradients of Linear / conv3x3 reuse the FORWARD implicit-GEMM kernels
// with transposed / tap-mirrored weights (dfw_gemm); the attention backward lives in attention_bwd.hip.
// Deterministic throughout: split reductions go through fp32 slabs folded in a fixed order, no float atomics.
#include "common.h"
#include "attention_common.h"

#ifndef DFW_TN_PRIO_ON
#define DFW_TN_PRIO_ON 1
#endif
#if DFW_TN_PRIO_ON
#define DFW_TN_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define DFW_TN_PRIO(x) ((void)0)
#endif

namespace dfw {

// ---------------------------------------------------------------------------------------------
// TN GEMM:  C[z][n][k] = sum_m A[m][n] * B[row(m, tap)][k]      (fp32 out)
//   Linear weight gradient:  A = dY [M][N], B = X [M][K]          (dW = dY^T X, torch.nn.Linear.weight layout)
//   conv3x3 weight gradient: taps = 9, B rows gathered with the forward conv's im2col addressing
//                            (stride / pad / fused nearest-2x upsample), C[n][tap][k] = packed [Cout][ky][kx][Cin]
// The reduction index m is the ROW index of both operands, so both MFMA fragments are transposed reads
// (ds_read_b64_tr_b16) of row-major [64 m][128 cols] LDS tiles; the image is guide image (b) (256-byte rows,
// 16-byte chunk ch of row r at 16 * (ch ^ (((r & 3) << 2) | ((r >> 2) & 3)))): conflict-free transposed reads.
// Workgroup = 4 waves = 128 x 128 output tile (wave: 64 x 64 = 2 x 2 MFMA 32x32x16 blocks); M is split over
// gridDim.y workgroups writing fp32 slabs, folded by tn_reduce_kernel in split order.
struct TnP {
  const char* A; const char* B; float* slab;
  uint32_t a_bytes, b_bytes;
  int M, N, Kc, lda, ldb;
  int taps, Hi, Wi, Ho, Wo, stride, pad, ups;
  int splits, mchunk;
  int batch2; long long strideA, strideB, strideA2, strideB2;   // z = (b1 * batch2 + b2) * taps + tap
  float* out; long long ldo_b, ldo_n, ldo_t; float scale; int accumulate;   // splits == 1: written directly, no slab pass
};

__device__ __forceinline__ uint32_t tn_off(int row, int ch) {
  return (uint32_t)(256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))));
}

template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_kernel(const TnP p) {
  constexpr int TILE = 64 * 256;   // bytes of one [64][128] tile
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];   // [buf][A|B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 1, wk = wave & 1;
  const int lh = lane >> 5;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const int ntk = (p.Kc + 127) >> 7;
  const int n0 = ((int)blockIdx.x / ntk) * 128, k0 = ((int)blockIdx.x % ntk) * 128;
  const int split = blockIdx.y, z = blockIdx.z;
  const int tap = z % p.taps, bb = z / p.taps;
  const int b1 = bb / p.batch2, b2 = bb - b1 * p.batch2;
  const char* Ab = p.A + ((size_t)b1 * p.strideA + (size_t)b2 * p.strideA2) * sizeof(T);
  const char* Bb = p.B + ((size_t)b1 * p.strideB + (size_t)b2 * p.strideB2) * sizeof(T);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ab, p.a_bytes), rb = make_rsrc(Bb, p.b_bytes);
  const int m_begin = split * p.mchunk, m_end = min(p.M, m_begin + p.mchunk);
  const int nsteps = (m_end - m_begin + 63) >> 6;
  const int ky = tap / 3, kx = tap - ky * 3;
  const unsigned limH = p.ups ? 2 * p.Hi : p.Hi, limW = p.ups ? 2 * p.Wi : p.Wi;
  const int ush = p.ups ? 1 : 0;

  i32x4 ga[4], gb[4];
  auto issue = [&](int step) {
    const int mb = m_begin + step * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, row = e >> 4, ch = e & 15;
      const int m = mb + row;
      const bool okm = m < m_end;
      const int na = n0 + ch * 8, kb = k0 + ch * 8;
      ga[i] = buf_load16(ra, (okm && na < p.N) ? (uint32_t)(((size_t)m * p.lda + na) * sizeof(T)) : kOOB);
      uint32_t off = kOOB;
      if (okm && kb < p.Kc) {
        if (p.taps == 1) {
          off = (uint32_t)(((size_t)m * p.ldb + kb) * sizeof(T));
        } else {
          const int hw = p.Ho * p.Wo;
          const int img = m / hw, rem = m - img * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
          int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
          if ((unsigned)iy < limH && (unsigned)ix < limW) {
            iy >>= ush;
            ix >>= ush;
            off = (uint32_t)((((size_t)img * p.Hi + iy) * p.Wi + ix) * p.ldb + kb) * (uint32_t)sizeof(T);
          }
        }
      }
      gb[i] = buf_load16(rb, off);
    }
  };
  auto write_lds = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, row = e >> 4, ch = e & 15;
      *(i32x4*)(buf + tn_off(row, ch)) = ga[i];
      *(i32x4*)(buf + TILE + tn_off(row, ch)) = gb[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nsteps > 0) {
    issue(0);
    write_lds(smem);
    __syncthreads();
  }
  int cur = 0;
  for (int s = 0; s < nsteps; ++s) {
    const bool more = s + 1 < nsteps;
    if (more) issue(s + 1);
    const char* at = smem + cur * 2 * TILE;
    const char* bt = at + TILE;
#pragma unroll
    for (int ss = 0; ss < 4; ++ss) {
      typename Tr<T>::v8 fa[2], fb[2];
      const int r0 = ss * 16 + 8 * lh + tq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ca = wn * 8 + i * 4 + 2 * tg + (tp >> 1), cb = wk * 8 + i * 4 + 2 * tg + (tp >> 1);
        const typename Tr<T>::v4 alo = lds_tr_read<T>(at + tn_off(r0, ca) + 8 * (tp & 1));
        const typename Tr<T>::v4 ahi = lds_tr_read<T>(at + tn_off(r0 + 4, ca) + 8 * (tp & 1));
        const typename Tr<T>::v4 blo = lds_tr_read<T>(bt + tn_off(r0, cb) + 8 * (tp & 1));
        const typename Tr<T>::v4 bhi = lds_tr_read<T>(bt + tn_off(r0 + 4, cb) + 8 * (tp & 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) { fa[i][j] = alo[j]; fa[i][4 + j] = ahi[j]; fb[i][j] = blo[j]; fb[i][4 + j] = bhi[j]; }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Tr<T>::mfma(fa[i], fb[j], acc[i][j]);
    }
    if (more) write_lds(smem + (cur ^ 1) * 2 * TILE);
    __syncthreads();
    cur ^= 1;
  }
  // D layout: col = lane & 31 (k), row = (r & 3) + 8 * (r >> 2) + 4 * lh (n)
  if (p.splits == 1) {
    float* o = p.out + (size_t)bb * p.ldo_b + (size_t)tap * p.ldo_t;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int k = k0 + wk * 64 + j * 32 + (lane & 31);
        if (k >= p.Kc) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (n < p.N) {
            float* d = o + (size_t)n * p.ldo_n + k;
            *d = (p.accumulate ? *d : 0.f) + p.scale * acc[i][j][r];
          }
        }
      }
    return;
  }
  float* out = p.slab + ((size_t)split * gridDim.z + z) * (size_t)p.N * p.Kc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 64 + j * 32 + (lane & 31);
      if (k >= p.Kc) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < p.N) out[(size_t)n * p.Kc + k] = acc[i][j][r];
      }
    }
}

// Same mathematics, fragments and LDS image as gemm_tn_kernel, different data movement and tile size.
//   * Output tile (NSA * 128) x (NSB * 128): 128 x 128 moves one operand byte per 64 FLOP through L2 -> LDS and tops out
//     near 0.65 PFLOP/s however large the problem (measured: ~10 TB/s of L2 reads); 256 x 256 halves that.  Every wave
//     owns a 64 x 64 block (4 accumulator tiles), so a workgroup has 4 * NSA * NSB waves.
//   * The operands arrive by LDS-DMA into a ring: stage = 32 reduction rows of every 128-column sub-tile (8 KiB each,
//     256-byte rows, the tn_off image), S = 4 stages, three in flight across the single barrier of a step behind a
//     counted s_waitcnt vmcnt (the wave's own DPS instructions per stage).  One DMA instruction covers 4 rows x 256 B of
//     one sub-tile: lane -> row lane >> 4, LDS slot lane & 15; the tn_off swizzle sits on the SOURCE chunk (LDS is written
//     linearly): slot c of row r holds chunk c ^ (((r & 3) << 2) | ((r >> 2) & 3)).  Instruction g = j * NW + wave of a
//     stage covers rows 4 * (g & 7) .. + 4 of sub-tile g >> 3, and NW is a multiple of 4, so (r >> 2) & 3 = wave & 3:
//     the source chunk is the lane constant c ^ (((lane >> 4) << 2) | (wave & 3)).
//   * CONV: the im2col gather keeps (image, oy, ox) of each of the lane's rows and advances them by 32 rows per stage with
//     predicated wraps (Wo >= 8 and Ho * Wo >= 32, checked on the host) -- no division, no divergent branch in the loop.
template <typename T, int NSA, int NSB, bool CONV>
__global__ __launch_bounds__(256 * NSA * NSB) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_ring_kernel(const TnP p) {
  constexpr int BKR = 32, S = 4, SUB = BKR * 256, NSUB = NSA + NSB, STAGE = NSUB * SUB;
  constexpr int NW = 4 * NSA * NSB, DPS = NSUB * 8 / NW;
  static_assert(NSUB * 8 % NW == 0, "DMA instructions divide evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / (2 * NSB), wk = wave % (2 * NSB);          // 64-column block of the tile along n / along k
  const int lh = lane >> 5;
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const int ntk = (p.Kc + NSB * 128 - 1) / (NSB * 128);
  // XCD-aware walk: workgroups go round-robin over the 8 XCDs in launch order, and every tile (and tap) of one split
  // reads the same rows of A and B -- so the launch-order index is re-read as (XCD c, its k-th workgroup) -> logical
  // index c * (total / 8) + k with the tile fastest, then the tap / batch, then the split: each XCD's L2 serves whole
  // splits instead of an eighth of every split.
  int tile_x = blockIdx.x, split = blockIdx.y, z = blockIdx.z;
  {
    const long long gx = gridDim.x, gy = gridDim.y, gz = gridDim.z, total = gx * gy * gz;
    if ((total & 7) == 0 && total >= 64) {
      const long long id = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
      long long l = (id & 7) * (total >> 3) + (id >> 3);
      tile_x = (int)(l % gx);
      l /= gx;
      z = (int)(l % gz);
      split = (int)(l / gz);
    }
  }
  const int n0 = (tile_x / ntk) * (NSA * 128), k0 = (tile_x % ntk) * (NSB * 128);
  const int tap = z % p.taps, bb = z / p.taps;
  const int b1 = bb / p.batch2, b2 = bb - b1 * p.batch2;
  const char* Ab = p.A + ((size_t)b1 * p.strideA + (size_t)b2 * p.strideA2) * sizeof(T);
  const char* Bb = p.B + ((size_t)b1 * p.strideB + (size_t)b2 * p.strideB2) * sizeof(T);
  const u32x4 ra = make_srd(Ab, p.a_bytes), rb = make_srd(Bb, p.b_bytes);
  const uint32_t lds0 = lds_addr(smem);
  const int m_begin = split * p.mchunk, m_end = min(p.M, m_begin + p.mchunk);
  const int nsteps = (m_end - m_begin + BKR - 1) / BKR;
  const int ky = tap / 3, kx = tap - ky * 3;
  const int limH = p.ups ? 2 * p.Hi : p.Hi, limW = p.ups ? 2 * p.Wi : p.Wi;
  const int ush = p.ups ? 1 : 0;

  // ---- loader state per DMA instruction of a stage: column offset (fixed), row, and (CONV, B operand) its pixel
  const int ch = (lane & 15) ^ (((lane >> 4) << 2) | (wave & 3));
  uint32_t colb[DPS];                      // byte offset of the lane's 16 bytes inside its row, kOOB if out of range
  int lm[DPS], loy[DPS], lox[DPS], limg[DPS];
#pragma unroll
  for (int j = 0; j < DPS; ++j) {
    const int g = j * NW + wave, t = g >> 3;
    const bool isA = t < NSA;
    const int col = isA ? n0 + t * 128 + ch * 8 : k0 + (t - NSA) * 128 + ch * 8;
    colb[j] = col < (isA ? p.N : p.Kc) ? (uint32_t)col * (uint32_t)sizeof(T) : kOOB;
    lm[j] = m_begin + 4 * (g & 7) + (lane >> 4);
    limg[j] = loy[j] = lox[j] = 0;
    if (CONV && !isA) {
      const int hw = p.Ho * p.Wo;
      limg[j] = lm[j] / hw;
      const int rem = lm[j] - limg[j] * hw;
      loy[j] = rem / p.Wo;
      lox[j] = rem - loy[j] * p.Wo;
    }
  }
  int ld_slot = 0;
  auto issue = [&]() __attribute__((always_inline)) {
    const uint32_t dst = lds0 + (uint32_t)ld_slot * STAGE;
    ld_slot = (ld_slot + 1) & (S - 1);
#pragma unroll
    for (int j = 0; j < DPS; ++j) {
      const int g = j * NW + wave, t = g >> 3;
      const bool isA = t < NSA;                       // wave-uniform
      const bool okm = lm[j] < m_end && colb[j] != kOOB;
      uint32_t off;
      if (isA) {
        off = (uint32_t)((size_t)lm[j] * p.lda * sizeof(T)) + colb[j];
      } else if (!CONV) {
        off = (uint32_t)((size_t)lm[j] * p.ldb * sizeof(T)) + colb[j];
      } else {
        int iy = loy[j] * p.stride - p.pad + ky, ix = lox[j] * p.stride - p.pad + kx;
        const bool in = (unsigned)iy < (unsigned)limH && (unsigned)ix < (unsigned)limW;
        iy >>= ush;
        ix >>= ush;
        off = (uint32_t)((((size_t)limg[j] * p.Hi + iy) * p.Wi + ix) * p.ldb * sizeof(T)) + colb[j];
        off = in ? off : kOOB;
        lox[j] += BKR;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const bool wr = lox[j] >= p.Wo;
          lox[j] -= wr ? p.Wo : 0;
          loy[j] += wr ? 1 : 0;
        }
        const bool wi = loy[j] >= p.Ho;
        loy[j] -= wi ? p.Ho : 0;
        limg[j] += wi ? 1 : 0;
      }
      dma16(isA ? ra : rb, okm ? off : kOOB, dst + (uint32_t)g * 1024u);
      lm[j] += BKR;
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int issued = 0;
#pragma unroll
  for (int i = 0; i < S - 1; ++i)
    if (issued < nsteps) { issue(); ++issued; }
  const int a_sub = (wn >> 1) * SUB, b_sub = (NSA + (wk >> 1)) * SUB;   // this wave's sub-tiles inside a stage
  const int wn1 = wn & 1, wk1 = wk & 1;
  for (int s = 0; s < nsteps; ++s) {
    const int younger = issued - s - 1;
    if (younger >= 2) wait_vm<2 * DPS>();
    else if (younger == 1) wait_vm<DPS>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (issued < nsteps) { issue(); ++issued; }
    const char* at = smem + (s & (S - 1)) * STAGE + a_sub;
    const char* bt = smem + (s & (S - 1)) * STAGE + b_sub;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      typename Tr<T>::v8 fa[2], fb[2];
      const int r0 = ss * 16 + 8 * lh + tq;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ca = wn1 * 8 + i * 4 + 2 * tg + (tp >> 1), cb = wk1 * 8 + i * 4 + 2 * tg + (tp >> 1);
        const typename Tr<T>::v4 alo = lds_tr_read<T>(at + tn_off(r0, ca) + 8 * (tp & 1));
        const typename Tr<T>::v4 ahi = lds_tr_read<T>(at + tn_off(r0 + 4, ca) + 8 * (tp & 1));
        const typename Tr<T>::v4 blo = lds_tr_read<T>(bt + tn_off(r0, cb) + 8 * (tp & 1));
        const typename Tr<T>::v4 bhi = lds_tr_read<T>(bt + tn_off(r0 + 4, cb) + 8 * (tp & 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) { fa[i][j] = alo[j]; fa[i][4 + j] = ahi[j]; fb[i][j] = blo[j]; fb[i][4 + j] = bhi[j]; }
      }
      DFW_TN_PRIO(1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = Tr<T>::mfma(fa[i], fb[j], acc[i][j]);
      DFW_TN_PRIO(0);
    }
  }
  // D layout: col = lane & 31 (k), row = (r & 3) + 8 * (r >> 2) + 4 * lh (n)
  float* o;
  long long ldn;
  float sc;
  bool accum;
  if (p.splits == 1) {
    o = p.out + (size_t)bb * p.ldo_b + (size_t)tap * p.ldo_t;
    ldn = p.ldo_n; sc = p.scale; accum = p.accumulate != 0;
  } else {
    o = p.slab + ((size_t)split * gridDim.z + z) * (size_t)p.N * p.Kc;
    ldn = p.Kc; sc = 1.f; accum = false;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + wk * 64 + j * 32 + (lane & 31);
      if (k >= p.Kc) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < p.N) {
          float* d = o + (size_t)n * ldn + k;
          *d = (accum ? *d : 0.f) + sc * acc[i][j][r];
        }
      }
    }
}

// out[b * ldo_b + n * ldo_n + tap * ldo_t + k] (+)= scale * sum_s slab[s][z][n][k], z = b * taps + tap
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* slab, float* out, int splits, int Z, int taps, int N,
                                                        int Kc, long long ldo_b, long long ldo_n, long long ldo_t,
                                                        float scale, int accumulate) {
  const long long per = (long long)N * Kc, total = per * Z;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int z = (int)(e / per);
    const long long r = e - (long long)z * per;
    const int n = (int)(r / Kc), k = (int)(r - (long long)n * Kc);
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += slab[(size_t)sp * total + e];
    float* o = out + (size_t)(z / taps) * ldo_b + (size_t)n * ldo_n + (size_t)(z % taps) * ldo_t + k;
    *o = (accumulate ? *o : 0.f) + scale * s;
  }
}

// ---------------------------------------------------------------------------------------------
// Column sums: part[seg][chunk][n] = sum over the chunk's rows of x[seg * rows_per_seg + row][n]; the fold over
// chunks (fixed order) writes out[seg][n].  Bias gradients (one segment) and the per-image time-embedding
// projection gradient (segment = image): d(rowbias)[img][n] = sum_pixels dY.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const char* x, float* part, int rows_per_seg, int N, int ldx, int rpc) {
  __shared__ float red[8][256];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = ((int)blockIdx.x * 32 + cl) * 8;
  const int chunk = blockIdx.y, seg = blockIdx.z;
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  if (n < N) {
    const int r1 = min(rows_per_seg, (chunk + 1) * rpc);
    for (int r = chunk * rpc + rl; r < r1; r += 8) {
      float f[8];
      unpack8<T>(*(const i32x4*)(x + (((size_t)seg * rows_per_seg + r) * ldx + n) * sizeof(T)), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += f[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cl * 8 + i] = s[i];
  __syncthreads();
  const int c = threadIdx.x;   // 256 columns of this block
  const int nn = (int)blockIdx.x * 256 + c;
  if (nn < N) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) a += red[r][c];
    part[((size_t)seg * gridDim.y + chunk) * N + nn] = a;
  }
}

// block = 16 columns x 16 lanes over the chunks (fixed order inside a lane, fixed pairwise combine): N / 16 workgroups, so
// a 320-column bias gradient folds its 512 partial rows on 20 CUs with 32 loads per thread (the 64 x 4 layout ran it on 5
// workgroups with 128 serial loads each: 12.8 us per call, 2.4 ms per training step)
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* part, float* out, int chunks, int N, int segs,
                                                          long long ldo, float scale, int accumulate) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, l16 = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + cl, seg = blockIdx.y;
  float a = 0.f;
  if (n < N)
    for (int c = l16; c < chunks; c += 16) a += part[((size_t)seg * chunks + c) * N + n];
  red[l16][cl] = a;
  __syncthreads();
  if (l16 == 0 && n < N) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = red[r][cl];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int r = 0; r < w; ++r) t[r] += t[r + w];
    float* o = out + (size_t)seg * ldo + n;
    *o = (accumulate ? *o : 0.f) + scale * t[0];
  }
}

// Every column sum of a backward walk in TWO launches (dfw_colsum_batch): a device table of items, each the arguments of one
// dfw_colsum call plus its place in the two flattened grids and its slice of the shared partial-sum workspace; a block finds
// its item by bisection.  ~190 bias / time-projection gradients per training step were 2 x 190 launches of 6-11 us each.
struct ColsumItem {
  long long x, out;                 // addresses
  long long rows_per_seg, segs, N, ldx, ldo;
  long long scale_bits, accumulate; // float bits of the scale; accumulate flag
  long long chunks, rpc, part_off;  // plan of the item; offset (floats) of its partial rows in the workspace
  long long block_begin1, block_begin2;
  long long reserved0, reserved1;
};

template <typename T>
__global__ __launch_bounds__(256) void colsum_batch_kernel(const ColsumItem* items, int n_items, float* ws) {
  __shared__ float red[8][256];
  const long long blk = blockIdx.x;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_begin1 <= blk) lo = mid; else hi = mid - 1;
  }
  const ColsumItem it = items[lo];
  const int N = (int)it.N, rows_per_seg = (int)it.rows_per_seg, ldx = (int)it.ldx, rpc = (int)it.rpc, chunks = (int)it.chunks;
  const int gx = (N + 255) / 256;
  long long l = blk - it.block_begin1;
  const int bx = (int)(l % gx);
  l /= gx;
  const int chunk = (int)(l % chunks), seg = (int)(l / chunks);
  const char* x = (const char*)it.x;
  float* part = ws + it.part_off;
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int n = (bx * 32 + cl) * 8;
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  if (n < N) {
    const int r1 = min(rows_per_seg, (chunk + 1) * rpc);
    for (int r = chunk * rpc + rl; r < r1; r += 8) {
      float f[8];
      unpack8<T>(*(const i32x4*)(x + (((size_t)seg * rows_per_seg + r) * ldx + n) * sizeof(T)), f);
#pragma unroll
      for (int i = 0; i < 8; ++i) s[i] += f[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cl * 8 + i] = s[i];
  __syncthreads();
  const int c = threadIdx.x;
  const int nn = bx * 256 + c;
  if (nn < N) {
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) a += red[r][c];
    part[((size_t)seg * chunks + chunk) * N + nn] = a;
  }
}

__global__ __launch_bounds__(256) void colsum_fold_batch_kernel(const ColsumItem* items, int n_items, const float* ws) {
  __shared__ float red[16][17];
  const long long blk = blockIdx.x;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_begin2 <= blk) lo = mid; else hi = mid - 1;
  }
  const ColsumItem it = items[lo];
  const int N = (int)it.N, chunks = (int)it.chunks;
  const int gx = (N + 15) / 16;
  const long long l = blk - it.block_begin2;
  const int bx = (int)(l % gx), seg = (int)(l / gx);
  const float* part = ws + it.part_off;
  const int cl = threadIdx.x & 15, l16 = threadIdx.x >> 4;
  const int n = bx * 16 + cl;
  float a = 0.f;
  if (n < N)
    for (int c = l16; c < chunks; c += 16) a += part[((size_t)seg * chunks + c) * N + n];
  red[l16][cl] = a;
  __syncthreads();
  if (l16 == 0 && n < N) {
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = red[r][cl];
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int r = 0; r < w; ++r) t[r] += t[r + w];
    float* o = (float*)it.out + (size_t)seg * it.ldo + n;
    const float scale = __builtin_bit_cast(float, (int)it.scale_bits);
    *o = (it.accumulate ? *o : 0.f) + scale * t[0];
  }
}

// ---------------------------------------------------------------------------------------------
// GroupNorm(+SiLU) backward on NHWC.  y = act(gamma * xhat + beta), xhat = (x - mean_g) * rstd_g.
//   dz = dy * act'(z)                      dgamma[c] = sum dz * xhat      dbeta[c] = sum dz
//   dx = rstd * (gamma * dz - mean_g(gamma * dz) - xhat * mean_g(gamma * dz * xhat))
// Pass 1: per (image, pixel chunk) per-channel (sum dz, sum dz * xhat); pass 2: fold chunks, parameter
// gradients and the two group means; pass 3: dx.  Thread = 8 fixed channels, like the forward kernels.
struct GnbP {
  const char* x; const char* dy; char* dx; const char* dxadd; const float* gamma; const float* beta; const float* mr;
  float* part;   // [B][chunks][C][2]
  float* gs;     // [B][groups][2]  (mean_g(gamma dz), mean_g(gamma dz xhat))
  float* dgamma; float* dbeta;
  int B, HW, C, groups, ldx, lddy, lddx, chunks, ppc, silu, accumulate;
  float gscale;   // parameter gradients are scaled by this (1 / loss scale)
};

__device__ __forceinline__ float silu_grad(float z) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  return sg * (1.0f + z * (1.0f - sg));
}

template <typename T>
__global__ void gn_bwd_stats_kernel(const GnbP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_b[];
  float* ls = (float*)smem_b;   // [slots][C][2]
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cpg = p.C / p.groups, tpp = p.C >> 3, slots = blockDim.x / tpp;
  const int cc = threadIdx.x % tpp, slot = threadIdx.x / tpp;
  const int p0 = chunk * p.ppc, p1 = min(p.HW, p0 + p.ppc);
  float mean[8], rstd[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cc * 8 + i, g = c / cpg;
    mean[i] = p.mr[(b * p.groups + g) * 2];
    rstd[i] = p.mr[(b * p.groups + g) * 2 + 1];
    ga[i] = p.gamma ? p.gamma[c] : 1.f;
    be[i] = p.beta ? p.beta[c] : 0.f;
    s1[i] = 0.f;
    s2[i] = 0.f;
  }
  for (int px = p0 + slot; px < p1; px += slots) {
    float fx[8], fd[8];
    unpack8<T>(*(const i32x4*)(p.x + (((size_t)b * p.HW + px) * p.ldx + cc * 8) * sizeof(T)), fx);
    unpack8<T>(*(const i32x4*)(p.dy + (((size_t)b * p.HW + px) * p.lddy + cc * 8) * sizeof(T)), fd);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (fx[i] - mean[i]) * rstd[i];
      const float dz = p.silu ? fd[i] * silu_grad(ga[i] * xh + be[i]) : fd[i];
      s1[i] += dz;
      s2[i] += dz * xh;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ls[((size_t)slot * p.C + cc * 8 + i) * 2 + 0] = s1[i];
    ls[((size_t)slot * p.C + cc * 8 + i) * 2 + 1] = s2[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
    float a = 0.f, a2 = 0.f;
    for (int sl = 0; sl < slots; ++sl) {
      a += ls[((size_t)sl * p.C + c) * 2 + 0];
      a2 += ls[((size_t)sl * p.C + c) * 2 + 1];
    }
    float* o = p.part + (((size_t)b * p.chunks + chunk) * p.C + c) * 2;
    o[0] = a;
    o[1] = a2;
  }
}

// fold the pixel chunks per (image, channel): block = 64 channels x 4 chunk lanes, grid (C / 64, B); the sums stay in
// part[b][0][c]; gn_bwd_group_param_kernel then adds the images (B terms per channel) into the parameter gradients
__global__ __launch_bounds__(256) void gn_bwd_fold_kernel(const GnbP p) {
  __shared__ float red[4][64][2];
  const int cl = threadIdx.x & 63, lane4 = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
  float a = 0.f, a2 = 0.f;
  if (c < p.C)
    for (int ch = lane4; ch < p.chunks; ch += 4) {
      const float* o = p.part + (((size_t)b * p.chunks + ch) * p.C + c) * 2;
      a += o[0];
      a2 += o[1];
    }
  red[lane4][cl][0] = a;
  red[lane4][cl][1] = a2;
  __syncthreads();
  if (lane4 == 0 && c < p.C) {
    float* o0 = p.part + (((size_t)b * p.chunks) * p.C + c) * 2;
    o0[0] = (red[0][cl][0] + red[1][cl][0]) + (red[2][cl][0] + red[3][cl][0]);
    o0[1] = (red[0][cl][1] + red[1][cl][1]) + (red[2][cl][1] + red[3][cl][1]);
  }
}

// One launch for the two small reductions after the fold: workgroups [0, B * groups) fold a group's channels into the two
// means the apply pass needs (gs[b][g][2]); the workgroups after them add the images into the parameter gradients, 64
// channels each.  (They were two launches of 4.7 us each, 61 times per training step.)
__global__ __launch_bounds__(64) void gn_bwd_group_param_kernel(const GnbP p) {
  const int ng = p.B * p.groups;
  if ((int)blockIdx.x >= ng) {
    const int c = ((int)blockIdx.x - ng) * 64 + threadIdx.x;
    if (c >= p.C || (!p.dgamma && !p.dbeta)) return;
    float dg = 0.f, db = 0.f;
    for (int b = 0; b < p.B; ++b) {
      const float* o0 = p.part + (((size_t)b * p.chunks) * p.C + c) * 2;
      db += o0[0];
      dg += o0[1];
    }
    if (p.dgamma) p.dgamma[c] = (p.accumulate ? p.dgamma[c] : 0.f) + p.gscale * dg;
    if (p.dbeta) p.dbeta[c] = (p.accumulate ? p.dbeta[c] : 0.f) + p.gscale * db;
    return;
  }
  const int bg = blockIdx.x, b = bg / p.groups, g = bg - b * p.groups;
  const int cpg = p.C / p.groups;
  float a = 0.f, a2 = 0.f;
  for (int i = threadIdx.x; i < cpg; i += 64) {
    const int c = g * cpg + i;
    const float w = p.gamma ? p.gamma[c] : 1.f;
    const float* o = p.part + (((size_t)b * p.chunks) * p.C + c) * 2;
    a += w * o[0];
    a2 += w * o[1];
  }
  a = wave_sum(a);
  a2 = wave_sum(a2);
  if (threadIdx.x == 0) {
    const float n = (float)p.HW * cpg;
    p.gs[bg * 2 + 0] = a / n;
    p.gs[bg * 2 + 1] = a2 / n;
  }
}

template <typename T>
__global__ void gn_bwd_apply_kernel(const GnbP p) {
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cpg = p.C / p.groups, tpp = p.C >> 3, slots = blockDim.x / tpp;
  const int cc = threadIdx.x % tpp, slot = threadIdx.x / tpp;
  const int p0 = chunk * p.ppc, p1 = min(p.HW, p0 + p.ppc);
  float mean[8], rstd[8], ga[8], be[8], m1[8], m2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cc * 8 + i, g = c / cpg;
    mean[i] = p.mr[(b * p.groups + g) * 2];
    rstd[i] = p.mr[(b * p.groups + g) * 2 + 1];
    ga[i] = p.gamma ? p.gamma[c] : 1.f;
    be[i] = p.beta ? p.beta[c] : 0.f;
    m1[i] = p.gs[(b * p.groups + g) * 2];
    m2[i] = p.gs[(b * p.groups + g) * 2 + 1];
  }
  for (int px = p0 + slot; px < p1; px += slots) {
    float fx[8], fd[8], o[8];
    unpack8<T>(*(const i32x4*)(p.x + (((size_t)b * p.HW + px) * p.ldx + cc * 8) * sizeof(T)), fx);
    unpack8<T>(*(const i32x4*)(p.dy + (((size_t)b * p.HW + px) * p.lddy + cc * 8) * sizeof(T)), fd);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (fx[i] - mean[i]) * rstd[i];
      const float dz = p.silu ? fd[i] * silu_grad(ga[i] * xh + be[i]) : fd[i];
      o[i] = rstd[i] * (ga[i] * dz - m1[i] - xh * m2[i]);
    }
    if (p.dxadd) {       // the gradient x already has from its other consumer: summed here in fp32, one rounding
      float pv[8];
      unpack8<T>(*(const i32x4*)(p.dxadd + (((size_t)b * p.HW + px) * p.lddx + cc * 8) * sizeof(T)), pv);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] += pv[i];
    }
    *(i32x4*)(p.dx + (((size_t)b * p.HW + px) * p.lddx + cc * 8) * sizeof(T)) = pack8<T>(o);
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm backward: one wave per row (row, dy and the statistics in registers), rows strided over the grid;
// each wave keeps per-channel partial (dgamma, dbeta) for its rows, folded per block then by ln_bwd_fold_kernel.
struct LnbP {
  const char* x; const char* dy; char* dx; const char* dxadd; const float* gamma; float* part; float* dgamma; float* dbeta;
  int rows, C, ldx, lddy, lddx, nblocks, accumulate;
  float eps, gscale;
};

template <typename T, int MAXC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnbP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_l[];
  float* red = (float*)smem_l;   // [4 waves][C][2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = p.C >> 3;
  float pg[MAXC][8], pb[MAXC][8], gm[MAXC][8];
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pg[i][j] = 0.f;
      pb[i][j] = 0.f;
      gm[i][j] = ch < nch ? p.gamma[ch * 8 + j] : 0.f;
    }
  }
  for (int row = blockIdx.x * 4 + wave; row < p.rows; row += gridDim.x * 4) {
    float f[MAXC][8], d[MAXC][8];
    const char* xr = p.x + (size_t)row * p.ldx * sizeof(T);
    const char* dr = p.dy + (size_t)row * p.lddy * sizeof(T);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        unpack8<T>(*(const i32x4*)(xr + ch * 16), f[i]);
        unpack8<T>(*(const i32x4*)(dr + ch * 16), d[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += f[i][j];
      }
    }
    const float mean = wave_sum(s) / p.C;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float t = f[i][j] - mean; v += t * t; }
      }
    }
    const float rstd = rsqrtf(wave_sum(v) / p.C + p.eps);
    float a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (f[i][j] - mean) * rstd;
          f[i][j] = xh;
          pb[i][j] += d[i][j];
          pg[i][j] += d[i][j] * xh;
          const float gd = gm[i][j] * d[i][j];
          d[i][j] = gd;
          a1 += gd;
          a2 += gd * xh;
        }
      }
    }
    const float m1 = wave_sum(a1) / p.C, m2 = wave_sum(a2) / p.C;
    char* orow = p.dx + (size_t)row * p.lddx * sizeof(T);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (d[i][j] - m1 - f[i][j] * m2);
        if (p.dxadd) {
          float pv[8];
          unpack8<T>(*(const i32x4*)(p.dxadd + (size_t)row * p.lddx * sizeof(T) + ch * 16), pv);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += pv[j];
        }
        *(i32x4*)(orow + ch * 16) = pack8<T>(o);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[((size_t)wave * p.C + ch * 8 + j) * 2 + 0] = pg[i][j];
        red[((size_t)wave * p.C + ch * 8 + j) * 2 + 1] = pb[i][j];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < p.C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[((size_t)w * p.C + c) * 2 + 0];
      b += red[((size_t)w * p.C + c) * 2 + 1];
    }
    p.part[((size_t)blockIdx.x * p.C + c) * 2 + 0] = a;
    p.part[((size_t)blockIdx.x * p.C + c) * 2 + 1] = b;
  }
}

__global__ __launch_bounds__(256) void ln_bwd_fold_kernel(const LnbP p) {   // block = 16 channels x 16 lanes over the blocks
  // (64 channels x 4 lanes left the launch with C / 64 = 5..20 workgroups whose threads walked 64 partial rows one after
  // the other: 17 us per call, 48 calls per training step)
  __shared__ float red[16][16][2];
  const int cl = threadIdx.x & 15, ln = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a = 0.f, b = 0.f;
  if (c < p.C) {
    int k = ln;
    for (; k + 48 < p.nblocks; k += 64) {      // four independent loads in flight per lane
      const float2 v0 = *(const float2*)(p.part + ((size_t)k * p.C + c) * 2);
      const float2 v1 = *(const float2*)(p.part + ((size_t)(k + 16) * p.C + c) * 2);
      const float2 v2 = *(const float2*)(p.part + ((size_t)(k + 32) * p.C + c) * 2);
      const float2 v3 = *(const float2*)(p.part + ((size_t)(k + 48) * p.C + c) * 2);
      a += (v0.x + v1.x) + (v2.x + v3.x);
      b += (v0.y + v1.y) + (v2.y + v3.y);
    }
    for (; k < p.nblocks; k += 16) {
      const float2 v = *(const float2*)(p.part + ((size_t)k * p.C + c) * 2);
      a += v.x;
      b += v.y;
    }
  }
  red[ln][cl][0] = a;
  red[ln][cl][1] = b;
  __syncthreads();
  if (ln == 0 && c < p.C) {
    a = 0.f; b = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a += red[i][cl][0]; b += red[i][cl][1]; }
    p.dgamma[c] = (p.accumulate ? p.dgamma[c] : 0.f) + p.gscale * a;
    p.dbeta[c] = (p.accumulate ? p.dbeta[c] : 0.f) + p.gscale * b;
  }
}

// ---------------------------------------------------------------------------------------------
// GEGLU (diffusers GEGLU.forward: hidden, gate = proj(x).chunk(2); hidden * gelu(gate)) on the PACKED column
// order of packing.pack_geglu: 64-column groups of 32 value columns followed by their 32 gate columns.
__device__ __forceinline__ float gelu_grad(float g) {
  const float cdf = 0.5f * (1.0f + erf_as(g * 0.70710678118654752f));
  return cdf + g * 0.3989422804014327f * __expf(-0.5f * g * g);
}

template <typename T>
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const char* pre, char* out, long long rows, int H) {
  // thread = 8 output columns (one 16-byte piece of a 32-column value block)
  const long long per = H >> 3, total = rows * per;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long m = e / per;
    const int oc = (int)(e - m * per) * 8, grp = oc >> 5, j = oc & 31;
    float a[8], g[8], o[8];
    unpack8<T>(*(const i32x4*)(pre + ((size_t)m * 2 * H + grp * 64 + j) * sizeof(T)), a);
    unpack8<T>(*(const i32x4*)(pre + ((size_t)m * 2 * H + grp * 64 + 32 + j) * sizeof(T)), g);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = a[i] * gelu_erf(g[i]);
    *(i32x4*)(out + ((size_t)m * H + oc) * sizeof(T)) = pack8<T>(o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const char* pre, const char* dout, char* dpre, long long rows, int H) {
  const long long per = H >> 3, total = rows * per;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long m = e / per;
    const int oc = (int)(e - m * per) * 8, grp = oc >> 5, j = oc & 31;
    float a[8], g[8], d[8], da[8], dg[8];
    unpack8<T>(*(const i32x4*)(pre + ((size_t)m * 2 * H + grp * 64 + j) * sizeof(T)), a);
    unpack8<T>(*(const i32x4*)(pre + ((size_t)m * 2 * H + grp * 64 + 32 + j) * sizeof(T)), g);
    unpack8<T>(*(const i32x4*)(dout + ((size_t)m * H + oc) * sizeof(T)), d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      da[i] = d[i] * gelu_erf(g[i]);
      dg[i] = d[i] * a[i] * gelu_grad(g[i]);
    }
    *(i32x4*)(dpre + ((size_t)m * 2 * H + grp * 64 + j) * sizeof(T)) = pack8<T>(da);
    *(i32x4*)(dpre + ((size_t)m * 2 * H + grp * 64 + 32 + j) * sizeof(T)) = pack8<T>(dg);
  }
}

// ---------------------------------------------------------------------------------------------
// Element-wise data movement of the backward graph (all on 16-byte pieces).
//   mode 0: y = a + b                       (gradient of a tensor used twice: residual / skip connections)
//   mode 1: y = a[:, c0 : c0 + C]           (backward of torch.cat([h, skip], dim=1), U:1226: a has ld columns)
//   mode 2: zero-stuff  y[b][2y][2x] = a[b][y][x], 0 elsewhere     (data gradient of a stride-2 conv = stride-1
//           conv of the zero-stuffed output gradient with mirrored taps)
//   mode 3: y[b][y][x] = sum of the 2x2 block of a[b][2y..][2x..]  (backward of the nearest-2x upsample fused
//           into Upsample2D's conv)
template <typename T>
__global__ __launch_bounds__(256) void ew_kernel(const char* a, const char* b, char* y, long long rows, int C, int lda,
                                                 int c0, int H, int W, int mode) {
  const int per = C >> 3;
  const long long total = rows * per;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / per;
    const int c = (int)(e - r * per) * 8;
    if (mode == 0) {
      float fa[8], fb[8];
      unpack8<T>(*(const i32x4*)(a + ((size_t)r * C + c) * sizeof(T)), fa);
      unpack8<T>(*(const i32x4*)(b + ((size_t)r * C + c) * sizeof(T)), fb);
#pragma unroll
      for (int i = 0; i < 8; ++i) fa[i] += fb[i];
      *(i32x4*)(y + ((size_t)r * C + c) * sizeof(T)) = pack8<T>(fa);
    } else if (mode == 1) {
      *(i32x4*)(y + ((size_t)r * C + c) * sizeof(T)) = *(const i32x4*)(a + ((size_t)r * lda + c0 + c) * sizeof(T));
    } else if (mode == 2) {     // rows = B * 2H * 2W output pixels; H, W = input size
      const long long hw2 = 4LL * H * W;
      const long long img = r / hw2;
      const int rem = (int)(r - img * hw2), oy = rem / (2 * W), ox = rem - oy * 2 * W;
      i32x4 v = {0, 0, 0, 0};
      if (((oy | ox) & 1) == 0) v = *(const i32x4*)(a + ((((size_t)img * H + (oy >> 1)) * W + (ox >> 1)) * C + c) * sizeof(T));
      *(i32x4*)(y + ((size_t)r * C + c) * sizeof(T)) = v;
    } else {                    // rows = B * H * W output pixels; input is 2H x 2W
      const long long hw = (long long)H * W;
      const long long img = r / hw;
      const int rem = (int)(r - img * hw), oy = rem / W, ox = rem - oy * W;
      float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          float f[8];
          unpack8<T>(*(const i32x4*)(a + ((((size_t)img * 2 * H + 2 * oy + dy) * 2 * W + 2 * ox + dx) * C + c) * sizeof(T)), f);
#pragma unroll
          for (int i = 0; i < 8; ++i) s[i] += f[i];
        }
      *(i32x4*)(y + ((size_t)r * C + c) * sizeof(T)) = pack8<T>(s);
    }
  }
}

// NCHW fp32 [B][C][H][W] -> NHWC storage dtype [B][H][W][Cp] (Cp >= C, zero padded), times `scale`
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* x, T* y, int B, int C, int HW, int Cp, float scale) {
  const long long total = (long long)B * HW * Cp;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % Cp);
    const long long pix = e / Cp;
    const long long b = pix / HW;
    const int p = (int)(pix - b * HW);
    y[e] = c < C ? (T)(x[((size_t)b * C + c) * HW + p] * scale) : (T)0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// Loss: mean((pred - target)^2) over all elements (F.mse_loss(reduction='mean'), T:1384) and its gradient
// dpred = 2 (pred - target) / numel * loss_scale, written as NHWC storage dtype [B][H][W][8] (channels padded to
// 8) for the conv_out backward.  Two-level deterministic reduction of the loss value.
template <typename T>
__global__ __launch_bounds__(256) void mse_kernel(const float* pred, const float* target, T* dpred, float* dnchw, float* part,
                                                  int B, int C, int HW, float gscale) {
  __shared__ float red[4];
  const long long total = (long long)B * C * HW;
  float acc = 0.f;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const float d = pred[e] - target[e];
    acc += d * d;
    const long long bc = e / HW;
    const int p = (int)(e - bc * HW);
    const long long b = bc / C;
    const int c = (int)(bc - b * C);
    const T g = (T)(d * gscale);
    dpred[((size_t)b * HW + p) * 8 + c] = g;
    if (dnchw) dnchw[e] = (float)g;       // the same (rounded) values, NCHW fp32, for the direct data-gradient conv
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// External loss gradient (torch autograd hands d loss / d pred to the UNet's backward): dpred = (T)(g * scale) as NHWC
// [B][HW][8] and the same rounded values as NCHW fp32 -- what mse_kernel emits for the built-in loss.
template <typename T>
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* g, T* dpred, float* dnchw, int B, int C, int HW, float scale) {
  const long long total = (long long)B * C * HW;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long bc = e / HW;
    const int p = (int)(e - bc * HW);
    const long long b = bc / C;
    const int c = (int)(bc - b * C);
    const T r = (T)(g[e] * scale);
    dpred[((size_t)b * HW + p) * 8 + c] = r;
    if (dnchw) dnchw[e] = (float)r;
  }
}

// storage dtype -> fp32 with a scale (the bf16 gradient all-reduce: sum in bf16 on the wire, back to fp32 * 1/world)
template <typename T>
__global__ __launch_bounds__(256) void to_f32_kernel(const char* x, float* y, long long n8, float scale) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long long)gridDim.x * 256) {
    float f[8];
    unpack8<T>(*(const i32x4*)(x + e * 8 * sizeof(T)), f);
    *(f32x4*)(y + e * 8) = f32x4{f[0] * scale, f[1] * scale, f[2] * scale, f[3] * scale};
    *(f32x4*)(y + e * 8 + 4) = f32x4{f[4] * scale, f[5] * scale, f[6] * scale, f[7] * scale};
  }
}

__global__ __launch_bounds__(64) void fold_scalar_kernel(const float* part, float* out, int n, float scale) {
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) a += part[i];   // lane-strided, then a fixed butterfly
  a = wave_sum(a);
  if (threadIdx.x == 0) *out = a * scale;
}

// sum of squares of an fp32 vector (gradient norm for clip_grad_norm_, T:1393): partials per block
__global__ __launch_bounds__(256) void sumsq_kernel(const float* x, float* part, long long n) {
  __shared__ float red[4];
  float acc = 0.f;
  const long long n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n >> 2 : 0;     // 16-byte loads, four partial sums
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n4; e += (long long)gridDim.x * 256) {
    const f32x4 v = ((const f32x4*)x)[e];
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] += v[i] * v[i];
  }
  acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  for (long long e = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) acc += x[e] * x[e];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// AdamW (torch.optim.AdamW, T:1186-1194: decoupled weight decay) on flat fp32 master parameters, with the
// clip_grad_norm_ factor (T:1393) read from device memory: g *= min(1, max_norm / (sqrt(*sumsq) + 1e-6)).
struct AdamP {
  float* p; const float* g; float* m; float* v; const float* sumsq; void* shadow; int shadow_bf16;
  int* found_inf;
  long long n;
  float lr, beta1, beta2, eps, wd, bc1, bc2, max_norm;
};

__global__ __launch_bounds__(256) void adamw_kernel(const AdamP a) {
  float clip = 1.f;
  if (a.sumsq) {
    // overflow guard (GradScaler's skipped step under accelerate mixed_precision='fp16', T:1017 / T:1239): one inf / NaN
    // gradient makes the norm non-finite -- the whole update is then SKIPPED (master, both moments and the 16-bit shadow
    // stay untouched) and the flag is raised for the host's dynamic loss scale
    const float ssq = *a.sumsq;
    const bool bad = !(ssq == ssq) || ssq > 3.0e38f;
    if (a.found_inf && blockIdx.x == 0 && threadIdx.x == 0) *a.found_inf = bad ? 1 : 0;   // written every step: no memset needed
    if (bad) return;
  }
  if (a.sumsq && a.max_norm > 0.f) {
    const float c = a.max_norm / (sqrtf(*a.sumsq) + 1e-6f);
    clip = c < 1.f ? c : 1.f;
  }
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < a.n; e += (long long)gridDim.x * 256) {
    const float g = a.g[e] * clip;
    float w = a.p[e] * (1.f - a.lr * a.wd);
    const float m = a.beta1 * a.m[e] + (1.f - a.beta1) * g;
    const float v = a.beta2 * a.v[e] + (1.f - a.beta2) * g * g;
    a.m[e] = m;
    a.v[e] = v;
    const float denom = sqrtf(v) / sqrtf(a.bc2) + a.eps;
    w -= (a.lr / a.bc1) * (m / denom);
    a.p[e] = w;
    if (a.shadow) {        // the 16-bit copy the MFMA kernels read, refreshed in the same pass
      if (a.shadow_bf16) ((__bf16*)a.shadow)[e] = (__bf16)w;
      else ((_Float16*)a.shadow)[e] = (_Float16)w;
    }
  }
}

// y[(flip ? nb - 1 - b : b)][c][r] = x[b][r][c]   (16-bit elements; x rows at ldx / matrices at x_bs, y likewise):
// W^T of a Linear for its data-gradient GEMM (nb = 1) and, with nb = 9 and flip, the tap-mirrored channel-swapped
// conv3x3 weight  W'[ci][8 - tap][co] = W[co][tap][ci].  64 x 64 tiles through LDS, 16-byte global accesses.
__global__ __launch_bounds__(256) void relayout_kernel(const uint16_t* x, uint16_t* y, int R, int C, long long ldx,
                                                       long long ldy, long long x_bs, long long y_bs, int nb, int flip) {
  __shared__ uint16_t tile[64][72];
  const int b = blockIdx.z, yb = flip ? nb - 1 - b : b;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const uint16_t* xs = x + (size_t)b * x_bs;
  uint16_t* ys = y + (size_t)yb * y_bs;
  for (int e = threadIdx.x; e < 64 * 8; e += 256) {        // 64 rows x 8 chunks of 8 elements
    const int i = e >> 3, ch = e & 7, r = r0 + i, c = c0 + ch * 8;
    i32x4 v = {0, 0, 0, 0};
    if (r < R && c < C) v = *(const i32x4*)(xs + (size_t)r * ldx + c);
    *(i32x4*)(&tile[i][ch * 8]) = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 8; e += 256) {        // output row = input column
    const int i = e >> 3, ch = e & 7, c = c0 + i, r = r0 + ch * 8;
    if (c < C && r < R) {
      uint16_t o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = tile[ch * 8 + k][i];
      *(i32x4*)(ys + (size_t)c * ldy + r) = *(const i32x4*)o;
    }
  }
}

}  // namespace dfw

using namespace dfw;

static int grid_for(long long items, int cap = 4096) {
  long long b = (items + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// Plan of one TN GEMM: tile (nsa x 128) x (nsb x 128), ring or register-staged kernel, split of M.
struct TnPlan { int nsa, nsb, splits, mchunk, Z; bool ring, conv; };

static int tn_tile_dim(int d) {           // 256-wide tiles where they waste at most 20 % of their columns
  if (d < 256) return 1;
  const int t = (d + 255) / 256 * 256;
  return t * 4 <= d * 5 ? 2 : 1;
}

static int tn_plan(const dfw_gemm_tn_args* a, TnPlan& pl) {
  if (!a || !a->A || !a->B || !a->out) return DFW_EINVAL;
  if (a->M <= 0 || a->N <= 0 || a->Kc <= 0) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  if ((a->N % 8) || (a->Kc % 8) || (a->lda % 8) || (a->ldb % 8)) return DFW_ESHAPE;
  if (a->taps != 1 && a->taps != 9) return DFW_ESHAPE;
  if (a->taps == 9) {
    if (a->Hi <= 0 || a->Wi <= 0 || a->Ho <= 0 || a->Wo <= 0 || a->stride <= 0) return DFW_EINVAL;
    if (a->M % (a->Ho * a->Wo)) return DFW_ESHAPE;
  }
  if (a->a_elems <= 0 || a->b_elems <= 0 || a->a_elems >= (1ll << 30) || a->b_elems >= (1ll << 30)) return DFW_ERANGE;
  const int b1 = a->batch > 1 ? a->batch : 1, b2 = a->batch2 > 1 ? a->batch2 : 1;
  pl.Z = b1 * b2 * a->taps;
  pl.conv = a->taps == 9;
  pl.ring = !pl.conv || (a->Wo >= 8 && a->Ho * a->Wo >= 32);    // else: the register-staged 128 x 128 kernel (tiny maps)
  const bool big = pl.ring;
  pl.nsa = big ? tn_tile_dim(a->N) : 1;
  pl.nsb = big ? tn_tile_dim(a->Kc) : 1;
  const long long tiles = (long long)((a->N + pl.nsa * 128 - 1) / (pl.nsa * 128)) * ((a->Kc + pl.nsb * 128 - 1) / (pl.nsb * 128)) * pl.Z;
  // Split of M: every split costs a fp32 slab of N x Kc (written once, read once by the fold), every workgroup a fixed
  // start-up / store tail; few splits leave CUs idle.  Minimise  rounds * (steps / splits * t_step + t_fixed) + slab time
  // over the split count (times in us; t_step = one 32-row step of a workgroup at the occupancy the tile allows, from
  // scratch/bench_tn2.py on MI355X).
  const int nsub = pl.nsa + pl.nsb;
  const int slots = 256 * (nsub == 2 ? 2 : 1);                  // workgroups resident at once (LDS: 32 KiB x sub-tiles)
  const double t_step = pl.ring ? (nsub == 2 ? 0.8 : nsub == 3 ? 0.75 : 1.1) : 1.6, t_fixed = 4.0, slab_bw = 3.0e6;
  const int steps = (a->M + 31) / 32;
  const int smax = steps / 4 < 1 ? 1 : (steps / 4 > 64 ? 64 : steps / 4);
  double best = 1e30;
  int best_s = 1;
  for (int sp = 1; sp <= smax; ++sp) {
    const int spp = (steps + sp - 1) / sp;
    const int real = (steps + spp - 1) / spp;
    const long long rounds = (tiles * real + slots - 1) / slots;
    const double slab = real > 1 ? 2.0 * real * pl.Z * (double)a->N * a->Kc * 4.0 / slab_bw + 4.0 : 0.0;
    const double tt = rounds * (spp * t_step + t_fixed) + slab;
    if (tt < best) { best = tt; best_s = sp; }
  }
  int spp = (steps + best_s - 1) / best_s;
  if (!pl.ring) spp = (spp + 1) & ~1;                            // the register-staged kernel walks 64-row steps
  pl.mchunk = spp * 32;
  pl.splits = (a->M + pl.mchunk - 1) / pl.mchunk;
  return 0;
}

extern "C" size_t dfw_gemm_tn_workspace_bytes(const dfw_gemm_tn_args* a) {
  TnPlan pl;
  if (tn_plan(a, pl)) return 0;
  return (size_t)pl.splits * pl.Z * a->N * a->Kc * sizeof(float);
}

template <typename T, int NSA, int NSB, bool CONV>
static void launch_tn_ring(const TnP& p, dim3 grid, hipStream_t st) {
  constexpr int lds = 4 * (NSA + NSB) * 32 * 256;
  auto kfn = gemm_tn_ring_kernel<T, NSA, NSB, CONV>;
  (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(kfn, grid, dim3(256 * NSA * NSB), lds, st, p);
}

template <typename T>
static void launch_tn(const TnP& p, const TnPlan& pl, dim3 grid, hipStream_t st) {
  if (!pl.ring) { hipLaunchKernelGGL((gemm_tn_kernel<T>), grid, dim3(256), 0, st, p); return; }
  const int cfg = (pl.nsa - 1) * 2 + (pl.nsb - 1);
  if (pl.conv) {
    switch (cfg) {
      case 0: launch_tn_ring<T, 1, 1, true>(p, grid, st); break;
      case 1: launch_tn_ring<T, 1, 2, true>(p, grid, st); break;
      case 2: launch_tn_ring<T, 2, 1, true>(p, grid, st); break;
      default: launch_tn_ring<T, 2, 2, true>(p, grid, st); break;
    }
  } else {
    switch (cfg) {
      case 0: launch_tn_ring<T, 1, 1, false>(p, grid, st); break;
      case 1: launch_tn_ring<T, 1, 2, false>(p, grid, st); break;
      case 2: launch_tn_ring<T, 2, 1, false>(p, grid, st); break;
      default: launch_tn_ring<T, 2, 2, false>(p, grid, st); break;
    }
  }
}

extern "C" int dfw_gemm_tn(const dfw_gemm_tn_args* a, dfw_stream_t stream) {
  TnPlan pl;
  int rc = tn_plan(a, pl);
  if (rc) return rc;
  const int splits = pl.splits, Z = pl.Z;
  if (!a->workspace || a->workspace_bytes < (size_t)splits * Z * a->N * a->Kc * sizeof(float)) return DFW_EWORKSPACE;
  TnP p;
  p.A = (const char*)a->A; p.B = (const char*)a->B; p.slab = (float*)a->workspace;
  p.a_bytes = (uint32_t)(a->a_elems * 2); p.b_bytes = (uint32_t)(a->b_elems * 2);
  p.M = a->M; p.N = a->N; p.Kc = a->Kc; p.lda = a->lda; p.ldb = a->ldb;
  p.taps = a->taps; p.Hi = a->Hi; p.Wi = a->Wi; p.Ho = a->Ho; p.Wo = a->Wo;
  p.stride = a->stride; p.pad = a->pad; p.ups = a->ups;
  p.splits = splits; p.mchunk = pl.mchunk;
  p.batch2 = a->batch2 > 1 ? a->batch2 : 1;
  p.strideA = a->strideA; p.strideB = a->strideB; p.strideA2 = a->strideA2; p.strideB2 = a->strideB2;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(((a->N + pl.nsa * 128 - 1) / (pl.nsa * 128)) * ((a->Kc + pl.nsb * 128 - 1) / (pl.nsb * 128)), splits, Z);
  const long long total = (long long)Z * a->N * a->Kc;
  const long long ldo_n = a->ldo_n > 0 ? a->ldo_n : (long long)a->taps * a->Kc;
  const long long ldo_t = a->ldo_t > 0 ? a->ldo_t : a->Kc;
  const long long ldo_b = a->ldo_b > 0 ? a->ldo_b : (long long)a->N * a->taps * a->Kc;
  p.out = a->out; p.ldo_b = ldo_b; p.ldo_n = ldo_n; p.ldo_t = ldo_t; p.scale = a->scale; p.accumulate = a->accumulate;
  if (a->dtype == DFW_BF16) launch_tn<__bf16>(p, pl, grid, st);
  else launch_tn<_Float16>(p, pl, grid, st);
  DFW_CHECK_LAUNCH();
  if (splits == 1) return 0;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3(grid_for(total, 2048)), dim3(256), 0, st, (const float*)p.slab, a->out, splits, Z,
                     a->taps, a->N, a->Kc, ldo_b, ldo_n, ldo_t, a->scale, a->accumulate);
  DFW_CHECK_LAUNCH();
  return 0;
}

static void colsum_plan(int rows_per_seg, int& chunks, int& rpc) {
  // 64-row chunks (8 rows per thread) up to 1024 chunks per segment: a [32768][320] activation gradient becomes
  // 2 x 512 workgroups instead of 2 x 16 (the first version ran 750 launches per step at ~60 us each)
  chunks = (rows_per_seg + 63) / 64;
  if (chunks > 1024) chunks = 1024;
  if (chunks < 1) chunks = 1;
  rpc = (rows_per_seg + chunks - 1) / chunks;
  rpc = (rpc + 7) & ~7;
  chunks = (rows_per_seg + rpc - 1) / rpc;
}

extern "C" size_t dfw_colsum_workspace_bytes(int64_t rows_per_seg, int32_t segs, int32_t N) {
  if (rows_per_seg <= 0 || segs <= 0 || N <= 0) return 0;
  int chunks, rpc;
  colsum_plan((int)rows_per_seg, chunks, rpc);
  return (size_t)segs * chunks * N * sizeof(float);
}

extern "C" int dfw_colsum(const void* x, float* out, void* workspace, size_t workspace_bytes, int64_t rows_per_seg,
                          int32_t segs, int32_t N, int32_t ldx, int64_t ldo, float scale, int32_t accumulate,
                          int32_t dtype, dfw_stream_t stream) {
  if (!x || !out || !workspace || rows_per_seg <= 0 || segs <= 0 || N <= 0) return DFW_EINVAL;
  if ((N % 8) || (ldx % 8)) return DFW_ESHAPE;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  int chunks, rpc;
  colsum_plan((int)rows_per_seg, chunks, rpc);
  if (workspace_bytes < (size_t)segs * chunks * N * sizeof(float)) return DFW_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((N + 255) / 256, chunks, segs);
  if (dtype == DFW_BF16) hipLaunchKernelGGL((colsum_kernel<__bf16>), grid, dim3(256), 0, st, (const char*)x, (float*)workspace, (int)rows_per_seg, N, ldx, rpc);
  else hipLaunchKernelGGL((colsum_kernel<_Float16>), grid, dim3(256), 0, st, (const char*)x, (float*)workspace, (int)rows_per_seg, N, ldx, rpc);
  DFW_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((N + 15) / 16, segs), dim3(256), 0, st, (const float*)workspace, out,
                     chunks, N, segs, (long long)(ldo > 0 ? ldo : N), scale, accumulate);
  DFW_CHECK_LAUNCH();
  return 0;
}

// Device tables (dfw_colsum_batch, dfw_weight_relayout_batch) written through KERNEL ARGUMENTS: up to 24 records of sixteen
// int64 travel by value in the launch, so filling a table needs neither pinned host memory nor a memcpy node -- both are
// awkward inside a stream capture (hipHostMalloc and the host allocator's event queries are not permitted while capturing).
struct TableChunk { long long v[24 * 16]; };
__global__ __launch_bounds__(256) void table_write_kernel(long long* dst, const TableChunk c, int n) {
  for (int e = threadIdx.x; e < n; e += 256) dst[e] = c.v[e];
}

extern "C" int dfw_table_write(void* table, int64_t first_record, const int64_t* records, int32_t n_records, dfw_stream_t stream) {
  if (!table || !records || n_records <= 0 || n_records > 24 || first_record < 0) return DFW_EINVAL;
  TableChunk c;
  for (int i = 0; i < n_records * 16; ++i) c.v[i] = records[i];
  hipLaunchKernelGGL(table_write_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (long long*)table + first_record * 16, c, n_records * 16);
  DFW_CHECK_LAUNCH();
  return 0;
}

// HOST helper of dfw_colsum_batch: the plan of one item (chunks, rows per chunk) -- what dfw_colsum uses itself.
extern "C" int dfw_colsum_plan(int64_t rows_per_seg, int32_t* chunks, int32_t* rpc) {
  if (rows_per_seg <= 0 || !chunks || !rpc) return DFW_EINVAL;
  int c, r;
  colsum_plan((int)rows_per_seg, c, r);
  *chunks = c; *rpc = r;
  return 0;
}

extern "C" int dfw_colsum_batch(const void* items, int32_t n_items, int64_t total_blocks1, int64_t total_blocks2, void* workspace,
                                int32_t dtype, dfw_stream_t stream) {
  if (!items || !workspace || n_items <= 0 || total_blocks1 <= 0 || total_blocks2 <= 0) return DFW_EINVAL;
  if (total_blocks1 > 0x7fffffffll || total_blocks2 > 0x7fffffffll) return DFW_ERANGE;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFW_BF16) hipLaunchKernelGGL((colsum_batch_kernel<__bf16>), dim3((unsigned)total_blocks1), dim3(256), 0, st, (const ColsumItem*)items, n_items, (float*)workspace);
  else hipLaunchKernelGGL((colsum_batch_kernel<_Float16>), dim3((unsigned)total_blocks1), dim3(256), 0, st, (const ColsumItem*)items, n_items, (float*)workspace);
  DFW_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_fold_batch_kernel, dim3((unsigned)total_blocks2), dim3(256), 0, st, (const ColsumItem*)items, n_items, (const float*)workspace);
  DFW_CHECK_LAUNCH();
  return 0;
}

static int gnb_geometry(const dfw_groupnorm_bwd_args* a, int& chunks, int& ppc, int& threads, int& slots) {
  if (!a || a->B <= 0 || a->HW <= 0 || a->C <= 0 || a->groups <= 0) return DFW_EINVAL;
  if (a->C % 8 || a->C % a->groups || a->ldx % 8 || a->lddy % 8 || a->lddx % 8) return DFW_ESHAPE;
  const int tpp = a->C / 8;
  if (tpp > 1024) return DFW_ESHAPE;
  slots = 256 / tpp;
  if (slots < 1) slots = 1;
  threads = tpp * slots;
  int want = 1024 / a->B;
  if (want < 1) want = 1;
  ppc = (a->HW + want - 1) / want;
  if (ppc < slots * 4) ppc = slots * 4;
  chunks = (a->HW + ppc - 1) / ppc;
  return 0;
}

extern "C" size_t dfw_groupnorm_bwd_workspace_bytes(const dfw_groupnorm_bwd_args* a) {
  int chunks, ppc, threads, slots;
  if (gnb_geometry(a, chunks, ppc, threads, slots)) return 0;
  return ((size_t)a->B * chunks * a->C * 2 + (size_t)a->B * a->groups * 2) * sizeof(float);
}

extern "C" int dfw_groupnorm_bwd(const dfw_groupnorm_bwd_args* a, dfw_stream_t stream) {
  int chunks, ppc, threads, slots;
  int rc = gnb_geometry(a, chunks, ppc, threads, slots);
  if (rc) return rc;
  if (!a->x || !a->dy || !a->dx || !a->mean_rstd || !a->workspace) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  if (a->workspace_bytes < dfw_groupnorm_bwd_workspace_bytes(a)) return DFW_EWORKSPACE;
  const size_t lds = (size_t)slots * a->C * 2 * sizeof(float);
  if (lds > 64 * 1024) return DFW_ESHAPE;
  GnbP p;
  p.x = (const char*)a->x; p.dy = (const char*)a->dy; p.dx = (char*)a->dx; p.dxadd = (const char*)a->dx_add;
  p.gamma = a->gamma; p.beta = a->beta; p.mr = a->mean_rstd;
  p.part = (float*)a->workspace;
  p.gs = p.part + (size_t)a->B * chunks * a->C * 2;
  p.dgamma = a->dgamma; p.dbeta = a->dbeta;
  p.B = a->B; p.HW = a->HW; p.C = a->C; p.groups = a->groups; p.ldx = a->ldx; p.lddy = a->lddy; p.lddx = a->lddx;
  p.chunks = chunks; p.ppc = ppc; p.silu = a->silu; p.accumulate = a->accumulate; p.gscale = a->grad_scale;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(chunks, a->B);
  const bool bf = a->dtype == DFW_BF16;
  if (bf) hipLaunchKernelGGL((gn_bwd_stats_kernel<__bf16>), grid, dim3(threads), lds, st, p);
  else hipLaunchKernelGGL((gn_bwd_stats_kernel<_Float16>), grid, dim3(threads), lds, st, p);
  DFW_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3((a->C + 63) / 64, a->B), dim3(256), 0, st, p);
  DFW_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_bwd_group_param_kernel, dim3(a->B * a->groups + ((a->dgamma || a->dbeta) ? (a->C + 63) / 64 : 0)), dim3(64), 0, st, p);
  DFW_CHECK_LAUNCH();
  if (bf) hipLaunchKernelGGL((gn_bwd_apply_kernel<__bf16>), grid, dim3(threads), 0, st, p);
  else hipLaunchKernelGGL((gn_bwd_apply_kernel<_Float16>), grid, dim3(threads), 0, st, p);
  DFW_CHECK_LAUNCH();
  return 0;
}

static int lnb_blocks(int rows) {
  int b = (rows + 3) / 4;
  return b > 256 ? 256 : (b < 1 ? 1 : b);
}

extern "C" size_t dfw_layernorm_bwd_workspace_bytes(int32_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)lnb_blocks(rows) * C * 2 * sizeof(float);
}

extern "C" int dfw_layernorm_bwd(const dfw_layernorm_bwd_args* a, dfw_stream_t stream) {
  if (!a || !a->x || !a->dy || !a->dx || !a->gamma || !a->dgamma || !a->dbeta || !a->workspace) return DFW_EINVAL;
  if (a->rows <= 0 || a->C <= 0) return DFW_EINVAL;
  if (a->C % 8 || a->ldx % 8 || a->lddy % 8 || a->lddx % 8 || a->C > 8 * 64 * 4) return DFW_ESHAPE;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  const int nb = lnb_blocks(a->rows);
  if (a->workspace_bytes < (size_t)nb * a->C * 2 * sizeof(float)) return DFW_EWORKSPACE;
  LnbP p;
  p.x = (const char*)a->x; p.dy = (const char*)a->dy; p.dx = (char*)a->dx; p.dxadd = (const char*)a->dx_add; p.gamma = a->gamma;
  p.part = (float*)a->workspace; p.dgamma = a->dgamma; p.dbeta = a->dbeta;
  p.rows = a->rows; p.C = a->C; p.ldx = a->ldx; p.lddy = a->lddy; p.lddx = a->lddx; p.nblocks = nb;
  p.accumulate = a->accumulate; p.eps = a->eps; p.gscale = a->grad_scale;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds = (size_t)4 * a->C * 2 * sizeof(float);
  const int nch = a->C / 8;
  const bool bf = a->dtype == DFW_BF16;
  if (nch <= 64) {
    if (bf) hipLaunchKernelGGL((ln_bwd_kernel<__bf16, 1>), dim3(nb), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((ln_bwd_kernel<_Float16, 1>), dim3(nb), dim3(256), lds, st, p);
  } else if (nch <= 128) {
    if (bf) hipLaunchKernelGGL((ln_bwd_kernel<__bf16, 2>), dim3(nb), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((ln_bwd_kernel<_Float16, 2>), dim3(nb), dim3(256), lds, st, p);
  } else {
    if (bf) hipLaunchKernelGGL((ln_bwd_kernel<__bf16, 4>), dim3(nb), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((ln_bwd_kernel<_Float16, 4>), dim3(nb), dim3(256), lds, st, p);
  }
  DFW_CHECK_LAUNCH();
  hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3((a->C + 15) / 16), dim3(256), 0, st, p);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_geglu(const void* pre, const void* dout, void* out, int64_t rows, int32_t H, int32_t dtype,
                         dfw_stream_t stream) {
  if (!pre || !out || rows <= 0 || H <= 0) return DFW_EINVAL;
  if (H % 32) return DFW_ESHAPE;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(rows * (H / 8));
  const bool bf = dtype == DFW_BF16;
  if (!dout) {
    if (bf) hipLaunchKernelGGL((geglu_fwd_kernel<__bf16>), dim3(g), dim3(256), 0, st, (const char*)pre, (char*)out, (long long)rows, H);
    else hipLaunchKernelGGL((geglu_fwd_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const char*)pre, (char*)out, (long long)rows, H);
  } else {
    if (bf) hipLaunchKernelGGL((geglu_bwd_kernel<__bf16>), dim3(g), dim3(256), 0, st, (const char*)pre, (const char*)dout, (char*)out, (long long)rows, H);
    else hipLaunchKernelGGL((geglu_bwd_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const char*)pre, (const char*)dout, (char*)out, (long long)rows, H);
  }
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_elementwise(int32_t mode, const void* a, const void* b, void* y, int64_t rows, int32_t C, int32_t lda,
                               int32_t c0, int32_t H, int32_t W, int32_t dtype, dfw_stream_t stream) {
  if (!a || !y || rows <= 0 || C <= 0 || mode < 0 || mode > 3) return DFW_EINVAL;
  if (mode == 0 && !b) return DFW_EINVAL;
  if ((C % 8) || (mode == 1 && ((lda % 8) || (c0 % 8) || c0 + C > lda))) return DFW_ESHAPE;
  if ((mode == 2 || mode == 3) && (H <= 0 || W <= 0)) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for(rows * (C / 8));
  if (dtype == DFW_BF16) hipLaunchKernelGGL((ew_kernel<__bf16>), dim3(g), dim3(256), 0, st, (const char*)a, (const char*)b, (char*)y, (long long)rows, C, lda, c0, H, W, mode);
  else hipLaunchKernelGGL((ew_kernel<_Float16>), dim3(g), dim3(256), 0, st, (const char*)a, (const char*)b, (char*)y, (long long)rows, C, lda, c0, H, W, mode);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_nchw_to_nhwc(const float* x, void* y, int32_t B, int32_t C, int32_t HW, int32_t Cp, float scale,
                                int32_t dtype, dfw_stream_t stream) {
  if (!x || !y || B <= 0 || C <= 0 || HW <= 0 || Cp < C) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int g = grid_for((long long)B * HW * Cp);
  if (dtype == DFW_BF16) hipLaunchKernelGGL((nchw_to_nhwc_kernel<__bf16>), dim3(g), dim3(256), 0, st, x, (__bf16*)y, B, C, HW, Cp, scale);
  else hipLaunchKernelGGL((nchw_to_nhwc_kernel<_Float16>), dim3(g), dim3(256), 0, st, x, (_Float16*)y, B, C, HW, Cp, scale);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_mse_loss(const float* pred, const float* target, void* dpred, float* dpred_nchw, float* loss,
                            float* workspace, int32_t B, int32_t C, int32_t HW, float loss_scale, int32_t dtype,
                            dfw_stream_t stream) {
  if (!pred || !target || !dpred || !loss || !workspace || B <= 0 || C <= 0 || C > 8 || HW <= 0) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long total = (long long)B * C * HW;
  const int nb = 256;   // workspace: 256 floats
  const float gscale = 2.0f / (float)total * loss_scale;
  // channels C..7 of dpred are never written by the kernel: the caller passes a zero-initialised buffer
  if (dtype == DFW_BF16) hipLaunchKernelGGL((mse_kernel<__bf16>), dim3(nb), dim3(256), 0, st, pred, target, (__bf16*)dpred, dpred_nchw, workspace, B, C, HW, gscale);
  else hipLaunchKernelGGL((mse_kernel<_Float16>), dim3(nb), dim3(256), 0, st, pred, target, (_Float16*)dpred, dpred_nchw, workspace, B, C, HW, gscale);
  DFW_CHECK_LAUNCH();
  hipLaunchKernelGGL(fold_scalar_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, loss, nb, 1.0f / (float)total);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_loss_grad(const float* g, void* dpred, float* dpred_nchw, int32_t B, int32_t C, int32_t HW, float scale,
                             int32_t dtype, dfw_stream_t stream) {
  if (!g || !dpred || B <= 0 || C <= 0 || C > 8 || HW <= 0) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = grid_for((long long)B * C * HW);
  if (dtype == DFW_BF16) hipLaunchKernelGGL((loss_grad_kernel<__bf16>), dim3(nb), dim3(256), 0, st, g, (__bf16*)dpred, dpred_nchw, B, C, HW, scale);
  else hipLaunchKernelGGL((loss_grad_kernel<_Float16>), dim3(nb), dim3(256), 0, st, g, (_Float16*)dpred, dpred_nchw, B, C, HW, scale);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_convert_to_f32(const void* x, float* y, int64_t n, float scale, int32_t dtype, dfw_stream_t stream) {
  if (!x || !y || n <= 0) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  if (n % 8 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return DFW_ESHAPE;
  const int nb = grid_for(n / 8, 8192);
  if (dtype == DFW_BF16) hipLaunchKernelGGL((to_f32_kernel<__bf16>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const char*)x, y, (long long)(n / 8), scale);
  else hipLaunchKernelGGL((to_f32_kernel<_Float16>), dim3(nb), dim3(256), 0, (hipStream_t)stream, (const char*)x, y, (long long)(n / 8), scale);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_sumsq(const float* x, float* out, float* workspace, int64_t n, dfw_stream_t stream) {
  if (!x || !out || !workspace || n <= 0) return DFW_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = 1024;    // workspace: 1024 floats
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, st, x, workspace, (long long)n);
  DFW_CHECK_LAUNCH();
  hipLaunchKernelGGL(fold_scalar_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, out, nb, 1.0f);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_adamw(const dfw_adamw_args* a, dfw_stream_t stream) {
  if (!a || !a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || a->n <= 0 || a->step <= 0) return DFW_EINVAL;
  AdamP p;
  p.p = a->param; p.g = a->grad; p.m = a->exp_avg; p.v = a->exp_avg_sq; p.sumsq = a->grad_sumsq;
  p.shadow = a->shadow; p.shadow_bf16 = a->shadow_dtype == DFW_BF16;
  if (a->shadow && a->shadow_dtype != DFW_BF16 && a->shadow_dtype != DFW_F16) return DFW_EINVAL;
  p.n = a->n; p.lr = a->lr; p.beta1 = a->beta1; p.beta2 = a->beta2; p.eps = a->eps; p.wd = a->weight_decay;
  p.bc1 = 1.0f - powf(a->beta1, (float)a->step);
  p.bc2 = 1.0f - powf(a->beta2, (float)a->step);
  p.max_norm = a->max_grad_norm;
  p.found_inf = a->found_inf;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(a->n, 8192)), dim3(256), 0, (hipStream_t)stream, p);
  DFW_CHECK_LAUNCH();
  return 0;
}

// All re-layouts of a step in ONE launch: a device table of items (the dfw_weight_relayout arguments + the first block
// of the item in the flattened grid); a block finds its item by bisection.  ~200 items per step in training, each a
// 3-7 us launch on its own.
struct RelayoutItem { long long x, y, R, C, ldx, ldy, x_bs, y_bs, nb, flip, block_begin, reserved; };

__global__ __launch_bounds__(256) void relayout_batch_kernel(const RelayoutItem* items, int n_items) {
  __shared__ uint16_t tile[64][72];
  const long long blk = blockIdx.x;
  int lo = 0, hi = n_items - 1;
  while (lo < hi) {                                   // last item whose block_begin <= blk
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].block_begin <= blk) lo = mid; else hi = mid - 1;
  }
  const RelayoutItem it = items[lo];
  const int R = (int)it.R, C = (int)it.C, nb = (int)it.nb;
  const int gx = (C + 63) / 64, gy = (R + 63) / 64;
  long long l = blk - it.block_begin;
  const int bx = (int)(l % gx);
  l /= gx;
  const int by = (int)(l % gy), b = (int)(l / gy);
  if (b >= nb) return;
  const int yb = it.flip ? nb - 1 - b : b;
  const int r0 = by * 64, c0 = bx * 64;
  const uint16_t* xs = (const uint16_t*)it.x + (size_t)b * it.x_bs;
  uint16_t* ys = (uint16_t*)it.y + (size_t)yb * it.y_bs;
  for (int e = threadIdx.x; e < 64 * 8; e += 256) {
    const int i = e >> 3, ch = e & 7, r = r0 + i, c = c0 + ch * 8;
    i32x4 v = {0, 0, 0, 0};
    if (r < R && c < C) v = *(const i32x4*)(xs + (size_t)r * it.ldx + c);
    *(i32x4*)(&tile[i][ch * 8]) = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 8; e += 256) {
    const int i = e >> 3, ch = e & 7, c = c0 + i, r = r0 + ch * 8;
    if (c < C && r < R) {
      uint16_t o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = tile[ch * 8 + k][i];
      *(i32x4*)(ys + (size_t)c * it.ldy + r) = *(const i32x4*)o;
    }
  }
}

extern "C" int dfw_weight_relayout_batch(const void* items, int32_t n_items, int64_t total_blocks, dfw_stream_t stream) {
  if (!items || n_items <= 0 || total_blocks <= 0 || total_blocks >= (1ll << 31)) return DFW_EINVAL;
  hipLaunchKernelGGL(relayout_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const RelayoutItem*)items, n_items);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_weight_relayout(const void* x, void* y, int32_t R, int32_t C, int64_t ldx, int64_t ldy, int32_t nb,
                                   int64_t x_bs, int64_t y_bs, int32_t flip, dfw_stream_t stream) {
  if (!x || !y || R <= 0 || C <= 0 || nb <= 0) return DFW_EINVAL;
  if ((R % 8) || (C % 8) || (ldx % 8) || (ldy % 8) || (x_bs % 8) || (y_bs % 8)) return DFW_ESHAPE;
  if (((uintptr_t)x | (uintptr_t)y) & 15) return DFW_ESHAPE;
  dim3 grid((C + 63) / 64, (R + 63) / 64, nb);
  hipLaunchKernelGGL(relayout_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, R, C,
                     (long long)ldx, (long long)ldy, (long long)x_bs, (long long)y_bs, nb, flip);
  DFW_CHECK_LAUNCH();
  return 0;
}
