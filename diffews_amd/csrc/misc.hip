// Small HBM-bound kernels around the MFMA path: boundary convs with tiny channel counts, row
// softmax, transpose, channel concat, timestep embedding, segmentation post-processing + metric.
#include "common.h"
#include <stdlib.h>

namespace dfw {

// ---------------------------------------------------------------------------------------------
// Direct conv, Cin <= 8, NCHW fp32 input.  thread = (pixel, block of 8 output channels).
struct CsP {
  const float* x; const float* W; const float* bias; char* y;
  int B, Cin, H, Wd, Cout, taps, ldy, out_mode;
  float in_scale, out_scale;
  int iters;   // conv_small8w: pixel groups per thread
  float* gn_partial; int gn_groups, gn_chunks;   // fused GroupNorm partial sums of the NHWC output (optional)
  long long ybs;   // NCHW fp32 output: floats between images
  const float* x1; const float* x2; int b0, b1;   // images [b0, b1) come from x1, [b1, B) from x2 (x1 == nullptr: all from x)
};

// first input plane of image b (the batch may be spread over up to three buffers)
__device__ __forceinline__ const float* cs_image(const CsP& p, int b) {
  const size_t per = (size_t)p.Cin * p.H * p.Wd;
  if (p.x1 == nullptr || b < p.b0) return p.x + (size_t)b * per;
  if (b < p.b1) return p.x1 + (size_t)(b - p.b0) * per;
  return p.x2 + (size_t)(b - p.b1) * per;
}

template <typename T>
__global__ __launch_bounds__(256) void conv_small_kernel(const CsP p) {
  __shared__ float ws[9 * 8 * 8];  // [tap][c][8 outputs]
  __shared__ float bs[8];
  const int co0 = blockIdx.y * 8;
  const int tc = p.taps * p.Cin;
  for (int e = threadIdx.x; e < tc * 8; e += 256) {
    const int o = e & 7, k = e >> 3;
    ws[e] = (co0 + o < p.Cout) ? p.W[(size_t)(co0 + o) * tc + k] : 0.f;
  }
  if (threadIdx.x < 8) bs[threadIdx.x] = (p.bias && co0 + threadIdx.x < p.Cout) ? p.bias[co0 + threadIdx.x] : 0.f;
  __syncthreads();
  const int HW = p.H * p.Wd;
  const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (long long)p.B * HW) return;
  const int b = (int)(pix / HW), rem = (int)(pix - (long long)b * HW);
  const int y = rem / p.Wd, x = rem - y * p.Wd;
  float acc[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = 0.f;
  const int pad = p.taps == 9 ? 1 : 0;
  for (int t = 0; t < p.taps; ++t) {
    const int ky = p.taps == 9 ? t / 3 : 0, kx = p.taps == 9 ? t - ky * 3 : 0;
    const int iy = y + ky - pad, ix = x + kx - pad;
    if ((unsigned)iy >= (unsigned)p.H || (unsigned)ix >= (unsigned)p.Wd) continue;
    for (int c = 0; c < p.Cin; ++c) {
      const float v = cs_image(p, b)[((size_t)c * p.H + iy) * p.Wd + ix] * p.in_scale;
      const float* w = ws + (t * p.Cin + c) * 8;
#pragma unroll
      for (int o = 0; o < 8; ++o) acc[o] += v * w[o];
    }
  }
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o] = (acc[o] + bs[o]) * p.out_scale;
  if (p.out_mode == DFW_OUT_T) {
    *(i32x4*)(p.y + ((size_t)pix * p.ldy + co0) * sizeof(T)) = pack8<T>(acc);
  } else if (p.out_mode == DFW_OUT_F32) {   // NHWC fp32 (the fp32 residual stream)
    float* o = (float*)p.y + (size_t)pix * p.ldy + co0;
    *(f32x4*)o = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *(f32x4*)(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  } else {
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (co0 + o < p.Cout) ((float*)p.y)[(size_t)b * p.ybs + (size_t)(co0 + o) * HW + rem] = acc[o];
  }
}

// Same conv, 4 adjacent output pixels x 8 output channels per thread (W % 4 == 0): each input row
// segment is loaded once as float4 + 2 halo scalars and reused by the 3 horizontal taps and the 4
// pixels, i.e. 96 FMAs per 3 load instructions instead of 8 per load.
template <typename T>
__global__ __launch_bounds__(256) void conv_small4_kernel(const CsP p) {
  __shared__ float ws[9 * 8 * 8];  // [tap][c][8 outputs]
  __shared__ float bs[8];
  const int co0 = blockIdx.y * 8;
  const int tc = p.taps * p.Cin;
  for (int e = threadIdx.x; e < tc * 8; e += 256) {
    const int o = e & 7, k = e >> 3;
    ws[e] = (co0 + o < p.Cout) ? p.W[(size_t)(co0 + o) * tc + k] : 0.f;
  }
  if (threadIdx.x < 8) bs[threadIdx.x] = (p.bias && co0 + threadIdx.x < p.Cout) ? p.bias[co0 + threadIdx.x] : 0.f;
  __syncthreads();
  const int HW = p.H * p.Wd, W4 = p.Wd >> 2;
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
  if (q >= (long long)p.B * p.H * W4) return;
  const int b = (int)(q / (p.H * W4)), rem = (int)(q - (long long)b * p.H * W4);
  const int y = rem / W4, x0 = (rem - y * W4) * 4;
  float acc[4][8];
#pragma unroll
  for (int px = 0; px < 4; ++px)
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[px][o] = 0.f;
  const int nky = p.taps == 9 ? 3 : 1, pad = p.taps == 9 ? 1 : 0;
  for (int ky = 0; ky < nky; ++ky) {
    const int iy = y + ky - pad;
    if ((unsigned)iy >= (unsigned)p.H) continue;
    for (int c = 0; c < p.Cin; ++c) {
      const float* row = cs_image(p, b) + ((size_t)c * p.H + iy) * p.Wd;
      const f32x4 mid = *(const f32x4*)(row + x0);
      float in[6];
      in[0] = (pad && x0 > 0) ? row[x0 - 1] * p.in_scale : 0.f;
      in[1] = mid[0] * p.in_scale; in[2] = mid[1] * p.in_scale; in[3] = mid[2] * p.in_scale; in[4] = mid[3] * p.in_scale;
      in[5] = (pad && x0 + 4 < p.Wd) ? row[x0 + 4] * p.in_scale : 0.f;
      if (p.taps == 9) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* w = ws + ((ky * 3 + kx) * p.Cin + c) * 8;
#pragma unroll
          for (int o = 0; o < 8; ++o) {
            const float wv = w[o];
#pragma unroll
            for (int px = 0; px < 4; ++px) acc[px][o] += in[px + kx] * wv;
          }
        }
      } else {
        const float* w = ws + c * 8;
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
          for (int px = 0; px < 4; ++px) acc[px][o] += in[px + 1] * w[o];
      }
    }
  }
  const int pix = y * p.Wd + x0;
  if (p.out_mode == DFW_OUT_T) {
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float v[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) v[o] = (acc[px][o] + bs[o]) * p.out_scale;
      *(i32x4*)(p.y + (((size_t)b * HW + pix + px) * p.ldy + co0) * sizeof(T)) = pack8<T>(v);
    }
  } else if (p.out_mode == DFW_OUT_F32) {
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      float v[8];
#pragma unroll
      for (int o = 0; o < 8; ++o) v[o] = (acc[px][o] + bs[o]) * p.out_scale;
      float* o2 = (float*)p.y + ((size_t)b * HW + pix + px) * p.ldy + co0;
      *(f32x4*)o2 = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)(o2 + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
  } else {
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (co0 + o < p.Cout) {
        f32x4 v;
#pragma unroll
        for (int px = 0; px < 4; ++px) v[px] = (acc[px][o] + bs[o]) * p.out_scale;
        *(f32x4*)((float*)p.y + (size_t)b * p.ybs + (size_t)(co0 + o) * HW + pix) = v;
      }
  }
}

// NHWC-output variant: the 16 lanes of a pixel group own the 16 channel octets of a 128-channel block,
// so every store instruction writes whole 256-byte pixel rows (the per-octet grid of conv_small4
// scattered 16-byte pieces of each row over 16 workgroups and ran ~6x off the write roofline on the
// VAE's 3 -> 128 conv_in at 512x512).  Weights of the channel block sit in LDS as [tap*cin][128].
template <typename T, int TAPS>
__global__ __launch_bounds__(256) void conv_small8w_kernel(const CsP p) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem_cs[];
  float* ws = (float*)smem_cs;            // [tc][128]
  float* bs = ws + TAPS * p.Cin * 128;    // [128]
  const int cb0 = blockIdx.y * 128;
  const int tc = TAPS * p.Cin;
  for (int e = threadIdx.x; e < tc * 128; e += 256) {
    const int o = e & 127, k = e >> 7;
    ws[e] = (cb0 + o < p.Cout) ? p.W[(size_t)(cb0 + o) * tc + k] : 0.f;
  }
  if (threadIdx.x < 128) bs[threadIdx.x] = (p.bias && cb0 + threadIdx.x < p.Cout) ? p.bias[cb0 + threadIdx.x] : 0.f;
  __syncthreads();
  constexpr int NK = TAPS == 9 ? 3 : 1, PAD = TAPS == 9 ? 1 : 0;
  const int HW = p.H * p.Wd, W8 = p.Wd >> 3;
  const int oct = threadIdx.x & 15;
  if (cb0 + oct * 8 >= p.Cout) return;
  const long long nq = (long long)p.B * p.H * W8;
  float gsum[8], gsq[8];
#pragma unroll
  for (int o = 0; o < 8; ++o) { gsum[o] = 0.f; gsq[o] = 0.f; }
  // thread = 8 adjacent pixels x 8 channels (each LDS weight read feeds 64 FMAs; with 4 pixels the
  // LDS pipe was as busy as the VALU); p.iters pixel groups per thread amortise the weight staging
  for (int it = 0; it < p.iters; ++it) {
    const long long q = ((long long)blockIdx.x * p.iters + it) * 16 + (threadIdx.x >> 4);
    if (q >= nq) return;
    const int b = (int)(q / (p.H * W8)), rem = (int)(q - (long long)b * p.H * W8);
    const int y = rem / W8, x0 = (rem - y * W8) * 8;
    const float* xim = cs_image(p, b);
    f32x2 acc[8][4];                       // [pixel][channel pair]: packed fp32 FMAs
    // (fused GroupNorm statistics: host guarantees every thread is live in every iteration)
#pragma unroll
    for (int px = 0; px < 8; ++px)
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[px][o] = f32x2{0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < NK; ++ky) {
      const int iy = y + ky - PAD;
      if ((unsigned)iy >= (unsigned)p.H) continue;
      for (int c = 0; c < p.Cin; ++c) {
        const float* row = xim + ((size_t)c * p.H + iy) * p.Wd;
        const f32x4 m0 = *(const f32x4*)(row + x0), m1 = *(const f32x4*)(row + x0 + 4);
        float in[10];
        in[0] = (PAD && x0 > 0) ? row[x0 - 1] * p.in_scale : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { in[1 + e] = m0[e] * p.in_scale; in[5 + e] = m1[e] * p.in_scale; }
        in[9] = (PAD && x0 + 8 < p.Wd) ? row[x0 + 8] * p.in_scale : 0.f;
#pragma unroll
        for (int kx = 0; kx < NK; ++kx) {
          const float* w = ws + ((ky * NK + kx) * p.Cin + c) * 128 + oct * 8;
          const f32x4 w0 = *(const f32x4*)w, w1 = *(const f32x4*)(w + 4);
          const f32x2 wp[4] = {f32x2{w0[0], w0[1]}, f32x2{w0[2], w0[3]}, f32x2{w1[0], w1[1]}, f32x2{w1[2], w1[3]}};
#pragma unroll
          for (int px = 0; px < 8; ++px) {
            const float iv = in[px + (TAPS == 9 ? kx : 1)];
            const f32x2 ivv = f32x2{iv, iv};
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[px][o] = __builtin_elementwise_fma(ivv, wp[o], acc[px][o]);
          }
        }
      }
    }
    const int pix = y * p.Wd + x0;
#pragma unroll
    for (int px = 0; px < 8; ++px) {
      float v[8];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        v[2 * o] = (acc[px][o][0] + bs[oct * 8 + 2 * o]) * p.out_scale;
        v[2 * o + 1] = (acc[px][o][1] + bs[oct * 8 + 2 * o + 1]) * p.out_scale;
      }
      if (p.out_mode == DFW_OUT_F32) {      // NHWC fp32 (the fp32 residual stream): 16 lanes write 512-byte pixel rows
        float* o2 = (float*)p.y + ((size_t)b * HW + pix + px) * p.ldy + cb0 + oct * 8;
        *(f32x4*)o2 = f32x4{v[0], v[1], v[2], v[3]};
        *(f32x4*)(o2 + 4) = f32x4{v[4], v[5], v[6], v[7]};
        continue;
      }
      const i32x4 pk = pack8<T>(v);
      *(i32x4*)(p.y + (((size_t)b * HW + pix + px) * p.ldy + cb0 + oct * 8) * sizeof(T)) = pk;
      if (p.gn_partial) {                 // statistics of the STORED (rounded) values
        float r[8];
        unpack8<T>(pk, r);
#pragma unroll
        for (int o = 0; o < 8; ++o) { gsum[o] += r[o]; gsq[o] += r[o] * r[o]; }
      }
    }
  }
  if (p.gn_partial) {
    // [pixel group 0..15][128 channels][2] in LDS, then one thread per group folds channels and pixel
    // groups in a fixed order: deterministic partial (sum, sum of squares) per (image, workgroup, group)
    float* red = bs + 128;
    const int pg = threadIdx.x >> 4;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      red[(pg * 128 + oct * 8 + o) * 2 + 0] = gsum[o];
      red[(pg * 128 + oct * 8 + o) * 2 + 1] = gsq[o];
    }
    __syncthreads();
    const int cpg = p.Cout / p.gn_groups, gpb = 128 / cpg;      // channels per group, groups per 128-channel block
    if ((int)threadIdx.x < gpb) {
      float a = 0.f, a2 = 0.f;
      for (int g2 = 0; g2 < 16; ++g2)
        for (int c = threadIdx.x * cpg; c < ((int)threadIdx.x + 1) * cpg; ++c) {
          a += red[(g2 * 128 + c) * 2 + 0];
          a2 += red[(g2 * 128 + c) * 2 + 1];
        }
      const long long per_blk = 16LL * p.iters, per_img = (long long)p.H * W8;
      const long long first = (long long)blockIdx.x * per_blk;
      const int img = (int)(first / per_img), chunk = (int)((first - (long long)img * per_img) / per_blk);
      float* o2 = p.gn_partial + (((size_t)img * p.gn_chunks + chunk) * p.gn_groups + cb0 / cpg + threadIdx.x) * 2;
      o2[0] = a;
      o2[1] = a2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Row softmax: fp32 scores -> T probabilities.  One workgroup per row.
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
  return r;
}

// Row held in registers (L <= 4096, L % 4 == 0): one read of the fp32 scores instead of three.
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_reg_kernel(const float* x, T* y, int L, float c) {
  __shared__ float red[4];
  const f32x4* xr = (const f32x4*)(x + (size_t)blockIdx.x * L);
  T* yr = y + (size_t)blockIdx.x * L;
  const int nv = L >> 2;
  f32x4 v[4];
  float m = -1e30f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = threadIdx.x + 256 * i;
    if (j < nv) {
      v[i] = xr[j];
      m = fmaxf(fmaxf(fmaxf(m, v[i][0]), fmaxf(v[i][1], v[i][2])), v[i][3]);
    }
  }
  m = block_reduce(m, true, red);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = threadIdx.x + 256 * i;
    if (j < nv) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] = __builtin_amdgcn_exp2f((v[i][e] - m) * c);
        s += v[i][e];
      }
    }
  }
  s = block_reduce(s, false, red);
  const float inv = 1.0f / s;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = threadIdx.x + 256 * i;
    if (j < nv) {
      float o[4] = {v[i][0] * inv, v[i][1] * inv, v[i][2] * inv, v[i][3] * inv};
      *(i32x2*)(yr + 4 * j) = pack4<T>(o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* x, T* y, int L, float c) {
  __shared__ float red[4];
  const float* xr = x + (size_t)blockIdx.x * L;
  T* yr = y + (size_t)blockIdx.x * L;
  float m = -1e30f;
  for (int i = threadIdx.x; i < L; i += 256) m = fmaxf(m, xr[i]);
  m = block_reduce(m, true, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < L; i += 256) s += __builtin_amdgcn_exp2f((xr[i] - m) * c);
  s = block_reduce(s, false, red);
  const float inv = 1.0f / s;
  for (int i = threadIdx.x; i < L; i += 256) yr[i] = from_f<T>(__builtin_amdgcn_exp2f((xr[i] - m) * c) * inv);
}

// ---------------------------------------------------------------------------------------------
// Batched transpose of 2-byte elements through a padded LDS tile.
__global__ __launch_bounds__(256) void transpose_kernel(const uint16_t* x, uint16_t* y, int R, int C) {
  __shared__ uint16_t tile[64][66];
  const size_t base = (size_t)blockIdx.z * R * C;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    if (r < R && c < C) tile[i][tx] = x[base + (size_t)r * C + c];
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (r < R && c < C) y[base + (size_t)c * R + r] = tile[tx][i];
  }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void concat_kernel(const i32x4* a, const i32x4* b, i32x4* y, long long rows,
                                                       int cha, int chb) {
  const int nch = cha + chb;
  const long long total = rows * nch;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const long long r = e / nch;
    const int c = (int)(e - r * nch);
    y[e] = c < cha ? a[r * cha + c] : b[r * chb + (c - cha)];
  }
}

// ---------------------------------------------------------------------------------------------
// fp32 -> storage dtype, 8 elements per thread (the 16-bit MFMA-operand copy of an fp32 residual-stream tensor).
template <typename T>
__global__ __launch_bounds__(256) void convert_f32_kernel(const float* x, char* y, long long n8) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long long)gridDim.x * 256) {
    const f32x4 a = __builtin_nontemporal_load((const f32x4*)(x + e * 8)), b = __builtin_nontemporal_load((const f32x4*)(x + e * 8 + 4));
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *(i32x4*)(y + e * 8 * sizeof(T)) = pack8<T>(f);
  }
}

// Zero fill as a KERNEL (16-byte stores): scratch and gradient-seed buffers inside captured steps must not become memset
// nodes (a small memset node replayed next to plain launches received another launch's arguments).
__global__ __launch_bounds__(256) void zero_kernel(i32x4* p, long long n16) {
  const i32x4 z = {0, 0, 0, 0};
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n16; e += (long long)gridDim.x * 256) p[e] = z;
}

// fp32 -> (hi, lo) storage-dtype pair with hi = (T)x, lo = (T)(x - hi): x = hi + lo to ~2^-22 relative (fp16) -- an
// fp32 residual-stream tensor consumed AS A GEMM OPERAND (conv_shortcut, Downsample2D / Upsample2D convs) is fed as two
// 16-bit operands whose products are summed in the fp32 accumulator / fp32 residual epilogue, so the stream's 16-bit
// rounding never enters the result.
template <typename T>
__global__ __launch_bounds__(256) void split_f32_kernel(const float* x, char* hi, char* lo, long long n8) {
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n8; e += (long long)gridDim.x * 256) {
    const f32x4 a = __builtin_nontemporal_load((const f32x4*)(x + e * 8)), b = __builtin_nontemporal_load((const f32x4*)(x + e * 8 + 4));
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    float l[8];
    typename Tr<T>::v8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      h[i] = (T)f[i];
      l[i] = f[i] - (float)h[i];
    }
    *(i32x4*)(hi + e * 8 * sizeof(T)) = as_i4<T>(h);
    *(i32x4*)(lo + e * 8 * sizeof(T)) = pack8<T>(l);
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void timestep_embedding_kernel(const float* t, T* out, int B, int dim, int flip, float shift) {
  const int half = dim >> 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * half) return;
  const int b = e / half, j = e - b * half;
  const float freq = expf(-9.210340371976184f * (float)j / ((float)half - shift));
  const float arg = t[b] * freq;
  const float sn = sinf(arg), cs = cosf(arg);
  T* o = out + (size_t)b * dim;
  if (flip) { o[j] = from_f<T>(cs); o[half + j] = from_f<T>(sn); }
  else { o[j] = from_f<T>(sn); o[half + j] = from_f<T>(cs); }
}

// ---------------------------------------------------------------------------------------------
// Softmax over `L` consecutive fp32 scores per (row, group) -> T probabilities; columns >= groups*L
// of the ld-wide output row are zeroed.  Used by the folded prompt attention (L = 2 prompt tokens,
// groups = heads): the scores row is [heads*L <= ld] wide.
template <typename T>
__global__ __launch_bounds__(256) void softmax_groups_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                             long long rows, int ld, int groups, int L) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  const int per_row = ld / L;                       // slots per row, including the zeroed padding
  if (e >= rows * per_row) return;
  const long long r = e / per_row;
  const int g = (int)(e - r * per_row);
  const float* xi = x + r * ld + (long long)g * L;
  T* yo = y + r * ld + (long long)g * L;
  if (g >= groups) {
    for (int l = 0; l < L; ++l) yo[l] = (T)0.f;
    return;
  }
  float m = xi[0];
  for (int l = 1; l < L; ++l) m = fmaxf(m, xi[l]);
  float sum = 0.f;
  for (int l = 0; l < L; ++l) sum += __expf(xi[l] - m);
  const float inv = 1.0f / sum;
  for (int l = 0; l < L; ++l) yo[l] = (T)(__expf(xi[l] - m) * inv);
}

// ---------------------------------------------------------------------------------------------
// Segmentation post-processing.  Pass 1: uint8 image + per-image max.  Pass 2: threshold + counts.
__device__ __forceinline__ uint32_t seg_quant(float v) {
  v = fminf(fmaxf(v, -1.0f), 1.0f);
  v = (v * 0.5f + 0.5f) * 255.0f;
  v = fminf(fmaxf(v, 0.0f), 255.0f);
  return (uint32_t)v;  // truncation == numpy astype(uint8) on [0,255]
}

// 4 values per thread (float4 in, one 32-bit store out) when per_img % 4 == 0; scalar otherwise.
__global__ __launch_bounds__(256) void seg_u8_kernel(const float* x, uint8_t* u8, uint32_t* mx, int per_img) {
  const int b = blockIdx.y;
  uint32_t m = 0;
  const float* xb = x + (size_t)b * per_img;
  uint8_t* ub = u8 + (size_t)b * per_img;
  if ((per_img & 3) == 0 && (((uintptr_t)xb & 15) == 0) && (((uintptr_t)ub & 3) == 0)) {
    const int n4 = per_img >> 2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n4; e += gridDim.x * 256) {
      const f32x4 v = *(const f32x4*)(xb + 4 * (size_t)e);
      const uint32_t q0 = seg_quant(v[0]), q1 = seg_quant(v[1]), q2 = seg_quant(v[2]), q3 = seg_quant(v[3]);
      *(uint32_t*)(ub + 4 * (size_t)e) = q0 | (q1 << 8) | (q2 << 16) | (q3 << 24);
      m = max(max(m, q0), max(max(q1, q2), q3));
    }
  } else {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < per_img; e += gridDim.x * 256) {
      const uint32_t q = seg_quant(xb[e]);
      ub[e] = (uint8_t)q;
      m = max(m, q);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
  // one atomic per workgroup (thousands of same-address atomics serialise at ~30 ns each in L2)
  __shared__ uint32_t wmax[4];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(mx + b, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}

// pred = mean_c(u8 / 255) > thr, same fp32 expressions as the reference's to_tensor + mean
// (byte / 255.0f through a per-block table: 256 IEEE divisions per block instead of 3 per pixel).
__global__ __launch_bounds__(256) void seg_count_kernel(const uint8_t* u8, const uint8_t* gt, const uint32_t* mx,
                                                        long long* counts, int HW, float r_thr, float fixed_thr,
                                                        int batch_max) {
  __shared__ float lut[256];
  lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
  __syncthreads();
  const int b = blockIdx.y;
  // r_thr > 0: dynamic threshold max * r_threshold (main_oss.py:129-132) with the max taken over this image
  // or (batch_max) over the whole batch tensor, as `pred_mask.max()` literally does; else the fixed
  // `--threshold` (main_oss.py:134-135)
  uint32_t m = mx[b];
  if (batch_max)
    for (int i = 0; i < (int)gridDim.y; ++i) m = max(m, mx[i]);
  const float thr = r_thr > 0.f ? ((float)m / 255.0f) * r_thr : fixed_thr;
  // inter0, inter1, pred0, pred1, gt0, gt1
  unsigned c[6] = {0, 0, 0, 0, 0, 0};
  const uint8_t* ub = u8 + (size_t)b * 3 * HW;
  const uint8_t* gb = gt + (size_t)b * HW;
  auto pixel = [&](uint32_t u0, uint32_t u1, uint32_t u2, uint32_t g) {
    if (g == 255) return;  // ignore index: dropped from every histogram
    const float mean = ((lut[u0] + lut[u1]) + lut[u2]) / 3.0f;
    const int pr = mean > thr ? 1 : 0;
    c[2 + pr]++;
    c[4 + (g ? 1 : 0)]++;
    if (pr == (g ? 1 : 0)) c[pr]++;
  };
  if ((HW & 3) == 0 && (((uintptr_t)ub | (uintptr_t)gb) & 3) == 0) {
    const int n4 = HW >> 2;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n4; e += gridDim.x * 256) {
      const uint32_t w0 = *(const uint32_t*)(ub + 4 * (size_t)e), w1 = *(const uint32_t*)(ub + HW + 4 * (size_t)e);
      const uint32_t w2 = *(const uint32_t*)(ub + 2 * (size_t)HW + 4 * (size_t)e), wg = *(const uint32_t*)(gb + 4 * (size_t)e);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        pixel((w0 >> (8 * k)) & 255u, (w1 >> (8 * k)) & 255u, (w2 >> (8 * k)) & 255u, (wg >> (8 * k)) & 255u);
    }
  } else {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < HW; e += gridDim.x * 256) pixel(ub[e], ub[HW + e], ub[2 * HW + e], gb[e]);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    unsigned v = c[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
    c[k] = v;
  }
  // one set of integer atomics per workgroup
  __shared__ unsigned wc[4][6];
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) wc[threadIdx.x >> 6][k] = c[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = wc[0][k] + wc[1][k] + wc[2][k] + wc[3][k];
    // counts[b] = {inter0, inter1, union0, union1}; union = pred + gt - inter
    atomicAdd((unsigned long long*)(counts + b * 4 + 0), (unsigned long long)t[0]);
    atomicAdd((unsigned long long*)(counts + b * 4 + 1), (unsigned long long)t[1]);
    atomicAdd((unsigned long long*)(counts + b * 4 + 2), (unsigned long long)(t[2] + t[4] - t[0]));
    atomicAdd((unsigned long long*)(counts + b * 4 + 3), (unsigned long long)(t[3] + t[5] - t[1]));
  }
}

// Zeroes the per-image max words and the count cells.  A kernel of the library, not hipMemsetAsync: inside a
// captured HIP graph (ROCm 7.2) a small memset node replayed next to ordinary launches on the same stream was
// observed to deposit another launch's kernel arguments in its destination (the per-image max then read as a
// pointer and every prediction fell below the threshold).
__global__ void seg_zero_kernel(uint32_t* mx, int B, unsigned long long* counts) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < B) mx[e] = 0u;
  if (counts && e < 4 * B) counts[e] = 0ull;
}

__global__ void meter_update_kernel(const long long* counts, const long long* cls, unsigned long long* ib,
                                    unsigned long long* ub, int B, int nclass) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * 2) return;
  const int b = e >> 1, k = e & 1;
  const long long c = cls[b];
  if (c < 0 || c >= nclass) return;
  atomicAdd(ib + (size_t)k * nclass + c, (unsigned long long)counts[b * 4 + k]);
  atomicAdd(ub + (size_t)k * nclass + c, (unsigned long long)counts[b * 4 + 2 + k]);
}

}  // namespace dfw

using namespace dfw;

// pixel groups per thread of conv_small8w for this shape (>= ~2048 workgroups in flight)
static long long cs8w_iters(const dfw_conv_small_args* a) {
  const long long pix = (long long)a->B * a->H * a->Wd;
  const long long cblocks = (a->Cout + 127) / 128, groups16 = (pix / 8 + 15) / 16;
  long long iters = groups16 * cblocks / 2048;
  return iters < 1 ? 1 : (iters > 8 ? 8 : iters);
}
static bool cs8w_ok(const dfw_conv_small_args* a) {
  return a->Wd % 8 == 0 && ((uintptr_t)a->x % 16) == 0 && (a->out_mode == DFW_OUT_T || a->out_mode == DFW_OUT_F32) && a->Cout % 8 == 0 &&
         a->taps * a->Cin <= 72 && (a->taps == 9 || a->taps == 1);
}

extern "C" int32_t dfw_conv_small_gn_chunks(const dfw_conv_small_args* a) {
  if (!a || a->gn_groups <= 0 || !cs8w_ok(a) || a->out_mode != DFW_OUT_T || a->Cout % 128 != 0 || a->Cout % a->gn_groups != 0) return 0;
  const int cpg = a->Cout / a->gn_groups;
  if (128 % cpg != 0) return 0;
  const long long per_img = (long long)a->H * (a->Wd / 8), per_blk = 16 * cs8w_iters(a);
  if (per_img % per_blk != 0) return 0;          // workgroups must not straddle images (and no idle threads)
  return (int32_t)(per_img / per_blk);
}

extern "C" int dfw_conv_small(const dfw_conv_small_args* a, dfw_stream_t stream) {
  if (!a || !a->x || !a->W || !a->y) return DFW_EINVAL;
  if (a->B <= 0 || a->H <= 0 || a->Wd <= 0 || a->Cout <= 0) return DFW_EINVAL;
  if (a->Cin <= 0 || a->Cin > 8 || (a->taps != 1 && a->taps != 9)) return DFW_ESHAPE;
  if (a->out_mode != DFW_OUT_NCHW_F32 && (a->Cout % 8 != 0 || a->ldy % 8 != 0)) return DFW_ESHAPE;
  if (a->out_mode != DFW_OUT_T && a->out_mode != DFW_OUT_F32 && a->out_mode != DFW_OUT_NCHW_F32) return DFW_ESHAPE;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  CsP p;
  p.x = a->x; p.W = a->W; p.bias = a->bias; p.y = (char*)a->y;
  p.B = a->B; p.Cin = a->Cin; p.H = a->H; p.Wd = a->Wd; p.Cout = a->Cout; p.taps = a->taps;
  p.ldy = a->ldy; p.out_mode = a->out_mode; p.in_scale = a->in_scale; p.out_scale = a->out_scale;
  p.x1 = a->x1; p.x2 = a->x2; p.b0 = a->b0; p.b1 = a->b1;
  if (a->x1) {
    if (a->b0 <= 0 || a->b1 < a->b0 || a->b1 > a->B || (a->b1 < a->B && !a->x2)) return DFW_EINVAL;
    if (((uintptr_t)a->x1 % 16) != 0 || (a->x2 && ((uintptr_t)a->x2 % 16) != 0)) return DFW_ESHAPE;
  }
  p.ybs = a->y_bstride > 0 ? a->y_bstride : (long long)a->Cout * a->H * a->Wd;
  if (a->out_mode == DFW_OUT_NCHW_F32 && (p.ybs < (long long)a->Cout * a->H * a->Wd || (p.ybs & 3) || ((uintptr_t)a->y & 15)))
    return DFW_ESHAPE;
  const long long pix = (long long)a->B * a->H * a->Wd;
  hipStream_t st = (hipStream_t)stream;
  p.gn_partial = nullptr; p.gn_groups = 0; p.gn_chunks = 0;
  if (a->gn_partial) {
    p.gn_chunks = dfw_conv_small_gn_chunks(a);
    if (p.gn_chunks <= 0) return DFW_ESHAPE;     // ask dfw_conv_small_gn_chunks() first
    p.gn_partial = a->gn_partial; p.gn_groups = a->gn_groups;
  }
  if (cs8w_ok(a)) {
    const size_t lds = ((size_t)a->taps * a->Cin + 1) * 128 * sizeof(float) + (p.gn_partial ? 16 * 128 * 2 * sizeof(float) : 0);
    const long long cblocks = (a->Cout + 127) / 128, groups16 = (pix / 8 + 15) / 16;
    const long long iters = cs8w_iters(a);
    p.iters = (int)iters;
    dim3 grid((unsigned)((groups16 + iters - 1) / iters), (unsigned)cblocks);
    const bool bf = a->dtype == DFW_BF16;
    if (a->taps == 9) {
      if (bf) hipLaunchKernelGGL((conv_small8w_kernel<__bf16, 9>), grid, dim3(256), lds, st, p);
      else hipLaunchKernelGGL((conv_small8w_kernel<_Float16, 9>), grid, dim3(256), lds, st, p);
    } else {
      if (bf) hipLaunchKernelGGL((conv_small8w_kernel<__bf16, 1>), grid, dim3(256), lds, st, p);
      else hipLaunchKernelGGL((conv_small8w_kernel<_Float16, 1>), grid, dim3(256), lds, st, p);
    }
    DFW_CHECK_LAUNCH();
    return 0;
  }
  if (a->Wd % 4 == 0 && ((uintptr_t)a->x % 16) == 0) {
    dim3 grid((unsigned)((pix / 4 + 255) / 256), (a->Cout + 7) / 8);
    if (a->dtype == DFW_BF16) hipLaunchKernelGGL((conv_small4_kernel<__bf16>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_small4_kernel<_Float16>), grid, dim3(256), 0, st, p);
    DFW_CHECK_LAUNCH();
    return 0;
  }
  dim3 grid((unsigned)((pix + 255) / 256), (a->Cout + 7) / 8);
  if (a->dtype == DFW_BF16) hipLaunchKernelGGL((conv_small_kernel<__bf16>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((conv_small_kernel<_Float16>), grid, dim3(256), 0, st, p);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_softmax_rows(const float* x, void* y, int64_t rows, int32_t L, float scale, int32_t dtype,
                                dfw_stream_t stream) {
  if (!x || !y || rows <= 0 || L <= 0 || rows > 0x7fffffff) return DFW_EINVAL;
  const float c = scale * 1.4426950408889634f;
  hipStream_t st = (hipStream_t)stream;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  if (L <= 4096 && (L & 3) == 0) {
    if (dtype == DFW_BF16) hipLaunchKernelGGL((softmax_rows_reg_kernel<__bf16>), dim3((unsigned)rows), dim3(256), 0, st, x, (__bf16*)y, L, c);
    else hipLaunchKernelGGL((softmax_rows_reg_kernel<_Float16>), dim3((unsigned)rows), dim3(256), 0, st, x, (_Float16*)y, L, c);
  } else if (dtype == DFW_BF16) hipLaunchKernelGGL((softmax_rows_kernel<__bf16>), dim3((unsigned)rows), dim3(256), 0, st, x, (__bf16*)y, L, c);
  else hipLaunchKernelGGL((softmax_rows_kernel<_Float16>), dim3((unsigned)rows), dim3(256), 0, st, x, (_Float16*)y, L, c);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_softmax_groups(const float* x, void* y, int64_t rows, int32_t ld, int32_t groups, int32_t L,
                                  int32_t dtype, dfw_stream_t stream) {
  if (!x || !y || rows <= 0 || ld <= 0 || groups <= 0 || L <= 0) return DFW_EINVAL;
  if (ld % L != 0 || groups * L > ld) return DFW_ESHAPE;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  const long long total = rows * (ld / L);
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFW_BF16) hipLaunchKernelGGL((softmax_groups_kernel<__bf16>), grid, dim3(256), 0, st, x, (__bf16*)y, (long long)rows, ld, groups, L);
  else hipLaunchKernelGGL((softmax_groups_kernel<_Float16>), grid, dim3(256), 0, st, x, (_Float16*)y, (long long)rows, ld, groups, L);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_transpose(const void* x, void* y, int32_t batch, int32_t R, int32_t C, int32_t dtype,
                             dfw_stream_t stream) {
  if (!x || !y || batch <= 0 || R <= 0 || C <= 0) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  dim3 grid((C + 63) / 64, (R + 63) / 64, batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, R, C);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_concat_channels(const void* a, const void* b, void* y, int64_t rows, int32_t Ca, int32_t Cb,
                                   int32_t dtype, dfw_stream_t stream) {
  if (!a || !b || !y || rows <= 0 || Ca <= 0 || Cb <= 0) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  if (Ca % 8 != 0 || Cb % 8 != 0) return DFW_ESHAPE;
  const long long total = rows * ((Ca + Cb) / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(concat_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const i32x4*)a,
                     (const i32x4*)b, (i32x4*)y, (long long)rows, Ca / 8, Cb / 8);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_convert_f32(const float* x, void* y, int64_t n, int32_t dtype, dfw_stream_t stream) {
  if (!x || !y || n <= 0) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  if (n % 8 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return DFW_ESHAPE;
  const long long n8 = n / 8;
  long long blocks = (n8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == DFW_BF16) hipLaunchKernelGGL((convert_f32_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (char*)y, n8);
  else hipLaunchKernelGGL((convert_f32_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (char*)y, n8);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_zero(void* p, int64_t bytes, dfw_stream_t stream) {
  if (!p || bytes <= 0) return DFW_EINVAL;
  if ((bytes & 15) || ((uintptr_t)p & 15)) return DFW_ESHAPE;
  const long long n16 = bytes / 16;
  long long blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (i32x4*)p, n16);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_split_f32(const float* x, void* hi, void* lo, int64_t n, int32_t dtype, dfw_stream_t stream) {
  if (!x || !hi || !lo || n <= 0) return DFW_EINVAL;
  if (dtype != DFW_BF16 && dtype != DFW_F16) return DFW_EINVAL;
  if (n % 8 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)hi & 15) || ((uintptr_t)lo & 15)) return DFW_ESHAPE;
  const long long n8 = n / 8;
  long long blocks = (n8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (dtype == DFW_BF16) hipLaunchKernelGGL((split_f32_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (char*)hi, (char*)lo, n8);
  else hipLaunchKernelGGL((split_f32_kernel<_Float16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (char*)hi, (char*)lo, n8);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_timestep_embedding(const float* timesteps, void* out, int32_t B, int32_t dim,
                                      int32_t flip_sin_to_cos, float freq_shift, int32_t dtype,
                                      dfw_stream_t stream) {
  if (!timesteps || !out || B <= 0 || dim <= 0 || (dim & 1)) return DFW_EINVAL;
  const int n = B * (dim / 2);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == DFW_BF16) hipLaunchKernelGGL((timestep_embedding_kernel<__bf16>), dim3((n + 255) / 256), dim3(256), 0, st, timesteps, (__bf16*)out, B, dim, flip_sin_to_cos, freq_shift);
  else if (dtype == DFW_F16) hipLaunchKernelGGL((timestep_embedding_kernel<_Float16>), dim3((n + 255) / 256), dim3(256), 0, st, timesteps, (_Float16*)out, B, dim, flip_sin_to_cos, freq_shift);
  else return DFW_EINVAL;
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_seg_postprocess(const float* x, uint8_t* seg_u8, const uint8_t* gt, int64_t* counts,
                                   uint32_t* scratch, int32_t B, int32_t H, int32_t Wd, float r_threshold,
                                   dfw_stream_t stream) {
  return dfw_seg_postprocess_ex(x, seg_u8, gt, counts, scratch, B, H, Wd, r_threshold, 0.0f, 0, stream);
}

extern "C" int dfw_seg_postprocess_ex(const float* x, uint8_t* seg_u8, const uint8_t* gt, int64_t* counts,
                                      uint32_t* scratch, int32_t B, int32_t H, int32_t Wd, float r_threshold,
                                      float threshold, int32_t batch_max, dfw_stream_t stream) {
  if (!x || !seg_u8 || !scratch || B <= 0 || H <= 0 || Wd <= 0) return DFW_EINVAL;
  if (gt && !counts) return DFW_EINVAL;
  // the reference leaves the 3-channel mask un-thresholded when both flags are <= 0 and then fails its
  // shape assert (main_oss.py:137): reject that here instead of counting something undefined
  if (gt && !(r_threshold > 0.f) && !(threshold > 0.f)) return DFW_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * Wd, per_img = 3 * HW;
  hipLaunchKernelGGL(seg_zero_kernel, dim3((4 * B + 255) / 256), dim3(256), 0, st, scratch, B,
                     (unsigned long long*)(gt ? counts : nullptr));
  DFW_CHECK_LAUNCH();
  int bx = (per_img / 4 + 255) / 256;
  if (bx > 128) bx = 128;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(seg_u8_kernel, dim3(bx, B), dim3(256), 0, st, x, seg_u8, scratch, per_img);
  DFW_CHECK_LAUNCH();
  if (gt) {
    int cx = (HW / 4 + 255) / 256;
    if (cx > 64) cx = 64;
    if (cx < 1) cx = 1;
    hipLaunchKernelGGL(seg_count_kernel, dim3(cx, B), dim3(256), 0, st, (const uint8_t*)seg_u8, gt,
                       (const uint32_t*)scratch, (long long*)counts, HW, r_threshold, threshold, batch_max ? 1 : 0);
    DFW_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int dfw_meter_update(const int64_t* counts, const int64_t* class_id, int64_t* inter_buf, int64_t* union_buf,
                                int32_t B, int32_t nclass, dfw_stream_t stream) {
  if (!counts || !class_id || !inter_buf || !union_buf || B <= 0 || nclass <= 0) return DFW_EINVAL;
  hipLaunchKernelGGL(meter_update_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                     (const long long*)counts, (const long long*)class_id, (unsigned long long*)inter_buf,
                     (unsigned long long*)union_buf, B, nclass);
  DFW_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Host-side inspection of a captured hipGraph_t: number of memset nodes (and the node count), recursing into child graphs.
static int count_memset_nodes(hipGraph_t g, int32_t* total, int depth) {
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) return -1;
  if (n == 0) return 0;
  hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
  if (!nodes) return -1;
  int cnt = 0;
  if (hipGraphGetNodes(g, nodes, &n) != hipSuccess) { free(nodes); return -1; }
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) { free(nodes); return -1; }
    if (total) ++*total;
    if (t == hipGraphNodeTypeMemset) ++cnt;
    else if (t == hipGraphNodeTypeGraph && depth < 4) {
      hipGraph_t child;
      if (hipGraphChildGraphNodeGetGraph(nodes[i], &child) == hipSuccess) {
        const int c = count_memset_nodes(child, total, depth + 1);
        if (c < 0) { free(nodes); return -1; }
        cnt += c;
      }
    }
  }
  free(nodes);
  return cnt;
}

extern "C" int dfw_graph_memset_nodes(void* graph, int32_t* n_nodes) {
  if (!graph) return DFW_EINVAL;
  if (n_nodes) *n_nodes = 0;
  const int c = count_memset_nodes((hipGraph_t)graph, n_nodes, 0);
  return c < 0 ? DFW_EINVAL : c;
}

static const dfw_config kDefaultCfg = {3, 1, 0, 0, 0, 0, 0, 1, 0, 192, 3};
static dfw_config g_cfg = kDefaultCfg;
namespace dfw { const dfw_config& cfg() { return g_cfg; } }

extern "C" int dfw_configure(const dfw_config* c) {
  if (!c) { g_cfg = kDefaultCfg; return 0; }
  if (c->conv_patch < 0 || c->conv_patch > 4 || c->fsa_force_splits < 0 || c->big_min_tiles < 1 || c->k8 < 0 || c->k8 > 3) return DFW_EINVAL;
  if (c->gemm_bm && !((c->gemm_bm == 128 && (c->gemm_bn == 128 || c->gemm_bn == 64)) || (c->gemm_bm == 64 && c->gemm_bn == 64))) return DFW_EINVAL;
  g_cfg = *c;
  return 0;
}

extern "C" void dfw_get_config(dfw_config* out) {
  if (out) *out = g_cfg;
}

extern "C" int dfw_version(void) { return 103; }

extern "C" const char* dfw_error_string(int code) {
  switch (code) {
    case 0: return "success";
    case DFW_EINVAL: return "invalid argument (null pointer or non-positive size)";
    case DFW_ESHAPE: return "shape not supported by the gfx950 kernel (alignment / multiple-of constraint)";
    case DFW_ERANGE: return "tensor exceeds the 2 GiB buffer-descriptor range";
    case DFW_EWORKSPACE: return "workspace missing or too small";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
