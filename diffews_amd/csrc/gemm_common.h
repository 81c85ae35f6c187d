// Shared between gemm.hip (128-wide tiles) and gemm_big.hip (256-row tiles): kernel parameters,
// the epilogue for 4 consecutive output channels, tile coordinates.
#pragma once
#include "common.h"

namespace dfw {

struct GemmP {
  const char* A; const char* W; char* C;
  const char* Wblk;     // optional [tap][Cin/32][N][32] copy of a conv3x3 weight (dfw_gemm_args.W_blocked)
  const float* bias; const float* rowbias; const char* residual; float* partial;
  uint32_t a_bytes, w_bytes;
  int M, N, K, lda, ldc, ldr, ldrb;
  int taps, Cin, Hi, Wi, Ho, Wo, stride, pad, ups, rows_per_img;
  float out_scale;
  float cs; int cs_n;   // output columns n < cs_n are scaled by cs instead of out_scale (0: none)
  int act, geglu, out_mode, splitk, batch;
  int res_f32;          // residual is fp32 [M][ldr floats] (fp32 residual stream)
  long long strideA, strideW, strideC;
  int nk, cpt, ntn, ntm;
  int plan_bm, plan_bn, dtype_bf16;
  float* gn_partial; int gn_groups, gn_chunks;   // fused GroupNorm partial sums of the output (optional)
  int conv_chunk_major;                          // gemm_big conv: K walk = (channel chunk, tap) instead of (tap, chunk)
  int tw, tw_log2, tpr, tpi;  // 2-D output-pixel tiles (conv): tile width, tiles per row / per image; tw == 0: linear rows
};

// Epilogue for 4 consecutive output channels n..n+3 of output row m (raw fp32 accumulators in v).
template <typename T, bool RF32 = false>   // RF32: fp32 residual (only the split-K reduce pass instantiates it)
__device__ __forceinline__ void epilogue4(const GemmP& p, char* Cb, int m, int n, float* v) {
  const bool vec = (p.N & 3) == 0;
  if (vec) {
    if (p.bias) {
      f32x4 b = *(const f32x4*)(p.bias + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += b[i];
    }
    if (p.rowbias) {
      f32x4 b = *(const f32x4*)(p.rowbias + (size_t)(m / p.rows_per_img) * p.ldrb + n);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += b[i];
    }
    if (p.residual) {
      float r[4];
      if constexpr (RF32) {
        const f32x4 t = *(const f32x4*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(float));
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = t[i];
      } else {
        unpack4<T>(*(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T)), r);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += r[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (n + i < p.N) {
        if (p.bias) v[i] += p.bias[n + i];
        if (p.rowbias) v[i] += p.rowbias[(size_t)(m / p.rows_per_img) * p.ldrb + n + i];
        if (p.residual) v[i] += RF32 ? ((const float*)p.residual)[(size_t)m * p.ldr + n + i]
                                     : to_f(((const T*)p.residual)[(size_t)m * p.ldr + n + i]);
      }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[i] *= (n < p.cs_n ? p.cs : p.out_scale);
    if (p.act == DFW_ACT_SILU) v[i] = silu_f(v[i]);
    else if (p.act == DFW_ACT_CLAMP1) v[i] = fminf(fmaxf(v[i], -1.0f), 1.0f);
  }
  if (p.out_mode == DFW_OUT_T && vec) {
    *(i32x2*)(Cb + ((size_t)m * p.ldc + n) * sizeof(T)) = pack4<T>(v);
  } else if (p.out_mode == DFW_OUT_F32 && vec) {
    f32x4 o = {v[0], v[1], v[2], v[3]};
    *(f32x4*)(Cb + ((size_t)m * p.ldc + n) * sizeof(float)) = o;
  } else if (p.out_mode == DFW_OUT_NCHW_F32) {
    const int img = m / p.rows_per_img, pix = m - img * p.rows_per_img;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (n + i < p.N) ((float*)Cb)[((size_t)img * p.N + n + i) * p.rows_per_img + pix] = v[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (n + i < p.N) {
        if (p.out_mode == DFW_OUT_T) ((T*)Cb)[(size_t)m * p.ldc + n + i] = from_f<T>(v[i]);
        else ((float*)Cb)[(size_t)m * p.ldc + n + i] = v[i];
      }
  }
}

// Epilogue of one 32-row block of a lane (N % 4 == 0): the lane owns row m and, for each of its NB
// 32-column blocks, four groups of 4 consecutive channels n = ncol0 + 32*j + 8*g.
// Every load (bias, per-image row bias, residual) is issued BEFORE the first store: on gfx9xx vmcnt
// counts stores too and retires in order, so a load issued after a store waits for that store's
// round trip -- interleaved load/store pairs made the epilogue as long as 10-12 K-steps per tile.
template <typename T, int NB, bool SLIM = false>   // SLIM: storage-dtype NHWC output, no activation
__device__ __forceinline__ void epi_block(const GemmP& p, char* Cb, int m, int img, int ncol0,
                                          const f32x16 (&a)[NB]) {
  f32x4 add[NB][4];
  i32x2 res[NB][4];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ncol0 + 32 * j + 8 * g;
      const bool ok = n < p.N;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (p.bias && ok) b = *(const f32x4*)(p.bias + n);
      if (p.rowbias && ok) {
        const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)img * p.ldrb + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] += r[e];
      }
      add[j][g] = b;
      res[j][g] = i32x2{0, 0};
      if (p.residual && ok) res[j][g] = *(const i32x2*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(T));
    }
  const int rimg = (!SLIM && p.out_mode == DFW_OUT_NCHW_F32) ? m / p.rows_per_img : 0;
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ncol0 + 32 * j + 8 * g;
      if (n >= p.N) continue;
      float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
      if (p.residual) unpack4<T>(res[j][g], r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = (a[j][4 * g + e] + add[j][g][e] + r[e]) * (n < p.cs_n ? p.cs : p.out_scale);
        if (!SLIM && p.act == DFW_ACT_SILU) v[e] = silu_f(v[e]);
        else if (!SLIM && p.act == DFW_ACT_CLAMP1) v[e] = fminf(fmaxf(v[e], -1.0f), 1.0f);
      }
      if (SLIM || p.out_mode == DFW_OUT_T) {
        *(i32x2*)(Cb + ((size_t)m * p.ldc + n) * sizeof(T)) = pack4<T>(v);
      } else if (p.out_mode == DFW_OUT_F32) {
        const f32x4 o = {v[0], v[1], v[2], v[3]};
        *(f32x4*)(Cb + ((size_t)m * p.ldc + n) * sizeof(float)) = o;
      } else {
        const int pix = m - rimg * p.rows_per_img;
#pragma unroll
        for (int e = 0; e < 4; ++e) ((float*)Cb)[((size_t)rimg * p.N + n + e) * p.rows_per_img + pix] = v[e];
      }
    }
}

// The same block with an fp32 residual [M][ldr floats] (dfw_gemm_args.residual_f32: the fp32 residual stream): the
// residual is summed in fp32 next to bias / row bias; output per out_mode (fp32 for the stream, storage dtype where the
// result only feeds the next GEMM).  Kept apart from epi_block so that the 16-bit default path compiles unchanged.
template <typename T, int NB>
__device__ __forceinline__ void epi_block_rf32(const GemmP& p, char* Cb, int m, int img, int ncol0,
                                               const f32x16 (&a)[NB]) {
  f32x4 add[NB][4];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ncol0 + 32 * j + 8 * g;
      f32x4 b = {0.f, 0.f, 0.f, 0.f};
      if (n < p.N) {
        b = *(const f32x4*)(p.residual + ((size_t)m * p.ldr + n) * sizeof(float));
        if (p.bias) {
          const f32x4 r = *(const f32x4*)(p.bias + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] += r[e];
        }
        if (p.rowbias) {
          const f32x4 r = *(const f32x4*)(p.rowbias + (size_t)img * p.ldrb + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) b[e] += r[e];
        }
      }
      add[j][g] = b;
    }
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ncol0 + 32 * j + 8 * g;
      if (n >= p.N) continue;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (a[j][4 * g + e] + add[j][g][e]) * p.out_scale;
      if (p.out_mode == DFW_OUT_F32) {
        const f32x4 o = {v[0], v[1], v[2], v[3]};
        *(f32x4*)(Cb + ((size_t)m * p.ldc + n) * sizeof(float)) = o;
      } else {
        *(i32x2*)(Cb + ((size_t)m * p.ldc + n) * sizeof(T)) = pack4<T>(v);
      }
    }
}

// GroupNorm partial sums of a wave's (rows x 64 channels) output block taken from REGISTERS: the 16x16x32 accumulator layout
// gives lane (l15 = lane & 15, l4 = lane >> 4) row l15 and channels 16 j + 4 l4 .. + 3 of every 16-row block.  Used by the
// fp32-output epilogues (the stored values ARE the fp32 values, so these are the statistics of the stored tensor).
// add(): once per 16-row block and channel block j with the 4 stored values; store(): after the last block -- sums over
// the 16 rows (lanes xor 1, 2, 4, 8), then over the quads of a channel group (cpg = 4 .. 64, a power of two), fixed order.
struct GnRegSums {
  float s[4], q[4];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j < 4; ++j) { s[j] = 0.f; q[j] = 0.f; }
  }
  __device__ __forceinline__ void add(int j, const f32x4& v) {
    s[j] += (v[0] + v[1]) + (v[2] + v[3]);
    q[j] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
  }
  // out: p.gn_partial + ((img * gn_chunks + chunk) * gn_groups) * 2; n_w0: first channel of the wave's 64
  __device__ __forceinline__ void store(const GemmP& p, float* out, int n_w0, int lane) {
    const int cpg = p.N / p.gn_groups;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        s[j] += __shfl_xor(s[j], o, 64);
        q[j] += __shfl_xor(q[j], o, 64);
      }
      if (cpg >= 8) { s[j] += __shfl_xor(s[j], 16, 64); q[j] += __shfl_xor(q[j], 16, 64); }
      if (cpg >= 16) { s[j] += __shfl_xor(s[j], 32, 64); q[j] += __shfl_xor(q[j], 32, 64); }
    }
    if (cpg >= 32) { s[0] += s[1]; q[0] += q[1]; s[2] += s[3]; q[2] += q[3]; }
    if (cpg >= 64) { s[0] += s[2]; q[0] += q[2]; }
    const int l15 = lane & 15, l4 = lane >> 4;
    if (l15 != 0) return;
    const int lstep = cpg >= 16 ? 4 : cpg / 4;          // quads (l4 values) per group inside a 16-channel block
    if (l4 % lstep) return;
    const int jstep = cpg >= 64 ? 4 : (cpg >= 32 ? 2 : 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j % jstep) continue;
      const int grp = (n_w0 + j * 16 + 4 * l4) / cpg;
      out[grp * 2] = s[j];
      out[grp * 2 + 1] = q[j];
    }
  }
};

// Tile coordinates of one output tile (uniform per workgroup).
struct TileC {
  int m0, n0;                 // first output row (linear tiles) / first output channel
  int img, oy0, ox0;          // 2-D conv tiles: image and top-left output pixel
};

int launch_gemm_big(const GemmP& p, hipStream_t st);  // gemm_big.hip
bool gemm_big_eligible(const GemmP& p, int& bm, int& bn, int& bk);
int gemm_big_gn_chunks(const GemmP& p);   // > 0: the planned big kernel can emit GroupNorm partials
int launch_gemm8(const GemmP& p, hipStream_t st, int bn);   // gemm8.hip
bool gemm8_eligible(const GemmP& p, int bn);
bool gemm8_n160_eligible(const GemmP& p);   // the 256 x 160 Linear tile (N = 320 at the UNet's 64^2 level)
int launch_conv_patch(const GemmP& p, hipStream_t st);  // conv_patch.hip
bool conv_patch_eligible(const GemmP& p, int& bm, int& bn);
int conv_patch_gn_chunks(const GemmP& p);
int launch_conv_patch8(const GemmP& p, hipStream_t st);  // conv_patch8.hip
bool conv_patch8_eligible(const GemmP& p, int& bn);
int conv_patch8_gn_chunks(const GemmP& p);

}  // namespace dfw
