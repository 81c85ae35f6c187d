// GroupNorm(+SiLU) and LayerNorm on NHWC / [rows][C] activations for gfx950.
// HBM-bound: 16-byte vector access, fp32 statistics, deterministic reductions (no atomics).
#include "common.h"

namespace dfw {

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics.  grid = (chunks, B); block = tpp*slots threads where tpp = C/8 threads
// cover one pixel (8 channels each) and `slots` pixels are processed per step.  Each thread keeps
// per-channel fp32 sum / sum-of-squares for its fixed 8 channels; the block then folds slots and
// the channels of each group in a fixed order and writes one (sum, sumsq) pair per group:
// part[b][chunk][g][2].
struct GnP {
  const char* x; char* y; const float* gamma; const float* beta; float* part; float* mr;
  int B, HW, C, groups, ldx, ldy, chunks, ppc;  // ppc = pixels per chunk
  float eps;
  int silu;
  int nt;   // gn_apply: non-temporal stores
  int rev;  // statistics / apply: walk images and pixel chunks back to front
};

// 8 consecutive channels of the input as floats: X = the storage type (one 16-byte load) or float (two) -- the
// fp32 residual stream (residual_dtype=torch.float32) is normalised straight from fp32, so the stream itself is
// never rounded to 16 bits; only this kernel's OUTPUT (an MFMA operand) is.
template <typename X> struct In8 {
  using raw = i32x4;
  static __device__ __forceinline__ raw ld(const char* p) { return *(const i32x4*)p; }
  static __device__ __forceinline__ raw ldnt(const char* p) { return __builtin_nontemporal_load((const i32x4*)p); }
  static __device__ __forceinline__ void unpack(const raw& r, float* f) { unpack8<X>(r, f); }
};
template <> struct In8<float> {
  struct raw { f32x4 a, b; };
  static __device__ __forceinline__ raw ld(const char* p) { return raw{*(const f32x4*)p, *(const f32x4*)(p + 16)}; }
  static __device__ __forceinline__ raw ldnt(const char* p) {
    return raw{__builtin_nontemporal_load((const f32x4*)p), __builtin_nontemporal_load((const f32x4*)(p + 16))};
  }
  static __device__ __forceinline__ void unpack(const raw& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = r.a[i]; f[4 + i] = r.b[i]; }
  }
};

template <typename T, typename X = T>
__global__ void gn_stats_kernel(const GnP p) {
  extern __shared__ __attribute__((aligned(16))) char smem_n[];
  float* ls = (float*)smem_n;  // [slots][C][2]
  const int tpp = p.C >> 3, slots = blockDim.x / tpp;
  const int cc = threadIdx.x % tpp, slot = threadIdx.x / tpp;
  // walk the tensor from its END: the producer (a conv that wrote it front to back) has just left its
  // tail in L2 / Infinity Cache, and this kernel's own output then ends at the front, where the next conv
  // starts reading
  const int b = p.rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int chunk = p.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int p0 = chunk * p.ppc, p1 = min(p.HW, p0 + p.ppc);
  float s[8], ss[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.f; ss[i] = 0.f; }
  const char* xb = p.x + ((size_t)b * p.HW * p.ldx + cc * 8) * sizeof(X);
  int px = p0 + slot;
  for (; px + 3 * slots < p1; px += 4 * slots) {   // four loads in flight (same order of accumulation)
    typename In8<X>::raw raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = In8<X>::ld(xb + (size_t)(px + u * slots) * p.ldx * sizeof(X));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      In8<X>::unpack(raw[u], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[i] += f[i]; ss[i] += f[i] * f[i]; }
    }
  }
  for (; px < p1; px += slots) {
    float f[8];
    In8<X>::unpack(In8<X>::ld(xb + (size_t)px * p.ldx * sizeof(X)), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] += f[i]; ss[i] += f[i] * f[i]; }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    ls[((size_t)slot * p.C + cc * 8 + i) * 2 + 0] = s[i];
    ls[((size_t)slot * p.C + cc * 8 + i) * 2 + 1] = ss[i];
  }
  __syncthreads();
  const int cpg = p.C / p.groups;
  for (int g = threadIdx.x; g < p.groups; g += blockDim.x) {
    float a = 0.f, a2 = 0.f;
    for (int sl = 0; sl < slots; ++sl)
      for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        a += ls[((size_t)sl * p.C + c) * 2 + 0];
        a2 += ls[((size_t)sl * p.C + c) * 2 + 1];
      }
    float* o = p.part + (((size_t)b * p.chunks + chunk) * p.groups + g) * 2;
    o[0] = a;
    o[1] = a2;
  }
}

// Finalize: one wave per (image, group) folds the chunk partials in fp64 (lane-strided, then a fixed
// butterfly) into mean / rstd: mr[b][g][2].
__global__ __launch_bounds__(256) void gn_finalize_kernel(const GnP p) {
  const int lane = threadIdx.x & 63;
  const int bg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bg >= p.B * p.groups) return;
  const int b = bg / p.groups, g = bg - b * p.groups;
  double a = 0.0, a2 = 0.0;
  for (int ch = lane; ch < p.chunks; ch += 64) {
    const float* o = p.part + (((size_t)b * p.chunks + ch) * p.groups + g) * 2;
    a += (double)o[0];
    a2 += (double)o[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    a2 += __shfl_xor(a2, o, 64);
  }
  if (lane == 0) {
    const double n = (double)p.HW * (p.C / p.groups);
    const double mean = a / n;
    double var = a2 / n - mean * mean;
    if (var < 0.0) var = 0.0;
    p.mr[bg * 2 + 0] = (float)mean;
    p.mr[bg * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
  }
}

// Apply: thread = fixed 8 channels (scale/shift in registers), streams its pixel range.
template <typename T, typename X = T>
__global__ void gn_apply_kernel(const GnP p) {
  const int b = p.rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int chunk = p.rev ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
  const int cpg = p.C / p.groups;
  const int tpp = p.C >> 3, slots = blockDim.x / tpp;
  const int cc = threadIdx.x % tpp, slot = threadIdx.x / tpp;
  const int p0 = chunk * p.ppc, p1 = min(p.HW, p0 + p.ppc);
  float rs[8], rh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = cc * 8 + i, g = c / cpg;
    const float mean = p.mr[(b * p.groups + g) * 2], rstd = p.mr[(b * p.groups + g) * 2 + 1];
    const float w = p.gamma ? p.gamma[c] : 1.f, bb = p.beta ? p.beta[c] : 0.f;
    rs[i] = rstd * w;
    rh[i] = bb - mean * rs[i];
  }
  const char* xb = p.x + ((size_t)b * p.HW * p.ldx + cc * 8) * sizeof(X);
  char* yb = p.y + ((size_t)b * p.HW * p.ldy + cc * 8) * sizeof(T);
  // four independent 16-byte loads in flight per thread: one load per iteration left the kernel
  // latency-bound at ~3.5 TB/s
  int px = p0 + slot;
  for (; px + 3 * slots < p1; px += 4 * slots) {
    typename In8<X>::raw raw[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) raw[u] = In8<X>::ldnt(xb + (size_t)(px + u * slots) * p.ldx * sizeof(X));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      In8<X>::unpack(raw[u], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = f[i] * rs[i] + rh[i];
        f[i] = p.silu ? silu_f(v) : v;
      }
      if (p.nt) __builtin_nontemporal_store(pack8<T>(f), (i32x4*)(yb + (size_t)(px + u * slots) * p.ldy * sizeof(T)));
      else *(i32x4*)(yb + (size_t)(px + u * slots) * p.ldy * sizeof(T)) = pack8<T>(f);
    }
  }
  for (; px < p1; px += slots) {
    float f[8];
    In8<X>::unpack(In8<X>::ld(xb + (size_t)px * p.ldx * sizeof(X)), f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float v = f[i] * rs[i] + rh[i];
      f[i] = p.silu ? silu_f(v) : v;
    }
    *(i32x4*)(yb + (size_t)px * p.ldy * sizeof(T)) = pack8<T>(f);
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, row held in registers (C <= 8*64*MAXC), two-pass statistics.
struct LnP {
  const char* x; char* y; const float* gamma; const float* beta;
  int rows, C, ldx, ldy;
  float eps;
};

template <typename T, int MAXC, typename X = T>
__global__ __launch_bounds__(256) void ln_kernel(const LnP p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= p.rows) return;
  const int nch = p.C >> 3;
  float f[MAXC][8];
  const char* xr = p.x + (size_t)row * p.ldx * sizeof(X);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      In8<X>::unpack(In8<X>::ld(xr + ch * 8 * sizeof(X)), f[i]);
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
      for (int j = 0; j < 8; ++j) { const float d = f[i][j] - mean; v += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(v) / p.C + p.eps);
  char* yr = p.y + (size_t)row * p.ldy * sizeof(T);
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float o[8];
      const f32x4 g0 = *(const f32x4*)(p.gamma + ch * 8), g1 = *(const f32x4*)(p.gamma + ch * 8 + 4);
      const f32x4 b0 = *(const f32x4*)(p.beta + ch * 8), b1 = *(const f32x4*)(p.beta + ch * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = (f[i][j] - mean) * rstd * g0[j] + b0[j];
        o[4 + j] = (f[i][4 + j] - mean) * rstd * g1[j] + b1[j];
      }
      *(i32x4*)(yr + ch * 16) = pack8<T>(o);
    }
  }
}

}  // namespace dfw

using namespace dfw;

static int gn_geometry(const dfw_groupnorm_args* a, int& chunks, int& ppc, int& threads, int& slots) {
  if (!a) return DFW_EINVAL;
  if (a->B <= 0 || a->HW <= 0 || a->C <= 0 || a->groups <= 0) return DFW_EINVAL;
  if (a->C % 8 != 0 || a->C % a->groups != 0 || a->ldx % 8 != 0 || a->ldy % 8 != 0) return DFW_ESHAPE;
  const int tpp = a->C / 8;
  if (tpp > 1024) return DFW_ESHAPE;
  slots = 256 / tpp;
  if (slots < 1) slots = 1;
  threads = tpp * slots;
  // aim for ~2048 workgroups over the batch, at least `slots*4` pixels per chunk
  int want = 2048 / a->B;
  if (want < 1) want = 1;
  ppc = (a->HW + want - 1) / want;
  const int minp = slots * 4;
  if (ppc < minp) ppc = minp;
  chunks = (a->HW + ppc - 1) / ppc;
  return 0;
}

extern "C" size_t dfw_groupnorm_workspace_bytes(const dfw_groupnorm_args* a) {
  int chunks, ppc, threads, slots;
  if (gn_geometry(a, chunks, ppc, threads, slots)) return 0;
  return ((size_t)a->B * chunks * a->groups * 2 + (size_t)a->B * a->groups * 2) * sizeof(float);
}

extern "C" int dfw_groupnorm(const dfw_groupnorm_args* a, dfw_stream_t stream) {
  int chunks, ppc, threads, slots;
  int rc = gn_geometry(a, chunks, ppc, threads, slots);
  if (rc) return rc;
  if (!a->stats_ws || !a->x || !a->y) return DFW_EINVAL;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  if (a->stats_ws_bytes < ((size_t)a->B * chunks * a->groups * 2 + (size_t)a->B * a->groups * 2) * sizeof(float))
    return DFW_EWORKSPACE;
  GnP p;
  p.x = (const char*)a->x; p.y = (char*)a->y; p.gamma = a->gamma; p.beta = a->beta;
  p.part = (float*)a->stats_ws;
  p.mr = p.part + (size_t)a->B * chunks * a->groups * 2;
  p.B = a->B; p.HW = a->HW; p.C = a->C; p.groups = a->groups; p.ldx = a->ldx; p.ldy = a->ldy;
  p.chunks = chunks; p.ppc = ppc; p.eps = a->eps; p.silu = a->silu;
  // back-to-front walk (see gn_stats_kernel); plain stores: non-temporal ones made this kernel 15-30 % faster in isolation
  // (4.7 -> 5.5 TB/s) but the step no faster (47.2 vs 47.0 ms): the consumer conv then misses L2 / Infinity Cache
  p.rev = 1;
  p.nt = 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t lds1 = (size_t)slots * a->C * 2 * sizeof(float);
  if (lds1 > 64 * 1024) return DFW_ESHAPE;
  const dim3 gridf((a->B * a->groups + 3) / 4);
  dim3 grid(chunks, a->B);
  const bool pre = a->pre_partial != nullptr && a->pre_chunks > 0;
  if (!pre && !a->x) return DFW_EINVAL;
  if (pre) {   // statistics were fused into the conv that produced x
    p.part = const_cast<float*>(a->pre_partial);
    p.chunks = a->pre_chunks;
  }
  GnP pa = p;  // apply pass keeps its own pixel chunking
  pa.chunks = chunks;
  if (a->x_f32) {   // fp32 residual stream in, storage dtype out
    if (!pre) hipLaunchKernelGGL((gn_stats_kernel<__bf16, float>), grid, dim3(threads), lds1, st, p);   // T unused by the statistics
    DFW_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_finalize_kernel, gridf, dim3(256), 0, st, p);
    DFW_CHECK_LAUNCH();
    if (a->dtype == DFW_BF16) hipLaunchKernelGGL((gn_apply_kernel<__bf16, float>), grid, dim3(threads), 0, st, pa);
    else hipLaunchKernelGGL((gn_apply_kernel<_Float16, float>), grid, dim3(threads), 0, st, pa);
  } else if (a->dtype == DFW_BF16) {
    if (!pre) hipLaunchKernelGGL((gn_stats_kernel<__bf16>), grid, dim3(threads), lds1, st, p);
    DFW_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_finalize_kernel, gridf, dim3(256), 0, st, p);
    DFW_CHECK_LAUNCH();
    hipLaunchKernelGGL((gn_apply_kernel<__bf16>), grid, dim3(threads), 0, st, pa);
  } else {
    if (!pre) hipLaunchKernelGGL((gn_stats_kernel<_Float16>), grid, dim3(threads), lds1, st, p);
    DFW_CHECK_LAUNCH();
    hipLaunchKernelGGL(gn_finalize_kernel, gridf, dim3(256), 0, st, p);
    DFW_CHECK_LAUNCH();
    hipLaunchKernelGGL((gn_apply_kernel<_Float16>), grid, dim3(threads), 0, st, pa);
  }
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_layernorm(const dfw_layernorm_args* a, dfw_stream_t stream) {
  if (!a || !a->x || !a->y || !a->gamma || !a->beta) return DFW_EINVAL;
  if (a->rows <= 0 || a->C <= 0) return DFW_EINVAL;
  if (a->C % 8 != 0 || a->ldx % 8 != 0 || a->ldy % 8 != 0 || a->C > 8 * 64 * 4) return DFW_ESHAPE;
  if (a->dtype != DFW_BF16 && a->dtype != DFW_F16) return DFW_EINVAL;
  LnP p;
  p.x = (const char*)a->x; p.y = (char*)a->y; p.gamma = a->gamma; p.beta = a->beta;
  p.rows = a->rows; p.C = a->C; p.ldx = a->ldx; p.ldy = a->ldy; p.eps = a->eps;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((a->rows + 3) / 4);
  const int nch = a->C / 8;
  if (a->x_f32) {
    const bool bf = a->dtype == DFW_BF16;
    if (nch <= 64) { if (bf) hipLaunchKernelGGL((ln_kernel<__bf16, 1, float>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((ln_kernel<_Float16, 1, float>), grid, dim3(256), 0, st, p); }
    else if (nch <= 128) { if (bf) hipLaunchKernelGGL((ln_kernel<__bf16, 2, float>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((ln_kernel<_Float16, 2, float>), grid, dim3(256), 0, st, p); }
    else { if (bf) hipLaunchKernelGGL((ln_kernel<__bf16, 4, float>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((ln_kernel<_Float16, 4, float>), grid, dim3(256), 0, st, p); }
  } else if (a->dtype == DFW_BF16) {
    if (nch <= 64) hipLaunchKernelGGL((ln_kernel<__bf16, 1>), grid, dim3(256), 0, st, p);
    else if (nch <= 128) hipLaunchKernelGGL((ln_kernel<__bf16, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((ln_kernel<__bf16, 4>), grid, dim3(256), 0, st, p);
  } else {
    if (nch <= 64) hipLaunchKernelGGL((ln_kernel<_Float16, 1>), grid, dim3(256), 0, st, p);
    else if (nch <= 128) hipLaunchKernelGGL((ln_kernel<_Float16, 2>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((ln_kernel<_Float16, 4>), grid, dim3(256), 0, st, p);
  }
  DFW_CHECK_LAUNCH();
  return 0;
}
