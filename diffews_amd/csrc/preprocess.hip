// Episode input transform on the GPU: what evaluation_util/data/dataset.py:36-40
// and coco.py:36-46 do on the host per image -- PIL bilinear Resize((S,S)) -> ToTensor ->
// Normalize(0.5, 0.5), and nearest resize of the binarised class mask -- as three small integer
// kernels, bit-exact with Pillow's ImagingResample (8 bits per channel) and ATen's nearest.
//
// Pillow's resize is separable with a uint8 intermediate: horizontal pass, then vertical, each
// output sample = clip8((2^21 + sum_x pixel[xmin + x] * k[x]) >> 22) with fixed-point weights
// k = int(0.5 + w * 2^22) of a triangle filter whose support scales with the reduction factor.
// The weights are computed on the host in double exactly as Resample.c does
// (dfw_resample_coeffs) and travel with the image bytes in one H2D copy; ToTensor + Normalize of a
// byte is a 256-entry table supplied by the caller (computed by torch itself -> same bits).
// HBM-bound integer/byte work: a 640x480 JPEG is 0.9 MB in, 3 MB out.
#include "common.h"
#include <math.h>

namespace dfw {

constexpr int kPrecisionBits = 32 - 8 - 2;

// horizontal: thread = (row y, output column xo), 3 channels
__global__ __launch_bounds__(256) void resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp,
                                                         const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ coef, int ksize, int H, int W,
                                                         int out_w) {
  const int xo = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (xo >= out_w) return;
  const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
  const int32_t* k = coef + (size_t)xo * ksize;
  int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
  const uint8_t* row = src + ((size_t)y * W + x0) * 3;
  for (int x = 0; x < n; ++x) {
    const int kv = k[x];
    a0 += row[3 * x] * kv;
    a1 += row[3 * x + 1] * kv;
    a2 += row[3 * x + 2] * kv;
  }
  uint8_t* o = tmp + ((size_t)y * out_w + xo) * 3;
  o[0] = (uint8_t)min(max(a0 >> kPrecisionBits, 0), 255);
  o[1] = (uint8_t)min(max(a1 >> kPrecisionBits, 0), 255);
  o[2] = (uint8_t)min(max(a2 >> kPrecisionBits, 0), 255);
}

// vertical + ToTensor/Normalize table: thread = (output row yo, output column xo), planar fp32 out
__global__ __launch_bounds__(256) void resample_v_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ dst,
                                                         const int32_t* __restrict__ bounds,
                                                         const int32_t* __restrict__ coef, int ksize, int out_h,
                                                         int out_w, const float* __restrict__ lut) {
  const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y;
  if (xo >= out_w) return;
  const int y0 = bounds[2 * yo], n = bounds[2 * yo + 1];
  const int32_t* k = coef + (size_t)yo * ksize;
  int a0 = 1 << (kPrecisionBits - 1), a1 = a0, a2 = a0;
  for (int y = 0; y < n; ++y) {
    const uint8_t* px = tmp + ((size_t)(y0 + y) * out_w + xo) * 3;
    const int kv = k[y];
    a0 += px[0] * kv;
    a1 += px[1] * kv;
    a2 += px[2] * kv;
  }
  const size_t plane = (size_t)out_h * out_w, o = (size_t)yo * out_w + xo;
  dst[o] = lut[min(max(a0 >> kPrecisionBits, 0), 255)];
  dst[plane + o] = lut[min(max(a1 >> kPrecisionBits, 0), 255)];
  dst[2 * plane + o] = lut[min(max(a2 >> kPrecisionBits, 0), 255)];
}

// class-id map -> binary (== class_value) -> nearest resize; +-1 on three planes and/or 0/1 bytes
template <typename M>
__global__ __launch_bounds__(256) void mask_nearest_kernel(const M* __restrict__ mask, int H, int W, int class_value,
                                                           float sy, float sx, int out_h, int out_w,
                                                           float* __restrict__ pm1, uint8_t* __restrict__ bin) {
  const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y;
  if (xo >= out_w) return;
  const int iy = min((int)floorf((float)yo * sy), H - 1), ix = min((int)floorf((float)xo * sx), W - 1);
  const int on = (int)mask[(size_t)iy * W + ix] == class_value;
  const size_t plane = (size_t)out_h * out_w, o = (size_t)yo * out_w + xo;
  if (bin) bin[o] = (uint8_t)on;
  if (pm1) {
    const float v = on ? 1.f : -1.f;
    pm1[o] = v;
    pm1[plane + o] = v;
    pm1[2 * plane + o] = v;
  }
}

}  // namespace dfw

using namespace dfw;

// ---- host: Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter, whole-image box
extern "C" int32_t dfw_resample_ksize(int32_t in_size, int32_t out_size) {
  if (in_size <= 0 || out_size <= 0) return 0;
  double filterscale = (double)in_size / (double)out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  return (int32_t)ceil(support) * 2 + 1;
}

extern "C" int dfw_resample_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* coeffs) {
  if (in_size <= 0 || out_size <= 0 || !bounds || !coeffs) return DFW_EINVAL;
  const double scale = (double)in_size / (double)out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 1.0 * filterscale;
  const int ksize = (int)ceil(support) * 2 + 1;
  const double ss = 1.0 / filterscale;
  double* w = (double*)malloc(sizeof(double) * ksize);
  if (!w) return DFW_EINVAL;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      w[x] = a < 1.0 ? 1.0 - a : 0.0;
      ww += w[x];
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) w[x] /= ww;
    for (int x = xmax; x < ksize; ++x) w[x] = 0.0;
    int32_t* k = coeffs + (size_t)xx * ksize;
    for (int x = 0; x < ksize; ++x)
      k[x] = w[x] < 0 ? (int32_t)(-0.5 + w[x] * (1 << kPrecisionBits)) : (int32_t)(0.5 + w[x] * (1 << kPrecisionBits));
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  free(w);
  return 0;
}

extern "C" int dfw_image_to_tensor(const dfw_image_args* a, dfw_stream_t stream) {
  if (!a || !a->src || !a->tmp || !a->dst || !a->lut) return DFW_EINVAL;
  if (!a->xbounds || !a->xcoef || !a->ybounds || !a->ycoef) return DFW_EINVAL;
  if (a->H <= 0 || a->W <= 0 || a->out_h <= 0 || a->out_w <= 0) return DFW_EINVAL;
  if (a->xk != dfw_resample_ksize(a->W, a->out_w) || a->yk != dfw_resample_ksize(a->H, a->out_h)) return DFW_ESHAPE;
  if (a->H > 65535 || a->out_h > 65535) return DFW_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 gh((a->out_w + 255) / 256, a->H), gv((a->out_w + 255) / 256, a->out_h);
  hipLaunchKernelGGL(resample_h_kernel, gh, dim3(256), 0, st, a->src, a->tmp, a->xbounds, a->xcoef, a->xk, a->H, a->W,
                     a->out_w);
  DFW_CHECK_LAUNCH();
  hipLaunchKernelGGL(resample_v_kernel, gv, dim3(256), 0, st, (const uint8_t*)a->tmp, a->dst, a->ybounds, a->ycoef,
                     a->yk, a->out_h, a->out_w, a->lut);
  DFW_CHECK_LAUNCH();
  return 0;
}

extern "C" int dfw_mask_to_tensor(const void* mask, int32_t elem_bytes, int32_t H, int32_t W, int32_t class_value,
                                  int32_t out_h, int32_t out_w, float* dst_pm1, uint8_t* dst_bin,
                                  dfw_stream_t stream) {
  if (!mask || (!dst_pm1 && !dst_bin) || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return DFW_EINVAL;
  if (elem_bytes != 1 && elem_bytes != 4) return DFW_EINVAL;
  if (out_h > 65535) return DFW_ERANGE;
  const float sy = (float)H / out_h, sx = (float)W / out_w;   // ATen compute_scales_value<float>
  hipStream_t st = (hipStream_t)stream;
  const dim3 g((out_w + 255) / 256, out_h);
  if (elem_bytes == 1)
    hipLaunchKernelGGL((mask_nearest_kernel<uint8_t>), g, dim3(256), 0, st, (const uint8_t*)mask, H, W, class_value, sy,
                       sx, out_h, out_w, dst_pm1, dst_bin);
  else
    hipLaunchKernelGGL((mask_nearest_kernel<int32_t>), g, dim3(256), 0, st, (const int32_t*)mask, H, W, class_value, sy,
                       sx, out_h, out_w, dst_pm1, dst_bin);
  DFW_CHECK_LAUNCH();
  return 0;
}
