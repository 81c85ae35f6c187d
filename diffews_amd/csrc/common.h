// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16, 16-byte vector access).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffews_hip.h"

namespace dfw {

const dfw_config& cfg();   // the record of dfw_configure() (misc.hip)

using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x2 = int __attribute__((ext_vector_type(2)));
using bf16x8 = __bf16 __attribute__((ext_vector_type(8)));
using f16x8 = _Float16 __attribute__((ext_vector_type(8)));
using bf16x4 = __bf16 __attribute__((ext_vector_type(4)));
using f16x4 = _Float16 __attribute__((ext_vector_type(4)));

template <typename T> struct Tr;
template <> struct Tr<__bf16> {
  using v8 = bf16x8;
  using v4 = bf16x4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct Tr<_Float16> {
  using v8 = f16x8;
  using v4 = f16x4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <typename T> __device__ __forceinline__ float to_f(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x) { return (T)x; }

template <typename T> __device__ __forceinline__ typename Tr<T>::v8 as_v8(i32x4 r) {
  return __builtin_bit_cast(typename Tr<T>::v8, r);
}
template <typename T> __device__ __forceinline__ i32x4 as_i4(typename Tr<T>::v8 r) {
  return __builtin_bit_cast(i32x4, r);
}

// 8 storage elements (16 B) <-> 8 floats
template <typename T> __device__ __forceinline__ void unpack8(i32x4 r, float* f) {
  typename Tr<T>::v8 v = as_v8<T>(r);
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ i32x4 pack8(const float* f) {
  typename Tr<T>::v8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (T)f[i];
  return as_i4<T>(v);
}
template <typename T> __device__ __forceinline__ i32x2 pack4(const float* f) {
  typename Tr<T>::v4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (T)f[i];
  return __builtin_bit_cast(i32x2, v);
}
template <typename T> __device__ __forceinline__ void unpack4(i32x2 r, float* f) {
  typename Tr<T>::v4 v = __builtin_bit_cast(typename Tr<T>::v4, r);
#pragma unroll
  for (int i = 0; i < 4; ++i) f[i] = (float)v[i];
}

// Bounds-checked 16-byte load through a buffer descriptor: offsets >= num_records read as zero,
// which is how conv halos, ragged M/N tiles and key padding are produced without branches.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ i32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}
constexpr uint32_t kOOB = 0x80000000u;  // always >= num_records (tensors are < 2 GiB, checked on host)

// ---- LDS-DMA from inline asm (invisible to hipcc's waitcnt bookkeeping on purpose: stages stay in
// flight across barriers and are retired by counted waits).
using u32x4 = unsigned __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 make_srd(const void* p, uint32_t bytes) {
  const uint64_t a = (uint64_t)p;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}

// One LDS-DMA wave-instruction: 64 lanes x 16 B -> LDS [m0 .. m0 + 1 KiB), out-of-range lanes
// write zeros.  M0 is written in the same statement that uses it (hipcc does not preserve it around
// asm); s_nop covers the SALU-write-M0 -> LDS-DMA hazard.
__device__ __forceinline__ void dma16(u32x4 srd, uint32_t voff, uint32_t lds_byte) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :: "s"(__builtin_amdgcn_readfirstlane(lds_byte)), "v"(voff), "s"(srd) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

// x * sigmoid(x) with the hardware reciprocal (1 ulp; the IEEE divide expands to ~10 VALU ops, which
// matters where SiLU runs inside a conv kernel).  Every SiLU in the library goes through this.
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// Exact (erf) GELU as diffusers' GEGLU uses (F.gelu default), with erf from Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7, far below the storage dtype's rounding): the library erff costs ~3x more VALU
// and made the GEGLU GEMM epilogue longer than its K loop.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float y = 1.0f - poly * __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace dfw

#define DFW_CHECK_LAUNCH()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)
