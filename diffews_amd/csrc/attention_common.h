// Helpers shared by the attention kernels (forward and backward) and the TN GEMM: transposed LDS reads and
// the cross-half exchange of a wave.
#pragma once
#include "common.h"

namespace dfw {

template <typename T>
__device__ __forceinline__ typename Tr<T>::v4 lds_tr_read(const char* p);
template <>
__device__ __forceinline__ bf16x4 lds_tr_read<__bf16>(const char* p) {
  using s16x4 = short __attribute__((ext_vector_type(4)));
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
  return __builtin_bit_cast(bf16x4, r);
}
template <>
__device__ __forceinline__ f16x4 lds_tr_read<_Float16>(const char* p) {
  using s16x4 = short __attribute__((ext_vector_type(4)));
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
  return __builtin_bit_cast(f16x4, r);
}

// Exchange a value with lane^32.  v_permlane32_swap swaps vdst[32..63] with src[0..31]; fed the
// same value twice it leaves {own | low-half copy} in one register and {high-half copy | own} in
// the other, so max/sum of the two is the cross-half reduction in every lane.
// Written as inline asm: with the builtin, hipcc (ROCm 7.2) copy-propagates the second result
// away when both inputs are copies of one value (r[1] is replaced by r[0]).  The s_nop covers the
// VALU-write -> v_permlane read hazard (2 wait states), which hipcc does not pad inside asm.
__device__ __forceinline__ void half_swap(float v, float& r0, float& r1) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  r0 = a;
  r1 = b;
}
__device__ __forceinline__ float half_swap_max(float v) {
  float r0, r1;
  half_swap(v, r0, r1);
  return fmaxf(r0, r1);
}
__device__ __forceinline__ float half_swap_sum(float v) {
  float r0, r1;
  half_swap(v, r0, r1);
  return r0 + r1;
}


}  // namespace dfw
