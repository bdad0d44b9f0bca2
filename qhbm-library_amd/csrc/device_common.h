// device_common.h -- small gfx950 device helpers shared by the kernel translation units
// (kernels.hip: the pass kernels; observable.hip: the block-grouped Pauli-sum kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qhbm {

// One amplitude = one 64-bit VGPR pair (re, im); the packed-fp32 VOP3P sequences work on these.
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Sum over the 64 lanes of a wave (the result is wave-uniform).  Rows of 16 lanes are
// reduced with DPP (no LDS traffic, no waitcnt); the four row sums are combined
// through readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_step(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true);
  return v + __int_as_float(t);
}
__device__ __forceinline__ float wave_sum(float v) {
  v = dpp_step<0xB1>(v);   // quad_perm:[1,0,3,2]
  v = dpp_step<0x4E>(v);   // quad_perm:[2,3,0,1]
  v = dpp_step<0x141>(v);  // row_half_mirror
  v = dpp_step<0x140>(v);  // row_mirror  -> every lane holds its 16-lane row sum
  const int iv = __float_as_int(v);
  return __int_as_float(__builtin_amdgcn_readlane(iv, 0)) + __int_as_float(__builtin_amdgcn_readlane(iv, 16)) +
         __int_as_float(__builtin_amdgcn_readlane(iv, 32)) + __int_as_float(__builtin_amdgcn_readlane(iv, 48));
}

// Fixed-point image of a partial expectation value (two's complement in an unsigned word).
__device__ __forceinline__ unsigned long long to_fixed(float v, float scale) {
  return static_cast<unsigned long long>(__float2ll_rn(v * scale));
}

}  // namespace qhbm
