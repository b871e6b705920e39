// Buffer-descriptor addressing for epilogues: a tensor region (one sample's channel tile) behind a descriptor, a lane's address
// one 32-bit byte offset.  An offset at or beyond the descriptor's size -- kOutside -- makes a load return zero and a store
// vanish in the hardware's bounds check, so a ragged tile runs the code of a full one: no store under a divergent branch (around
// which the compiler spills live accumulators: conv3d_wf.hip's epilogue, profiles/r04_wf_experiments.txt) and no 64-bit address
// arithmetic per lane.
#pragma once
#include <hip/hip_runtime.h>

namespace tmdiff {
namespace buf {

constexpr unsigned kOutside = 0xFFFFFFF0u;

#if defined(__HIP_DEVICE_COMPILE__)   // the builtins exist in the device pass only
using rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc make(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ float4 load4(rsrc r, unsigned voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void store4(rsrc r, unsigned voff, float4 t) {
  const u32x4 v = {__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
__device__ __forceinline__ void store4(rsrc r, unsigned voff, uint4 t) {
  const u32x4 v = {t.x, t.y, t.z, t.w};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
#else
struct rsrc {};
__device__ __forceinline__ rsrc make(const void*, unsigned) { return {}; }
__device__ __forceinline__ float4 load4(rsrc, unsigned) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void store4(rsrc, unsigned, float4) {}
__device__ __forceinline__ void store4(rsrc, unsigned, uint4) {}
#endif
// base + add where the base is inside (kOutside + anything would wrap around into the descriptor)
__device__ __forceinline__ unsigned at(unsigned base, unsigned add) { return base >= kOutside ? kOutside : base + add; }

}  // namespace buf
}  // namespace tmdiff
