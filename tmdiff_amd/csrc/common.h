// Shared host-side helpers for libtmdiff_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "tmdiff_hip.h"

namespace tmdiff {

constexpr int kWave = 64;  // CDNA wavefront

char* last_error_buf();  // thread-local, 512 bytes (abi.cpp)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(TMDIFF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return TMDIFF_OK;
}

inline hipStream_t as_stream(tmdiff_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float silu_f(float v) {
  // x * sigmoid(x); __expf -> v_exp_f32 on a pre-scaled argument, 1 ulp-level accurate for our range
  return v / (1.0f + __expf(-v));
}

// Dropout keep factor of element `idx` under `seed`: 1/(1-p) with probability 1-p, else 0.  Counter-based (splitmix64
// finaliser of idx + seed * golden ratio): every kernel that needs the mask recomputes it.  thresh = p * 2^32.
__device__ __forceinline__ float drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32) >= thresh ? inv_keep : 0.f;
}
inline uint32_t drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t;
}

// x' = act(x + shift[b,c]) * scale[b,c] * mask of the descriptor's (possibly segmented) input, written densely as
// [B, Cin, N, H, W] (backward.hip).  Used by the weight-gradient kernel and the staged forward convolution.
int launch_prologue_apply(const tmdiff_conv3d_desc* d, float* xp, hipStream_t st);

// Tile configuration and split-K factor of a fp32 3x3x3 launch -- one rule for the fused kernel (conv3d.hip), the staged
// kernel (conv3d_dma.hip) and the workspace query, so that the two kernels always split identically.
//   tile 0: 128 positions x 64 channels (2x8x8 box)     tile 1: 256 x 64 (4x8x8)
//   tile 2: 512 positions x 32 channels (4x8x16 box)    tile 3: 256 x 32 (4x8x8)
struct Conv3Plan {
  int tile;
  long blocks;   // workgroups without splitting
  int ksplit;    // 1 = no split; else the chunks (4 input channels each) are divided into ksplit equal ranges
};
Conv3Plan plan_conv3(const tmdiff_conv3d_desc* d);
// the dwordx4 epilogue (epilogue.h) applies: W % 4 == 0, 16-byte aligned outputs / residual (TMDIFF_EPILOGUE_VEC=0: never)
bool epilogue_vec_ok(const tmdiff_conv3d_desc* d);

// sum of the split-K partials + epilogue (conv3d.hip)
struct SplitKReduceArgs {
  const float* part;
  int ksplit;
  int B, Cout;
  long plane;
  const float* bias;
  float bias_scale;
  const float* residual;
  float out_scale;
  float* y;
  float* y2;
  const float* y2_shift;
  const float* y2_scale;
  int y2_shift_stride, y2_scale_stride, y2_act;
};
int launch_splitk_reduce(const SplitKReduceArgs& r, hipStream_t st);

// 1x1x1 forward through the LDS-free bandwidth kernel (conv1.hip); TMDIFF_E_UNSUPPORTED = shape not taken.
int conv1_fp32_try(const tmdiff_conv3d_desc* d, hipStream_t st, bool dry = false);   // dry: no launch, TMDIFF_OK = the 16-byte kernel on a raw input

}  // namespace tmdiff

#define TMDIFF_REQUIRE(cond, ...) \
  do {                            \
    if (!(cond)) return tmdiff::fail(TMDIFF_E_INVALID, __VA_ARGS__); \
  } while (0)
