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

// x' = act(x + shift[b,c]) * scale[b,c] * mask of the descriptor's (possibly segmented) input, written densely as
// [B, Cin, N, H, W] (backward.hip).  Used by the weight-gradient kernel and the staged forward convolution.
int launch_prologue_apply(const tmdiff_conv3d_desc* d, float* xp, hipStream_t st);

// 1x1x1 forward through the LDS-free bandwidth kernel (conv1.hip); TMDIFF_E_UNSUPPORTED = shape not taken.
int conv1_fp32_try(const tmdiff_conv3d_desc* d, hipStream_t st);

}  // namespace tmdiff

#define TMDIFF_REQUIRE(cond, ...) \
  do {                            \
    if (!(cond)) return tmdiff::fail(TMDIFF_E_INVALID, __VA_ARGS__); \
  } while (0)
