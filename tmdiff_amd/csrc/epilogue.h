// Vector epilogue of the fp32 3x3x3 / 1x1x1 MFMA convolution kernels (conv3d.hip, conv3d_dma.hip).
//
// The accumulators come out of v_mfma_f32_32x32x2_f32 as D[channel row][position column]: a lane holds ONE position and 16
// channels, so written straight from that layout every register is one dword store of 32-byte (TW = 8) or 64-byte (TW = 16)
// runs -- and one dword load per residual element, one more dword store per second-output element: 16-48 vector-memory
// instructions per 32 x 32 sub-tile.  A vector-memory instruction costs its wave about the same whatever its width, and at
// 32 input channels a tile's fixed costs (set-up, first chunk, epilogue) are a third of its MFMA time (per-layer
// efficiency 64 % at Cin = 32, 77 % at 64, 85 % at 128, 87 % at 256: one constant per tile explains all four).
// Here a sub-tile goes through a wave-private 4 KB LDS tile T[channel][position] (ds_write_b32 in the D layout,
// ds_read_b128 along the positions) and leaves as dwordx4: lane (tc, tq) = (lane >> 3, lane & 7) owns channels tc + 8j
// (j = 0..3) and positions 4 tq .. 4 tq + 3 of the sub-tile, which are consecutive in w because the sub-tile is a
// run of 32 positions in (n, h, w) order with TW % 4 == 0.  Bias, residual, scale and the second output (the consumer's
// prologue) are applied in that layout with the same operations in the same order as the scalar epilogue, so both give
// the same bits.  The LDS traffic of one wave is processed in order: no barrier, only the compiler kept from reordering.
// Needs W % 4 == 0 and 16-byte aligned y / y2 / residual (Args::vec4, set by the entry points).
#pragma once
#include "bufaddr.h"
#include "common.h"

namespace tmdiff {

__device__ __forceinline__ float lane_value(float v, int src_lane) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane * 4, __float_as_int(v)));
}

// Every tensor is addressed through a buffer descriptor over this tile's MSUB x 32 channels of sample b (bufaddr.h): a lane's
// address is one 32-bit byte offset, a lane outside the image carries an offset outside the descriptor -- ragged tiles run the
// code of full ones and no store sits under a branch.  Needs planes of at most 2^23 positions (epilogue_vec_ok).
// xxx_v[m]: per-lane vectors, lane l (and l + 32) holds the value of channel co0 + m*32 + (l & 31).
template <int NS, int MSUB, int TH, int TW, bool Y, bool RES, bool Y2, class Args, class Acc>
__device__ __forceinline__ void epilogue_vec_v(const Args& a, Acc (&acc)[NS][MSUB], const float (&bias_v)[MSUB],
                                               const float (&sh2_v)[MSUB], const float (&sc2_v)[MSUB], int b, int g, int co0,
                                               int n0, int h0, int w0, int wv, int lane, long plane, float* T,
                                               const int* sub_base = nullptr) {
#pragma clang fp contract(off)   // (every instantiation rounds alike: without y the compiler would fuse "* scale" and "+ shift")
  // sub_base[s]: linear index (over the TN x TH x TW output tile, w fastest) of the first position of sub-tile s; by
  // default the waves' sub-tiles follow one another ((wv * NS + s) * 32)
  static_assert(TW % 4 == 0, "a quad of positions lies in one row");
  const int l31 = lane & 31, khalf = lane >> 5, tq = lane & 7, tc = lane >> 3;
  const long cb = ((long)b * a.Cout + g * a.cout_g + co0) * plane;
  const unsigned span = (unsigned)((long)MSUB * 32 * plane * 4);
  const buf::rsrc ry = buf::make(a.y ? a.y + cb : nullptr, a.y ? span : 0u);
  const buf::rsrc rr = buf::make(a.residual ? a.residual + cb : nullptr, a.residual ? span : 0u);
  const buf::rsrc r2 = buf::make(a.y2 ? a.y2 + cb : nullptr, a.y2 ? span : 0u);
  unsigned toff[NS];      // byte offset of the lane's quad of channel tc inside the descriptor
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p = (sub_base ? sub_base[s] : (wv * NS + s) * 32) + 4 * tq;
    const int n = n0 + p / (TW * TH), h = h0 + (p / TW) % TH, w = w0 + p % TW;
    const bool ok = n < a.N && h < a.H && w < a.W;      // (W % 4 == 0: the four positions stand or fall together)
    toff[s] = ok ? (unsigned)((tc * (int)plane + (n * a.H + h) * a.W + w) * 4) : buf::kOutside;
  }
  // residual quads of sub-tile (s, m): requested one sub-tile ahead of their use
  float4 rs[NS * MSUB + 1][4];
  auto load_res = [&](int i) __attribute__((always_inline)) {
    const int m = i / NS, s = i % NS;
#pragma unroll
    for (int j = 0; j < 4; ++j)   // (outside the image: zero, never stored)
      rs[i][j] = buf::load4(rr, buf::at(toff[s], (unsigned)(m * 32 + 8 * j) * (unsigned)plane * 4u));
  };
  if constexpr (RES) load_res(0);
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    float bias_t[4], sh2_t[4], sc2_t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bias_t[j] = lane_value(bias_v[m], tc + 8 * j);
      if constexpr (Y2) sh2_t[j] = lane_value(sh2_v[m], tc + 8 * j), sc2_t[j] = lane_value(sc2_v[m], tc + 8 * j);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (RES) {
        if (m * NS + s + 1 < NS * MSUB) load_res(m * NS + s + 1);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * khalf) * 32 + l31] = acc[s][m][r];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(T + (tc + 8 * j) * 32 + tq * 4);
        float v[4] = {t.x, t.y, t.z, t.w};
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (RES) q[0] = rs[m * NS + s][j].x, q[1] = rs[m * NS + s][j].y, q[2] = rs[m * NS + s][j].z, q[3] = rs[m * NS + s][j].w;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] + bias_t[j] + q[e]) * a.out_scale;   // as the scalar epilogue
        const unsigned o = buf::at(toff[s], (unsigned)(m * 32 + 8 * j) * (unsigned)plane * 4u);
        if constexpr (Y) buf::store4(ry, o, make_float4(v[0], v[1], v[2], v[3]));
        if constexpr (Y2) {
          float u[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float x = v[e] + sh2_t[j];
            const float xa = silu_f(x);
            u[e] = (a.y2_act ? xa : x) * sc2_t[j];
          }
          buf::store4(r2, o, make_float4(u[0], u[1], u[2], u[3]));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
  }
}

// Dispatch on what is read and written (wave-uniform).
template <int NS, int MSUB, int TN, int TH, int TW, class Args, class Acc>
__device__ __forceinline__ void epilogue_vec(const Args& a, Acc (&acc)[NS][MSUB], const float (&bias_v)[MSUB],
                                             const float (&sh2_v)[MSUB], const float (&sc2_v)[MSUB], int b, int g, int co0,
                                             int n0, int h0, int w0, int wv, int lane, long plane, float* T,
                                             const int* sub_base = nullptr) {
#define TMDIFF_EPI(Y, R, Y2) \
  epilogue_vec_v<NS, MSUB, TH, TW, Y, R, Y2>(a, acc, bias_v, sh2_v, sc2_v, b, g, co0, n0, h0, w0, wv, lane, plane, T, sub_base)
  if (a.y) {
    if (a.residual) { if (a.y2) TMDIFF_EPI(true, true, true); else TMDIFF_EPI(true, true, false); }
    else            { if (a.y2) TMDIFF_EPI(true, false, true); else TMDIFF_EPI(true, false, false); }
  } else {
    if (a.residual) TMDIFF_EPI(false, true, true); else TMDIFF_EPI(false, false, true);
  }
#undef TMDIFF_EPI
}

}  // namespace tmdiff
