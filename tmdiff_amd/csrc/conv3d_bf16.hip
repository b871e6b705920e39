// conv3d 3x3x3 with bf16 operands and fp32 accumulation on the bf16 matrix cores of gfx950
// (v_mfma_f32_32x32x16_bf16: 32 cycles per 32x32x16 product, 16x the rate of the exact-fp32 MFMA).
// This is the "bf16 compute / fp32 accumulate" mode of the WorldView-3 inference configuration
// (SURVEY 8d, config 3); the default path stays the exact-fp32 kernel in conv3d.hip.
//
// Same contract as tmdiff_conv3d_fwd (include/tmdiff_hip.h): activations are fp32 in HBM, the fused
// prologue  x' = act(x + shift[b,c]) * scale[b,c]  is evaluated in fp32 and x' is rounded to bf16
// (round-to-nearest-even) when it is staged in LDS; the weights are rounded once at pack time;
// products are accumulated in fp32 and the epilogue (bias, residual, scale) is fp32.
//
// GEMM view per (batch b, group g):  D[co, pos] = sum_{ci,tap} W[co][ci][tap] * X'[ci][pos + tap]
//   MFMA rows (A) = 32 output channels, cols (B) = 32 output positions = 4 rows x 8 columns of one band plane,
//   K = 16 = 8 input channels x 2 taps: lanes 0-31 take tap 2p, lanes 32-63 tap 2p+1 (27 taps + one zero tap).
// A workgroup (4 waves) owns 4 bands x TH x TW positions x CO channels; wave w works on band w.
// Per chunk of 8 input channels the LDS holds
//   x : [halo position][8 ch] bf16 = one 16-byte unit per position, so a lane's B operand is ONE ds_read_b128;
//       rows are 24 units apart (== 8 mod 16), which makes the 4x8-position reads bank-conflict free; for
//       TW == 8 two band planes share a row (units 0-9 and 12-21);
//   w : [tap][co][8 ci] bf16, copied verbatim from the packed weights (16-byte global loads).
// Pipeline: the next chunk is loaded into registers while the MFMAs of this one run; between chunks the
// registers go through the prologue into LDS (two barriers per chunk).  The other workgroups on the CU cover
// that hand-off.
#include <cstdint>
#include <type_traits>

#include "bufaddr.h"
#include "common.h"

// Experiment switches of the packed-input kernel (tools/build_variant.sh; every one of them gives wrong results):
// 1 = no LDS-DMA inside the chunk loop, 2 = no weight pieces inside the loop, 4 = no output stores, 8 = no barriers,
// 16 = no input pieces inside the loop.
#ifndef TMDIFF_BF16_DEBUG
#define TMDIFF_BF16_DEBUG 0
#endif
// Diagnostic build (-DTMDIFF_BF16_STAMPS=1, tools/bf16_stamps.py): every wave of the packed-input kernel adds up its
// s_memtime cycles per phase into d->splitk_ws (8 counters per wave); no stamp exists in the production build.
#ifndef TMDIFF_BF16_STAMPS
#define TMDIFF_BF16_STAMPS 0
#endif

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

struct BfArgs {
  int B, N, H, W;
  int Cin, Cout, cin_g, cout_g, groups;
  int seg_c[3];
  const float* seg_x[3];
  const uint4* wp;  // [g][chunk][28 taps][cout_g] units of 8 bf16
  const float* bias;
  float bias_scale;
  const float* in_shift;
  const float* in_scale;
  int shift_stride, scale_stride;
  int in_act;
  const float* residual;
  float out_scale;
  float* y;               // may be NULL when only y2 is wanted
  uint4* y2;              // optional second output: bf16 units [B][Cout/8][plane] of act2(y + shift2) * scale2
  const float* y2_shift;
  const float* y2_scale;
  int y2_shift_stride, y2_scale_stride, y2_act;
  int tiles_n, tiles_h, tiles_w, tiles_co;
  unsigned total_blocks;
  int vec4;                     // W % 4 == 0 and y / residual 16-byte aligned: dwordx4 epilogue through LDS
  unsigned long long* stamps;   // diagnostic builds only
};

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, k = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// stand-ins for absent shift / scale rows, so that the hand-off has no branches
__device__ const float kZeros[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
__device__ const float kOnes[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};

constexpr int TN = 4;    // bands per workgroup = waves
constexpr int RS = 24;   // LDS row stride in 16-byte units
constexpr int TAPS2 = 28;

template <int TH, int TW>
struct XLayout {
  static constexpr bool ZPAIR = TW == 8;
  static constexpr int HN = TN + 2, HH = TH + 2, HW = TW + 2;
  static constexpr int UNITS = (ZPAIR ? HN / 2 : HN) * HH * RS;
  static_assert(TW == 8 || TW == 16, "row of 24 units holds one 18-wide or two 10-wide halo rows");
  __device__ static __forceinline__ int unit(int z, int y, int x) {
    return ZPAIR ? ((z >> 1) * HH + y) * RS + (z & 1) * 12 + x : (z * HH + y) * RS + x;
  }
};

// Lanes 32-63 of x trade places with lanes 0-31 of y (v_permlane32_swap_b32).
__device__ __forceinline__ void swap_halves(unsigned& x, unsigned& y) {
#if defined(__HIP_DEVICE_COMPILE__)
  const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  x = r[0], y = r[1];
#endif
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  union { __attribute__((ext_vector_type(2))) __bf16 h; unsigned u; } p;
  p.h[0] = (__bf16)lo, p.h[1] = (__bf16)hi;
  return p.u;
}

// SiLU of the bf16 mode: the quotient by v_rcp_f32 (1 ulp) instead of the IEEE division sequence (a dozen instructions
// per element, which made the second-output epilogue cost more than the tile's MFMAs); the result is rounded to bf16
// right after.  Every prologue evaluation of this file uses it, so producer- and consumer-side packing stay bit-equal.
__device__ __forceinline__ float silu_bf(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
#else
  return v;
#endif
}

// Epilogue, D layout: col = lane&31 (position), row (r&3) + 8*(r>>2) + 4*kg (channel).  xxx_v[m] hold (in lane l31) the
// scaled bias / second-output shift / scale of channel co0 + m*32 + l31.
// The consumer's prologue on 16 finished values of one position, rounded to bf16 and regrouped into units of 8
// consecutive channels: register quads (0-3, 4-7) hold channels {0-3, 8-11} in lanes 0-31 and {4-7, 12-15} in lanes
// 32-63; trading halves gives lanes 0-31 channels 0-7 and lanes 32-63 channels 8-15 (same for quads 8-11, 12-15).
__device__ __forceinline__ void second_output_units(const f32x16& v, const float (&sh)[16], const float (&sc)[16], bool act,
                                                    uint4& lo, uint4& hi) {
  unsigned d[8];
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    float t[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float u = v[r + e] + sh[r + e];
      t[e] = (act ? silu_bf(u) : u) * sc[r + e];
    }
    d[r / 2] = pack_bf16x2(t[0], t[1]);
  }
  swap_halves(d[0], d[2]); swap_halves(d[1], d[3]);   // quads 0 / 1
  swap_halves(d[4], d[6]); swap_halves(d[5], d[7]);   // quads 2 / 3
  lo = make_uint4(d[0], d[1], d[2], d[3]);   // channel octet kg
  hi = make_uint4(d[4], d[5], d[6], d[7]);   // channel octet 2 + kg
}

// this lane's 16 per-register channel constants out of the per-lane vector (channel l31)
__device__ __forceinline__ void lane_rows(float v, int kg, float (&out)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2);
    const float x0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), row));
    const float x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), row + 4));
    out[r] = kg ? x1 : x0;
  }
}

// Generic epilogue (any width, any alignment): one dword per register.
template <int NS, int MSUB, int TW>
__device__ __forceinline__ void store_tile(const BfArgs& a, f32x16 (&acc)[NS][MSUB], const float (&bias_v)[MSUB],
                                           const float (&sh2_v)[MSUB], const float (&sc2_v)[MSUB], int b, int g, int co0,
                                           int n0, int h0, int w0, int wv, int l31, int kg, long plane) {
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    float bias_r[16], sh2_r[16], sc2_r[16];
    lane_rows(bias_v[m], kg, bias_r);
    lane_rows(sh2_v[m], kg, sh2_r);
    lane_rows(sc2_v[m], kg, sc2_r);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      constexpr int per_row = TW / 8;
      const int n = n0 + wv, h = h0 + (s / per_row) * 4 + (l31 >> 3), w = w0 + (s % per_row) * 8 + (l31 & 7);
      const bool pok = n < a.N && h < a.H && w < a.W;
      const long sp = pok ? ((long)n * a.H + h) * a.W + w : 0;
      const long obase = ((long)b * a.Cout + g * a.cout_g + co0 + m * 32 + 4 * kg) * plane + sp;
      float res[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        res[r] = (a.residual && pok) ? a.residual[obase + row * plane] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const float v = (acc[s][m][r] + bias_r[r] + res[r]) * a.out_scale;
        if (pok && a.y) a.y[obase + row * plane] = v;
        acc[s][m][r] = v;
      }
      if (a.y2) {
        uint4 lo, hi;
        second_output_units(acc[s][m], sh2_r, sc2_r, a.y2_act != 0, lo, hi);
        const long ubase = ((long)b * (a.Cout / 8) + (g * a.cout_g + co0 + m * 32) / 8 + kg) * plane + sp;
        if (pok) a.y2[ubase] = lo, a.y2[ubase + 2 * plane] = hi;
      }
    }
  }
}

// Vector epilogue (a.vec4: W % 4 == 0, y / residual 16-byte aligned), specialised on what is read and written and on the
// tile lying wholly inside the image (FULL: no predicates).  A vector-memory instruction costs its wave a fixed price
// whatever its width, and straight from the D layout every register is one dword store (and one residual load) of
// 32-byte runs -- 136 instructions per wave for four sub-tiles, 12-39 k cycles against 7 k cycles of MFMAs (s_memtime
// stamps, tools/bf16_stamps.py).  Here a 32 x 32 sub-tile goes through a wave-private 4 KB LDS tile T[channel][position]
// (ds_write_b32 in the D layout, ds_read_b128 along the positions) and leaves as dwordx4: lane (tc, tq) owns channels
// tc + 8j and the 4 consecutive w of quarter-row tq; all residual quads of the tile are requested first.  When a second
// output is wanted on top of a residual the finished values make the trip back (ds_write_b128 / ds_read_b32) into the D
// layout, which is what the packed bf16 units want (one position, 8 channels per lane).  LDS traffic of one wave is
// processed in order, so the tile needs no barrier, only the compiler kept from reordering (wavefront fence).
// Every tensor is addressed through a buffer descriptor over this tile's channels of sample b (bufaddr.h): a lane outside the
// image carries an offset outside the descriptor, ragged tiles run the code of full ones, no store sits under a branch.
template <int NS, int MSUB, int TW, bool Y, bool RES, bool Y2>
__device__ __forceinline__ void store_tile_v(const BfArgs& a, f32x16 (&acc)[NS][MSUB], const float (&bias_v)[MSUB],
                                             const float (&sh2_v)[MSUB], const float (&sc2_v)[MSUB], int b, int g, int co0,
                                             int n0, int h0, int w0, int wv, int l31, int kg, long plane, float* T) {
  namespace bf = tmdiff::buf;
  constexpr int per_row = TW / 8;
  const int lane = l31 + 32 * kg, tq = lane & 7, tc = lane >> 3;
  const int n = n0 + wv;
  // descriptors: MSUB x 32 channels of sample b from channel g * cout_g + co0 on (fp32 tensors); the packed second output:
  // their MSUB x 4 units of 8 channels, 16 bytes per (unit, position)
  const long cb = ((long)b * a.Cout + g * a.cout_g + co0) * plane;
  const unsigned span = (unsigned)((long)MSUB * 32 * plane * 4);
  const bf::rsrc ry = bf::make(a.y ? a.y + cb : nullptr, a.y ? span : 0u);
  const bf::rsrc rr = bf::make(a.residual ? a.residual + cb : nullptr, a.residual ? span : 0u);
  const bf::rsrc r2 = bf::make(a.y2 ? a.y2 + ((long)b * (a.Cout / 8) + (g * a.cout_g + co0) / 8) * plane : nullptr,
                               a.y2 ? (unsigned)((long)MSUB * 4 * plane * 16) : 0u);
  unsigned toff[NS], poff[NS];      // byte offsets: the transposed lane's quad of channel tc, the D-layout lane's position of unit kg
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int hs = h0 + (s / per_row) * 4, ws = w0 + (s % per_row) * 8;
    const int ht = hs + (tq >> 1), wt = ws + (tq & 1) * 4;      // (W % 4 == 0: the four positions stand or fall together)
    const int h = hs + (l31 >> 3), w = ws + (l31 & 7);
    const bool tok = n < a.N && ht < a.H && wt < a.W, pok = n < a.N && h < a.H && w < a.W;
    toff[s] = tok ? (unsigned)((tc * (int)plane + (n * a.H + ht) * a.W + wt) * 4) : bf::kOutside;
    poff[s] = pok ? (unsigned)((kg * (int)plane + (n * a.H + h) * a.W + w) * 16) : bf::kOutside;
  }
  // residual quads of sub-tile (s, m): requested one sub-tile ahead of their use
  float4 rs[NS * MSUB + 1][4];
  auto load_res = [&](int i) __attribute__((always_inline)) {
    const int m = i / NS, s = i % NS;
#pragma unroll
    for (int j = 0; j < 4; ++j)   // (outside the image: zero, never stored)
      rs[i][j] = bf::load4(rr, bf::at(toff[s], (unsigned)(m * 32 + 8 * j) * (unsigned)plane * 4u));
  };
  if constexpr (RES) load_res(0);
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    float bias_r[16], sh2_r[16], sc2_r[16];
    lane_rows(bias_v[m], kg, bias_r);
    if constexpr (Y2) {
      lane_rows(sh2_v[m], kg, sh2_r);
      lane_rows(sc2_v[m], kg, sc2_r);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if constexpr (RES) {
        if (m * NS + s + 1 < NS * MSUB) load_res(m * NS + s + 1);
      }
      if constexpr (Y || RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + l31] = acc[s][m][r] + bias_r[r];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 t = *reinterpret_cast<const float4*>(T + (tc + 8 * j) * 32 + tq * 4);
          if constexpr (RES) {
            const float4 q = rs[m * NS + s][j];
            t.x += q.x, t.y += q.y, t.z += q.z, t.w += q.w;
          }
          t.x *= a.out_scale, t.y *= a.out_scale, t.z *= a.out_scale, t.w *= a.out_scale;
          if constexpr (Y) bf::store4(ry, bf::at(toff[s], (unsigned)(m * 32 + 8 * j) * (unsigned)plane * 4u), t);
          if constexpr (Y2 && RES) *reinterpret_cast<float4*>(T + (tc + 8 * j) * 32 + tq * 4) = t;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      }
      if constexpr (Y2) {
        if constexpr (RES) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[s][m][r] = T[((r & 3) + 8 * (r >> 2) + 4 * kg) * 32 + l31];
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[s][m][r] = (acc[s][m][r] + bias_r[r]) * a.out_scale;
        }
        uint4 lo, hi;
        second_output_units(acc[s][m], sh2_r, sc2_r, a.y2_act != 0, lo, hi);
        const unsigned o2 = bf::at(poff[s], (unsigned)(m * 4) * (unsigned)plane * 16u);
        bf::store4(r2, o2, lo);
        bf::store4(r2, bf::at(o2, 2u * (unsigned)plane * 16u), hi);
      }
    }
  }
}

template <int NS, int MSUB, int TW>
__device__ __forceinline__ void store_tile_vec(const BfArgs& a, f32x16 (&acc)[NS][MSUB], const float (&bias_v)[MSUB],
                                               const float (&sh2_v)[MSUB], const float (&sc2_v)[MSUB], int b, int g, int co0,
                                               int n0, int h0, int w0, int wv, int l31, int kg, long plane, float* T) {
#define TMDIFF_EPI(Y, R, Y2) \
  store_tile_v<NS, MSUB, TW, Y, R, Y2>(a, acc, bias_v, sh2_v, sc2_v, b, g, co0, n0, h0, w0, wv, l31, kg, plane, T)
  if (a.y) {
    if (a.residual) { if (a.y2) TMDIFF_EPI(true, true, true); else TMDIFF_EPI(true, true, false); }
    else            { if (a.y2) TMDIFF_EPI(true, false, true); else TMDIFF_EPI(true, false, false); }
  } else {
    if (a.residual) TMDIFF_EPI(false, true, true); else TMDIFF_EPI(false, false, true);
  }
#undef TMDIFF_EPI
}

template <int NS, int MSUB, int TH, int TW, bool ACT>
__global__ void __launch_bounds__(256, 2) conv3d_bf16_kernel(const BfArgs a) {
  using XL = XLayout<TH, TW>;
  constexpr int CO = 32 * MSUB;
  constexpr int HH = XL::HH, HW = XL::HW;
  constexpr int HALO_POS = XL::HN * HH * HW;
  constexpr int XI = (HALO_POS + 255) / 256;  // halo positions per thread
  constexpr int WUNITS = TAPS2 * CO;
  constexpr int WI = (WUNITS + 255) / 256;    // weight units per thread
  constexpr int SINK = XL::UNITS + WUNITS;    // unit that absorbs the stores of idle staging slots
  static_assert(TH * TW == 32 * NS, "a wave covers one band plane = NS tiles of 4x8 positions");
  __shared__ uint4 lds[SINK + 1];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, kg = lane >> 5;

  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int tw_i = __builtin_amdgcn_readfirstlane(id % a.tiles_w); id /= a.tiles_w;
  const int th_i = __builtin_amdgcn_readfirstlane(id % a.tiles_h); id /= a.tiles_h;
  const int tn_i = __builtin_amdgcn_readfirstlane(id % a.tiles_n); id /= a.tiles_n;
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups);
  const int b = __builtin_amdgcn_readfirstlane(id / a.groups);
  const int n0 = tn_i * TN, h0 = th_i * TH, w0 = tw_i * TW;
  const int co0 = co_tile * CO;
  const long plane = (long)a.N * a.H * a.W;
  const int nchunks = a.cin_g / 8;

  // ---- staging pattern of this thread: XI halo positions (all 8 channels of a chunk each), WI weight units ----
  int goff[XI], xdst[XI];
  bool inb[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int e = tid + 256 * i;
    const int xz = e % HW, yz = (e / HW) % HH, zz = e / (HW * HH);
    const int n = n0 + zz - 1, h = h0 + yz - 1, w = w0 + xz - 1;
    const bool valid = e < HALO_POS;
    inb[i] = valid && n >= 0 && n < a.N && h >= 0 && h < a.H && w >= 0 && w < a.W;
    goff[i] = inb[i] ? (n * a.H + h) * a.W + w : 0;
    xdst[i] = valid ? XL::unit(zz, yz, xz) : SINK;
  }
  int wsrc[WI], wdst[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i) {
    const int u = tid + 256 * i;
    const bool valid = u < WUNITS;
    const int tap = valid ? u / CO : 0, j = u % CO;
    wsrc[i] = tap * a.cout_g + co0 + j;
    wdst[i] = valid ? XL::UNITS + u : SINK;
  }
  const uint4* wg = a.wp + (long)g * nchunks * TAPS2 * a.cout_g;

  // ---- operand addresses (16-byte units) ---------------------------------------------------------------------
  // B: position (band wv, row r, column xl) of sub-tile s, tap (dz,dy,dx): unit(wv+dz, y0(s)+r+dy, x0(s)+xl+dx).
  const int lane_pos = (l31 >> 3) * RS + (l31 & 7);
  int baddr[TAPS2 / 2];
#pragma unroll
  for (int p = 0; p < TAPS2 / 2; ++p) {
    const int t0 = 2 * p, t1 = 2 * p + 1 < 27 ? 2 * p + 1 : 26;  // tap 27 has zero weights; read anything valid
    const int u0 = XL::unit(wv + t0 / 9, (t0 / 3) % 3, t0 % 3);
    const int u1 = XL::unit(wv + t1 / 9, (t1 / 3) % 3, t1 % 3);
    baddr[p] = lane_pos + (kg ? u1 : u0);
  }
  const int aaddr = XL::UNITS + kg * CO + l31;  // + (2p*CO + m*32)

  // bias of channel co0 + m*32 + l31 (used in the epilogue through readlane)
  float bias_v[MSUB];
#pragma unroll
  for (int m = 0; m < MSUB; ++m) bias_v[m] = a.bias ? a.bias[g * a.cout_g + co0 + m * 32 + l31] * a.bias_scale : 0.f;
  float sh2_v[MSUB], sc2_v[MSUB];  // second-output shift / scale of channel co0 + m*32 + l31
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int col = g * a.cout_g + co0 + m * 32 + l31;
    sh2_v[m] = (a.y2 && a.y2_shift) ? a.y2_shift[(long)b * a.y2_shift_stride + col] : 0.f;
    sc2_v[m] = (a.y2 && a.y2_scale) ? a.y2_scale[(long)b * a.y2_scale_stride + col] : 1.f;
  }

  f32x16 acc[NS][MSUB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  float xr[XI][8];
  unsigned wr[WI][4];  // scalars, not uint4[]: vector-typed register arrays end up in scratch
  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    // the 8 channels of a chunk lie in one input segment (segment sizes are multiples of 8)
    const int cg = g * a.cin_g + c * 8;
    const float* src;
    int cl;
    if (cg < a.seg_c[0]) src = a.seg_x[0], cl = cg;
    else if (cg < a.seg_c[0] + a.seg_c[1]) src = a.seg_x[1], cl = cg - a.seg_c[0];
    else src = a.seg_x[2], cl = cg - a.seg_c[0] - a.seg_c[1];
    const int segc = cg < a.seg_c[0] ? a.seg_c[0] : (cg < a.seg_c[0] + a.seg_c[1] ? a.seg_c[1] : a.seg_c[2]);
    const float* base = src + ((long)b * segc + cl) * plane;
#pragma unroll
    for (int i = 0; i < XI; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[i][j] = base[goff[i] + j * plane];
    const uint4* wc = wg + (long)c * TAPS2 * a.cout_g;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const uint4 t = wc[wsrc[i]];
      wr[i][0] = t.x, wr[i][1] = t.y, wr[i][2] = t.z, wr[i][3] = t.w;
    }
  };
  auto stage_chunk = [&](int c) __attribute__((always_inline)) {
    const int cg = g * a.cin_g + c * 8;
    const float* shp = a.in_shift ? a.in_shift + (long)b * a.shift_stride + cg : kZeros;
    const float* scp = a.in_scale ? a.in_scale + (long)b * a.scale_stride + cg : kOnes;
    float sh[8], sc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[j] = shp[j], sc[j] = scp[j];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      union { bf16x8 h; uint4 u; } pk;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = xr[i][j] + sh[j];
        if constexpr (ACT) v = silu_bf(v);
        v *= sc[j];
        pk.h[j] = (__bf16)(inb[i] ? v : 0.f);
      }
      lds[xdst[i]] = pk.u;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) lds[wdst[i]] = make_uint4(wr[i][0], wr[i][1], wr[i][2], wr[i][3]);
  };

  load_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
    stage_chunk(c);
    __syncthreads();
    if (c + 1 < nchunks) load_chunk(c + 1);
#pragma unroll
    for (int p = 0; p < TAPS2 / 2; ++p) {
      union { bf16x8 h; uint4 u; } av[MSUB], bv[NS];
#pragma unroll
      for (int m = 0; m < MSUB; ++m) av[m].u = lds[aaddr + 2 * p * CO + m * 32];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        constexpr int per_row = TW / 8;  // sub-tiles side by side in a plane
        bv[s].u = lds[baddr[p] + (s / per_row) * 4 * RS + (s % per_row) * 8];
      }
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int m = 0; m < MSUB; ++m)
          acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m].h, bv[s].h, acc[s][m], 0, 0, 0);
    }
    __syncthreads();
  }

  store_tile<NS, MSUB, TW>(a, acc, bias_v, sh2_v, sc2_v, b, g, co0, n0, h0, w0, wv, l31, kg, plane);
}

// ---- two-kernel variant: prologue + bf16 packing once, then a staging-free convolution -----------------------
// pack_x: xp[b][chunk][pos] (one 16-byte unit = 8 channels, bf16) = bf16(prologue(x[b][chunk*8 .. +7][pos])).
// The convolution below then moves BOTH operands HBM/L2 -> LDS with global_load_lds_dwordx4 (no registers, no
// VALU work) into two LDS stages and spends its issue slots on ds_read_b128 + MFMA only.  Worth it when the
// input is reused by several channel tiles (the fused kernel re-evaluates the prologue per tile and halo).
__device__ const uint4 kZeroUnit = {0u, 0u, 0u, 0u};  // source of the zero padding

// One wave-wide piece (64 x 16 bytes): global src (per lane) -> LDS dst + lane*16 (dst is wave-uniform, via M0).
__device__ __forceinline__ void dma_piece(const uint4* src, uint4* dst) {
#if defined(__HIP_DEVICE_COMPILE__)  // the builtin exists in the device pass only
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#endif
}

template <bool ACT>
__global__ void __launch_bounds__(256) pack_x_bf16_kernel(const BfArgs a, uint4* __restrict__ xp) {
  const long plane = (long)a.N * a.H * a.W;
  const int nch = a.Cin / 8;
  const int bc = blockIdx.y;  // b * nch + chunk
  const int b = bc / nch, ch = bc % nch;
  const int cg = ch * 8;
  const float* src;
  int cl, segc;
  if (cg < a.seg_c[0]) src = a.seg_x[0], cl = cg, segc = a.seg_c[0];
  else if (cg < a.seg_c[0] + a.seg_c[1]) src = a.seg_x[1], cl = cg - a.seg_c[0], segc = a.seg_c[1];
  else src = a.seg_x[2], cl = cg - a.seg_c[0] - a.seg_c[1], segc = a.seg_c[2];
  const float* base = src + ((long)b * segc + cl) * plane;
  const float* shp = a.in_shift ? a.in_shift + (long)b * a.shift_stride + cg : kZeros;
  const float* scp = a.in_scale ? a.in_scale + (long)b * a.scale_stride + cg : kOnes;
  float sh[8], sc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) sh[j] = shp[j], sc[j] = scp[j];
  for (long pos = blockIdx.x * 256L + threadIdx.x; pos < plane; pos += 256L * gridDim.x) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = base[pos + j * plane];
    union { bf16x8 h; uint4 u; } pk;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = v[j] + sh[j];
      if constexpr (ACT) t = silu_bf(t);
      pk.h[j] = (__bf16)(t * sc[j]);
    }
    xp[(long)bc * plane + pos] = pk.u;
  }
}

template <int NS, int MSUB, int TH, int TW>
__global__ void __launch_bounds__(256, 2) conv3d_bf16_dma_kernel(const BfArgs a, const uint4* __restrict__ xp) {
  using XL = XLayout<TH, TW>;
  constexpr int CO = 32 * MSUB;
  constexpr int HH = XL::HH, HW = XL::HW;
  constexpr int XINST = (XL::UNITS + 63) / 64;   // wave-wide 1 KiB pieces of the x image
  constexpr int WINST = TAPS2 * CO / 64;         // ... of the weight slab
  constexpr int XU = XINST * 64;
  constexpr int XK = (XINST + 3) / 4, WK = (WINST + 3) / 4;  // pieces per wave
  constexpr int STAGE = XU + TAPS2 * CO;
  static_assert(TH * TW == 32 * NS, "a wave covers one band plane = NS tiles of 4x8 positions");
  __shared__ uint4 st0[STAGE];
  __shared__ uint4 st1[STAGE];
#if TMDIFF_BF16_STAMPS
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime(), rt_entry = __builtin_amdgcn_s_memrealtime();
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, kg = lane >> 5;

  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int tw_i = __builtin_amdgcn_readfirstlane(id % a.tiles_w); id /= a.tiles_w;
  const int th_i = __builtin_amdgcn_readfirstlane(id % a.tiles_h); id /= a.tiles_h;
  const int tn_i = __builtin_amdgcn_readfirstlane(id % a.tiles_n); id /= a.tiles_n;
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups);
  const int b = __builtin_amdgcn_readfirstlane(id / a.groups);
  const int n0 = tn_i * TN, h0 = th_i * TH, w0 = tw_i * TW;
  const int co0 = co_tile * CO;
  const long plane = (long)a.N * a.H * a.W;
  const int nchunks = a.cin_g / 8;

  // ---- DMA sources of this lane: x piece q = wv + 4k covers LDS units q*64 + lane ------------------------------
  int xpos[XK];  // position offset inside a chunk plane, or -1 = zero unit (padding / layout filler)
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int u = (wv + 4 * k) * 64 + lane;
    const int row = u / RS, col = u % RS;
    int zz, yz, xz;
    if constexpr (XL::ZPAIR) {
      zz = 2 * (row / HH) + (col >= 12), yz = row % HH, xz = col >= 12 ? col - 12 : col;
    } else {
      zz = row / HH, yz = row % HH, xz = col;
    }
    const int n = n0 + zz - 1, h = h0 + yz - 1, w = w0 + xz - 1;
    const bool ok = u < XL::UNITS && xz < HW && n >= 0 && n < a.N && h >= 0 && h < a.H && w >= 0 && w < a.W;
    xpos[k] = ok ? (n * a.H + h) * a.W + w : -1;
  }
  const uint4* xg = xp + ((long)b * (a.Cin / 8) + (long)g * nchunks) * plane;
  // weight piece q covers slab units q*64 + lane = tap-major [tap][CO]
  int wsrc[WK];
#pragma unroll
  for (int k = 0; k < WK; ++k) {
    const int u = (wv + 4 * k) * 64 + lane;
    wsrc[k] = (u / CO) * a.cout_g + co0 + u % CO;
  }
  const uint4* wg = a.wp + (long)g * nchunks * TAPS2 * a.cout_g;

  // piece i of this wave for chunk c: i < XK = input pieces, then the weight pieces
  constexpr int NPIECE = XK + WK;
  static_assert(NPIECE <= TAPS2 / 2, "one piece is issued per tap pair");
  auto issue_piece = [&](auto ic, int c, uint4* st) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < XK) {
      constexpr int k = i;
      const int q = wv + 4 * k;
      if (XINST % 4 == 0 || q < XINST) {
        const uint4* src = xpos[k] >= 0 ? xg + (long)c * plane + xpos[k] : &kZeroUnit;
        dma_piece(src, st + q * 64);
      }
    } else if constexpr (i < NPIECE) {
      constexpr int k = i - XK;
      const int q = wv + 4 * k;
      if (WINST % 4 == 0 || q < WINST) dma_piece(wg + (long)c * TAPS2 * a.cout_g + wsrc[k], st + XU + q * 64);
    }
  };

  // the first chunk is requested as soon as its addresses exist; the rest of the set-up runs under its flight time
#if TMDIFF_BF16_STAMPS
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  static_for<0, NPIECE>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, 0, st0); });
  __builtin_amdgcn_sched_barrier(0);

  // ---- operand addresses (16-byte units inside a stage) ---------------------------------------------------------
  const int lane_pos = (l31 >> 3) * RS + (l31 & 7);
  int baddr[TAPS2 / 2];
#pragma unroll
  for (int p = 0; p < TAPS2 / 2; ++p) {
    const int t0 = 2 * p, t1 = 2 * p + 1 < 27 ? 2 * p + 1 : 26;
    const int u0 = XL::unit(wv + t0 / 9, (t0 / 3) % 3, t0 % 3);
    const int u1 = XL::unit(wv + t1 / 9, (t1 / 3) % 3, t1 % 3);
    baddr[p] = lane_pos + (kg ? u1 : u0);
  }
  const int aaddr = XU + kg * CO + l31;

  float bias_v[MSUB];
#pragma unroll
  for (int m = 0; m < MSUB; ++m) bias_v[m] = a.bias ? a.bias[g * a.cout_g + co0 + m * 32 + l31] * a.bias_scale : 0.f;
  float sh2_v[MSUB], sc2_v[MSUB];  // second-output shift / scale of channel co0 + m*32 + l31
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int col = g * a.cout_g + co0 + m * 32 + l31;
    sh2_v[m] = (a.y2 && a.y2_shift) ? a.y2_shift[(long)b * a.y2_shift_stride + col] : 0.f;
    sc2_v[m] = (a.y2 && a.y2_scale) ? a.y2_scale[(long)b * a.y2_scale_stride + col] : 1.f;
  }

  f32x16 acc[NS][MSUB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  // While chunk c is multiplied, one piece of chunk c_next is issued behind the last MFMA of each of the first NPIECE
  // tap pairs: issued in a burst at the start of a chunk, the pieces cost the wave ~0.6 us in which it feeds no
  // MFMA (measured: MFMA-only 1.5 us + DMA-only 1.15 us per chunk added up to 2.6 us).
  auto mfma_chunk = [&](const uint4* st, int c_next, uint4* st_next) __attribute__((always_inline)) {
    constexpr int per_row = TW / 8;
    union Frag { bf16x8 h; uint4 u; };
    Frag av[2][MSUB], bv[2][NS];
    auto fetch = [&](auto pc) __attribute__((always_inline)) {
      constexpr int p = decltype(pc)::value;
#pragma unroll
      for (int m = 0; m < MSUB; ++m) av[p & 1][m].u = st[aaddr + 2 * p * CO + m * 32];
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[p & 1][s].u = st[baddr[p] + (s / per_row) * 4 * RS + (s % per_row) * 8];
    };
    fetch(std::integral_constant<int, 0>{});
    // The operands of tap pair p+1 are requested right after the first MFMA of pair p (pinned with sched_barrier: the
    // scheduler otherwise sinks the reads behind the MFMAs).  The wait the compiler puts in front of pair p+1 is a
    // full lgkmcnt(0), so the reads must be old by then: three MFMAs (~100 cycles) cover the LDS latency.
    static_for<0, TAPS2 / 2>([&](auto pc) __attribute__((always_inline)) {
      constexpr int p = decltype(pc)::value;
      static_for<0, NS * MSUB>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int sx = j / MSUB, m = j % MSUB;
        acc[sx][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[p & 1][m].h, bv[p & 1][sx].h, acc[sx][m], 0, 0, 0);
        if constexpr (j == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (p + 1 < TAPS2 / 2) fetch(std::integral_constant<int, p + 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (j == NS * MSUB - 1 && p < NPIECE && !(TMDIFF_BF16_DEBUG & 1) && !((TMDIFF_BF16_DEBUG & 2) && p >= XK) &&
                      !((TMDIFF_BF16_DEBUG & 16) && p < XK)) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(std::integral_constant<int, p>{}, c_next, st_next);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // two stages, one barrier per chunk: while chunk c is multiplied out of one stage the pieces of chunk c+1 land in
  // the other; the barrier at the end of a chunk says "everyone has read this stage and my pieces have landed".
#if TMDIFF_BF16_STAMPS
  unsigned long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // prologue, mfma phase, landing wait, barrier, epilogue,
                                                                      // total, HW_ID, LDS_ALLOC, realtime in / out, setup
  unsigned long long t_a, t_b;
#define STAMP_PHASE(i) (t_b = __builtin_amdgcn_s_memtime(), tk[i] += t_b - t_a, t_a = t_b)
#else
#define STAMP_PHASE(i) ((void)0)
#endif
  __syncthreads();
#if TMDIFF_BF16_STAMPS
  t_a = __builtin_amdgcn_s_memtime();
  tk[0] = t_a - t_start;
#endif
  auto chunk_end = [&]() __attribute__((always_inline)) {
#if TMDIFF_BF16_STAMPS
    STAMP_PHASE(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's pieces have landed
    STAMP_PHASE(2);
#endif
    if constexpr (!(TMDIFF_BF16_DEBUG & 8)) __syncthreads();
    STAMP_PHASE(3);
  };
  for (int c = 0; c < nchunks; c += 2) {
    // (past the last chunk the pieces of chunk 0 are fetched again into the idle stage: valid addresses, nobody
    //  reads them, and the MFMA stream stays free of branches)
    mfma_chunk(st0, c + 1 < nchunks ? c + 1 : 0, st1);
    chunk_end();
    if (c + 1 < nchunks) {
      mfma_chunk(st1, c + 2 < nchunks ? c + 2 : 0, st0);
      chunk_end();
    }
  }
  if constexpr (TMDIFF_BF16_DEBUG & 4) {   // keep the accumulators alive, store (practically) nothing
    float t = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int m = 0; m < MSUB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[s][m][r];
    if (t == 123456.789f && a.y) a.y[0] = t;
    return;
  }
  // (the chunk loop ends with a barrier: nobody reads the stages any more)
  static_assert(sizeof(st0) >= 4 * 4096, "the epilogue borrows 4 KB of LDS per wave");
  float* const T = reinterpret_cast<float*>(st0) + wv * 1024;
  if (!a.vec4)
    store_tile<NS, MSUB, TW>(a, acc, bias_v, sh2_v, sc2_v, b, g, co0, n0, h0, w0, wv, l31, kg, plane);
  else
    store_tile_vec<NS, MSUB, TW>(a, acc, bias_v, sh2_v, sc2_v, b, g, co0, n0, h0, w0, wv, l31, kg, plane, T);
#if TMDIFF_BF16_STAMPS
  __builtin_amdgcn_s_waitcnt(0x0F70);     // (stores issued and acknowledged)
  STAMP_PHASE(4);
  tk[5] = t_a - t_start;
  tk[6] = __builtin_amdgcn_s_getreg(0xF804), tk[7] = __builtin_amdgcn_s_getreg(0xF806);
  tk[8] = rt_entry, tk[9] = __builtin_amdgcn_s_memrealtime(), tk[10] = t_start - t_entry;
  if (a.stamps && lane == 0)
    for (int i = 0; i < 12; ++i) a.stamps[((long)blockIdx.x * 4 + wv) * 12 + i] = tk[i];
#endif
#undef STAMP_PHASE
}

// ---- 1x1x1 convolution, bf16 operands -------------------------------------------------------------------------
// A bandwidth kernel (2*Cin*Cout/(4*(Cin+Cout)) FLOP per byte): no LDS at all.  MFMA columns = 32 consecutive
// positions, K = 16 input channels: lane (col, kg) loads its 8 channels of position col straight from the fp32
// activations (each load coalesced over the positions), applies the prologue and converts -- that IS the B
// operand.  The A operand (weights [chunk][co][8 ci]) comes from L2 as one 16-byte load per lane.
// A workgroup = 4 waves x (NS x 32 positions) x (MSUB x 32 channels).
template <int NS, int MSUB, bool ACT>
__global__ void __launch_bounds__(256, 2) conv1_bf16_kernel(const BfArgs a) {
  constexpr int CO = 32 * MSUB;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, kg = lane >> 5;
  const long plane = (long)a.N * a.H * a.W;
  unsigned id = blockIdx.x;
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int ptile = __builtin_amdgcn_readfirstlane(id % a.tiles_w); id /= a.tiles_w;   // tiles_w = position tiles here
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups);
  const int b = __builtin_amdgcn_readfirstlane(id / a.groups);
  const int co0 = co_tile * CO;
  const int nsteps = a.cin_g / 16;
  const long pos0 = (long)ptile * (4 * NS * 32) + wv * (NS * 32) + l31;
  long pos[NS];
  bool pok[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    pos[s] = pos0 + s * 32;
    pok[s] = pos[s] < plane;
    if (!pok[s]) pos[s] = plane - 1;  // clamped loads, no store
  }
  const uint4* wg = a.wp + (long)g * (a.cin_g / 8) * a.cout_g + co0 + l31;

  f32x16 acc[NS][MSUB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  float xr[2][NS][8];
  unsigned wr[2][MSUB][4];
  float shs[2][8], scs[2][8];
  auto load_step = [&](int q, int buf) __attribute__((always_inline)) {
    const int cg = g * a.cin_g + q * 16 + kg * 8;  // this lane's octet (global channel index)
    const float* src;
    int cl, segc;
    if (cg < a.seg_c[0]) src = a.seg_x[0], cl = cg, segc = a.seg_c[0];
    else if (cg < a.seg_c[0] + a.seg_c[1]) src = a.seg_x[1], cl = cg - a.seg_c[0], segc = a.seg_c[1];
    else src = a.seg_x[2], cl = cg - a.seg_c[0] - a.seg_c[1], segc = a.seg_c[2];
    const float* base = src + ((long)b * segc + cl) * plane;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) xr[buf][s][j] = base[pos[s] + j * plane];
    const float* shp = a.in_shift ? a.in_shift + (long)b * a.shift_stride + cg : kZeros;
    const float* scp = a.in_scale ? a.in_scale + (long)b * a.scale_stride + cg : kOnes;
#pragma unroll
    for (int j = 0; j < 8; ++j) shs[buf][j] = shp[j], scs[buf][j] = scp[j];
#pragma unroll
    for (int m = 0; m < MSUB; ++m) {
      const uint4 t = wg[(long)(2 * q + kg) * a.cout_g + m * 32];
      wr[buf][m][0] = t.x, wr[buf][m][1] = t.y, wr[buf][m][2] = t.z, wr[buf][m][3] = t.w;
    }
  };
  auto mfma_step = [&](int buf) __attribute__((always_inline)) {
    union Frag { bf16x8 h; uint4 u; };
    Frag bv[NS], av[MSUB];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = xr[buf][s][j] + shs[buf][j];
        if constexpr (ACT) v = silu_bf(v);
        bv[s].h[j] = (__bf16)(v * scs[buf][j]);
      }
#pragma unroll
    for (int m = 0; m < MSUB; ++m) av[m].u = make_uint4(wr[buf][m][0], wr[buf][m][1], wr[buf][m][2], wr[buf][m][3]);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int m = 0; m < MSUB; ++m)
        acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m].h, bv[s].h, acc[s][m], 0, 0, 0);
  };
  load_step(0, 0);
  for (int q = 0; q < nsteps; q += 2) {
    if (q + 1 < nsteps) load_step(q + 1, 1);
    mfma_step(0);
    if (q + 1 < nsteps) {
      if (q + 2 < nsteps) load_step(q + 2, 0);
      mfma_step(1);
    }
  }

  // epilogue: col = position, row (r&3) + 8*(r>>2) + 4*kg = channel
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int cbase = g * a.cout_g + co0 + m * 32 + 4 * kg;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const long obase = ((long)b * a.Cout + cbase) * plane + pos[s];
      float res[16], bs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        res[r] = a.residual ? a.residual[obase + row * plane] : 0.f;
        bs[r] = a.bias ? a.bias[cbase + row] * a.bias_scale : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        if (pok[s]) a.y[obase + row * plane] = (acc[s][m][r] + bs[r] + res[r]) * a.out_scale;
      }
    }
  }
}

template <int NS, int MSUB>
int launch_k1(BfArgs& a, hipStream_t st) {
  const long plane = (long)a.N * a.H * a.W;
  a.tiles_w = (int)((plane + 4 * NS * 32 - 1) / (4 * NS * 32));
  a.tiles_co = a.cout_g / (32 * MSUB);
  const long blocks = (long)a.B * a.groups * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv1_bf16: grid of %ld blocks", blocks);
  if (a.in_act) conv1_bf16_kernel<NS, MSUB, true><<<(unsigned)blocks, 256, 0, st>>>(a);
  else conv1_bf16_kernel<NS, MSUB, false><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_fwd_bf16 (1x1x1)");
}

// packed[g][chunk][tap][co][8 ci] (bf16, RNE) <- w[g*cout_g + co][chunk*8 + ci][tap]
// 3x3x3: 28 tap slots, slot 27 = 0 (taps are consumed in pairs); 1x1x1: one slot.
__global__ void __launch_bounds__(256) pack_weights_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ packed,
                                                                int cout_g, int cin_g, int groups, int taps, int slots,
                                                                long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += 256L * gridDim.x) {
    const int ci8 = (int)(i % 8);
    long r = i / 8;
    const int co = (int)(r % cout_g); r /= cout_g;
    const int tap = (int)(r % slots); r /= slots;
    const int chunk = (int)(r % (cin_g / 8));
    const int g = (int)(r / (cin_g / 8));
    float v = 0.f;
    if (tap < taps) v = w[(((long)g * cout_g + co) * cin_g + chunk * 8 + ci8) * taps + tap];
    const __bf16 h = (__bf16)v;
    packed[i] = *reinterpret_cast<const uint16_t*>(&h);
  }
}

template <int NS, int MSUB, int TH, int TW>
int launch(BfArgs& a, hipStream_t st) {
  constexpr int CO = 32 * MSUB;
  a.tiles_n = (a.N + TN - 1) / TN;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  a.tiles_co = a.cout_g / CO;
  const long blocks = (long)a.B * a.groups * a.tiles_n * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d_bf16: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  if (a.in_act)
    conv3d_bf16_kernel<NS, MSUB, TH, TW, true><<<(unsigned)blocks, 256, 0, st>>>(a);
  else
    conv3d_bf16_kernel<NS, MSUB, TH, TW, false><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_fwd_bf16");
}

template <int NS, int MSUB, int TH, int TW>
int launch_dma(BfArgs& a, uint4* xp, hipStream_t st, bool do_pack = true) {
  constexpr int CO = 32 * MSUB;
  const long plane = (long)a.N * a.H * a.W;
  if (do_pack) {
    long pb = (plane + 255) / 256;
    if (pb > 1024) pb = 1024;
    const dim3 pgrid((unsigned)pb, (unsigned)(a.B * (a.Cin / 8)));
    if (a.in_act) pack_x_bf16_kernel<true><<<pgrid, 256, 0, st>>>(a, xp);
    else pack_x_bf16_kernel<false><<<pgrid, 256, 0, st>>>(a, xp);
  }
  a.tiles_n = (a.N + TN - 1) / TN;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  a.tiles_co = a.cout_g / CO;
  const long blocks = (long)a.B * a.groups * a.tiles_n * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d_bf16: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  conv3d_bf16_dma_kernel<NS, MSUB, TH, TW><<<(unsigned)blocks, 256, 0, st>>>(a, xp);
  return tmdiff::check_launch("conv3d_fwd_bf16 (packed input)");
}

}  // namespace

extern "C" size_t tmdiff_conv3d_bf16_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!d || d->B <= 0 || d->Cin <= 0 || d->Cin % 8) return 0;
  return (size_t)d->B * d->Cin * d->N * d->H * d->W * 2;
}

extern "C" size_t tmdiff_conv3d_packed_bf16_bytes(int32_t Cout, int32_t Cin, int32_t ksize, int32_t groups) {
  if (groups < 1 || Cout <= 0 || Cin <= 0 || Cin % groups || Cout % groups || (ksize != 1 && ksize != 3)) return 0;
  if ((Cin / groups) % (ksize == 3 ? 8 : 16)) return 0;
  return (size_t)Cin / 8 * (ksize == 3 ? TAPS2 : 1) * (Cout / groups) * 16;
}

extern "C" int tmdiff_conv3d_pack_weights_bf16(const float* w, void* packed, int32_t Cout, int32_t Cin, int32_t ksize,
                                               int32_t groups, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(w && packed, "pack_weights_bf16: NULL pointer");
  TMDIFF_REQUIRE(ksize == 1 || ksize == 3, "pack_weights_bf16: ksize=%d (1 or 3)", ksize);
  TMDIFF_REQUIRE(groups >= 1 && Cout > 0 && Cin > 0 && Cout % groups == 0 && Cin % groups == 0,
                 "pack_weights_bf16: Cout=%d Cin=%d groups=%d", Cout, Cin, groups);
  TMDIFF_REQUIRE((Cin / groups) % 8 == 0, "pack_weights_bf16: Cin/groups=%d is not a multiple of 8", Cin / groups);
  const int taps = ksize == 3 ? 27 : 1, slots = ksize == 3 ? TAPS2 : 1;
  const long total = (long)(Cin / 8) * slots * (Cout / groups) * 8;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  pack_weights_bf16_kernel<<<(int)blocks, 256, 0, as_stream(stream)>>>(w, static_cast<uint16_t*>(packed), Cout / groups,
                                                                      Cin / groups, groups, taps, slots, total);
  return check_launch("conv3d_pack_weights_bf16");
}

extern "C" int tmdiff_conv3d_fwd_bf16(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d != nullptr, "conv3d_fwd_bf16: NULL descriptor");
  TMDIFF_REQUIRE(d->B >= 0 && d->N > 0 && d->H > 0 && d->W > 0, "conv3d_fwd_bf16: bad extents B=%d N=%d H=%d W=%d", d->B,
                 d->N, d->H, d->W);
  TMDIFF_REQUIRE(d->groups == 1 || d->groups == 3, "conv3d_fwd_bf16: groups=%d (1 or 3)", d->groups);
  TMDIFF_REQUIRE(d->Cin > 0 && d->Cout > 0 && d->Cin % d->groups == 0 && d->Cout % d->groups == 0,
                 "conv3d_fwd_bf16: Cin=%d Cout=%d groups=%d", d->Cin, d->Cout, d->groups);
  TMDIFF_REQUIRE(d->nseg >= 1 && d->nseg <= 3, "conv3d_fwd_bf16: nseg=%d", d->nseg);
  TMDIFF_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv3d_fwd_bf16: ksize=%d (1 or 3)", d->ksize);
  if (d->in_mask || d->drop_p > 0.f) return fail(TMDIFF_E_UNSUPPORTED, "conv3d_fwd_bf16: dropout (training) is fp32 only");
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  if (cin_g % (d->ksize == 3 ? 8 : 16) || cout_g % 32)
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_fwd_bf16: Cin/g=%d (multiple of %d) Cout/g=%d (multiple of 32)", cin_g,
                d->ksize == 3 ? 8 : 16, cout_g);
  if (d->B == 0) return TMDIFF_OK;
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    TMDIFF_REQUIRE(d->seg_x[i] != nullptr && d->seg_c[i] > 0, "conv3d_fwd_bf16: segment %d is empty", i);
    if (d->seg_c[i] % 8) return fail(TMDIFF_E_UNSUPPORTED, "conv3d_fwd_bf16: segment of %d channels", d->seg_c[i]);
    csum += d->seg_c[i];
  }
  TMDIFF_REQUIRE(csum == d->Cin, "conv3d_fwd_bf16: segments hold %d channels, Cin=%d", csum, d->Cin);
  if (d->groups == 3)
    TMDIFF_REQUIRE(d->nseg == 1 || (d->nseg == 3 && d->seg_c[0] == d->seg_c[1] && d->seg_c[1] == d->seg_c[2]),
                   "conv3d_fwd_bf16: groups=3 wants 1 segment or 3 equal ones");
  TMDIFF_REQUIRE(d->w_packed && (d->y || d->y2), "conv3d_fwd_bf16: NULL weights/output");
  if (d->y2) {
    if (!d->y2_bf16) return fail(TMDIFF_E_UNSUPPORTED, "conv3d_fwd_bf16: the second output is written as bf16 units (y2_bf16)");
    if (d->ksize != 3) return fail(TMDIFF_E_UNSUPPORTED, "conv3d_fwd_bf16: second output for 3x3x3 only");
    TMDIFF_REQUIRE(aligned16(d->y2), "conv3d_fwd_bf16: y2 must be 16-byte aligned");
  }
  if (d->x_bf16) {
    TMDIFF_REQUIRE(d->ksize == 3 && d->nseg == 1 && !d->in_shift && !d->in_scale && !d->in_act && aligned16(d->seg_x[0]),
                   "conv3d_fwd_bf16: a bf16-packed input is one 16-byte aligned tensor without prologue, 3x3x3 only");
  }
  TMDIFF_REQUIRE(aligned16(d->w_packed), "conv3d_fwd_bf16: packed weights must be 16-byte aligned");
  TMDIFF_REQUIRE((long)d->N * d->H * d->W < (1L << 31), "conv3d_fwd_bf16: plane too large for 32-bit offsets");

  BfArgs a;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cout = d->Cout; a.groups = d->groups; a.cin_g = cin_g; a.cout_g = cout_g;
  for (int i = 0; i < 3; ++i) {
    a.seg_c[i] = i < d->nseg ? d->seg_c[i] : (1 << 28);
    a.seg_x[i] = i < d->nseg ? d->seg_x[i] : d->seg_x[0];
  }
  a.wp = reinterpret_cast<const uint4*>(d->w_packed);
  a.bias = d->bias; a.bias_scale = d->bias_scale;
  a.in_shift = d->in_shift; a.in_scale = d->in_scale; a.in_act = d->in_act;
  a.shift_stride = d->in_shift_stride > 0 ? d->in_shift_stride : (d->in_shift_stride < 0 ? 0 : d->Cin);
  a.scale_stride = d->in_scale_stride > 0 ? d->in_scale_stride : (d->in_scale_stride < 0 ? 0 : d->Cin);
  a.residual = d->residual; a.out_scale = d->out_scale; a.y = d->y;
  a.y2 = reinterpret_cast<uint4*>(d->y2); a.y2_shift = d->y2_shift; a.y2_scale = d->y2_scale; a.y2_act = d->y2_act;
  a.y2_shift_stride = d->y2_shift_stride > 0 ? d->y2_shift_stride : (d->y2_shift_stride < 0 ? 0 : d->Cout);
  a.y2_scale_stride = d->y2_scale_stride > 0 ? d->y2_scale_stride : (d->y2_scale_stride < 0 ? 0 : d->Cout);
  // (the dwordx4 epilogue addresses its tensors through descriptors of 32-bit offsets: planes of at most 2^24 positions)
  a.vec4 = d->W % 4 == 0 && (!d->y || aligned16(d->y)) && (!d->residual || aligned16(d->residual)) && (long)d->N * d->H * d->W <= (1L << 24);
  a.stamps = TMDIFF_BF16_STAMPS ? static_cast<unsigned long long*>(d->splitk_ws) : nullptr;
  hipStream_t st = as_stream(stream);
  if (d->x_bf16) {  // input already packed by its producer: straight to the staging-free kernel
    const uint4* xp = reinterpret_cast<const uint4*>(d->seg_x[0]);
    if (cout_g % 64 == 0) return launch_dma<2, 2, 8, 8>(a, const_cast<uint4*>(xp), st, false);
    return d->W >= 16 ? launch_dma<4, 1, 8, 16>(a, const_cast<uint4*>(xp), st, false)
                      : launch_dma<2, 1, 8, 8>(a, const_cast<uint4*>(xp), st, false);
  }
  if (d->ksize == 1) {  // bandwidth kernel, no workspace
    if (cout_g % 128 == 0) return launch_k1<1, 4>(a, st);
    return cout_g % 64 == 0 ? launch_k1<2, 2>(a, st) : launch_k1<4, 1>(a, st);
  }
  if (workspace) {  // two-kernel variant: pack the prologue output once, then the staging-free kernel
    TMDIFF_REQUIRE(aligned16(workspace), "conv3d_fwd_bf16: workspace must be 16-byte aligned");
    TMDIFF_REQUIRE(d->B * (long)(d->Cin / 8) <= 65535, "conv3d_fwd_bf16: B*Cin/8 = %ld exceeds the pack grid", d->B * (long)(d->Cin / 8));
    uint4* xp = static_cast<uint4*>(workspace);
    if (cout_g % 64 == 0) return launch_dma<2, 2, 8, 8>(a, xp, st);
    return d->W >= 16 ? launch_dma<4, 1, 8, 16>(a, xp, st) : launch_dma<2, 1, 8, 8>(a, xp, st);
  }
  if (cout_g % 64 == 0) return launch<2, 2, 8, 8>(a, st);
  return d->W >= 16 ? launch<4, 1, 8, 16>(a, st) : launch<2, 1, 8, 8>(a, st);
}
