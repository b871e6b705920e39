// 1x1x1 convolution on the exact-fp32 matrix cores, as a bandwidth kernel.
//
// res_conv (ResBlockModulateBEST, Hyper_unet_general.py:231, :248) and Conv_2 (WaveletUPorDown :361, :389) are
// 1x1x1: 2*Cin*Cout/(4*(Cin+Cout)) FLOP per byte, i.e. HBM-bound for the 32..256-channel layers.  The generic
// implicit-GEMM kernel stages them through LDS like a 3x3x3 convolution and reaches ~3 TB/s; this kernel has no LDS:
//   MFMA columns = 32 consecutive positions, K step = 2 input channels (v_mfma_f32_32x32x2_f32):
//   lane (col, khalf) loads x[channel 2q+khalf][position col] straight from the activations -- one dword, coalesced
//   over the positions -- applies the prologue, and that IS the B operand; the A operand (packed weights
//   [ci][co], tmdiff_conv3d_pack_weights) is one 4/8-byte load per lane from L2.
// A workgroup = 4 waves x (NS x 32 positions) x (MSUB x 32 channels); 16 input channels (8 K-steps) are in flight
// in registers while the previous 16 are multiplied.  Same accumulation order as the generic kernel.
// Three forms (conv1_fp32_try picks): the 16-byte kernel (a lane owns four consecutive positions; grids of at least 512 tiles),
// the dword kernel above (anything else), and -- round 4 -- the dword kernel with the input channels split over the four waves of
// a workgroup (KS = 4: small planes, where the channel loop was a chain of memory round trips).  The 16-byte and the KS = 4 form
// can also write act(x + shift) of their input as a by-product (desc.xp_*: the prologue output another convolution of the same
// input wants), stored through a buffer descriptor.
#include <cstdlib>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int G = 8;  // K-steps per register group (16 input channels)

__device__ const float kZeros16[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
__device__ const float kOnes16[16] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};

struct K1Args {
  int B, Cin, Cout, cin_g, cout_g, groups;
  int seg_c[3];
  const float* seg_x[3];
  const float* wp;
  const float* bias;
  float bias_scale;
  const float* in_shift;
  const float* in_scale;
  int shift_stride, scale_stride;
  const float* residual;
  float out_scale;
  float* y;
  long plane;
  int ptiles, tiles_co;
  // by-product (desc.xp_out): act(x + xp_shift[b, c]) of every input element, written densely as [B, Cin, plane] by the workgroups of
  // channel tile 0 -- the prologue output ANOTHER convolution of the same (segmented) input wants (a ResBlock's conv20 beside
  // its res_conv): this kernel loads every x element anyway
  float* xp_out;
  const float* xp_shift;
  int xp_shift_stride, xp_act;
};

// the by-product's stores (16-byte kernel): base + 32-bit byte offset through a buffer descriptor, offsets at or beyond its size are dropped
constexpr unsigned kOutsideXp = 0xFFFFFFF0u;
#if defined(__HIP_DEVICE_COMPILE__)
using xp_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ xp_rsrc xp_make_rsrc(float* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(base, 0, bytes, 0x00020000); }
__device__ __forceinline__ void xp_store4(xp_rsrc r, unsigned voff, float a, float b, float c, float d) {
  using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
  const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
__device__ __forceinline__ void xp_store1(xp_rsrc r, unsigned voff, float a) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), r, voff, 0, 0); }
#else
struct xp_rsrc {};
__device__ __forceinline__ xp_rsrc xp_make_rsrc(float*, unsigned) { return {}; }
__device__ __forceinline__ void xp_store4(xp_rsrc, unsigned, float, float, float, float) {}
__device__ __forceinline__ void xp_store1(xp_rsrc, unsigned, float) {}
#endif

// KS = 4 (small planes: the 8x8 / 16x16 levels, whose grids leave most CUs with one workgroup or none and whose waves then walk
// the whole channel loop one memory round trip at a time -- 768 -> 128 at 8x8x8: 48 round trips, 81 us for 58 MB): the FOUR WAVES of
// a workgroup share the same NS x 32 positions and take every fourth 16-channel group each; their partial sums meet in LDS (fixed
// order: wave 0 + 1 + 2 + 3), and wave (s, m) finishes accumulator (s, m).  Four times the workgroups, a quarter of the round trips.
// XP (with KS = 4 only, where every wave holds other channels of the same positions): the by-product desc.xp_out, as in the 16-byte
// kernel below.
template <int NS, int MSUB, bool ACT, int KS = 1, bool XP = false>
__global__ void __launch_bounds__(256, 2) conv1_fp32_kernel(const K1Args a) {
  static_assert(!XP || KS == 4, "the by-product: channels split over the waves");
  constexpr int CO = 32 * MSUB;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  unsigned id = blockIdx.x;
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int ptile = __builtin_amdgcn_readfirstlane(id % a.ptiles); id /= a.ptiles;
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups);
  const int b = __builtin_amdgcn_readfirstlane(id / a.groups);
  const int co0 = co_tile * CO;
  const long plane = a.plane;
  const int ngroups = a.cin_g / (2 * G);

  unsigned pos[NS];
  bool pok[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const long p = KS == 1 ? (long)ptile * (4 * NS * 32) + (wv * NS + s) * 32 + l31 : (long)ptile * (NS * 32) + s * 32 + l31;
    pok[s] = p < plane;
    pos[s] = (unsigned)(pok[s] ? p : plane - 1);  // clamped loads, no store
  }
  // lane part of every x offset inside a 16-channel group: channel 2j + khalf, position pos[s]
  const unsigned chan_off = (unsigned)khalf * (unsigned)plane;
  const float* wg = a.wp + (long)g * a.cin_g * a.cout_g + (long)khalf * a.cout_g + co0 + l31 * MSUB;

  f32x16 acc[NS][MSUB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  float xr[2][G][NS], wr[2][G][MSUB], shr[2][G], scr[2][G], xsh[2][G];
  const bool xp_on = XP && co_tile == 0;
  const xp_rsrc xpr = xp_make_rsrc(XP ? a.xp_out + (long)b * a.Cin * plane : nullptr, XP ? (unsigned)((long)a.Cin * plane * 4) : 0u);
  auto load_group = [&](int q, int buf) __attribute__((always_inline)) {
    const int cg = g * a.cin_g + q * 2 * G;  // first of 16 channels; they lie in one input segment
    const float* src;
    int cl, segc;
    if (cg < a.seg_c[0]) src = a.seg_x[0], cl = cg, segc = a.seg_c[0];
    else if (cg < a.seg_c[0] + a.seg_c[1]) src = a.seg_x[1], cl = cg - a.seg_c[0], segc = a.seg_c[1];
    else src = a.seg_x[2], cl = cg - a.seg_c[0] - a.seg_c[1], segc = a.seg_c[2];
    const float* base = src + ((long)b * segc + cl) * plane;
    const float* shp = a.in_shift ? a.in_shift + (long)b * a.shift_stride + cg : kZeros16;
    const float* scp = a.in_scale ? a.in_scale + (long)b * a.scale_stride + cg : kOnes16;
    const float* wq = wg + (long)q * 2 * G * a.cout_g;
#pragma unroll
    for (int j = 0; j < G; ++j) {
#pragma unroll
      for (int s = 0; s < NS; ++s) xr[buf][j][s] = base[(unsigned)(2 * j) * (unsigned)plane + chan_off + pos[s]];
      if constexpr (!XP) {      // (XP: a raw input -- the host's condition -- so no prologue constants of its own to hold)
        shr[buf][j] = shp[2 * j + khalf];
        scr[buf][j] = scp[2 * j + khalf];
      } else {
        xsh[buf][j] = (xp_on && a.xp_shift) ? a.xp_shift[(long)b * a.xp_shift_stride + cg + 2 * j + khalf] : 0.f;
      }
      if constexpr (MSUB == 2) {
        const float2 t = *reinterpret_cast<const float2*>(wq + (long)(2 * j) * a.cout_g);
        wr[buf][j][0] = t.x, wr[buf][j][1] = t.y;
      } else {
        wr[buf][j][0] = wq[(long)(2 * j) * a.cout_g];
      }
    }
  };
  auto mfma_group = [&](int buf, int q) __attribute__((always_inline)) {
    if constexpr (XP) if (xp_on) {     // the by-product: channel g * cin_g + q * 16 + 2 j + khalf at this lane's NS positions
      const unsigned c0 = (unsigned)(g * a.cin_g + q * 2 * G + khalf) * (unsigned)plane;
#pragma unroll
      for (int j = 0; j < G; ++j)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float t = xr[buf][j][s] + xsh[buf][j];
          const float ta = tmdiff::silu_f(t);
          xp_store1(xpr, pok[s] ? (c0 + (unsigned)(2 * j) * (unsigned)plane + pos[s]) * 4u : kOutsideXp, a.xp_act ? ta : t);
        }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      float bv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if constexpr (XP) {
          bv[s] = xr[buf][j][s];
        } else {
          float v = xr[buf][j][s] + shr[buf][j];
          if constexpr (ACT) v = tmdiff::silu_f(v);
          bv[s] = v * scr[buf][j];
        }
      }
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int m = 0; m < MSUB; ++m)
          acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[buf][j][m], bv[s], acc[s][m], 0, 0, 0);
    }
  };
  // this wave's groups: q0, q0 + KS, ... (KS = 1: all of them)
  const int q0 = KS == 1 ? 0 : wv;
  if (q0 < ngroups) load_group(q0, 0);
  for (int q = q0; q < ngroups; q += 2 * KS) {
    if (q + KS < ngroups) load_group(q + KS, 1);
    mfma_group(0, q);
    if (q + KS < ngroups) {
      if (q + 2 * KS < ngroups) load_group(q + 2 * KS, 0);
      mfma_group(1, q + KS);
    }
  }
  if constexpr (KS > 1) {
    static_assert(KS == 4 && NS * MSUB <= 4, "one wave per accumulator");
    // partial sums through LDS: red[wave][accumulator][register][lane]; wave i then sums accumulator i over the waves in order
    __shared__ float red[4 * NS * MSUB * 16 * 64];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int m = 0; m < MSUB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wv * NS * MSUB + s * MSUB + m) * 16 + r) * 64 + lane] = acc[s][m][r];
    __syncthreads();
    if (wv >= NS * MSUB) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float t = red[((0 * NS * MSUB + wv) * 16 + r) * 64 + lane];
#pragma unroll
      for (int k = 1; k < 4; ++k) t += red[((k * NS * MSUB + wv) * 16 + r) * 64 + lane];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int m = 0; m < MSUB; ++m)
          if (s * MSUB + m == wv) acc[s][m][r] = t;      // (wv is wave-uniform: a scalar branch)
    }
  }

  // epilogue: col = position, row (r&3) + 8*(r>>2) + 4*khalf = channel
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int cbase = g * a.cout_g + co0 + m * 32 + 4 * khalf;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (KS > 1 && s * MSUB + m != wv) continue;       // (KS = 4: this wave's accumulator only)
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

// 16-byte variant (plane % 4 == 0, 16-byte aligned tensors: every production layer).  A lane owns FOUR consecutive
// positions: one dwordx4 load per channel, and MFMA sub-tile s takes element s of every lane, i.e. its 32 columns are the
// positions {4*lane + s} -- a column permutation the epilogue undoes for free (a lane's four sub-tile results of a row are
// again 4 consecutive positions = one dwordx4 store).  Four times fewer memory instructions than the dword kernel above,
// whose issue rate, not HBM, capped it at ~3.4 TB/s.  Same accumulation order, so the same bits.
constexpr int GV = 4;  // K-steps per register group (8 input channels)

template <int MSUB, bool ACT, bool XP>
__global__ void __launch_bounds__(256, 2) conv1_fp32_vec_kernel(const K1Args a) {
  constexpr int CO = 32 * MSUB, NS = 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;
  unsigned id = blockIdx.x;
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int ptile = __builtin_amdgcn_readfirstlane(id % a.ptiles); id /= a.ptiles;
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups);
  const int b = __builtin_amdgcn_readfirstlane(id / a.groups);
  const int co0 = co_tile * CO;
  const long plane = a.plane;
  const int ngroups = a.cin_g / (2 * GV);

  const long p0l = (long)ptile * 512 + wv * 128 + l31 * 4;      // first of this lane's four positions
  const bool pok = p0l < plane;                                   // plane % 4 == 0: all four or none
  const unsigned p0 = (unsigned)(pok ? p0l : plane - 4);          // clamped loads, no store
  const unsigned chan_off = (unsigned)khalf * (unsigned)plane;
  const float* wg = a.wp + (long)g * a.cin_g * a.cout_g + (long)khalf * a.cout_g + co0 + l31 * MSUB;

  f32x16 acc[NS][MSUB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  float xr[2][GV][NS], wr[2][GV][MSUB], shr[2][GV], scr[2][GV], xsh[2][GV];
  const bool xp_on = XP && co_tile == 0;      // (XP instantiations only: the by-product costs registers)
  const xp_rsrc xpr = xp_make_rsrc(XP ? a.xp_out + (long)b * a.Cin * plane : nullptr, XP ? (unsigned)((long)a.Cin * plane * 4) : 0u);
  auto load_group = [&](int q, int buf) __attribute__((always_inline)) {
    const int cg = g * a.cin_g + q * 2 * GV;  // first of 8 channels; they lie in one input segment
    const float* src;
    int cl, segc;
    if (cg < a.seg_c[0]) src = a.seg_x[0], cl = cg, segc = a.seg_c[0];
    else if (cg < a.seg_c[0] + a.seg_c[1]) src = a.seg_x[1], cl = cg - a.seg_c[0], segc = a.seg_c[1];
    else src = a.seg_x[2], cl = cg - a.seg_c[0] - a.seg_c[1], segc = a.seg_c[2];
    const float* base = src + ((long)b * segc + cl) * plane;
    const float* shp = a.in_shift ? a.in_shift + (long)b * a.shift_stride + cg : kZeros16;
    const float* scp = a.in_scale ? a.in_scale + (long)b * a.scale_stride + cg : kOnes16;
    const float* wq = wg + (long)q * 2 * GV * a.cout_g;
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(base + (unsigned)(2 * j) * (unsigned)plane + chan_off + p0);
      xr[buf][j][0] = t.x, xr[buf][j][1] = t.y, xr[buf][j][2] = t.z, xr[buf][j][3] = t.w;
      shr[buf][j] = shp[2 * j + khalf];
      scr[buf][j] = scp[2 * j + khalf];
      if constexpr (XP) xsh[buf][j] = (xp_on && a.xp_shift) ? a.xp_shift[(long)b * a.xp_shift_stride + cg + 2 * j + khalf] : 0.f;
      if constexpr (MSUB == 2) {
        const float2 w2 = *reinterpret_cast<const float2*>(wq + (long)(2 * j) * a.cout_g);
        wr[buf][j][0] = w2.x, wr[buf][j][1] = w2.y;
      } else {
        wr[buf][j][0] = wq[(long)(2 * j) * a.cout_g];
      }
    }
  };
  auto mfma_group = [&](int buf, int q) __attribute__((always_inline)) {
    if constexpr (XP) if (xp_on) {     // the by-product: channel g * cin_g + q * 8 + 2 j + khalf, this lane's four positions
      // (through a buffer descriptor over sample b: a lane beyond the plane carries an offset outside it and its store is dropped --
      //  no store under a divergent branch, around which the compiler would spill live accumulators: see conv3d_wf.hip's epilogue)
      const unsigned d0 = pok ? (unsigned)(((long)(g * a.cin_g + q * 2 * GV + khalf) * plane + p0) * 4) : kOutsideXp;
#pragma unroll
      for (int j = 0; j < GV; ++j) {
        float u[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const float t = xr[buf][j][s] + xsh[buf][j];
          const float ta = tmdiff::silu_f(t);
          u[s] = a.xp_act ? ta : t;
        }
        xp_store4(xpr, pok ? d0 + (unsigned)(2 * j) * (unsigned)plane * 4u : kOutsideXp, u[0], u[1], u[2], u[3]);
      }
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      float bv[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float v = xr[buf][j][s] + shr[buf][j];
        if constexpr (ACT) v = tmdiff::silu_f(v);
        bv[s] = v * scr[buf][j];
      }
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int m = 0; m < MSUB; ++m)
          acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[buf][j][m], bv[s], acc[s][m], 0, 0, 0);
    }
  };
  load_group(0, 0);
  for (int q = 0; q < ngroups; q += 2) {
    if (q + 1 < ngroups) load_group(q + 1, 1);
    mfma_group(0, q);
    if (q + 1 < ngroups) {
      if (q + 2 < ngroups) load_group(q + 2, 0);
      mfma_group(1, q + 1);
    }
  }

  // epilogue: column l31 of sub-tile s = position p0 + s; row (r&3) + 8*(r>>2) + 4*khalf = channel
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int cbase = g * a.cout_g + co0 + m * 32 + 4 * khalf;
    const long obase = ((long)b * a.Cout + cbase) * plane + p0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {     // eight rows at a time: bounded register use
      float4 res[8];
      float bs[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = half * 8 + k, row = (r & 3) + 8 * (r >> 2);
        res[k] = a.residual ? *reinterpret_cast<const float4*>(a.residual + obase + row * plane) : make_float4(0.f, 0.f, 0.f, 0.f);
        bs[k] = a.bias ? a.bias[cbase + row] * a.bias_scale : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = half * 8 + k, row = (r & 3) + 8 * (r >> 2);
        if (pok)
          *reinterpret_cast<float4*>(a.y + obase + row * plane) =
              make_float4((acc[0][m][r] + bs[k] + res[k].x) * a.out_scale, (acc[1][m][r] + bs[k] + res[k].y) * a.out_scale,
                          (acc[2][m][r] + bs[k] + res[k].z) * a.out_scale, (acc[3][m][r] + bs[k] + res[k].w) * a.out_scale);
      }
    }
  }
}

template <int MSUB>
int launch_vec(K1Args& a, int in_act, hipStream_t st) {
  a.ptiles = (int)((a.plane + 511) / 512);
  a.tiles_co = a.cout_g / (32 * MSUB);
  const long blocks = (long)a.B * a.groups * a.ptiles * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv1: grid of %ld blocks", blocks);
  if (a.xp_out) {          // (the by-product goes with a raw input: res_conv has no prologue of its own)
    if (in_act) return tmdiff::fail(TMDIFF_E_UNSUPPORTED, "conv1: xp_out with an activated input");
    if ((long)a.Cin * a.plane >= (1L << 30)) return tmdiff::fail(TMDIFF_E_UNSUPPORTED, "conv1: xp_out sample too large for 32-bit byte offsets");
    conv1_fp32_vec_kernel<MSUB, false, true><<<(unsigned)blocks, 256, 0, st>>>(a);
  } else if (in_act) conv1_fp32_vec_kernel<MSUB, true, false><<<(unsigned)blocks, 256, 0, st>>>(a);
  else conv1_fp32_vec_kernel<MSUB, false, false><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_fwd (1x1x1 bandwidth kernel, 16-byte)");
}

template <int NS, int MSUB, int KS = 1>
int launch(K1Args& a, int in_act, hipStream_t st) {
  a.ptiles = (int)((a.plane + (KS == 1 ? 4 : 1) * NS * 32 - 1) / ((KS == 1 ? 4 : 1) * NS * 32));
  a.tiles_co = a.cout_g / (32 * MSUB);
  const long blocks = (long)a.B * a.groups * a.ptiles * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv1: grid of %ld blocks", blocks);
  if (a.xp_out) {
    if constexpr (KS == 4) {
      if (in_act || a.in_shift || a.in_scale) return tmdiff::fail(TMDIFF_E_UNSUPPORTED, "conv1: xp_out on a small grid goes with a raw input (no prologue of its own)");
      if ((long)a.Cin * a.plane >= (1L << 30)) return tmdiff::fail(TMDIFF_E_UNSUPPORTED, "conv1: xp_out sample too large for 32-bit byte offsets");
      conv1_fp32_kernel<NS, MSUB, false, 4, true><<<(unsigned)blocks, 256, 0, st>>>(a);
      return tmdiff::check_launch("conv3d_fwd (1x1x1 bandwidth kernel, channels over the waves + x')");
    } else {
      return tmdiff::fail(TMDIFF_E_UNSUPPORTED, "conv1: xp_out needs the 16-byte kernel or the small-grid kernel");
    }
  }
  if (in_act) conv1_fp32_kernel<NS, MSUB, true, KS><<<(unsigned)blocks, 256, 0, st>>>(a);
  else conv1_fp32_kernel<NS, MSUB, false, KS><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_fwd (1x1x1 bandwidth kernel)");
}

}  // namespace

// 1x1x1 forward through the bandwidth kernel.  Returns TMDIFF_E_UNSUPPORTED (without touching the error string)
// for shapes it does not take; tmdiff_conv3d_fwd then uses the generic kernel.  `d` has been validated by the caller.
int tmdiff::conv1_fp32_try(const tmdiff_conv3d_desc* d, hipStream_t st, bool dry) {
  if (d->ksize != 1 || d->in_mask || d->drop_p > 0.f || d->y2 || !d->y) return TMDIFF_E_UNSUPPORTED;
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  if (cin_g % (2 * G) || cout_g % 32) return TMDIFF_E_UNSUPPORTED;
  for (int i = 0; i < d->nseg; ++i)
    if (d->seg_c[i] % (2 * G)) return TMDIFF_E_UNSUPPORTED;
  const long plane = (long)d->N * d->H * d->W;
  if (plane * 2 * G >= (1L << 31)) return TMDIFF_E_UNSUPPORTED;
  K1Args a;
  a.B = d->B; a.Cin = d->Cin; a.Cout = d->Cout; a.cin_g = cin_g; a.cout_g = cout_g; a.groups = d->groups;
  for (int i = 0; i < 3; ++i) {
    a.seg_c[i] = i < d->nseg ? d->seg_c[i] : (1 << 28);
    a.seg_x[i] = i < d->nseg ? d->seg_x[i] : d->seg_x[0];
  }
  a.wp = d->w_packed; a.bias = d->bias; a.bias_scale = d->bias_scale;
  a.in_shift = d->in_shift; a.in_scale = d->in_scale;
  a.shift_stride = d->in_shift_stride > 0 ? d->in_shift_stride : (d->in_shift_stride < 0 ? 0 : d->Cin);
  a.scale_stride = d->in_scale_stride > 0 ? d->in_scale_stride : (d->in_scale_stride < 0 ? 0 : d->Cin);
  a.residual = d->residual; a.out_scale = d->out_scale; a.y = d->y;
  a.plane = plane;
  a.xp_out = d->xp_out; a.xp_shift = d->xp_shift; a.xp_act = d->xp_act;
  a.xp_shift_stride = d->xp_shift_stride > 0 ? d->xp_shift_stride : (d->xp_shift_stride < 0 ? 0 : d->Cin);
  // channel tiles follow the weight packing: 64-channel interleaved rows when cout_g % 64 == 0
  // (its 512-position tiles must still fill the chip: the 8x8x8 level keeps the dword kernel's 256-position tiles)
  const long blocks_vec = (long)d->B * d->groups * ((plane + 511) / 512) * (cout_g % 64 == 0 ? cout_g / 64 : cout_g / 32);
  static const bool force_vec = getenv("TMDIFF_CONV1_VEC") != nullptr;     // experiments / tests: wherever it is legal
  bool vec = plane % 4 == 0 && (blocks_vec >= 512 || force_vec) && aligned16(d->y) && aligned16(d->residual) && aligned16(d->xp_out);
  for (int i = 0; i < d->nseg; ++i) vec = vec && aligned16(d->seg_x[i]);
  static const bool no_vec = getenv("TMDIFF_CONV1_DWORD") != nullptr;      // experiments: the dword kernel everywhere
  // small grids (fewer than two workgroups per CU of 256-position tiles) with at least 8 channel groups: the four waves of a
  // workgroup split the channels (KS = 4).  TMDIFF_CONV1_KSPLIT=0: never (experiments; tests force either form).
  static const int ksplit_mode = [] {
    const char* e = getenv("TMDIFF_CONV1_KSPLIT");
    return e ? atoi(e) : -1;      // -1: by grid size, 0: never, 1: wherever it is legal
  }();
  const long blocks_dword = (long)d->B * d->groups * ((plane + 255) / 256) * (cout_g % 64 == 0 ? cout_g / 64 : cout_g / 32);
  const bool ks = ksplit_mode != 0 && cin_g / (2 * G) >= 8 && (ksplit_mode == 1 || blocks_dword < 512);
  // (xp_supported: a raw input on the 16-byte kernel or on the small-grid kernel)
  if (dry) return (((vec && !no_vec) || (ks && !d->in_shift && !d->in_scale)) && !d->in_act && (long)d->Cin * plane < (1L << 30)) ? TMDIFF_OK : TMDIFF_E_UNSUPPORTED;
  if (vec && !no_vec) return cout_g % 64 == 0 ? launch_vec<2>(a, d->in_act, st) : launch_vec<1>(a, d->in_act, st);
  if (cout_g % 64 == 0) return ks ? launch<2, 2, 4>(a, d->in_act, st) : launch<2, 2>(a, d->in_act, st);
  return ks ? launch<2, 1, 4>(a, d->in_act, st) : launch<2, 1>(a, d->in_act, st);
}

// 1 when tmdiff_conv3d_fwd would write the by-product d->xp_out for this descriptor (the 16-byte bandwidth kernel takes it), else 0.
extern "C" int tmdiff_conv3d_fwd_xp_supported(const tmdiff_conv3d_desc* d) {
  if (!d || !d->xp_out || d->x_bf16 || d->groups <= 0 || d->Cin <= 0 || d->Cout <= 0 || d->Cin % d->groups || d->Cout % d->groups ||
      d->nseg < 1 || d->nseg > 3 || d->B <= 0 || !d->w_packed)
    return 0;
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    if (!d->seg_x[i] || d->seg_c[i] <= 0) return 0;
    csum += d->seg_c[i];
  }
  return csum == d->Cin && tmdiff::conv1_fp32_try(d, nullptr, true) == TMDIFF_OK ? 1 : 0;
}
