// 3x3x3 convolution, Winograd F(4,3) along the band axis, with the INPUT TRANSFORM INSIDE THE KERNEL ("wf"): the kernel
// reads the plain convolution input x' (prologue already applied by its producer) -- there is no transformed copy V of the
// input in HBM and no transform pass in front of the convolution (conv3d_wino.hip has both: 4 B read + 6.6 B written per
// input element and launch, 10 % of the benchmark step).
//
// Mathematics as conv3d_wino.hip: for a tile of 4 output bands and its 6 input bands d (one band of halo either side, zero
// outside the image), v = B^T d per element, u = G g per (ci, co, dh, dw), m_k = sum_{ci,dh,dw} u_k v_k (six 3x3 convolutions
// in (h, w)), y = A^T m: 54 instead of 4 x 27 multiply-adds.  The whole band axis lies inside one workgroup (N = 8: two
// tiles; N = 4: one), so the band halo never crosses workgroups.
//
// Workgroup = 256 threads = 4 waves, 256 output positions (TT tiles x TH x TW) x 32 output channels; two workgroups per CU.
// Every wave holds ALL six planes of its 64 positions (2 sub-tiles of 32) = 12 accumulators, so the output transform
// y = A^T m happens in registers: no exchange between waves, no barrier after the last chunk, every wave runs the shared
// dwordx4 epilogue on its own.
//
// Per chunk of KC = 2 input channels, LDS holds
//   raw  [region (kc, row)][band][RW]   the haloed box of x' itself: rows h0-1 .. h0+TH, columns w0-4 .. w0+TW+3 (the aligned
//                                       superset of w0-1 .. w0+TW: 16-byte LDS-DMA pieces, whole quads inside or outside
//                                       the image, no bounds checks inside a quad), ONE buffer;
//   V    [kc][tile][plane][row][PW]     its transform (columns w0-1 .. w0+TW), two stages;
//   U    [kc][(dh,dw)][plane][32]       the weight slab, two stages (LDS-DMA).
// While the MFMAs of chunk c run from stage c % 2, each wave (1) transforms the regions of raw(c+1) that IT fetched into V of
// the other stage -- 6 ds_read_b128, ~90 VALU operations and 24 ds_write_b32 per wave and chunk, placed between the MFMAs --
// then (2) requests raw(c+2) into the same regions, and (3) requests the weights of chunk c+1.  A region is written (DMA) and
// read (transform) by one wave only, so the single raw buffer needs no barrier of its own: one barrier per chunk.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "epilogue.h"

#ifndef TMDIFF_WF_ABLATE
#define TMDIFF_WF_ABLATE 0       // timing experiments (WRONG results): 1 = no input transform in the loop, 2 = ... and no raw DMA,
#endif                           // 3 = ... and no weight DMA
#ifndef TMDIFF_WF_STAMPS
#define TMDIFF_WF_STAMPS 0       // diagnostic build: per-wave s_memrealtime stamps (tools/wino_stamps.py)
#endif

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

struct WfArgs {
  int B, N, H, W;
  int Cin, Cout, cin_g, cout_g, groups;
  const float* x;       // x' [B, Cin, N, H, W] ...
  const float* xgrp[3]; // ... or (grouped convolution whose three segments ARE its three groups' inputs) one tensor
  int grouped_segs;     //     [B, cin_g, N, H, W] per group: the concatenation never has to exist
  const float* wp;      // U packed [g][ci][9][6][co], natural column order (tmdiff_conv3d_wino_pack_weights, mode | 2)
  const float* bias;
  float bias_scale;
  const float* residual;
  float out_scale;
  float* y;
  float* y2;
  const float* y2_shift;
  const float* y2_scale;
  int y2_shift_stride, y2_scale_stride, y2_act;
  int y2_s2d;                 // the second output in "space to depth" form (see epilogue_wf)
  int tiles_h, tiles_w, tiles_co;
  unsigned total_blocks;
  int vec4;
  int ksplit, split_chunks;   // split-K (small grids): ksplit ranges of split_chunks chunks each; 1 = no split
  float* part;                // ... whose partial outputs go to part[split][B][Cout][plane] (splitk_reduce_kernel finishes)
  unsigned long long* stamps;
  int stagger;
  unsigned first_round;
  // folded 1x1x1 "residual convolution" (desc.rc_*): y += rc_w^T rc_x (+ its bias, folded into `bias` by the host)
  float* yll;                 // optional third output: the halved LL band of y, [B, Cout, N, H/2, W/2] (epilogue_wf; y itself not written)
  float* yhi[3];              // ... with these: the whole Haar transform of y -- LH, HL, HH beside an LL band that went through the
                              //     consumer's prologue (the y2 constants); no full-resolution output at all (desc.y_hi)
  const float* rc_x;          // [B, rc_cin, N, H, W] or NULL
  const float* rc_w;          // the weight as PyTorch holds it: [Cout][rc_cin]
  int rc_cin;
};

// The kernel's (only) argument where it lies in the kernel-argument segment: loads through this pointer are scalar loads at the
// place of use; the compiler neither shares them with the argument values it already holds nor keeps them afterwards.
#if defined(__HIP_DEVICE_COMPILE__)
using KArgs = const WfArgs __attribute__((address_space(4)))*;
__device__ __forceinline__ KArgs kargs() {
  KArgs k = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(k));
  return k;
}
#else
using KArgs = const WfArgs*;
__device__ __forceinline__ KArgs kargs() { return nullptr; }
#endif

// LDS-DMA through a buffer descriptor: 16 bytes per lane from base + voff (bytes) to dst + 16 * lane.  An offset at or beyond
// the descriptor's size -- kOutside -- reads as zero: the zero padding of the convolution costs a select, not a second
// source pointer (the address is one 32-bit register per lane; the base lives in scalar registers).
constexpr unsigned kOutside = 0xFFFFFFF0u;
// The compiler hoists every address that does not change from chunk to chunk out of the chunk loop and then, with 192 of the
// 256 registers holding accumulators, spills it -- a scratch reload (and its vmcnt(0), which also waits for the LDS-DMA in
// flight) in the middle of the MFMA stream.  opaque() hides a value's loop invariance: the few VALU operations that derive an
// address from it are redone where they are used.
template <class T>
__device__ __forceinline__ T opaque(T v) {
  asm volatile("" : "+v"(v));
  return v;
}
#if defined(__HIP_DEVICE_COMPILE__)  // the builtins exist in the device pass only
using buf_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ buf_rsrc make_rsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void dma_b128(buf_rsrc r, unsigned voff, float* dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, 0, 0, 0);
}
// one dword per lane from base + voff + soff (soff uniform): the address is a 32-bit register, nothing 64-bit per lane
__device__ __forceinline__ float buf_load(buf_rsrc r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// the epilogue's quads: 16 / 8 bytes per lane at base + voff; a lane at kOutside reads zero / stores nothing
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
__device__ __forceinline__ float4 buf_load4(buf_rsrc r, unsigned voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void buf_store4(buf_rsrc r, unsigned voff, float a, float b, float c, float d) {
  const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, 0, 0);
}
__device__ __forceinline__ void buf_store2(buf_rsrc r, unsigned voff, float a, float b) {
  const u32x2 v = {__float_as_uint(a), __float_as_uint(b)};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, 0, 0);
}
#else
struct buf_rsrc {};
__device__ __forceinline__ buf_rsrc make_rsrc(const float*, unsigned) { return {}; }
__device__ __forceinline__ void dma_b128(buf_rsrc, unsigned, float*) {}
__device__ __forceinline__ float buf_load(buf_rsrc, unsigned, unsigned) { return 0.f; }
__device__ __forceinline__ float4 buf_load4(buf_rsrc, unsigned) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void buf_store4(buf_rsrc, unsigned, float, float, float, float) {}
__device__ __forceinline__ void buf_store2(buf_rsrc, unsigned, float, float) {}
#endif

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, k = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// see conv3d_wino.hip: the second resident workgroup of a CU (LDS allocation not at 0) may wait once (experiments)
__device__ __forceinline__ void stagger_start(int cycles, unsigned first_round) {
  if (cycles <= 0 || blockIdx.x >= first_round) return;
  const unsigned lds_alloc = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (11 << 11));
  if (lds_alloc == 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)cycles) __builtin_amdgcn_s_sleep(32);
}

// PAIR (planes of exactly 8 columns -- the 8x8 level): the tile's 16 columns are TWO IMAGES side by side, 8 columns each.  The
// raw box then needs no halo columns at all (left and right of an 8-column image is the zero padding of the convolution), a V
// row holds [0, image 0, 0, 0, image 1, 0] (20 columns: each image its own zero halo, written once and never touched again),
// and a lane of the second image reads its operands 2 columns further right.  An 8x16 tile over one 8-column plane would be
// half empty.
// LLM (`Conv_0` + halved LL band as one convolution, with Winograd along the bands on top): the input is the producer's
// space-to-depth second output X[(c, ph, pw)][n][i][j] = x'[c][n][2i + ph][2j + pw], four virtual channels per real one at half
// the resolution.  The composed 3x4x4 stride-2 kernel (conv3d_ll.hip) is, per virtual channel, a stride-1 convolution with
// 2 x 2 of the 3 x 3 taps: rows {0, +1} (ph = 0) or {-1, 0} (ph = 1), columns likewise by pw.  A chunk = the two virtual
// channels (pw = 0, 1) of one (c, ph): everything up to the V planes is the plain kernel on X; the K loop has 4 taps x 6 planes
// = 24 steps instead of 54 (row taps by the chunk's parity = its stage, column taps by the lane's K half = pw), i.e.
// 16 x 6 / 4 = 24 multiply-adds per output where conv3d_ll executes 48 and the convolution + DWT pair 108.
template <int TT, int TH, int TW, bool PAIR = false, bool LLM = false>
struct GeoF {
  static constexpr int NP = 6, MO = 4, NB = TT * MO, KC = 2, CO = 32, NS = 2;
  static constexpr int HH = TH + 2;
  static constexpr int RW = PAIR ? TW : TW + 8, RQ = RW / 4;   // raw row: columns w0-4 .. w0+TW+3 as RQ aligned quads (PAIR: 2 x 8 columns)
  static constexpr int PW = PAIR ? TW + 4 : TW + 2;            // V row: columns w0-1 .. w0+TW (PAIR: two images with their own halo columns)
  static constexpr int REG_QUADS = NB * RQ;               // one region = (kc, row): every band's raw row
  static constexpr int PLANES = PAIR ? 32 : 48;           // lanes of a DMA piece
  static constexpr int RPP = PLANES / REG_QUADS;          // regions per piece
  static constexpr int NREG = KC * HH;
  static constexpr int XP = (NREG + RPP - 1) / RPP;       // raw pieces per chunk
  static constexpr int XK = (XP + 3) / 4;                 // ... per wave
  static constexpr int PSTRIDE = PLANES * 4 + 4;          // floats between pieces (+ 4: consecutive pieces of a wave 16 banks apart)
  static constexpr int RAW_FLOATS = XP * PSTRIDE;
  // one band tile's planes; with two tiles the stride is padded to 16 mod 32 banks: a sub-tile of MFMA columns is then 16
  // positions of one row in BOTH band tiles, and its operand read touches every bank once (two rows of one tile, 18 floats
  // apart, would share two banks: one more LDS cycle on every operand read)
  static constexpr bool SPLIT = TT == 2;
  static constexpr int TSTRIDE = SPLIT ? ((NP * HH * PW + 15) / 32) * 32 + 16 : NP * HH * PW;
  static constexpr int V_ELEMS = (TT - 1) * TSTRIDE + NP * HH * PW;       // one channel
  static_assert(!SPLIT || (TSTRIDE % 32 == 16 && TSTRIDE >= NP * HH * PW), "tile stride");
  static constexpr int V_FLOATS = KC * V_ELEMS;
  static constexpr int W_TAPS = (LLM ? 4 : 9) * NP;
  static constexpr int W_FLOATS = KC * W_TAPS * CO;
  static constexpr int W_UNITS = W_FLOATS / 4;
  static constexpr int WP = (W_UNITS + 63) / 64, WK = (WP + 3) / 4;
  static constexpr int STAGE = V_FLOATS + WP * 256;
  static constexpr int TCOLS = PAIR ? TW : PW;            // V columns the transform writes per (region, band tile)
  static_assert(REG_QUADS * RPP == PLANES, "a piece holds whole regions");
  static_assert(RPP * TT == 2, "a piece holds two (region, band tile) pairs");
  static_assert(TT * TH * TW == 4 * NS * 32, "4 waves x NS sub-tiles x 32 positions");
  static_assert(TW == 16, "a sub-tile is two rows of 16 positions");
  static_assert(!PAIR || (TT == 2 && TH == 8), "pair mode: two band tiles of 8 rows");
  static_assert(V_FLOATS % 4 == 0, "stage alignment");
};

// Epilogue of a wave: its 8 output blocks (2 sub-tiles x 4 bands, each 32 channels x 32 positions in the MFMA D layout: a lane
// = one position, 16 channels) leave as in epilogue.h -- through a wave-private 4 KB LDS tile T[channel][position]
// (ds_write_b32 in the D layout, ds_read_b128 along the positions), lane (tc, tq) = (lane >> 3, lane & 7) then owns channels
// tc + 8 j and four consecutive positions: bias, residual, scale, the consumer's prologue, dwordx4 stores -- with two
// differences.  The residual quads are requested THREE blocks ahead (the output transform has already freed a third of the
// accumulators, so there are registers for it): with one block of look-ahead a wave waited out most of an HBM round trip
// per block, eight times per tile: 22 of the 33 us of a tile's epilogue at 32 input channels (tools/wino_stamps.py).
// And every output / residual tensor is addressed through a BUFFER DESCRIPTOR over this tile's 32 channels of sample b (PAIR:
// up to the same channels of sample b + 1): a lane's address is one 32-bit byte offset, and a lane outside the image carries
// the offset kOutside -- its loads return zero, its stores are dropped by the bounds check.  No store sits under a branch:
// ragged tiles run the code of full ones, the wave never diverges (with 190 registers of live accumulators the compiler
// spills around divergent stores, and one experimental build computed ragged tiles wrongly in exactly the lanes such stores
// mask off: profiles/r04_wf_experiments.txt), and no address is 64-bit arithmetic per lane.
// (PAIR: the lane's four positions belong to image b or b + 1 -- pimg_ok says whether the second exists -- and the second
// output's per-(b, channel) shift / scale come in both versions, sh2_l / sc2_l for b and sh2_m / sc2_m for b + 1)
template <bool Y, bool RES, bool Y2, int TT, int TH, int TW, bool PAIR>
__device__ __forceinline__ void epilogue_wf(KArgs a, f32x16 (&out)[8], float bias_l, float sh2_l, float sc2_l, int b, int g,
                                            int co0, int h0, int w0, int wv, int lane, long plane, float* T, float* ydst,
                                            float oscale, float sh2_m = 0.f, float sc2_m = 1.f, bool pimg_ok = true) {
#pragma clang fp contract(off)   // (every instantiation rounds alike: without y the compiler would fuse "* scale" and "+ shift")
  constexpr int MO = 4, DEPTH = 3;
  const int l31 = lane & 31, khalf = lane >> 5, tq = lane & 7, tc = lane >> 3;
  const int hw = a->H * a->W;
  // descriptors: 32 channels of sample b from channel g * cout_g + co0 on (PAIR and a second image: up to its 32 channels)
  const long cb = ((long)b * a->Cout + g * a->cout_g + co0) * plane;
  const unsigned span = (unsigned)(((PAIR && pimg_ok ? (long)a->Cout * plane : 0) + 32 * plane) * 4);
  const buf_rsrc ry = make_rsrc(ydst ? ydst + cb : nullptr, ydst ? span : 0);
  const buf_rsrc rr = make_rsrc(a->residual ? a->residual + cb : nullptr, a->residual ? span : 0);
  const buf_rsrc r2 = make_rsrc(a->y2 ? a->y2 + cb : nullptr, a->y2 ? span : 0);
  unsigned toff[2], s2off[2];     // byte offsets of the lane's quad inside the descriptor (channel tc, band 0 of its band tile)
  const unsigned qplane4 = (unsigned)plane;                 // a quarter plane in bytes
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    // the lane's four positions 4 tq .. 4 tq + 3 of sub-tile s (see the operand offsets of the kernel): band tile, row, column
    const int bt = TT == 2 ? tq >> 2 : 0;
    const int h = h0 + (TT == 2 ? 2 * wv + s : 4 * wv + 2 * s + (tq >> 2));
    bool ok;
    int e, e2;
    if constexpr (PAIR) {      // columns 0-7 of the tile: image b, 8-15: image b + 1 (a whole sample further on)
      const int img = (tq & 3) >> 1, w = 4 * (tq & 1);
      ok = h < a->H && (img == 0 || pimg_ok);
      e = img * a->Cout * (int)plane + (MO * bt) * hw + h * a->W + w;
      e2 = img * a->Cout * (int)plane + (h & 1) * 2 * (int)(plane >> 2) + (MO * bt) * (hw >> 2) + (h >> 1) * (a->W >> 1) + (w >> 1);
    } else {
      const int w = w0 + 4 * (tq & 3);
      ok = h < a->H && w < a->W;                    // (W % 4 == 0: the four positions stand or fall together)
      e = (MO * bt) * hw + h * a->W + w;
      // space-to-depth form of the second output: channel (co, ph, pw) = 4 co + 2 ph + pw at half the resolution -- the channel
      // base is the same (4 channels of a quarter plane each); the lane's four columns are two of either column parity
      e2 = (h & 1) * 2 * (int)(plane >> 2) + (MO * bt) * (hw >> 2) + (h >> 1) * (a->W >> 1) + (w >> 1);
    }
    toff[s] = ok ? (unsigned)((tc * (int)plane + e) * 4) : kOutside;
    s2off[s] = ok ? (unsigned)((tc * (int)plane + e2) * 4) : kOutside;
  }
  // (offset of channel tc + 8 j, band n from the lane's base: added only where the base is inside -- kOutside + anything would wrap)
  auto at = [&](unsigned base, unsigned add) __attribute__((always_inline)) { return base >= kOutside ? kOutside : base + add; };
  float bias_t[4], sh2_t[4], sc2_t[4];
  const bool second = PAIR && ((tq & 3) >> 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bias_t[j] = tmdiff::lane_value(bias_l, tc + 8 * j);
    if constexpr (Y2) {
      sh2_t[j] = tmdiff::lane_value(sh2_l, tc + 8 * j), sc2_t[j] = tmdiff::lane_value(sc2_l, tc + 8 * j);
      if constexpr (PAIR) {
        const float s1 = tmdiff::lane_value(sh2_m, tc + 8 * j), c1 = tmdiff::lane_value(sc2_m, tc + 8 * j);
        sh2_t[j] = second ? s1 : sh2_t[j], sc2_t[j] = second ? c1 : sc2_t[j];
      }
    }
  }
  // The wave's eight blocks are taken in the order (band n, sub-tile s = 0, 1): step p is block i = s * MO + n with s = p & 1,
  // n = p >> 1 -- the two rows 2 wv, 2 wv + 1 of one band (TT == 2) follow each other, which is what the LL output below needs.
  // Third output (a->yll; only where y itself is not wanted, TT == 2, not PAIR): the HALVED LL BAND of y, (a + b + c + d) / 4 over
  // the 2 x 2 pixel block -- what a down block's Conv_2 path reads of its ResBlock's output (Hyper_unet_general.py:374, :390, :396):
  // a lane holds four columns of row 2 wv of a band, then the same four of row 2 wv + 1: two LL values per channel, one float2.
  float4 rs[DEPTH + 1][4];          // step p in slot p % (DEPTH + 1)
  auto load_res = [&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value, s = p & 1, n = p >> 1;
#pragma unroll
    for (int j = 0; j < 4; ++j)     // (outside the image: zero, never stored)
      rs[p % (DEPTH + 1)][j] = buf_load4(rr, at(toff[s], (unsigned)((8 * j * (int)plane + n * hw) * 4)));
  };
  constexpr bool LL_OK = !Y && TT == 2 && !PAIR;
  const bool want_ll = LL_OK && a->yll != nullptr;
  // the quarter-size outputs: the same 32 channels, a quarter plane each (the LL pair of a lane: both rows inside or neither)
  const long cbq = cb >> 2;
  const unsigned spanq = (unsigned)(32 * plane);
  const buf_rsrc rll = make_rsrc(want_ll ? a->yll + cbq : nullptr, want_ll ? spanq : 0);
  const bool want_hi = want_ll && a->yhi[0] != nullptr;
  const buf_rsrc rh0 = make_rsrc(want_hi ? a->yhi[0] + cbq : nullptr, want_hi ? spanq : 0);
  const buf_rsrc rh1 = make_rsrc(want_hi ? a->yhi[1] + cbq : nullptr, want_hi ? spanq : 0);
  const buf_rsrc rh2 = make_rsrc(want_hi ? a->yhi[2] + cbq : nullptr, want_hi ? spanq : 0);
  // (s2off[0] of row 2 wv, whose row-parity term is zero, IS the quad's place in a quarter plane; channel stride plane / 4)
  const unsigned lloff = (toff[0] >= kOutside || toff[1] >= kOutside) ? kOutside : s2off[0] - (unsigned)(tc * (int)plane * 4) + (unsigned)(tc * (int)plane);
  float vk[4][4];                   // row 2 wv of the current band (LL output only)
  if constexpr (RES) static_for<0, DEPTH>([&](auto pc) __attribute__((always_inline)) { load_res(pc); });
  static_for<0, 8>([&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value, s = p & 1, n = p >> 1, i = s * MO + n;
    if constexpr (RES && p + DEPTH < 8) load_res(std::integral_constant<int, p + DEPTH>{});
#pragma unroll
    for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * khalf) * 32 + l31] = out[i][r];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 t = *reinterpret_cast<const float4*>(T + (tc + 8 * j) * 32 + tq * 4);
      float v[4] = {t.x, t.y, t.z, t.w};
      float q[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (RES) {
        const float4 rq = rs[p % (DEPTH + 1)][j];
        q[0] = rq.x, q[1] = rq.y, q[2] = rq.z, q[3] = rq.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (v[e] + bias_t[j] + q[e]) * oscale;   // as the scalar epilogue
      const unsigned cj = (unsigned)((8 * j * (int)plane + n * hw) * 4);
      if constexpr (Y) buf_store4(ry, at(toff[s], cj), v[0], v[1], v[2], v[3]);
      if constexpr (LL_OK) {
        if (want_ll) {      // (uniform)
          if constexpr (s == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vk[j][e] = v[e];
          } else {
            // band MO * bt + n, row (h0 + 2 wv) / 2, columns (w0 + 4 (tq & 3)) / 2 and the next of the quarter-size plane
            const unsigned ol = at(lloff, (unsigned)((8 * j * (int)(plane >> 2) + n * (hw >> 2)) * 4));
            const float ll0 = ((vk[j][0] + vk[j][1]) + (v[0] + v[1])) * 0.25f, ll1 = ((vk[j][2] + vk[j][3]) + (v[2] + v[3])) * 0.25f;
            if (want_hi) {  // (uniform)
              // the whole transform of the 2 x 2 blocks (a b / c d): LL / 2 through the consumer's prologue, LH = (a - b + c - d) / 2,
              // HL = (a + b - c - d) / 2, HH = (a - b - c + d) / 2 (DWT_IDWT_Functions.py:47-57 in closed form)
              if constexpr (Y2) {
                const float x0 = ll0 + sh2_t[j], x1 = ll1 + sh2_t[j];
                const float a0 = tmdiff::silu_f(x0), a1 = tmdiff::silu_f(x1);
                buf_store2(rll, ol, (a->y2_act ? a0 : x0) * sc2_t[j], (a->y2_act ? a1 : x1) * sc2_t[j]);
                buf_store2(rh0, ol, ((vk[j][0] - vk[j][1]) + (v[0] - v[1])) * 0.5f, ((vk[j][2] - vk[j][3]) + (v[2] - v[3])) * 0.5f);
                buf_store2(rh1, ol, ((vk[j][0] + vk[j][1]) - (v[0] + v[1])) * 0.5f, ((vk[j][2] + vk[j][3]) - (v[2] + v[3])) * 0.5f);
                buf_store2(rh2, ol, ((vk[j][0] - vk[j][1]) - (v[0] - v[1])) * 0.5f, ((vk[j][2] - vk[j][3]) - (v[2] - v[3])) * 0.5f);
              }
            } else {
              buf_store2(rll, ol, ll0, ll1);
            }
          }
        }
      }
      if constexpr (Y2) if (a->y2) {       // (uniform)
        float u[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x = v[e] + sh2_t[j];
          const float xa = tmdiff::silu_f(x);
          u[e] = (a->y2_act ? xa : x) * sc2_t[j];
        }
        if (a->y2_s2d) {                   // (uniform)
          const unsigned o2 = at(s2off[s], (unsigned)((8 * j * (int)plane + n * (hw >> 2)) * 4));
          buf_store2(r2, o2, u[0], u[2]);
          buf_store2(r2, at(o2, qplane4), u[1], u[3]);
        } else {
          buf_store4(r2, at(toff[s], cj), u[0], u[1], u[2], u[3]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  });
}

template <int TT, int TH, int TW, bool PAIR, bool LLM>
__global__ void __launch_bounds__(256, 2) conv3d_wf_kernel(const WfArgs a) {
  using G = GeoF<TT, TH, TW, PAIR, LLM>;
  constexpr int NP = G::NP, MO = G::MO, KC = G::KC, CO = G::CO, NS = G::NS, HH = G::HH, RQ = G::RQ, PW = G::PW;
  constexpr int W_TAPS = G::W_TAPS;
  __shared__ __attribute__((aligned(16))) float lds[2 * G::STAGE + G::RAW_FLOATS];
  float* const st0 = lds;
  float* const st1 = lds + G::STAGE;
  float* const raw = lds + 2 * G::STAGE;
  static_assert(2 * G::STAGE + G::RAW_FLOATS >= 4 * 1024, "the epilogue borrows 4 KB of LDS per wave");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
#if TMDIFF_WF_STAMPS
  unsigned long long stamp_t[6];
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime();
  stamp_t[0] = __builtin_amdgcn_s_memrealtime();
#define WF_STAMP(i) stamp_t[i] = __builtin_amdgcn_s_memrealtime()
#else
#define WF_STAMP(i)
#endif
  stagger_start(a.stagger, a.first_round);
  WF_STAMP(1);

  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int split = __builtin_amdgcn_readfirstlane(id % a.ksplit); id /= a.ksplit;      // (1: no split-K)
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int tw_i = __builtin_amdgcn_readfirstlane(id % a.tiles_w); id /= a.tiles_w;
  const int th_i = __builtin_amdgcn_readfirstlane(id % a.tiles_h); id /= a.tiles_h;
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups);
  const int b = __builtin_amdgcn_readfirstlane(id / a.groups) * (PAIR ? 2 : 1);     // (PAIR: images b and b + 1)
  const bool pimg_ok = !PAIR || b + 1 < a.B;
  const int h0 = th_i * TH, w0 = tw_i * TW;
  const int co0 = co_tile * CO;
  const int hw = a.H * a.W;
  const long plane = (long)a.N * hw;                           // one channel of x' / y
  const int nchunks = a.split_chunks;                          // this workgroup's range of input-channel chunks
  const float* xg = (a.grouped_segs ? (g == 0 ? a.xgrp[0] : (g == 1 ? a.xgrp[1] : a.xgrp[2])) + (long)b * a.cin_g * plane
                                    : a.x + ((long)b * a.Cin + (long)g * a.cin_g) * plane) + (long)split * nchunks * KC * plane;
  const float* wg = a.wp + ((long)g * a.cin_g + (long)split * nchunks * KC) * W_TAPS * a.cout_g + co0;

  // ---- DMA: raw pieces (48 lanes: RPP regions of NB bands x RQ quads) and weight pieces ------------------------------
  // descriptors per chunk: the chunk's KC channels of x' (everything outside reads as zero) / its KC * 54 weight rows
  // (PAIR: the descriptor also spans the same channels of image b + 1, one sample stride further on)
  const long xbs = PAIR && pimg_ok ? (long)(a.grouped_segs ? a.cin_g : a.Cin) * plane : 0;
  const unsigned xbytes = (unsigned)((xbs + KC * plane) * 4), wbytes = (unsigned)(KC * W_TAPS * a.cout_g * 4);
  const int xq = lane % G::REG_QUADS, xsub = lane / G::REG_QUADS;          // (lanes >= PLANES issue nothing)
  unsigned xlane_off;
  if constexpr (PAIR) {     // quad (image, half row): no halo columns, they are the padding
    const int img = (xq % RQ) >> 1;
    xlane_off = (lane < G::PLANES && (img == 0 || pimg_ok)) ? (unsigned)((img * xbs + (xq / RQ) * hw + 4 * (xq & 1)) * 4) : kOutside;
  } else {
    const int xwq = w0 - 4 + 4 * (xq % RQ);
    xlane_off = (lane < G::PLANES && xwq >= 0 && xwq < a.W) ? (unsigned)(((xq / RQ) * hw + xwq) * 4) : kOutside;
  }
  auto issue_raw = [&](auto kc_, int c) __attribute__((always_inline)) {
    constexpr int k = decltype(kc_)::value;
    const int p = wv + 4 * k;
    if (G::XP % 4 == 0 || p < G::XP) {
      const buf_rsrc r = make_rsrc(xg + (long)c * KC * plane, xbytes);
      const int region = p * G::RPP + (G::RPP > 1 ? xsub : 0);
      const int kc = region / HH, row = region - kc * HH;
      const int h = h0 - 1 + row;
      const bool ok = h >= 0 && h < a.H && region < G::NREG;
      const unsigned lo = opaque(xlane_off);
      const unsigned voff = ok ? lo + (unsigned)((kc * (int)plane + h * a.W) * 4) : kOutside;
      if (lane < G::PLANES) dma_b128(r, lo >= kOutside ? kOutside : voff, raw + p * G::PSTRIDE);
    }
  };
  const unsigned wlane_off = (unsigned)(((lane >> 3) * a.cout_g + (lane & 7) * 4) * 4);
  auto issue_w = [&](auto kc_, int c, float* st) __attribute__((always_inline)) {
    constexpr int k = decltype(kc_)::value;
    const int q = wv + 4 * k;
    if (G::WP % 4 == 0 || q < G::WP) {
      const buf_rsrc r = make_rsrc(wg + (long)c * KC * W_TAPS * a.cout_g, wbytes);
      // (rows beyond the chunk's slab -- the tail lanes of the last piece -- read as zero into the stage's padding)
      dma_b128(r, opaque(wlane_off) + (unsigned)(q * 8 * a.cout_g * 4), st + G::V_FLOATS + q * 256);
    }
  };

  // ---- input transform.  A task = one column of one (region, band tile): six input bands in, six planes out.  A wave has
  // XK pieces x RPP regions x TT tiles x PW columns = 180 tasks, all inside the regions it fetched itself; lane L takes tasks
  // L, L + 64, L + 128, one per work item (6 ds_read_b32, 16 VALU operations, 6 ds_write_b32 with immediate offsets: nothing
  // conditional, nothing kept in registers between work items but two addresses per task -- every VALU instruction between
  // two MFMAs costs the matrix pipe its issue cycles).  Tasks beyond the last repeat an earlier one (same values, same place).
  constexpr int TC = G::TCOLS;
  constexpr int TPP = G::RPP * TT * TC;                 // tasks per piece (36; PAIR: 32, the halo columns stay zero)
  constexpr int NTASK = G::XK * TPP, NTR = (NTASK + 63) / 64;
  int trd[NTR], twr[NTR];
  bool tlo[NTR], thi[NTR];   // the tile has a band below / above inside the image (else that input band is zero padding)
#pragma unroll
  for (int j = 0; j < NTR; ++j) {
    int id = j * 64 + lane;
    if (id >= NTASK) id -= TPP;
    if (wv + 4 * (id / TPP) >= G::XP) id -= TPP;        // (a wave without a last piece)
    const int tpi = id / TPP, rem = id % TPP, sub = rem / TC, col = rem % TC;
    const int tp = wv + 4 * tpi;
    const int ttile = G::RPP == 1 ? sub : 0, trsub = G::RPP == 1 ? 0 : sub;
    const int tregion = tp * G::RPP + trsub;
    const int tkc = tregion / HH, trow = tregion - tkc * HH;
    // raw column of V column c is c + 3 (V columns w0-1 .. w0+TW are raw columns 3 .. TW+4); band i of the tile at + (i - 1) * RQ * 4
    // (PAIR: raw column = tile column, V column = 1 + column of image 0 / 11 + column of image 1)
    trd[j] = tp * G::PSTRIDE + trsub * (G::REG_QUADS * 4) + (MO * ttile) * (RQ * 4) + col + (PAIR ? 0 : 3);
    twr[j] = tkc * G::V_ELEMS + ttile * G::TSTRIDE + trow * PW + (PAIR ? col + 1 + 2 * (col >> 3) : col);      // + k * HH * PW
    tlo[j] = ttile > 0, thi[j] = ttile < TT - 1;
  }
  float tin[NP];
  auto tr_load = [&](auto jc) __attribute__((always_inline)) {      // one work item: the task's six input bands ...
    constexpr int j = decltype(jc)::value;
    const float* src = raw + opaque(trd[j]);
#pragma unroll
    for (int i = 0; i < NP; ++i) tin[i] = src[(i - 1) * (RQ * 4)];
  };
  auto tr_col = [&](auto jc, float* st) __attribute__((always_inline)) {   // ... the next: its six planes
    constexpr int j = decltype(jc)::value;
    const float d0 = tlo[j] ? tin[0] : 0.f, d1 = tin[1], d2 = tin[2], d3 = tin[3], d4 = tin[4], d5 = thi[j] ? tin[5] : 0.f;
    // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1] with the rows' shared
    // sums taken once (14 instead of 22 operations)
    const float a42 = d4 - 4.f * d2, b31 = d3 - 4.f * d1, c42 = d4 - d2, e31 = d3 - d1;
    float* dst = st + opaque(twr[j]);
    dst[0 * HH * PW] = (4.f * d0 + d4) - 5.f * d2;
    dst[1 * HH * PW] = a42 + b31;
    dst[2 * HH * PW] = a42 - b31;
    dst[3 * HH * PW] = c42 + 2.f * e31;
    dst[4 * HH * PW] = c42 - 2.f * e31;
    dst[5 * HH * PW] = (4.f * d1 + d5) - 5.f * d3;
  };

  // ---- per-lane MFMA operand offsets (floats inside a stage) ---------------------------------------------------------
  // wave wv owns positions wv * 64 .. wv * 64 + 63 of the TT x TH x TW tile (w fastest): sub-tile s = two rows of 16
  // N = 8 (two band tiles): wave wv owns rows 2 wv, 2 wv + 1 of the tile in both band tiles; sub-tile s = row 2 wv + s, MFMA
  // column l31 = (band tile l31 >> 4, column l31 & 15).  N = 4: wave wv owns rows 4 wv .. 4 wv + 3, sub-tile s = two rows of 16.
  // Either way sub-tile 1 lies a constant behind sub-tile 0: one base address, one ds_read2_b32 for both.
  constexpr int SUB_STEP = G::SPLIT ? PW : 2 * PW;
  // (LLM: the K half is the column parity pw; its two column taps are V columns {+1, +2} (pw = 0) or {0, +1} (pw = 1))
  const int boff = khalf * G::V_ELEMS + (G::SPLIT ? (l31 >> 4) * G::TSTRIDE + (2 * wv) * PW + (l31 & 15) + (PAIR ? 2 * ((l31 & 15) >> 3) : 0)
                                                  : (4 * wv + (l31 >> 4)) * PW + (l31 & 15)) + (LLM ? 1 - khalf : 0);
  const int aoff = G::V_FLOATS + khalf * W_TAPS * CO + l31;

  float bias_l, sh2_l, sc2_l, sh2_m = 0.f, sc2_m = 1.f;
  {
    const int col = g * a.cout_g + co0 + l31;
    bias_l = a.bias ? a.bias[col] * a.bias_scale : 0.f;
    const bool has2 = a.y2 || a.yhi[0];     // (the Haar-output mode applies the second output's constants to its LL band)
    sh2_l = (has2 && a.y2_shift) ? a.y2_shift[(long)b * a.y2_shift_stride + col] : 0.f;
    sc2_l = (has2 && a.y2_scale) ? a.y2_scale[(long)b * a.y2_scale_stride + col] : 1.f;
    if constexpr (PAIR) {
      const int b1 = pimg_ok ? b + 1 : b;
      sh2_m = (a.y2 && a.y2_shift) ? a.y2_shift[(long)b1 * a.y2_shift_stride + col] : 0.f;
      sc2_m = (a.y2 && a.y2_scale) ? a.y2_scale[(long)b1 * a.y2_scale_stride + col] : 1.f;
    }
  }
  if constexpr (PAIR) {     // the four halo columns of every V row, both stages: zero, for good
    constexpr int ROWS = G::V_FLOATS / PW;
    static_assert(ROWS * PW == G::V_FLOATS, "the planes of a stage are one run of rows");
    for (int i = tid; i < 2 * ROWS * 4; i += 256) {
      const int stg = i / (ROWS * 4), r = (i >> 2) % ROWS, q = i & 3;
      lds[stg * G::STAGE + r * PW + (q == 0 ? 0 : (q == 1 ? 9 : (q == 2 ? 10 : 19)))] = 0.f;
    }
  }

  f32x16 acc[NP * NS];    // [k * NS + s]
#pragma unroll
  for (int i = 0; i < NP * NS; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- first chunk: fetch, transform, and request the second chunk's raw box ------------------------------------------
  static_for<0, G::XK>([&](auto k) __attribute__((always_inline)) { issue_raw(k, 0); });
  static_for<0, G::WK>([&](auto k) __attribute__((always_inline)) { issue_w(k, 0, st0); });
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): this wave's own pieces have landed (it reads its own regions only)
  static_for<0, NTR>([&](auto jc) __attribute__((always_inline)) { tr_load(jc); tr_col(jc, st0); });
  static_for<0, G::XK>([&](auto k) __attribute__((always_inline)) { issue_raw(k, nchunks > 1 ? 1 : 0); });
  __syncthreads();
  WF_STAMP(2);

  constexpr int KSTEPS = W_TAPS;            // KC = 2: one K pair per tap; tap = (dh * 3 + dw) * NP + k
  // work items between the MFMAs of a chunk (slot = K-step after whose MFMAs the item is placed)
  // (no LDS-DMA is outstanding when the raw box is read: the compiler puts vmcnt(0) in front of LDS reads that a pending
  // DMA might alias)
  // (LLM: 24 K-steps; the raw pieces go out right behind the last transform task, the weights behind them)
  constexpr int SLOT_COL0 = 0, COL_STEP = LLM ? 4 : 6, SLOT_W0 = LLM ? 12 : 20, SLOT_RAW0 = LLM ? 14 : 26, STEP = LLM ? 1 : 2;
  static_assert(SLOT_COL0 + NTR * COL_STEP <= SLOT_W0 + 1 && SLOT_W0 + G::WK <= SLOT_RAW0 && SLOT_RAW0 + G::XK * STEP <= KSTEPS - (LLM ? 4 : 8),
                "work items in order, raw pieces early enough to land");
  auto mfma_chunk = [&](auto phc, const float* st, int c1, int c2, float* st_next) __attribute__((always_inline)) {
    constexpr int PH = decltype(phc)::value;      // LLM: row parity of this chunk's virtual channels (its row taps)
    // st: the stage of this chunk; c1 / c2: the chunks whose weights / raw box are requested now (next, next but one)
    // operands are fetched two K-steps (four MFMAs) ahead: the weights of two consecutive K-steps by ONE ds_read2_b32 (their
    // rows lie 32 floats apart), the two sub-tiles' positions by another
    float av[4], bv[3][NS];
    auto fetch_a2 = [&](auto ksc) __attribute__((always_inline)) {      // weights of K-steps ks, ks + 1 (ks even)
      constexpr int ks = decltype(ksc)::value;
      av[ks % 4] = st[aoff + ks * CO];
      if constexpr (ks + 1 < KSTEPS) av[(ks + 1) % 4] = st[aoff + (ks + 1) * CO];
    };
    auto fetch_b = [&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int k = ks % NP;
      constexpr int dh = LLM ? (ks / NP) / 2 + (PH == 0 ? 1 : 0) : (ks / NP) / 3, dw = LLM ? (ks / NP) % 2 : (ks / NP) % 3;
      constexpr int toff = (k * HH + dh) * PW + dw;
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[ks % 3][s] = st[boff + toff + s * SUB_STEP];
    };
    fetch_a2(std::integral_constant<int, 0>{});
    fetch_b(std::integral_constant<int, 0>{});
    fetch_b(std::integral_constant<int, 1>{});
    static_for<0, KSTEPS>([&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int k = ks % NP;
      acc[k * NS + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks % 4], bv[ks % 3][0], acc[k * NS + 0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks % 2 == 0 && ks + 2 < KSTEPS) fetch_a2(std::integral_constant<int, ks + 2>{});
      if constexpr (ks + 2 < KSTEPS) fetch_b(std::integral_constant<int, ks + 2>{});
      __builtin_amdgcn_sched_barrier(0);
      acc[k * NS + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks % 4], bv[ks % 3][1], acc[k * NS + 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (TMDIFF_WF_ABLATE < 1 && ks >= SLOT_COL0 && (ks - SLOT_COL0) % COL_STEP == 0 && (ks - SLOT_COL0) / COL_STEP < NTR)
        tr_load(std::integral_constant<int, (ks - SLOT_COL0) / COL_STEP>{});
      if constexpr (TMDIFF_WF_ABLATE < 1 && ks > SLOT_COL0 && (ks - 1 - SLOT_COL0) % COL_STEP == 0 && (ks - 1 - SLOT_COL0) / COL_STEP < NTR)
        tr_col(std::integral_constant<int, (ks - 1 - SLOT_COL0) / COL_STEP>{}, st_next);
      if constexpr (TMDIFF_WF_ABLATE < 3 && ks >= SLOT_W0 && ks < SLOT_W0 + G::WK) issue_w(std::integral_constant<int, ks - SLOT_W0>{}, c1, st_next);
      if constexpr (TMDIFF_WF_ABLATE < 2 && ks >= SLOT_RAW0 && (ks - SLOT_RAW0) % STEP == 0 && (ks - SLOT_RAW0) / STEP < G::XK)
        issue_raw(std::integral_constant<int, (ks - SLOT_RAW0) / STEP>{}, c2);
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // (LLM: chunk (c, ph) has index 2 c + ph and every range of chunks starts at an even one: stage 0 <-> ph = 0)
  // (the LDS-DMA hand-over between waves -- the weight pieces of the next stage are read by every wave -- needs this wave's
  //  pieces to have LANDED before the barrier; the compiler places s_waitcnt vmcnt(0) there today, the explicit one makes the
  //  source say so: ADVICE r3)
  for (int c = 0; c < nchunks; c += 2) {
    mfma_chunk(std::integral_constant<int, 0>{}, st0, c + 1 < nchunks ? c + 1 : 0, c + 2 < nchunks ? c + 2 : 0, st1);
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
    __syncthreads();
    if (c + 1 < nchunks) {
      mfma_chunk(std::integral_constant<int, 1>{}, st1, c + 2 < nchunks ? c + 2 : 0, c + 3 < nchunks ? c + 3 : 0, st0);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  }
  WF_STAMP(3);

  // ---- output transform in registers, y_j = sum_k A^T[j][k] m_k, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1],
  // IN PLACE for both sub-tiles first (band j takes the place of plane j; planes 4 and 5 die: a third of the accumulator
  // registers is free for the epilogue's residual look-ahead), then the wave's eight output blocks.  No barrier: the waves
  // of a workgroup finish independently. --------------------------------------------------------------------------------
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float m0 = acc[0 * NS + s][r], m1 = acc[1 * NS + s][r], m2 = acc[2 * NS + s][r], m3 = acc[3 * NS + s][r],
                  m4 = acc[4 * NS + s][r], m5 = acc[5 * NS + s][r];
      const float p12 = m1 + m2, d12 = m1 - m2, p34 = m3 + m4, d34 = m3 - m4;
      acc[0 * NS + s][r] = (m0 + p12) + p34;
      acc[1 * NS + s][r] = d12 + 2.f * d34;
      acc[2 * NS + s][r] = p12 + 4.f * p34;
      acc[3 * NS + s][r] = (d12 + 8.f * d34) + m5;
    }
  f32x16 out[8];                  // block (s, n) = sub-tile s, band n of the wave's band tile
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = acc[(i % MO) * NS + i / MO];
  // ---- folded residual convolution (ResBlock res_conv, Hyper_unet_general.py:231, :248): out += W1^T x, a 1x1x1 convolution of the
  // block's RAW input, accumulated by the matrix pipe straight into the output blocks -- no launch of its own, its result is never
  // written and read back.  K step = 2 input channels (lane half = channel parity, as in the main loop); the weight slab
  // [rc_cin][32] of this channel tile goes through LDS once per workgroup (the stages are free), a lane's x values come
  // straight from global memory as its B operands (positions of a block = the MFMA columns: coalesced rows of 16), sixteen
  // K-steps of the next work item in flight while the current one is multiplied.
  // (a grid that splits its input channels: the workgroups of range 0 add it to THEIR partial sums -- the reduction kernel sums the
  //  ranges and applies bias / scale / second output as ever)
  if constexpr (!PAIR && !LLM) {
    if (a.rc_x && split == 0) {
      __syncthreads();                                    // every wave is done with the stages: they hold the weight slab now
      float* wl = lds;                                    // [rc_cin][32]
      for (int e = tid; e < a.rc_cin * 32; e += 256) {    // from the PyTorch-layout weight [Cout][rc_cin]: thread = (co, ci), ci fastest
        const int co = e / a.rc_cin, ci = e - co * a.rc_cin;
        wl[ci * 32 + co] = a.rc_w[(long)(co0 + co) * a.rc_cin + ci];
      }
      __syncthreads();
      // x through a buffer descriptor over this sample's rc_cin channels: per-lane byte offset (channel parity, band, row,
      // column) in one 32-bit register per output block, the channel pair of a K-step as a scalar offset
      const buf_rsrc xr = make_rsrc(a.rc_x + (long)b * a.rc_cin * plane, (unsigned)((long)a.rc_cin * plane * 4));
      unsigned xo[8];                                     // lane offset of block (s, n): band, row, column of MFMA column l31
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int s_ = i / MO, n_ = i % MO;
        const int bt = TT == 2 ? l31 >> 4 : 0;
        const int h = h0 + (TT == 2 ? 2 * wv + s_ : 4 * wv + 2 * s_ + (l31 >> 4)), w = w0 + (l31 & 15);
        // (outside the image: any valid address, the value is never stored)
        xo[i] = (unsigned)((khalf * (int)plane + ((h < a.H && w < a.W) ? (MO * bt + n_) * hw + h * a.W + w : 0)) * 4);
      }
      constexpr int GJ = 16;                              // K-steps per work item (32 input channels of one output block)
      const int ngrp = a.rc_cin / (2 * GJ);               // rc_cin % 32 == 0 (host)
      const unsigned pair_bytes = (unsigned)(2 * plane * 4);
      float xv[2][GJ];
      auto fetch = [&](int jg, auto ic, auto bufc) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, buf = decltype(bufc)::value;
        const unsigned s0 = (unsigned)(jg * GJ) * pair_bytes;
#pragma unroll
        for (int j = 0; j < GJ; ++j) xv[buf][j] = buf_load(xr, xo[i], s0 + (unsigned)j * pair_bytes);
      };
      fetch(0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      for (int jg = 0; jg < ngrp; ++jg) {
        const float* wrow = wl + (jg * 2 * GJ + khalf) * 32 + l31;
        static_for<0, 8>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value, buf = i & 1;
          if constexpr (i < 7) fetch(jg, std::integral_constant<int, i + 1>{}, std::integral_constant<int, buf ^ 1>{});
          else if (jg + 1 < ngrp) fetch(jg + 1, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
#pragma unroll
          for (int j = 0; j < GJ; ++j) out[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[2 * j * 32], xv[buf][j], out[i], 0, 0, 0);
        });
      }
      __syncthreads();                                    // the slab is dead before the epilogue borrows the same LDS
    }
  }
  float* T = lds + wv * 1024;     // (every wave is past the last barrier: the stages are free)
  const KArgs ea = kargs();       // (the epilogue's arguments, read from the kernel-argument segment HERE: scalar registers are as scarce
                                  //  across the chunk loop as vector registers)
#define WF_EPI2(Y, R, Y2) \
  epilogue_wf<Y, R, Y2, TT, TH, TW, PAIR>(ea, out, bias_l, sh2_l, sc2_l, b, g, co0, h0, w0, wv, lane, plane, T, a.y, a.out_scale, sh2_m, sc2_m, pimg_ok)
  if (a.part) {     // split-K: this range's partial sums, bare (bias / residual / scale / second output: splitk_reduce_kernel)
    float* pd = a.part + (long)split * a.B * a.Cout * plane;
    epilogue_wf<true, false, false, TT, TH, TW, PAIR>(ea, out, 0.f, 0.f, 1.f, b, g, co0, h0, w0, wv, lane, plane, T, pd, 1.f, 0.f, 1.f, pimg_ok);
  } else if (a.y) {
    if (a.residual) { if (a.y2) WF_EPI2(true, true, true); else WF_EPI2(true, true, false); }
    else            { if (a.y2) WF_EPI2(true, false, true); else WF_EPI2(true, false, false); }
  } else {
    if (a.residual) WF_EPI2(false, true, true); else WF_EPI2(false, false, true);
  }
#undef WF_EPI2
#if TMDIFF_WF_STAMPS
  WF_STAMP(4);
  __builtin_amdgcn_s_waitcnt(0);
  WF_STAMP(5);
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((unsigned long long)blockIdx.x * 4 + wv) * 8;
    for (int i = 0; i < 6; ++i) o[i] = stamp_t[i];
    o[6] = ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(6 | (31 << 11)) |
           ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf) << 28);
    o[7] = __builtin_amdgcn_s_memtime() - stamp_c0;
  }
#endif
}

template <int TT, int TH, int TW, bool PAIR = false, bool LLM = false>
int launch(WfArgs& a, hipStream_t st) {
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = PAIR ? 1 : (a.W + TW - 1) / TW;
  a.tiles_co = a.cout_g / 32;
  const long blocks = (long)a.ksplit * (PAIR ? (a.B + 1) / 2 : a.B) * a.groups * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d_wf_fwd: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  static const double stagger_chunks = [] {
    const char* e = getenv("TMDIFF_WF_STAGGER");     // experiments: delay of every CU's second resident workgroup, in chunk times
    return e ? atof(e) : 0.0;
  }();
  a.first_round = 512;
  a.stagger = blocks > 512 ? (int)(stagger_chunks * 2.0 * 54 * 2 * 64) : 0;
  conv3d_wf_kernel<TT, TH, TW, PAIR, LLM><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_wf_fwd");
}

bool wf_shape_ok(const tmdiff_conv3d_desc* d) {
  if (!d || d->ksize != 3 || (d->groups != 1 && d->groups != 3) || d->in_mask || d->x_bf16 || d->y2_bf16) return false;
  if (!(d->drop_p >= 0.f && d->drop_p < 1.f)) return false;
  if (d->nseg < 1 || d->nseg > 3 || d->Cin <= 0 || d->Cout <= 0 || d->Cin % d->groups || d->Cout % d->groups) return false;
  return (d->Cin / d->groups) % 2 == 0 && (d->Cout / d->groups) % 32 == 0 && (d->N == 8 || d->N == 4) && d->H > 0 && d->W > 0 &&
         d->W % 4 == 0;
}

// three plain segments that are exactly the three groups' inputs (convH_0 on the three high bands of the skip connection)
bool wf_grouped_segs(const tmdiff_conv3d_desc* d) {
  return d->nseg == 3 && d->groups == 3 && d->seg_c[0] == d->seg_c[1] && d->seg_c[1] == d->seg_c[2] &&
         d->seg_c[0] * 3 == d->Cin;
}

// 8 bands x 8 columns (the 8x8 level): two images share a tile, see GeoF
bool wf_pair(const tmdiff_conv3d_desc* d) { return d->N == 8 && d->W == 8; }

long wf_tiles(const tmdiff_conv3d_desc* d) {
  const int th = d->N == 8 ? 8 : 16;
  const long units = wf_pair(d) ? (d->B + 1) / 2 : (long)d->B * ((d->W + 15) / 16);
  return units * d->groups * ((d->H + th - 1) / th) * (d->Cout / d->groups / 32);
}

// split-K factor for grids that cannot fill the chip: the smallest divisor of the chunk count that brings the grid to 256
// workgroups (TMDIFF_SPLITK=<n>: to n; one workgroup per CU unsplit beats two halves + a reduction kernel for short K loops:
// 384 cost 2 % of the finetune step), at least two chunks per range -- and once more by two where that still leaves fewer
// than 384 workgroups and a range is LONG (>= 1700 K-steps, i.e. 32 chunks: 128 -> 128 at the 16x16 level; the composed-LL
// launches of the 16x16 / 8x8 levels, 550 -> 472 us; benchmark step 23.97 -> 23.61 ms, finetune step unchanged):
// a workgroup alone on its CU leaves 15 % of the matrix pipe idle, which a long loop pays for the reduction several times over.
// TMDIFF_SPLITK=0: never.  pairs: the ranges must hold whole pairs of chunks (the composed-LL mode, whose chunks alternate in
// row parity).
int wf_ksplit(const tmdiff_conv3d_desc* d, bool pairs = false) {
  static const long target = [] {
    const char* e = getenv("TMDIFF_SPLITK");
    return e ? atol(e) : 256L;
  }();
  const long tiles = wf_tiles(d);
  const int nchunks = d->Cin / d->groups / 2;
  const int ksteps = pairs ? 24 : 54;
  if (target <= 0 || tiles <= 0) return 1;
  auto ok = [&](int s) { return s >= 1 && nchunks % s == 0 && nchunks / s >= 2 && !(pairs && (nchunks / s) % 2); };
  int best = 1;
  if (tiles < target)
    for (int s = 2; s <= nchunks / 2; ++s) {
      if (!ok(s)) continue;
      best = s;
      if (tiles * s >= target) break;
    }
  static const long long_range = [] {      // (experiments: TMDIFF_SPLITK_LONG=<K-steps per range from which a range counts as long>)
    const char* e = getenv("TMDIFF_SPLITK_LONG");
    return e ? atol(e) : 1700L;
  }();
  if (tiles * best < 384 && ok(2 * best) && (long)(nchunks / (2 * best)) * ksteps >= long_range) best *= 2;
  return best;
}

bool wf_plain(const tmdiff_conv3d_desc* d) {
  return (d->nseg == 1 || wf_grouped_segs(d)) && !d->in_shift && !d->in_scale && !d->in_act && !(d->drop_p > 0.f);
}

}  // namespace

extern "C" int tmdiff_conv3d_wf_supported(const tmdiff_conv3d_desc* d) { return wf_shape_ok(d) ? 1 : 0; }

/* the launch plan of tmdiff_conv3d_wf_fwd for a convolution of these extents, without a descriptor (host-side routing): returns
 * the split-K factor (1 = none; what the launch does when the split-K workspace is lent) and writes the number of output tiles
 * (workgroups = tiles x factor); 0 = shape not taken.  llm != 0: the composed Conv_0 + LL mode (Cin, H, W those of the
 * space-to-depth tensor: 4 x the channels, half the extents). */
extern "C" int32_t tmdiff_conv3d_wf_plan(int32_t B, int32_t Cin, int32_t Cout, int32_t N, int32_t H, int32_t W, int32_t groups,
                                         int32_t llm, int64_t* tiles) {
  tmdiff_conv3d_desc d = {};
  d.B = B; d.N = N; d.H = H; d.W = W; d.Cin = Cin; d.Cout = Cout; d.groups = groups; d.ksize = 3; d.nseg = 1; d.seg_c[0] = Cin;
  if (tiles) *tiles = 0;
  if (B <= 0 || !wf_shape_ok(&d)) return 0;
  if (tiles) *tiles = wf_tiles(&d);
  return wf_ksplit(&d, llm != 0);
}

/* workgroups of the grid when the split-K workspace is lent (tiles x split factor) */
extern "C" int64_t tmdiff_conv3d_wf_blocks(const tmdiff_conv3d_desc* d) {
  if (!wf_shape_ok(d) || d->B <= 0) return 0;
  return (int64_t)wf_tiles(d) * wf_ksplit(d);
}

/* bytes of partial outputs a small grid splits its input channels into (lend them through d->splitk_ws; 0: no split) */
extern "C" size_t tmdiff_conv3d_wf_splitk_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!wf_shape_ok(d) || d->B <= 0) return 0;
  const int ks = wf_ksplit(d);
  return ks > 1 ? (size_t)ks * d->B * d->Cout * d->N * d->H * d->W * sizeof(float) : 0;
}

/* bytes of the prologue output x' the entry point forms first when the input is not one plain tensor (0: plain input) */
extern "C" size_t tmdiff_conv3d_wf_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!wf_shape_ok(d) || d->B <= 0 || wf_plain(d)) return 0;
  return (size_t)d->B * d->Cin * d->N * d->H * d->W * sizeof(float);
}

namespace {
// llm: the composed Conv_0 + LL mode (d then describes the convolution on the space-to-depth tensor: 4 x the channels, half the
// extents); bias_mul: factor on the bias (2 * ll_scale there, 1 otherwise)
int wf_forward(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream, bool llm, float bias_mul) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d != nullptr, "conv3d_wf_fwd: NULL descriptor");
  if (!wf_shape_ok(d))
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wf_fwd: fp32 3x3x3, groups 1 or 3, N = 8 or 4, W %% 4 == 0, Cin/g %% 2 == 0, Cout/g %% 32 == 0, no mask");
  TMDIFF_REQUIRE(d->B >= 0, "conv3d_wf_fwd: bad extents");
  if (d->B == 0) return TMDIFF_OK;
  TMDIFF_REQUIRE(d->w_packed && (d->y || d->y2 || d->y_ll) && aligned16(d->w_packed), "conv3d_wf_fwd: NULL / unaligned weights or output");
  TMDIFF_REQUIRE((long)d->N * d->H * d->W <= (1L << 24), "conv3d_wf_fwd: plane too large for 32-bit offsets (32 channels of a sample per descriptor)");
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    TMDIFF_REQUIRE(d->seg_x[i] != nullptr && d->seg_c[i] > 0 && aligned16(d->seg_x[i]), "conv3d_wf_fwd: segment %d is empty / unaligned", i);
    csum += d->seg_c[i];
  }
  TMDIFF_REQUIRE(csum == d->Cin, "conv3d_wf_fwd: segments hold %d channels, Cin=%d", csum, d->Cin);
  if (!(aligned16(d->y) && aligned16(d->y2) && aligned16(d->residual)))
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wf_fwd: 16-byte aligned outputs / residual");
  hipStream_t st = as_stream(stream);
  const float* x = d->seg_x[0];
  if (!wf_plain(d)) {      // prologue / concatenation / dropout: one elementwise pass (8 B per input element) forms x'
    TMDIFF_REQUIRE(workspace && aligned16(workspace), "conv3d_wf_fwd: this input needs its workspace (tmdiff_conv3d_wf_workspace_bytes)");
    const int rc = launch_prologue_apply(d, static_cast<float*>(workspace), st);
    if (rc) return rc;
    x = static_cast<const float*>(workspace);
  }
  WfArgs a;
  a.grouped_segs = wf_plain(d) && d->nseg == 3;
  for (int i = 0; i < 3; ++i) a.xgrp[i] = a.grouped_segs ? d->seg_x[i] : nullptr;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cout = d->Cout; a.groups = d->groups; a.cin_g = d->Cin / d->groups; a.cout_g = d->Cout / d->groups;
  a.x = x; a.wp = d->w_packed;
  a.bias = d->bias; a.bias_scale = d->bias_scale * bias_mul;
  a.residual = d->residual; a.out_scale = d->out_scale; a.y = d->y;
  a.y2 = d->y2; a.y2_shift = d->y2_shift; a.y2_scale = d->y2_scale; a.y2_act = d->y2_act;
  a.y2_s2d = d->y2 && d->y2_s2d ? 1 : 0;
  if (a.y2_s2d) TMDIFF_REQUIRE(d->H % 2 == 0 && d->W % 4 == 0, "conv3d_wf_fwd: the space-to-depth second output needs even H and W %% 4 == 0");
  a.y2_shift_stride = d->y2_shift_stride > 0 ? d->y2_shift_stride : (d->y2_shift_stride < 0 ? 0 : d->Cout);
  a.y2_scale_stride = d->y2_scale_stride > 0 ? d->y2_scale_stride : (d->y2_scale_stride < 0 ? 0 : d->Cout);
  a.vec4 = 1;
  a.stamps = TMDIFF_WF_STAMPS ? static_cast<unsigned long long*>(d->splitk_ws) : nullptr;
  a.ksplit = 1; a.split_chunks = a.cin_g / 2; a.part = nullptr;
  if (!TMDIFF_WF_STAMPS) {
    const int ks = wf_ksplit(d, llm);
    const size_t need = (size_t)ks * d->B * d->Cout * d->N * d->H * d->W * sizeof(float);
    if (ks > 1 && d->splitk_ws && (size_t)d->splitk_ws_bytes >= need && aligned16(d->splitk_ws)) {
      // (only a launch that really splits is refused: without a lent workspace the grid runs unsplit and its own epilogue
      //  writes the second output -- ADVICE r3: TMDIFF_WF_SPLITK=0 used to raise here for shapes the host had routed)
      if (a.y2_s2d)
        return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wf_fwd: a grid that splits its input channels cannot write the space-to-depth second output");
      a.ksplit = ks; a.split_chunks = a.cin_g / 2 / ks; a.part = static_cast<float*>(d->splitk_ws);
    }
  }
  a.yll = nullptr; a.yhi[0] = a.yhi[1] = a.yhi[2] = nullptr;
  if (d->y_ll) {       // third output: the halved LL band of y (y itself not written); with y_hi: its whole Haar transform
    const bool dwt = d->y_hi[0] != nullptr;
    if (llm || wf_pair(d) || a.part || d->y || (dwt ? (d->y2 || !d->y_hi[1] || !d->y_hi[2]) : !d->y2) || d->N != 8 || d->H % 2 ||
        d->W % 4 || !aligned16(d->y_ll) || !aligned16(d->y_hi[0]) || !aligned16(d->y_hi[1]) || !aligned16(d->y_hi[2]))
      return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wf_fwd: the LL / Haar output needs 8 bands, even H, W %% 4 == 0, planes wider than 8 columns, "
                                        "y == NULL (with a second output, or with all three high bands and none), and a grid that "
                                        "does not split its input channels");
    a.yll = d->y_ll;
    for (int i = 0; i < 3; ++i) a.yhi[i] = dwt ? d->y_hi[i] : nullptr;
  }
  a.rc_x = nullptr; a.rc_w = nullptr; a.rc_cin = 0;
  if (d->rc_x) {       // the ResBlock's 1x1x1 res_conv folded into this launch's epilogue
    if (llm || wf_pair(d) || d->groups != 1 || d->residual || !d->rc_w || d->rc_cin <= 0 || d->rc_cin % 32 || d->rc_cin > 512 ||
        !aligned16(d->rc_x) || !aligned16(d->rc_w))
      return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wf_fwd: a folded residual convolution needs groups 1, rc_cin %% 32 == 0 (at most 512), no residual "
                                        "tensor and planes wider than 8 columns");
    TMDIFF_REQUIRE((long)d->rc_cin * d->N * d->H * d->W < (1L << 30), "conv3d_wf_fwd: rc_x sample too large for 32-bit byte offsets");
    a.rc_x = d->rc_x; a.rc_w = d->rc_w; a.rc_cin = d->rc_cin;
  }
  if (wf_pair(d))
    TMDIFF_REQUIRE(((long)d->Cin + 2) * d->N * d->H * d->W < (1L << 29), "conv3d_wf_fwd: sample too large for 32-bit offsets");
  int rc;
  if (llm) rc = wf_pair(d) ? launch<2, 8, 16, true, true>(a, st) : (d->N == 8 ? launch<2, 8, 16, false, true>(a, st) : launch<1, 16, 16, false, true>(a, st));
  else rc = wf_pair(d) ? launch<2, 8, 16, true>(a, st) : (d->N == 8 ? launch<2, 8, 16>(a, st) : launch<1, 16, 16>(a, st));
  if (rc || !a.part) return rc;
  SplitKReduceArgs r{a.part, a.ksplit, d->B, d->Cout, (long)d->N * d->H * d->W, d->bias, d->bias_scale, d->residual,
                     d->out_scale, d->y, d->y2, d->y2_shift, d->y2_scale, a.y2_shift_stride, a.y2_scale_stride, d->y2_act};
  return launch_splitk_reduce(r, st);
}

// the descriptor of the composed-LL convolution as the kernel sees it: the space-to-depth input has 4 x the channels at half
// the extents; outputs at the halved extents
bool wfll_desc(const tmdiff_conv3d_desc* d, tmdiff_conv3d_desc* e) {
  if (!d || d->ksize != 3 || d->groups != 1 || d->nseg != 1 || (d->N != 8 && d->N != 4) || d->H <= 0 || d->W <= 0 || d->H % 2 || d->W % 8) return false;
  if (d->in_shift || d->in_scale || d->in_act || d->in_mask || d->drop_p > 0.f || d->x_bf16 || d->y2_bf16 || d->y2_s2d) return false;
  if (d->Cin <= 0 || d->Cout <= 0 || d->Cout % 32 || d->seg_c[0] != d->Cin) return false;
  *e = *d;
  e->H = d->H / 2; e->W = d->W / 2; e->Cin = 4 * d->Cin; e->seg_c[0] = 4 * d->Cin;
  return wf_shape_ok(e);
}

// F(4,3) weight transform G (6 x 3), as WM<6> of conv3d_wino.hip
__constant__ float kG6[6][3] = {{0.25f, 0.f, 0.f},           {-1.f / 6.f, -1.f / 6.f, -1.f / 6.f}, {-1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                                {1.f / 24.f, 1.f / 12.f, 1.f / 6.f}, {1.f / 24.f, -1.f / 12.f, 1.f / 6.f}, {0.f, 0.f, 1.f}};

// One thread = one (co, ci) weight row (27 taps): the composed 3 x 4 x 4 kernel W'[dn][a][b] = s * sum_{P,Q in {0,1}} W[dn][a-P][b-Q]
// (conv3d_ll.hip), its four (a, b) taps per (row parity ph, column parity pw) -- a = 1 + 2 i (ph = 0) or 2 i (ph = 1), b likewise
// -- and G along dn:  packed[ci][ph][pw][i * 2 + j][k][co].
__global__ void __launch_bounds__(256) wfll_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, float s) {
  const long r = blockIdx.x * 256L + threadIdx.x;
  if (r >= (long)Cout * Cin) return;
  const int co = (int)(r % Cout), ci = (int)(r / Cout);
  const float* wk = w + ((long)co * Cin + ci) * 27;
  float t[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) t[i] = wk[i];
#pragma unroll
  for (int ph = 0; ph < 2; ++ph)
#pragma unroll
    for (int pw = 0; pw < 2; ++pw)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int a4 = ph == 0 ? 1 + 2 * i : 2 * i, b4 = pw == 0 ? 1 + 2 * j : 2 * j;
          float c3[3];
#pragma unroll
          for (int dn = 0; dn < 3; ++dn) {
            float acc = 0.f;
#pragma unroll
            for (int P = 0; P < 2; ++P)
#pragma unroll
              for (int Q = 0; Q < 2; ++Q) {
                const int dh = a4 - P, dw = b4 - Q;
                if (dh >= 0 && dh < 3 && dw >= 0 && dw < 3) acc += t[dn * 9 + dh * 3 + dw];
              }
            c3[dn] = s * acc;
          }
#pragma unroll
          for (int k = 0; k < 6; ++k)
            packed[(((((long)ci * 2 + ph) * 2 + pw) * 4 + i * 2 + j) * 6 + k) * Cout + co] = kG6[k][0] * c3[0] + kG6[k][1] * c3[1] + kG6[k][2] * c3[2];
        }
}

}  // namespace

extern "C" int tmdiff_conv3d_wf_fwd(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream) {
  return wf_forward(d, workspace, stream, false, 1.f);
}

/* ---- Conv_0 + halved LL band as one convolution WITH Winograd along the bands (see GeoF, LLM) ------------------------------ */
extern "C" int tmdiff_conv3d_wfll_supported(const tmdiff_conv3d_desc* d) {
  tmdiff_conv3d_desc e;
  return wfll_desc(d, &e) ? 1 : 0;
}

extern "C" size_t tmdiff_conv3d_wfll_packed_bytes(int32_t Cout, int32_t Cin) {
  if (Cout <= 0 || Cin <= 0 || Cout % 32) return 0;
  return (size_t)Cin * 96 * Cout * sizeof(float);
}

extern "C" int tmdiff_conv3d_wfll_pack_weights(const float* w, float* packed, int32_t Cout, int32_t Cin, float ll_scale,
                                               tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(w && packed && aligned16(packed), "conv3d_wfll_pack_weights: NULL / unaligned pointer");
  TMDIFF_REQUIRE(Cout > 0 && Cin > 0 && Cout % 32 == 0, "conv3d_wfll_pack_weights: Cout=%d (multiple of 32) Cin=%d", Cout, Cin);
  const long rows = (long)Cout * Cin;
  wfll_pack_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, as_stream(stream)>>>(w, packed, Cout, Cin, ll_scale * 0.5f);
  return check_launch("conv3d_wfll_pack_weights");
}

extern "C" size_t tmdiff_conv3d_wfll_splitk_workspace_bytes(const tmdiff_conv3d_desc* d) {
  tmdiff_conv3d_desc e;
  if (!wfll_desc(d, &e) || d->B <= 0) return 0;
  const int ks = wf_ksplit(&e, true);
  return ks > 1 ? (size_t)ks * e.B * e.Cout * e.N * e.H * e.W * sizeof(float) : 0;
}

extern "C" int tmdiff_conv3d_wfll_fwd(const tmdiff_conv3d_desc* d, float ll_scale, tmdiff_stream_t stream) {
  using namespace tmdiff;
  tmdiff_conv3d_desc e;
  if (!wfll_desc(d, &e))
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wfll_fwd: one plain fp32 space-to-depth input, 3x3x3, groups 1, 8 or 4 bands, even H, W %% 8 == 0, Cout %% 32 == 0");
  return wf_forward(&e, nullptr, stream, true, 2.f * ll_scale);
}
