// 3x3x3 convolution with Winograd F(m,3) along the band axis n (m = 4 or 2), on the exact-fp32 matrix cores.
//
// A tile of MO output bands (MO*t .. MO*t+MO-1) of one (h, w) column needs the NP = MO + 2 input bands MO*t-1 .. MO*t+MO.
// With v = B^T d (input transform, per element), u = G g (weight transform along the three n taps),
//     m_k = sum over (ci, dh, dw) of u_k * v_k          (NP 3x3 convolutions in (h, w)),      y = A^T m,
// the tile costs NP x 9 multiply-adds per (ci, co) instead of MO x 27: 2x fewer for F(4,3) (6 planes for 4 bands; taken
// when N % 4 == 0 -- N = 8 is exactly two tiles), 1.5x for F(2,3) (4 planes for 2 bands; N % 2 == 0):
//   F(2,3): B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
//   F(4,3): B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1],
//           G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1],
//           A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// fp32 error against the fp64 convolution (64 -> 64 channels, random data): direct 2.5e-7 relative L2, F(2,3) 3.9e-7,
// F(4,3) 1.0e-6 (max 2.3e-6 of the largest output) -- inside every tolerance of the parity suite.
//
// Pieces: wino_input_kernel writes V[b][c][t][k][h+1][w+1] with a zero border (prologue applied first, segments concatenated;
// NP / MO times the bytes of x, an HBM pass), tmdiff_conv3d_wino_pack_weights writes U as [g][ci][dh*3+dw][k][co], and
// conv3d_wino_kernel is the staged kernel (conv3d_dma.hip) over "taps" (dh, dw, k): the LDS box of an output tile holds, per
// channel, the TT x NP planes (t, k) with a one-pixel halo in (h, w); tap (dh, dw, k) multiplies plane k shifted by (dh, dw)
// into accumulator k; at the end of the tile the waves exchange partial output sums through LDS and hand the output bands
// to the shared vector epilogue.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "epilogue.h"

// Diagnostic build (-DTMDIFF_WINO_STAMPS=1, tools/wino_stamps.py): every wave records s_memrealtime (100 MHz, one clock for
// the whole chip) at its phase boundaries plus HW_REG_HW_ID / HW_REG_LDS_ALLOC, into the buffer lent through desc.splitk_ws
// (8 x u64 per wave; never read by the kernel, no output depends on it).
#ifndef TMDIFF_WINO_STAMPS
#define TMDIFF_WINO_STAMPS 0
#endif

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// transform matrices (NP = 4: F(2,3); NP = 6: F(4,3))
template <int NP> struct WM;
template <> struct WM<4> {
  static constexpr int MO = 2;
  static constexpr float BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
  static constexpr float G[4][3] = {{1, 0, 0}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0, 0, 1}};
  static constexpr float AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};
};
template <> struct WM<6> {
  static constexpr int MO = 4;
  static constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0},
                                     {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  static constexpr float G[6][3] = {{0.25f, 0, 0}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                                    {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0, 0, 1}};
  static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
};

struct WinoArgs {
  int B, N, H, W;       // output = input extents (N a multiple of MO)
  int Cin, Cout, cin_g, cout_g, groups;
  const float* v;       // V [B, Cin, N/MO, NP, H+2, W+4]
  const float* wp;      // U packed [g][ci][9][NP][co]
  const float* bias;
  float bias_scale;
  const float* residual;
  float out_scale;
  float* y;
  float* y2;
  const float* y2_shift;
  const float* y2_scale;
  int y2_shift_stride, y2_scale_stride, y2_act;
  int tiles_t, tiles_h, tiles_w, tiles_co;
  unsigned total_blocks;
  int vec4;
  unsigned long long* stamps;   // diagnostic builds only
  int stagger;          // cycles the SECOND resident workgroup of a CU waits before its first tile (0 = none): see stagger_start
  unsigned first_round; // workgroups resident at launch (2 per CU): only they can be in lockstep with their CU partner
};

struct WinoInArgs {
  int B, Cin, N, H, W, nseg;
  int seg_c[3];
  const float* seg_x[3];
  const float* in_shift;
  const float* in_scale;
  int shift_stride, scale_stride, in_act;
  float* v;
  float* xp;                 // optional: the prologue output x' itself [B, Cin, N, H, W] (kept for the weight gradient)
  uint64_t drop_seed;        // in-kernel dropout of x' (drop_inv > 0): common.h drop_keep, element index as in prologue_apply
  const uint64_t* drop_seed_dev;   // ... plus this device word (tmdiff_conv3d_desc.drop_seed_dev), or NULL
  uint32_t drop_thresh;
  float drop_inv;
};

// V is stored with a zero border: plane (t, k) is (H + 2) x (W + 4) floats, element (h, w) at row h + 1, column w + 1, so that
// the haloed box of a tile (rows h0 - 1 .., columns w0 - 1 ..; w0 a multiple of 8) starts on a 16-byte boundary and travels
// to LDS in 16-byte pieces without bounds checks (a dword piece per 64 floats cost the kernel its matrix-pipe time).
// one thread: one (b, c, t, h, quad j) -- source columns 4j-1 .. 4j+3 of the NP input bands of tile t (prologue applied),
// written as the padded quad 4j .. 4j+3 of the NP transformed planes (+ the trailing quad of a row, + the border rows)
template <int NP>
__global__ void __launch_bounds__(256) wino_input_kernel(const WinoInArgs a) {
  using M = WM<NP>;
  constexpr int MO = M::MO;
  const int bc = blockIdx.y, b = bc / a.Cin, c = bc % a.Cin;
  int cs = c, seg = 0;
  if (a.nseg > 1 && cs >= a.seg_c[0]) { cs -= a.seg_c[0]; seg = 1; }
  if (seg == 1 && a.nseg > 2 && cs >= a.seg_c[1]) { cs -= a.seg_c[1]; seg = 2; }
  const int segc = seg == 0 ? a.seg_c[0] : (seg == 1 ? a.seg_c[1] : a.seg_c[2]);
  const long hw = (long)a.H * a.W;
  const float* xs = (seg == 0 ? a.seg_x[0] : (seg == 1 ? a.seg_x[1] : a.seg_x[2])) + ((long)b * segc + cs) * a.N * hw;
  const float sh = a.in_shift ? a.in_shift[(long)b * a.shift_stride + c] : 0.f;
  const float sc = a.in_scale ? a.in_scale[(long)b * a.scale_stride + c] : 1.f;
  const bool drop = a.drop_inv > 0.f;
  const uint64_t dseed = a.drop_seed + (drop && a.drop_seed_dev ? *a.drop_seed_dev : 0ull);
  const bool plain = !a.in_shift && !a.in_scale && !a.in_act && !drop;
  const uint64_t ebase = (uint64_t)bc * (uint64_t)(a.N * hw);
  float* xpp = a.xp ? a.xp + (long)bc * a.N * hw : nullptr;
  const int T = a.N / MO, WP = a.W + 4, qrow = a.W / 4;
  const long pplane = (long)(a.H + 2) * WP;                    // one padded plane
  float* vp = a.v + (long)bc * T * NP * pplane;
  const long quads = hw / 4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < T * quads; i += 256L * gridDim.x) {
    const int t = (int)(i / quads);
    const long q = (i % quads) * 4;                            // offset of the source quad inside a band plane
    const int h = (int)(q / a.W), j = (int)((q % a.W) / 4);
    float d[NP][5];                                            // [band][source column 4j-1 .. 4j+3]
    auto prologue = [&](float x, long idx) __attribute__((always_inline)) {
      float u = x + sh;
      const float ua = tmdiff::silu_f(u);
      u = (a.in_act ? ua : u) * sc;
      if (drop) u *= tmdiff::drop_keep(dseed, ebase + (uint64_t)idx, a.drop_thresh, a.drop_inv);
      return u;
    };
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const int n = MO * t - 1 + k;
      if (n >= 0 && n < a.N) {
        const float4 x4 = *reinterpret_cast<const float4*>(xs + n * hw + q);
        d[k][1] = x4.x, d[k][2] = x4.y, d[k][3] = x4.z, d[k][4] = x4.w;
        if (!plain) {
#pragma unroll
          for (int e = 1; e < 5; ++e) d[k][e] = prologue(d[k][e], n * hw + q + e - 1);
        }
        if (xpp && k >= 1 && k <= MO)       // the tile's own bands: every element of x' exactly once
          *reinterpret_cast<float4*>(xpp + n * hw + q) = make_float4(d[k][1], d[k][2], d[k][3], d[k][4]);
      } else {
#pragma unroll
        for (int e = 1; e < 5; ++e) d[k][e] = 0.f;             // zero padding of the convolution along the bands
      }
      // source column 4j - 1: the last value of the previous quad = the previous lane's (consecutive lanes hold consecutive
      // quads of the same band; the first lane of a wave fetches it itself, the first quad of a row has the zero border)
      float prev = __shfl_up(d[k][4], 1);
      if ((threadIdx.x & 63) == 0 && j > 0 && n >= 0 && n < a.N) {
        const float x = xs[n * hw + q - 1];
        prev = plain ? x : prologue(x, n * hw + q - 1);
      }
      d[k][0] = j > 0 ? prev : 0.f;
    }
    float o[NP][5];
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
      for (int e = 0; e < 5; ++e) {
        float acc = 0.f;
#pragma unroll
        for (int jj = 0; jj < NP; ++jj)
          if (M::BT[k][jj] != 0.f) acc += M::BT[k][jj] * d[jj][e];
        o[k][e] = acc;
      }
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      float* row = vp + ((long)t * NP + k) * pplane + (long)(h + 1) * WP;
      *reinterpret_cast<float4*>(row + 4 * j) = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
      if (j == qrow - 1) *reinterpret_cast<float4*>(row + a.W) = make_float4(o[k][4], 0.f, 0.f, 0.f);
      if (h == 0) {                                            // border rows
        *reinterpret_cast<float4*>(row - WP + 4 * j) = z4;
        if (j == qrow - 1) *reinterpret_cast<float4*>(row - WP + a.W) = z4;
      }
      if (h == a.H - 1) {
        *reinterpret_cast<float4*>(row + WP + 4 * j) = z4;
        if (j == qrow - 1) *reinterpret_cast<float4*>(row + WP + a.W) = z4;
      }
    }
  }
}

__device__ const float4 kZero4 = {0.f, 0.f, 0.f, 0.f};  // source of zero padding / filler lanes

__device__ __forceinline__ void dma_b128(const float* src, float* dst) {
#if defined(__HIP_DEVICE_COMPILE__)  // the builtin exists in the device pass only
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#endif
}

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, k = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Two workgroups share a CU (LDS-limited) and one matrix pipe per SIMD.  Launched together and with equal tile times they
// stay in lockstep for the whole launch: both run their MFMA phases together (each at half the pipe) and both reach their
// epilogues together -- the pipe idles while every CU of the chip stores its outputs at once (HBM-bound burst: 64 -> 64 at
// 64x64, B = 32 writes / reads 0.8 GB per launch in four bursts).  Delaying the second resident workgroup of every CU ONCE,
// by about an epilogue, puts the pairs in anti-phase for good: one computes at the full pipe rate while its partner stores.
// The second resident is the one whose LDS allocation does not start at 0 (HW_REG_LDS_ALLOC, base field).
__device__ __forceinline__ void stagger_start(int cycles, unsigned first_round) {
  if (cycles <= 0 || blockIdx.x >= first_round) return;
  const unsigned lds_alloc = __builtin_amdgcn_s_getreg(6 | (0 << 6) | (11 << 11));   // hwreg(HW_REG_LDS_ALLOC, 0, 12): LDS_BASE
  if (lds_alloc == 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < (long long)cycles) __builtin_amdgcn_s_sleep(32);
}

template <int NS, int MSUB, int KC, int TT, int TH, int TW, int NP>
struct GeoW {
  static constexpr int CO = 32 * MSUB;
  static constexpr int W_TAPS = 9 * NP;                     // (dh, dw) x k
  static constexpr int HH = TH + 2, HW = TW + 4;            // rows of TW + 2 haloed columns, fetched as (TW + 4) / 4 quads
  static constexpr int PLANE = HH * HW;
  static constexpr int TILE_ELEMS = TT * NP * PLANE;        // [TT tiles][NP planes][HH][HW] of one channel
  static constexpr int LDS_IN = KC * TILE_ELEMS;
  static constexpr int XP = (LDS_IN / 4 + 63) / 64;         // 16-byte pieces (64 quads each)
  static constexpr int X_FLOATS = XP * 256;
  static constexpr int W_UNITS = KC * W_TAPS * CO / 4;      // weight slab [KC][9 NP][CO] in 16-byte units
  static constexpr int WP = (W_UNITS + 63) / 64;
  static constexpr int STAGE = X_FLOATS + WP * 256;
  static_assert(TT * TH * TW == 2 * NS * 32, "workgroup tile = 2 wave pairs x NS sub-tiles x 32 positions");
  static_assert(KC % 2 == 0, "K step is 2 channels");
  static_assert((TH * TW) % 32 == 0, "a sub-tile of 32 positions lies in one tile along n");
};

// Plane split: waves 0, 1 accumulate the planes k < NP/2 and waves 2, 3 the planes k >= NP/2 of the SAME 2 x NS x 32 positions
// (one weight read and NS input reads feed NS x MSUB MFMAs of one plane: 0.75 LDS reads per MFMA); at the end of the tile
// each pair forms its part of A^T m for every output band, keeps the parts of the output bands it will finish (the first
// pair the lower half of the bands, the second the upper half) and hands the others to its partner wave through LDS.
template <int NS, int MSUB, int KC, int TT, int TH, int TW, int NP>
__global__ void __launch_bounds__(256, 2) conv3d_wino_kernel(const WinoArgs a) {
  using G = GeoW<NS, MSUB, KC, TT, TH, TW, NP>;
  using M = WM<NP>;
  constexpr int CO = G::CO, MO = M::MO, W_TAPS = G::W_TAPS;
  constexpr int NX = NP / 2;                                   // planes per wave
  constexpr int OH = MO / 2;                                   // output bands a wave finishes
  constexpr int XK = (G::XP + 3) / 4, WK = (G::WP + 3) / 4;    // pieces per wave
  __shared__ __attribute__((aligned(16))) float st0[G::STAGE];
  __shared__ __attribute__((aligned(16))) float st1[G::STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
#if TMDIFF_WINO_STAMPS
  unsigned long long stamp_t[6];
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime();
  stamp_t[0] = __builtin_amdgcn_s_memrealtime();
#define WINO_STAMP(i) stamp_t[i] = __builtin_amdgcn_s_memrealtime()
#else
#define WINO_STAMP(i)
#endif
  stagger_start(a.stagger, a.first_round);
  WINO_STAMP(1);

  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int tw_i = __builtin_amdgcn_readfirstlane(id % a.tiles_w); id /= a.tiles_w;
  const int th_i = __builtin_amdgcn_readfirstlane(id % a.tiles_h); id /= a.tiles_h;
  const int tt_i = __builtin_amdgcn_readfirstlane(id % a.tiles_t); id /= a.tiles_t;
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups);
  const int b = __builtin_amdgcn_readfirstlane(id / a.groups);
  const int t0 = tt_i * TT, h0 = th_i * TH, w0 = tw_i * TW;
  const int co0 = co_tile * CO;
  const int T = a.N / MO;
  const long plane = (long)a.N * a.H * a.W;                    // output plane
  const int plane_v = T * NP * (a.H + 2) * (a.W + 4);          // one channel of V (planes with their zero border)
  const int nchunks = a.cin_g / KC;

  // ---- DMA sources of this lane (the same for every chunk) -----------------------------------------------------
  int xsrc[XK];  // float offset (of a quad) from the chunk base, or -1 = zero quad
  const int WPd = a.W + 4, HPd = a.H + 2;
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int f = ((wv + 4 * k) * 64 + lane) * 4;              // first float of this lane's quad inside the stage
    const int kc = f / G::TILE_ELEMS, e = f % G::TILE_ELEMS;
    const int tz = e / (NP * G::PLANE), xi = (e / G::PLANE) % NP, r = e % G::PLANE;
    const int hz = r / G::HW, wq = r % G::HW;                  // (wq a multiple of 4)
    const int t = t0 + tz, hp = h0 + hz, wp = w0 + wq;         // padded coordinates: row h + 1 = h0 - 1 + hz + 1, column w0 - 1 + wq + 1
    const bool ok = f < G::LDS_IN && t < T && hp < HPd && wp < WPd;
    xsrc[k] = ok ? kc * plane_v + ((t * NP + xi) * HPd + hp) * WPd + wp : -1;
  }
  int wsrc[WK];  // float offset inside the chunk's rows, or -1
#pragma unroll
  for (int k = 0; k < WK; ++k) {
    const int u = (wv + 4 * k) * 64 + lane;
    wsrc[k] = u < G::W_UNITS ? (u / (CO / 4)) * a.cout_g + (u % (CO / 4)) * 4 : -1;
  }
  const float* xg = a.v + ((long)b * a.Cin + (long)g * a.cin_g) * plane_v;
  const float* wg = a.wp + (long)g * a.cin_g * W_TAPS * a.cout_g + co0;
  const float* zero = reinterpret_cast<const float*>(&kZero4);

  constexpr int NPIECE = XK + WK;
  auto issue_piece = [&](auto ic, int c, float* st) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < XK) {
      constexpr int k = i;
      const int q = wv + 4 * k;
      if (G::XP % 4 == 0 || q < G::XP) dma_b128(xsrc[k] >= 0 ? xg + (long)c * KC * plane_v + xsrc[k] : zero, st + q * 256);
    } else if constexpr (i < NPIECE) {
      constexpr int k = i - XK;
      const int q = wv + 4 * k;
      if (G::WP % 4 == 0 || q < G::WP)
        dma_b128(wsrc[k] >= 0 ? wg + (long)c * KC * W_TAPS * a.cout_g + wsrc[k] : zero, st + G::X_FLOATS + q * 256);
    }
  };
  static_for<0, NPIECE>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, 0, st0); });
  __builtin_amdgcn_sched_barrier(0);

  // ---- per-lane operand offsets (floats inside a stage) ----------------------------------------------------------
  const int wpos = wv & 1;                                     // which positions of the tile this wave owns
  const int grp = wv >> 1;                                     // ... and which planes: k = NX * grp .. NX * grp + NX - 1
  const int xi_base = NX * grp;
  int boff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p = (wpos * NS + s) * 32 + l31;
    const int pw = p % TW, ph = (p / TW) % TH, pt = p / (TW * TH);
    boff[s] = ((pt * NP + xi_base) * G::HH + ph) * G::HW + pw + khalf * G::TILE_ELEMS;
  }
  const int aoff = G::X_FLOATS + khalf * W_TAPS * CO + xi_base * CO + l31 * MSUB;  // slab rows hold the tile's channels as [l31][m]

  float bias_v[MSUB], sh2_v[MSUB], sc2_v[MSUB];
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int col = g * a.cout_g + co0 + m * 32 + l31;
    bias_v[m] = a.bias ? a.bias[col] * a.bias_scale : 0.f;
    sh2_v[m] = (a.y2 && a.y2_shift) ? a.y2_shift[(long)b * a.y2_shift_stride + col] : 0.f;
    sc2_v[m] = (a.y2 && a.y2_scale) ? a.y2_scale[(long)b * a.y2_scale_stride + col] : 1.f;
  }

  f32x16 acc[NX * NS][MSUB];    // [(k - xi_base) * NS + s][m]
#pragma unroll
  for (int s = 0; s < NX * NS; ++s)
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  constexpr int MF = NS * MSUB;
  constexpr int WT = 9 * NX;                                  // taps of this wave per channel pair
  constexpr int KSTEPS = (KC / 2) * WT;
  constexpr int PSTRIDE = KSTEPS / NPIECE > 0 ? KSTEPS / NPIECE : 1;
  static_assert(NPIECE <= KSTEPS, "at most one piece per K-step");
  auto mfma_chunk = [&](const float* st, int c_next, float* st_next) __attribute__((always_inline)) {
    float av[2][MSUB], bv[2][NS];
    auto fetch = [&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int kp = ks / WT, tap = ks % WT;
      constexpr int xi = tap % NX, dh = (tap / NX) / 3, dw = (tap / NX) % 3;
      constexpr int toff = (xi * G::HH + dh) * G::HW + dw;
      const float* ap = st + aoff + (kp * 2 * W_TAPS + (tap / NX) * NP + xi) * CO;
      if constexpr (MSUB == 1) {
        av[ks & 1][0] = ap[0];
      } else {
        const float2 t2 = *reinterpret_cast<const float2*>(ap);
        av[ks & 1][0] = t2.x, av[ks & 1][1] = t2.y;
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[ks & 1][s] = st[boff[s] + kp * 2 * G::TILE_ELEMS + toff];
    };
    fetch(std::integral_constant<int, 0>{});
    static_for<0, KSTEPS>([&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int xi = (ks % WT) % NX;
      static_for<0, MF>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int s = j / MSUB, m = j % MSUB;
        acc[xi * NS + s][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][m], bv[ks & 1][s], acc[xi * NS + s][m], 0, 0, 0);
        if constexpr (j == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (ks + 1 < KSTEPS) fetch(std::integral_constant<int, ks + 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (j == MF - 1 && ks % PSTRIDE == 0 && ks / PSTRIDE < NPIECE) {
          __builtin_amdgcn_sched_barrier(0);
          issue_piece(std::integral_constant<int, ks / PSTRIDE>{}, c_next, st_next);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  __syncthreads();
  WINO_STAMP(2);
  for (int c = 0; c < nchunks; c += 2) {
    mfma_chunk(st0, c + 1 < nchunks ? c + 1 : 0, st1);
    __syncthreads();
    if (c + 1 < nchunks) {
      mfma_chunk(st1, c + 2 < nchunks ? c + 2 : 0, st0);
      __syncthreads();
    }
  }

  WINO_STAMP(3);
  // ---- output transform: y_j = sum_k A^T[j][k] m_k.  This wave holds m_k for its NX planes: it forms its part of every y_j,
  // keeps those of the OH bands it finishes (j = OH * grp .. + OH - 1) and sends the other OH to its partner (wave ^ 2), one
  // accumulator (1024 floats) per round through alternating stages, one barrier per round. -------------------------------
  static_assert(sizeof(st0) >= 4 * 4096, "the exchange and the epilogue borrow 4 KB of LDS per wave");
  auto part = [&](auto jc, auto gc, auto sc, auto mc, int r) __attribute__((always_inline)) {   // this wave's part of y_j (its group = gc)
    constexpr int j = decltype(jc)::value, gg = decltype(gc)::value, s = decltype(sc)::value, m = decltype(mc)::value;
    float v = 0.f;
#pragma unroll
    for (int xl = 0; xl < NX; ++xl)
      if (M::AT[j][NX * gg + xl] != 0.f) v += M::AT[j][NX * gg + xl] * acc[xl * NS + s][m][r];
    return v;
  };
  int round = 0;
  // sub-tile by sub-tile: exchange, then the epilogue of its OH output bands (the accumulators of a finished sub-tile are dead:
  // the register file never holds more than the accumulators plus one sub-tile's outputs)
  static_for<0, NS>([&](auto sc) __attribute__((always_inline)) {
    constexpr int s = decltype(sc)::value;
    f32x16 out[OH][MSUB];          // output band MO * pt + OH * grp + jj of this sub-tile
    int sub_base[OH];              // linear position (over the MO TT x TH x TW output tile) of the sub-tile's first position
    const int pb = (wpos * NS + s) * 32;
    const int pt = pb / (TW * TH), rem = pb % (TW * TH);
#pragma unroll
    for (int jj = 0; jj < OH; ++jj) sub_base[jj] = (MO * pt + OH * grp + jj) * (TW * TH) + rem;
    static_for<0, MSUB>([&](auto mc) __attribute__((always_inline)) {
      constexpr int m = decltype(mc)::value;
      static_for<0, OH>([&](auto jjc) __attribute__((always_inline)) {
        constexpr int jj = decltype(jjc)::value;
        float* buf = (round & 1) ? st1 : st0;
        // send the part of the partner's band jj (j = OH * (1 - grp) + jj), keep the part of the own band (j = OH * grp + jj)
        float keep[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float snd;
          if (grp == 0) {
            snd = part(std::integral_constant<int, OH + jj>{}, std::integral_constant<int, 0>{}, sc, mc, r);
            keep[r] = part(std::integral_constant<int, jj>{}, std::integral_constant<int, 0>{}, sc, mc, r);
          } else {
            snd = part(std::integral_constant<int, jj>{}, std::integral_constant<int, 1>{}, sc, mc, r);
            keep[r] = part(std::integral_constant<int, OH + jj>{}, std::integral_constant<int, 1>{}, sc, mc, r);
          }
          buf[(wv * 16 + r) * 64 + lane] = snd;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float got = buf[((wv ^ 2) * 16 + r) * 64 + lane];
          out[jj][m][r] = grp ? got + keep[r] : keep[r] + got;   // (planes in ascending order on either side)
        }
        ++round;
      });
    });
    // the epilogue's 4 KB tile: this wave's region of the stage the LAST round did not use (its partner read that region two
    // rounds ago; the next round writes other waves' regions of it only)
    float* T = ((round & 1) ? st1 : st0) + wv * 1024;
    tmdiff::epilogue_vec<OH, MSUB, MO * TT, TH, TW>(a, out, bias_v, sh2_v, sc2_v, b, g, co0, MO * t0, h0, w0, wv, lane, plane, T,
                                                    sub_base);
  });
#if TMDIFF_WINO_STAMPS
  WINO_STAMP(4);
  __builtin_amdgcn_s_waitcnt(0);          // every store of this wave has been acknowledged
  WINO_STAMP(5);
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((unsigned long long)blockIdx.x * 4 + wv) * 8;
    for (int i = 0; i < 6; ++i) o[i] = stamp_t[i];
    o[6] = ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32) | __builtin_amdgcn_s_getreg(6 | (31 << 11)) |
           ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xf) << 28);
    o[7] = __builtin_amdgcn_s_memtime() - stamp_c0;
  }
#endif
}

template <int NS, int MSUB, int KC, int TT, int TH, int TW, int NP>
int launch(WinoArgs& a, hipStream_t st) {
  constexpr int CO = 32 * MSUB, MO = WM<NP>::MO;
  a.tiles_t = (a.N / MO + TT - 1) / TT;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  a.tiles_co = a.cout_g / CO;
  const long blocks = (long)a.B * a.groups * a.tiles_t * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d_wino_fwd: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  // stagger (see stagger_start): TMDIFF_WINO_STAGGER = delay in units of one chunk's MFMA time of a workgroup pair
  // (2 x KSTEPS x MF x 64 cycles); only worth it when there are several rounds of workgroups
  static const double stagger_chunks = [] {
    const char* e = getenv("TMDIFF_WINO_STAGGER");
    return e ? atof(e) : 0.0;
  }();
  constexpr int NPIECE_UNUSED = 0; (void)NPIECE_UNUSED;
  const double chunk_cycles = 2.0 * (KC / 2) * 9 * (NP / 2) * NS * MSUB * 64;
  a.first_round = 512;
  a.stagger = blocks > 512 ? (int)(stagger_chunks * chunk_cycles) : 0;
  conv3d_wino_kernel<NS, MSUB, KC, TT, TH, TW, NP><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_wino_fwd");
}

// packed[g][ci][tap9][k][col(co)] = (G g)[k] of w[g * cout_g + co][ci][.][dh][dw]   (ci, co inside the group)
// (column order inside a 64-channel tile as tmdiff_conv3d_pack_weights: channel c at (c % 32) * 2 + c / 32; mode | 2: natural
// column order, for the 32-channel tiles of conv3d_wf.hip; mode | 1: the data-gradient form -- this convolution's (co, ci) are
// the forward one's (ci, co), every tap mirrored)
// One thread packs one WEIGHT ROW: the 27 taps of one (output column, input channel) pair -- 108 contiguous bytes in, 9 x NP
// values out, each of them in a run of consecutive columns written by consecutive lanes.  (One thread per packed ELEMENT read
// three words 36 bytes apart for every element, lanes a whole weight row apart: the re-pack of a finetune step took 0.9 ms.)
// row index r = (g * cin_g + ci) * cout_g + col of THIS convolution (for the data-gradient form the forward weight's
// dimensions swapped).
template <int NP>
__device__ __forceinline__ void wino_pack_row(const float* __restrict__ w, float* __restrict__ packed, long r, int cout_g, int cin_g, int mode) {
#pragma clang fp contract(off)     // (the single- and the multi-tensor kernel must give the same bits: no per-site FMA fusion)
  using M = WM<NP>;
  const int col = (int)(r % cout_g);
  r /= cout_g;
  const int ci = (int)(r % cin_g);
  const int g = (int)(r / cin_g);
  int co = col;
  if (cout_g % 64 == 0 && !(mode & 2)) {
    const int tile = col / 64, j = col % 64;
    co = tile * 64 + (j % 2) * 32 + j / 2;
  }
  const bool dg = (mode & 1) != 0;
  const float* wk = dg ? w + (((long)g * cin_g + ci) * cout_g + co) * 27 : w + (((long)g * cout_g + co) * cin_g + ci) * 27;
  float t[27];
#pragma unroll
  for (int i = 0; i < 27; ++i) t[i] = wk[i];
  float* dst = packed + (((long)g * cin_g + ci) * 9) * NP * cout_g + col;
#pragma unroll
  for (int tap9 = 0; tap9 < 9; ++tap9) {
    // forward form: taps (dn, tap9); data-gradient form: the mirrored taps (2 - dn, 8 - tap9)
    const float g0 = dg ? t[18 + 8 - tap9] : t[tap9], g1 = dg ? t[9 + 8 - tap9] : t[9 + tap9], g2 = dg ? t[8 - tap9] : t[18 + tap9];
#pragma unroll
    for (int k = 0; k < NP; ++k) dst[((long)tap9 * NP + k) * cout_g] = M::G[k][0] * g0 + M::G[k][1] * g1 + M::G[k][2] * g2;
  }
}

template <int NP>
__global__ void __launch_bounds__(256) wino_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int cout_g,
                                                                int cin_g, int mode, long rows) {
  for (long r = blockIdx.x * 256L + threadIdx.x; r < rows; r += 256L * gridDim.x) wino_pack_row<NP>(w, packed, r, cout_g, cin_g, mode);
}

// Multi-tensor form: every (weight, mode) of a network in ONE launch.  Workgroup k packs weight rows
// [chunk_index[k] * WINO_MT_CHUNK, + WINO_MT_CHUNK) of entry chunk_tensor[k].
constexpr int WINO_MT_CHUNK = 256;
__global__ void __launch_bounds__(256) wino_pack_weights_multi_kernel(const tmdiff_wino_pack_entry* __restrict__ entries,
                                                                      const int32_t* __restrict__ chunk_tensor,
                                                                      const int32_t* __restrict__ chunk_index) {
  const tmdiff_wino_pack_entry e = entries[chunk_tensor[blockIdx.x]];
  // Cout / Cin of THIS convolution (for the data-gradient form the forward weight's dimensions swapped)
  const int cout_g = ((e.mode & 1) ? e.Cin : e.Cout) / e.groups, cin_g = ((e.mode & 1) ? e.Cout : e.Cin) / e.groups;
  const long rows = (long)cin_g * cout_g * e.groups;
  const long r = (long)chunk_index[blockIdx.x] * WINO_MT_CHUNK + threadIdx.x;
  if (r >= rows) return;
  if (e.planes == 6) wino_pack_row<6>(e.w, e.packed, r, cout_g, cin_g, e.mode);
  else wino_pack_row<4>(e.w, e.packed, r, cout_g, cin_g, e.mode);
}

bool wino_ok(const tmdiff_conv3d_desc* d) {
  if (!d || d->ksize != 3 || (d->groups != 1 && d->groups != 3) || d->in_mask || d->x_bf16 || d->y2_bf16) return false;
  if (!(d->drop_p >= 0.f && d->drop_p < 1.f)) return false;
  if (d->nseg < 1 || d->nseg > 3 || d->Cin <= 0 || d->Cout <= 0 || d->Cin % d->groups || d->Cout % d->groups) return false;
  return (d->Cin / d->groups) % 2 == 0 && (d->Cout / d->groups) % 32 == 0 && d->N > 0 && d->N % 2 == 0 && d->H > 0 && d->W > 0 &&
         d->W % 4 == 0;
}

// F(4,3) when the band count is a multiple of four and its tiles fill the kernel's pairs of tiles at least as well as F(2,3)'s
// would (a workgroup covers TT = 2 tiles along the bands: with N = 4 the second F(4,3) tile would be empty -- 2 x 13.5
// multiply-adds per output against F(2,3)'s 18), else F(2,3); TMDIFF_WINO_F4=0 (experiments): always F(2,3)
int planes_for(int N) {
  static const bool f4 = [] {
    const char* e = getenv("TMDIFF_WINO_F4");
    return !(e && e[0] == '0');
  }();
  if (!f4 || N % 4) return 4;
  const int t6 = N / 4, t4 = N / 2;
  const double cost6 = 13.5 * (2 * ((t6 + 1) / 2)) / t6, cost4 = 18.0 * (2 * ((t4 + 1) / 2)) / t4;
  return cost6 <= cost4 ? 6 : 4;
}

}  // namespace

extern "C" int tmdiff_conv3d_wino_supported(const tmdiff_conv3d_desc* d) { return wino_ok(d) ? 1 : 0; }

/* planes of the transform this library uses for N bands (6: F(4,3), 4: F(2,3)): the `planes` argument of the weight packing */
extern "C" int32_t tmdiff_conv3d_wino_planes(int32_t N) { return N > 0 && N % 2 == 0 ? planes_for(N) : 0; }

// workgroups the convolution kernel would launch (0 = shape not supported): callers keep small grids on tmdiff_conv3d_fwd,
// whose split-K fills the chip
extern "C" int64_t tmdiff_conv3d_wino_blocks(const tmdiff_conv3d_desc* d) {
  if (!wino_ok(d) || d->B <= 0) return 0;
  const int T = d->N / (planes_for(d->N) - 2), cg = d->Cout / d->groups;
  if (cg % 64 == 0) return (int64_t)d->B * d->groups * ((T + 1) / 2) * ((d->H + 7) / 8) * ((d->W + 7) / 8) * (cg / 64);
  return (int64_t)d->B * d->groups * ((T + 1) / 2) * ((d->H + 7) / 8) * ((d->W + 15) / 16) * (cg / 32);
}

extern "C" size_t tmdiff_conv3d_wino_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!wino_ok(d) || d->B <= 0) return 0;
  return (size_t)d->B * d->Cin * (d->N / 2) * 4 * (d->H + 2) * (d->W + 4) * sizeof(float);   // (F(2,3): the larger of the two)
}

extern "C" size_t tmdiff_conv3d_wino_packed_bytes(int32_t Cout, int32_t Cin, int32_t groups, int32_t planes) {
  if (groups < 1 || Cout <= 0 || Cin <= 0 || Cout % groups || Cin % groups || (Cout / groups) % 32 || (planes != 4 && planes != 6)) return 0;
  return (size_t)(Cin / groups) * 9 * planes * Cout * sizeof(float);
}

extern "C" int tmdiff_conv3d_wino_pack_weights(const float* w, float* packed, int32_t Cout, int32_t Cin, int32_t groups,
                                               int32_t mode, int32_t planes, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(w && packed && aligned16(packed), "conv3d_wino_pack_weights: NULL / unaligned pointer");
  TMDIFF_REQUIRE(groups >= 1 && Cout > 0 && Cin > 0 && Cout % groups == 0 && Cin % groups == 0 && (Cout / groups) % 32 == 0,
                 "conv3d_wino_pack_weights: Cout=%d Cin=%d groups=%d (Cout/groups a multiple of 32)", Cout, Cin, groups);
  TMDIFF_REQUIRE(mode >= 0 && mode <= 3, "conv3d_wino_pack_weights: mode=%d (bit 0: data-gradient form, bit 1: natural column order)", mode);
  TMDIFF_REQUIRE(planes == 4 || planes == 6, "conv3d_wino_pack_weights: planes=%d (tmdiff_conv3d_wino_planes)", planes);
  const long rows = (long)(Cin / groups) * Cout;
  long blocks = (rows + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (planes == 6)
    wino_pack_weights_kernel<6><<<(int)blocks, 256, 0, as_stream(stream)>>>(w, packed, Cout / groups, Cin / groups, mode, rows);
  else
    wino_pack_weights_kernel<4><<<(int)blocks, 256, 0, as_stream(stream)>>>(w, packed, Cout / groups, Cin / groups, mode, rows);
  return check_launch("conv3d_wino_pack_weights");
}

extern "C" int32_t tmdiff_conv3d_wino_pack_weights_multi_chunk(void) { return WINO_MT_CHUNK; }

extern "C" int tmdiff_conv3d_wino_pack_weights_multi(const tmdiff_wino_pack_entry* entries_dev, const int32_t* chunk_tensor_dev,
                                                     const int32_t* chunk_index_dev, int32_t n_chunks, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(entries_dev && chunk_tensor_dev && chunk_index_dev && n_chunks >= 0, "conv3d_wino_pack_weights_multi: bad arguments");
  if (n_chunks == 0) return TMDIFF_OK;
  wino_pack_weights_multi_kernel<<<(unsigned)n_chunks, 256, 0, as_stream(stream)>>>(entries_dev, chunk_tensor_dev, chunk_index_dev);
  return check_launch("conv3d_wino_pack_weights_multi");
}

// stage: 0 = input transform + convolution, 1 = the transform pass alone (fills the workspace), 2 = the convolution alone
// (the workspace already holds this input's transform: a measurement, or several convolutions of one tensor)
// planes: 6 (N % 4 == 0 only) or 4 = the transform the weights were packed for; 0 = tmdiff_conv3d_wino_planes(N)
extern "C" int tmdiff_conv3d_wino_fwd_planes(const tmdiff_conv3d_desc* d, void* workspace, int32_t stage, float* xp_out,
                                             int32_t planes, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d != nullptr, "conv3d_wino_fwd: NULL descriptor");
  if (!wino_ok(d))
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wino_fwd: fp32 3x3x3, groups 1 or 3, even N, W %% 4 == 0, Cin/g %% 2 == 0, Cout/g %% 32 == 0, no mask");
  TMDIFF_REQUIRE(d->B >= 0, "conv3d_wino_fwd: bad extents");
  if (d->B == 0) return TMDIFF_OK;
  const int np = planes ? planes : planes_for(d->N), mo = np - 2;
  TMDIFF_REQUIRE((np == 6 && d->N % 4 == 0) || np == 4, "conv3d_wino_fwd: planes=%d with N=%d", planes, d->N);
  TMDIFF_REQUIRE(workspace && aligned16(workspace), "conv3d_wino_fwd: needs its workspace (tmdiff_conv3d_wino_workspace_bytes)");
  TMDIFF_REQUIRE(d->w_packed && (d->y || d->y2) && aligned16(d->w_packed), "conv3d_wino_fwd: NULL / unaligned weights or output");
  TMDIFF_REQUIRE((long)d->Cin * (d->N / mo) * np * (d->H + 2) * (d->W + 4) < (1L << 31) / 2, "conv3d_wino_fwd: input too large for 32-bit offsets");
  TMDIFF_REQUIRE((long)d->B * d->Cin <= 65535, "conv3d_wino_fwd: B*Cin = %ld exceeds the grid", (long)d->B * d->Cin);
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    TMDIFF_REQUIRE(d->seg_x[i] != nullptr && d->seg_c[i] > 0 && aligned16(d->seg_x[i]), "conv3d_wino_fwd: segment %d is empty / unaligned", i);
    csum += d->seg_c[i];
  }
  TMDIFF_REQUIRE(csum == d->Cin, "conv3d_wino_fwd: segments hold %d channels, Cin=%d", csum, d->Cin);
  hipStream_t st = as_stream(stream);

  WinoInArgs q;
  q.B = d->B; q.Cin = d->Cin; q.N = d->N; q.H = d->H; q.W = d->W; q.nseg = d->nseg;
  for (int i = 0; i < 3; ++i) { q.seg_c[i] = i < d->nseg ? d->seg_c[i] : 0; q.seg_x[i] = i < d->nseg ? d->seg_x[i] : nullptr; }
  q.in_shift = d->in_shift; q.in_scale = d->in_scale; q.in_act = d->in_act;
  q.shift_stride = d->in_shift_stride > 0 ? d->in_shift_stride : (d->in_shift_stride < 0 ? 0 : d->Cin);
  q.scale_stride = d->in_scale_stride > 0 ? d->in_scale_stride : (d->in_scale_stride < 0 ? 0 : d->Cin);
  q.v = static_cast<float*>(workspace);
  q.xp = xp_out;
  q.drop_seed = d->drop_seed; q.drop_seed_dev = d->drop_seed_dev; q.drop_thresh = drop_threshold(d->drop_p);
  q.drop_inv = d->drop_p > 0.f ? 1.0f / (1.0f - d->drop_p) : 0.f;
  TMDIFF_REQUIRE(!xp_out || aligned16(xp_out), "conv3d_wino_fwd: xp_out must be 16-byte aligned");
  TMDIFF_REQUIRE(stage >= 0 && stage <= 2, "conv3d_wino_fwd: stage=%d", stage);
  if (stage != 2) {
    long pb = ((long)(d->N / mo) * d->H * d->W / 4 + 255) / 256;
    if (pb > 64) pb = 64;
    const dim3 grid((unsigned)pb, (unsigned)(d->B * d->Cin));
    if (np == 6) wino_input_kernel<6><<<grid, 256, 0, st>>>(q);
    else wino_input_kernel<4><<<grid, 256, 0, st>>>(q);
    const int rc = check_launch("conv3d_wino_fwd (input transform)");
    if (rc || stage == 1) return rc;
  }

  WinoArgs a;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cout = d->Cout; a.groups = d->groups; a.cin_g = d->Cin / d->groups; a.cout_g = d->Cout / d->groups;
  a.v = q.v; a.wp = d->w_packed;
  a.bias = d->bias; a.bias_scale = d->bias_scale;
  a.residual = d->residual; a.out_scale = d->out_scale; a.y = d->y;
  a.y2 = d->y2; a.y2_shift = d->y2_shift; a.y2_scale = d->y2_scale; a.y2_act = d->y2_act;
  a.y2_shift_stride = d->y2_shift_stride > 0 ? d->y2_shift_stride : (d->y2_shift_stride < 0 ? 0 : d->Cout);
  a.y2_scale_stride = d->y2_scale_stride > 0 ? d->y2_scale_stride : (d->y2_scale_stride < 0 ? 0 : d->Cout);
  a.vec4 = 1;
  a.stamps = TMDIFF_WINO_STAMPS ? static_cast<unsigned long long*>(d->splitk_ws) : nullptr;
  if (!(d->W % 4 == 0 && aligned16(d->y) && aligned16(d->y2) && aligned16(d->residual) && (long)d->N * d->H * d->W <= (1L << 23)))
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wino_fwd: W %% 4 == 0, 16-byte aligned outputs / residual, planes of at most 2^23 positions");
  if (np == 6) {
    if (a.cout_g % 64 == 0) return launch<2, 2, 2, 2, 8, 8, 6>(a, st);
    return launch<4, 1, 2, 2, 8, 16, 6>(a, st);
  }
  if (a.cout_g % 64 == 0) return launch<2, 2, 2, 2, 8, 8, 4>(a, st);
  return launch<4, 1, 2, 2, 8, 16, 4>(a, st);
}

extern "C" int tmdiff_conv3d_wino_fwd_xp(const tmdiff_conv3d_desc* d, void* workspace, int32_t stage, float* xp_out,
                                         tmdiff_stream_t stream) {
  return tmdiff_conv3d_wino_fwd_planes(d, workspace, stage, xp_out, 0, stream);
}

extern "C" int tmdiff_conv3d_wino_fwd_stage(const tmdiff_conv3d_desc* d, void* workspace, int32_t stage, tmdiff_stream_t stream) {
  return tmdiff_conv3d_wino_fwd_planes(d, workspace, stage, nullptr, 0, stream);
}

extern "C" int tmdiff_conv3d_wino_fwd(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream) {
  return tmdiff_conv3d_wino_fwd_planes(d, workspace, 0, nullptr, 0, stream);
}
