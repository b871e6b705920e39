// "3x3x3 convolution, then the halved Haar LL band" as ONE strided convolution on the exact-fp32 matrix cores.
//
// Where: WaveletUPorDown(down=True) of the main (x_t) branch -- h = Conv_0(SiLU(x)), then only hLL / 2 of dwt(h) is used
// (reference Hyper_unet_general.py:371-372, :389, :396; the high bands of this branch are dropped, the up path takes the
// condition branch's).  The LL band is linear and local:
//     LL(y)[i][j] * ll_scale = s * sum_{P,Q in {0,1}} y[2i+P][2j+Q],      s = ll_scale / 2,
// so with y = conv3x3x3(x) the whole thing is a convolution with a 3 x 4 x 4 kernel and stride (1, 2, 2):
//     out[n][i][j] = sum_{dn} sum_{a,b in 0..3} W'[dn][a][b] x[n+dn-1][2i-1+a][2j-1+b],
//     W'[dn][a][b] = s * sum_{P,Q} W[dn][a-P][b-Q]   (terms with an index outside 0..2 dropped),
// 48 multiply-adds per (ci, co, output position) instead of 4 x 27 = 108 for the four full-resolution positions whose mean
// it is: 2.25x fewer FLOPs for the same numbers (up to fp32 summation order), and the full-resolution h is never written.
// The bias passes through unchanged when ll_scale = 1/2 (the halved LL band of a constant is the constant); in general it
// is multiplied by 2 * ll_scale (entry point).
//
// Kernel: conv3d_dma_kernel (conv3d_dma.hip) with another box geometry.  The stride-2 window of an output tile is kept in
// LDS de-interleaved by row / column parity ("space to depth", done by the dword LDS-DMA gather for free): group (p, q)
// holds the input rows 2(i0 + hz) - p and columns 2(j0 + wz) - q for hz = 0..TH, wz = 0..TW.  The 4 x 4 window of output
// (i, j) is then, per group, the 2 x 2 cells (hz, wz) = (i - i0 + e, j - j0 + f), e, f in {0, 1}: a tap is (group, dn, e, f)
// = 48 per input channel, every operand read has a compile-time offset, and consecutive lanes read consecutive LDS words.
// Weights arrive composed and packed as [ci][48 taps][co] (tmdiff_conv3d_ll_pack_weights).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "epilogue.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

struct LlArgs {
  int B, N, H, W;       // OUTPUT extents (H = Hi / 2, W = Wi / 2)
  int Hi, Wi;           // input extents
  int Cin, Cout, cout_g;
  const float* xq;      // x [B, Cin, N, Hi, Wi]
  const float* wp;      // composed + packed [ci][48][co]
  const float* bias;
  float bias_scale;
  const float* residual;
  float out_scale;
  float* y;
  float* y2;
  const float* y2_shift;
  const float* y2_scale;
  int y2_shift_stride, y2_scale_stride, y2_act;
  int tiles_n, tiles_h, tiles_w, tiles_co;
  unsigned total_blocks;
  int vec4;
  int ksplit, split_chunks;  // split-K over the input channels: ksplit ranges of split_chunks chunks (1, Cin / KC = no split)
  float* part;               // partial outputs [ksplit][B][Cout][plane] (NULL = no split), summed by splitk_reduce_kernel
};

__device__ const float4 kZero4 = {0.f, 0.f, 0.f, 0.f};  // source of zero padding / filler lanes

__device__ __forceinline__ void dma_b32(const float* src, float* dst) {
#if defined(__HIP_DEVICE_COMPILE__)  // the builtin exists in the device pass only
  __builtin_amdgcn_global_load_lds(src, dst, 4, 0, 0);
#endif
}
__device__ __forceinline__ void dma_b128(const float* src, float* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
#endif
}

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, k = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

constexpr int LL_TAPS = 48;   // (group p,q) x dn x e x f = 4 x 3 x 2 x 2

template <int NS, int MSUB, int KC, int TN, int TH, int TW>
struct GeoLL {
  static constexpr int CO = 32 * MSUB;
  static constexpr int HN = TN + 2, CH = TH + 1, CW = TW + 1;
  static constexpr int GRP = HN * CH * CW;                  // one parity group of one channel
  static constexpr int TILE_ELEMS = 4 * GRP;
  static constexpr int LDS_IN = KC * TILE_ELEMS;            // input box, [KC][4 groups][HN][CH][CW]
  static constexpr int XP = (LDS_IN + 63) / 64;             // dword pieces (64 floats each)
  static constexpr int X_FLOATS = XP * 64;
  static constexpr int W_UNITS = KC * LL_TAPS * CO / 4;     // weight slab [KC][48][CO] in 16-byte units
  static constexpr int WP = (W_UNITS + 63) / 64;
  static constexpr int STAGE = X_FLOATS + WP * 256;
  static_assert(TN * TH * TW == 4 * NS * 32, "workgroup tile = 4 waves x NS sub-tiles x 32 positions");
  static_assert(KC % 2 == 0, "K step is 2 channels");
};

template <int NS, int MSUB, int KC, int TN, int TH, int TW>
__global__ void __launch_bounds__(256, 2) conv3d_ll_kernel(const LlArgs a) {
  using G = GeoLL<NS, MSUB, KC, TN, TH, TW>;
  constexpr int CO = G::CO;
  constexpr int XK = (G::XP + 3) / 4, WK = (G::WP + 3) / 4;  // pieces per wave
  __shared__ __attribute__((aligned(16))) float st0[G::STAGE];
  __shared__ __attribute__((aligned(16))) float st1[G::STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;

  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int tw_i = __builtin_amdgcn_readfirstlane(id % a.tiles_w); id /= a.tiles_w;
  const int th_i = __builtin_amdgcn_readfirstlane(id % a.tiles_h); id /= a.tiles_h;
  const int tn_i = __builtin_amdgcn_readfirstlane(id % a.tiles_n); id /= a.tiles_n;
  const int b = __builtin_amdgcn_readfirstlane(id % a.B);
  const int split = __builtin_amdgcn_readfirstlane(id / a.B);   // split-K range of this workgroup (outermost index)
  constexpr int g = 0;
  const int n0 = tn_i * TN, h0 = th_i * TH, w0 = tw_i * TW;   // output coordinates
  const int co0 = co_tile * CO;
  const long plane = (long)a.N * a.H * a.W;                    // output plane
  const int plane_in = a.N * a.Hi * a.Wi;
  const int nchunks = a.split_chunks;                            // chunks of this workgroup: [split * nchunks, +nchunks)

  // ---- DMA sources of this lane (the same for every chunk) -----------------------------------------------------
  int xsrc[XK];  // float offset from the chunk base, or -1 = zero word
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int f = (wv + 4 * k) * 64 + lane;
    const int kc = f / G::TILE_ELEMS, e = f % G::TILE_ELEMS;
    const int grp = e / G::GRP, r = e % G::GRP;
    const int wz = r % G::CW, hz = (r / G::CW) % G::CH, nz = r / (G::CW * G::CH);
    const int n = n0 + nz - 1, h = 2 * (h0 + hz) - (grp >> 1), w = 2 * (w0 + wz) - (grp & 1);
    const bool ok = f < G::LDS_IN && n >= 0 && n < a.N && h >= 0 && h < a.Hi && w >= 0 && w < a.Wi;
    xsrc[k] = ok ? kc * plane_in + (n * a.Hi + h) * a.Wi + w : -1;
  }
  int wsrc[WK];  // float offset inside the chunk's rows, or -1
#pragma unroll
  for (int k = 0; k < WK; ++k) {
    const int u = (wv + 4 * k) * 64 + lane;
    wsrc[k] = u < G::W_UNITS ? (u / (CO / 4)) * a.cout_g + (u % (CO / 4)) * 4 : -1;
  }
  const long c_first = (long)split * nchunks * KC;               // first input channel of the range
  const float* xg = a.xq + ((long)b * a.Cin + c_first) * plane_in;
  const float* wg = a.wp + c_first * LL_TAPS * a.cout_g + co0;
  const float* zero = reinterpret_cast<const float*>(&kZero4);

  constexpr int NPIECE = XK + WK;
  auto issue_piece = [&](auto ic, int c, float* st) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < XK) {
      constexpr int k = i;
      const int q = wv + 4 * k;
      if (G::XP % 4 == 0 || q < G::XP) dma_b32(xsrc[k] >= 0 ? xg + (long)c * KC * plane_in + xsrc[k] : zero, st + q * 64);
    } else if constexpr (i < NPIECE) {
      constexpr int k = i - XK;
      const int q = wv + 4 * k;
      if (G::WP % 4 == 0 || q < G::WP)
        dma_b128(wsrc[k] >= 0 ? wg + (long)c * KC * LL_TAPS * a.cout_g + wsrc[k] : zero, st + G::X_FLOATS + q * 256);
    }
  };
  static_for<0, NPIECE>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, 0, st0); });
  __builtin_amdgcn_sched_barrier(0);

  // ---- per-lane operand offsets (floats inside a stage) ----------------------------------------------------------
  int boff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p = (wv * NS + s) * 32 + l31;
    const int pw = p % TW, ph = (p / TW) % TH, pn = p / (TW * TH);
    boff[s] = (pn * G::CH + ph) * G::CW + pw + khalf * G::TILE_ELEMS;
  }
  const int aoff = G::X_FLOATS + khalf * LL_TAPS * CO + l31 * MSUB;  // slab rows hold the tile's channels as [l31][m]

  float bias_v[MSUB], sh2_v[MSUB], sc2_v[MSUB];
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int col = co0 + m * 32 + l31;
    bias_v[m] = a.bias ? a.bias[col] * a.bias_scale : 0.f;
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

  constexpr int MF = NS * MSUB;
  constexpr int KSTEPS = (KC / 2) * LL_TAPS;
  constexpr int PSTRIDE = KSTEPS / NPIECE > 0 ? KSTEPS / NPIECE : 1;
  static_assert(NPIECE <= KSTEPS, "at most one piece per K-step");
  auto mfma_chunk = [&](const float* st, int c_next, float* st_next) __attribute__((always_inline)) {
    float av[2][MSUB], bv[2][NS];
    auto fetch = [&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int kp = ks / LL_TAPS, tap = ks % LL_TAPS;
      constexpr int grp = tap / 12, dn = (tap % 12) / 4, e = (tap / 2) % 2, f = tap % 2;
      constexpr int toff = ((grp * G::HN + dn) * G::CH + e) * G::CW + f;
      const float* ap = st + aoff + (kp * 2 * LL_TAPS + tap) * CO;
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
      static_for<0, MF>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int s = j / MSUB, m = j % MSUB;
        acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][m], bv[ks & 1][s], acc[s][m], 0, 0, 0);
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
  for (int c = 0; c < nchunks; c += 2) {
    // (past the last chunk the pieces of chunk 0 are fetched again into the idle stage: valid addresses, nobody reads them)
    mfma_chunk(st0, c + 1 < nchunks ? c + 1 : 0, st1);
    __syncthreads();
    if (c + 1 < nchunks) {
      mfma_chunk(st1, c + 2 < nchunks ? c + 2 : 0, st0);
      __syncthreads();
    }
  }

  if (a.part) {   // split-K: raw partial sums; splitk_reduce_kernel (conv3d.hip) adds them up and applies the epilogue
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int p = (wv * NS + s) * 32 + l31;
        const int n = n0 + p / (TW * TH), h = h0 + (p / TW) % TH, w = w0 + p % TW;
        const bool pok = n < a.N && h < a.H && w < a.W;
        const long sp = pok ? ((long)n * a.H + h) * a.W + w : 0;
        float* dst = a.part + (((long)split * a.B + b) * a.Cout + co0 + m * 32 + 4 * khalf) * plane + sp;
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (pok) dst[((r & 3) + 8 * (r >> 2)) * plane] = acc[s][m][r];
      }
    return;
  }
  if (a.vec4) {   // (the chunk loop ends with a barrier: nobody reads the stages any more)
    static_assert(sizeof(st0) >= 4 * 4096, "the epilogue borrows 4 KB of LDS per wave");
    tmdiff::epilogue_vec<NS, MSUB, TN, TH, TW>(a, acc, bias_v, sh2_v, sc2_v, b, g, co0, n0, h0, w0, wv, lane, plane,
                                               st0 + wv * 1024);
    return;
  }
  // ---- scalar epilogue (odd widths / unaligned tensors): D layout col = lane&31 (position), row = channel ---------
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int p = (wv * NS + s) * 32 + l31;
      const int n = n0 + p / (TW * TH), h = h0 + (p / TW) % TH, w = w0 + p % TW;
      const bool pok = n < a.N && h < a.H && w < a.W;
      const long sp = pok ? ((long)n * a.H + h) * a.W + w : 0;
      const long obase = ((long)b * a.Cout + co0 + m * 32 + 4 * khalf) * plane + sp;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const int ch = row + 4 * khalf;                      // channel inside the 32-channel sub-tile
        const float bias = tmdiff::lane_value(bias_v[m], ch);
        const float res = (a.residual && pok) ? a.residual[obase + row * plane] : 0.f;
        const float v = (acc[s][m][r] + bias + res) * a.out_scale;
        if (pok && a.y) a.y[obase + row * plane] = v;
        if (a.y2) {
          const float t = v + tmdiff::lane_value(sh2_v[m], ch);
          const float ta = tmdiff::silu_f(t);
          const float u = (a.y2_act ? ta : t) * tmdiff::lane_value(sc2_v[m], ch);
          if (pok) a.y2[obase + row * plane] = u;
        }
      }
    }
  }
}

template <int NS, int MSUB, int KC, int TN, int TH, int TW>
int launch(LlArgs& a, hipStream_t st) {
  constexpr int CO = 32 * MSUB;
  a.tiles_n = (a.N + TN - 1) / TN;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  a.tiles_co = a.Cout / CO;
  const long blocks = (long)a.ksplit * a.B * a.tiles_n * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d_ll_fwd: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  conv3d_ll_kernel<NS, MSUB, KC, TN, TH, TW><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_ll_fwd");
}

// packed[ci][tap][col(co)] = s * sum over the 3x3 taps that feed (group, e, f) of w[co][ci][dn][.][.]
// (column order inside a 64-channel tile as tmdiff_conv3d_pack_weights: channel c at (c % 32) * 2 + c / 32)
__global__ void __launch_bounds__(256) ll_pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout,
                                                              int Cin, float s, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += 256L * gridDim.x) {
    const int col = (int)(i % Cout);
    long r = i / Cout;
    const int tap = (int)(r % LL_TAPS);
    const int ci = (int)(r / LL_TAPS);
    const int tile = col / 64, j = col % 64;
    const int co = tile * 64 + (j % 2) * 32 + j / 2;
    const int grp = tap / 12, dn = (tap % 12) / 4, e = (tap / 2) % 2, f = tap % 2;
    const int p = grp >> 1, q = grp & 1;
    // rows of the 3x3 kernel that reach the window row (p, e):  p=0: e=0 -> {0,1}, e=1 -> {2};  p=1: e=0 -> {0}, e=1 -> {1,2}
    const int a_lo = p == 0 ? (e == 0 ? 0 : 2) : (e == 0 ? 0 : 1), a_hi = p == 0 ? (e == 0 ? 1 : 2) : (e == 0 ? 0 : 2);
    const int b_lo = q == 0 ? (f == 0 ? 0 : 2) : (f == 0 ? 0 : 1), b_hi = q == 0 ? (f == 0 ? 1 : 2) : (f == 0 ? 0 : 2);
    const float* wk = w + (((long)co * Cin + ci) * 3 + dn) * 9;
    float acc = 0.f;
    for (int aa = a_lo; aa <= a_hi; ++aa)
      for (int bb = b_lo; bb <= b_hi; ++bb) acc += wk[aa * 3 + bb];
    packed[i] = acc * s;
  }
}

bool ll_ok(const tmdiff_conv3d_desc* d) {
  if (!d || d->ksize != 3 || d->groups != 1 || d->nseg != 1) return false;
  if (d->in_shift || d->in_scale || d->in_mask || d->in_act || d->drop_p > 0.f || d->x_bf16 || d->y2_bf16) return false;
  return d->Cin > 0 && d->Cin % 2 == 0 && d->Cout > 0 && d->Cout % 64 == 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0;
}

// tile choice and split-K factor (as plan_conv3 for the stride-1 kernels): 256-position tiles, 128-position ones (two
// bands) when those would not give every CU two workgroups; below the split target the input channels are divided
struct LlPlan { bool small; long blocks; int ksplit; };
LlPlan plan_ll(const tmdiff_conv3d_desc* d) {
  const int N = d->N, H = d->H / 2, W = d->W / 2;
  LlPlan p{false, 0, 1};
  const long wg256 = (long)d->B * ((N + 3) / 4) * ((H + 7) / 8) * ((W + 7) / 8) * (d->Cout / 64);
  p.small = wg256 < 2 * 256 && N > 2;
  p.blocks = p.small ? (long)d->B * ((N + 1) / 2) * ((H + 7) / 8) * ((W + 7) / 8) * (d->Cout / 64) : wg256;
  static const long target = [] {
    const char* e = getenv("TMDIFF_SPLITK");
    return e ? atol(e) : 384L;
  }();
  if (target <= 0 || p.blocks >= target) return p;
  const int nchunks = d->Cin / 2;
  for (int s = 2; s <= nchunks / 2; ++s) {            // at least two chunks per range
    if (nchunks % s) continue;
    p.ksplit = s;
    if (p.blocks * s >= target) break;
  }
  return p;
}

}  // namespace

extern "C" int tmdiff_conv3d_ll_supported(const tmdiff_conv3d_desc* d) { return ll_ok(d) ? 1 : 0; }

extern "C" size_t tmdiff_conv3d_ll_splitk_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!ll_ok(d) || d->B <= 0) return 0;
  const LlPlan p = plan_ll(d);
  return p.ksplit > 1 ? (size_t)p.ksplit * d->B * d->Cout * d->N * (d->H / 2) * (d->W / 2) * sizeof(float) : 0;
}

extern "C" size_t tmdiff_conv3d_ll_packed_bytes(int32_t Cout, int32_t Cin) {
  if (Cout <= 0 || Cin <= 0 || Cout % 64) return 0;
  return (size_t)Cin * LL_TAPS * Cout * sizeof(float);
}

extern "C" int tmdiff_conv3d_ll_pack_weights(const float* w, float* packed, int32_t Cout, int32_t Cin, float ll_scale,
                                             tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(w && packed && aligned16(packed), "conv3d_ll_pack_weights: NULL / unaligned pointer");
  TMDIFF_REQUIRE(Cout > 0 && Cin > 0 && Cout % 64 == 0, "conv3d_ll_pack_weights: Cout=%d (multiple of 64) Cin=%d", Cout, Cin);
  const long total = (long)Cin * LL_TAPS * Cout;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  ll_pack_weights_kernel<<<(int)blocks, 256, 0, as_stream(stream)>>>(w, packed, Cout, Cin, ll_scale * 0.5f, total);
  return check_launch("conv3d_ll_pack_weights");
}

extern "C" int tmdiff_conv3d_ll_fwd(const tmdiff_conv3d_desc* d, float ll_scale, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d != nullptr, "conv3d_ll_fwd: NULL descriptor");
  if (!ll_ok(d))
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_ll_fwd: one plain fp32 input, 3x3x3, groups 1, even H and W, Cin %% 2 == 0, Cout %% 64 == 0");
  TMDIFF_REQUIRE(d->B >= 0 && d->N > 0, "conv3d_ll_fwd: bad extents");
  if (d->B == 0) return TMDIFF_OK;
  TMDIFF_REQUIRE(d->seg_x[0] && d->seg_c[0] == d->Cin, "conv3d_ll_fwd: the segment must hold all %d input channels", d->Cin);
  TMDIFF_REQUIRE(d->w_packed && (d->y || d->y2) && aligned16(d->w_packed), "conv3d_ll_fwd: NULL / unaligned weights or output");
  TMDIFF_REQUIRE((long)d->Cin * d->N * d->H * d->W < (1L << 31) / 2, "conv3d_ll_fwd: input too large for 32-bit offsets");
  LlArgs a;
  a.B = d->B; a.N = d->N; a.Hi = d->H; a.Wi = d->W; a.H = d->H / 2; a.W = d->W / 2;
  a.Cin = d->Cin; a.Cout = d->Cout; a.cout_g = d->Cout;
  a.xq = d->seg_x[0]; a.wp = d->w_packed;
  a.bias = d->bias; a.bias_scale = d->bias_scale * 2.0f * ll_scale;   // the (scaled) LL band of a constant
  a.residual = d->residual; a.out_scale = d->out_scale; a.y = d->y;
  a.y2 = d->y2; a.y2_shift = d->y2_shift; a.y2_scale = d->y2_scale; a.y2_act = d->y2_act;
  a.y2_shift_stride = d->y2_shift_stride > 0 ? d->y2_shift_stride : (d->y2_shift_stride < 0 ? 0 : d->Cout);
  a.y2_scale_stride = d->y2_scale_stride > 0 ? d->y2_scale_stride : (d->y2_scale_stride < 0 ? 0 : d->Cout);
  static const bool vec_on = [] {
    const char* e = getenv("TMDIFF_EPILOGUE_VEC");
    return !(e && e[0] == '0');
  }();
  a.vec4 = vec_on && a.W % 4 == 0 && aligned16(d->y) && aligned16(d->y2) && aligned16(d->residual) && (long)a.N * a.H * a.W <= (1L << 23);
  hipStream_t st = as_stream(stream);
  const LlPlan plan = plan_ll(d);
  a.ksplit = 1; a.split_chunks = a.Cin / 2; a.part = nullptr;
  const size_t need = (size_t)plan.ksplit * d->B * d->Cout * a.N * a.H * a.W * sizeof(float);
  if (plan.ksplit > 1 && d->splitk_ws && (size_t)d->splitk_ws_bytes >= need && aligned16(d->splitk_ws)) {
    a.ksplit = plan.ksplit; a.split_chunks = a.Cin / 2 / plan.ksplit; a.part = static_cast<float*>(d->splitk_ws);
  }
  const int rc = plan.small ? launch<1, 2, 2, 2, 8, 8>(a, st) : launch<2, 2, 2, 4, 8, 8>(a, st);
  if (rc || !a.part) return rc;
  SplitKReduceArgs r{a.part, a.ksplit, d->B, d->Cout, (long)a.N * a.H * a.W, d->bias, a.bias_scale, d->residual,
                     d->out_scale, d->y, d->y2, d->y2_shift, d->y2_scale, a.y2_shift_stride, a.y2_scale_stride, d->y2_act};
  return launch_splitk_reduce(r, st);
}
