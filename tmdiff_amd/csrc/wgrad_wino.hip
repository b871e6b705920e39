// Weight gradient of the 3x3x3 convolution in the Winograd domain along the band axis (the reference gets dL/dw from
// ATen's conv3d backward: GeneralModel/Hyper_unet_general.py:74, :244, :372; model.py:40-47).
//
// Along the bands the weight gradient of a tile of 4 output bands is a correlation of the tile's 6 input bands x (one band of
// halo either side, zero outside the image) with its 4 gradient bands g, three taps out:  dw[dn] = sum_n g[n] x[n + dn].  That
// is Winograd F(3, 4) -- the transpose of the forward's F(4, 3), same points 0, +-1, +-2, inf:
//     dw = A'^T [ (G' g) (.) (B^T x) ]      B^T 6x6 (the forward's), G' 6x4, A'^T 3x6
// 6 multiply-adds per tile, element and (dh, dw) instead of 12: per plane k = 0..5 the kernel accumulates a 3x3 (h, w)
// weight gradient  m_k[co][ci][dh][dw] = sum_{b, tile, h, w} g^_k[co][.] x^_k[ci][. + (dh, dw)]  on the matrix pipe, and a small
// kernel forms dw[co][ci][dn][dh][dw] = sum_k A'^T[dn][k] m_k from the partial sums.
//
// Both operands are transformed by one HBM pass each into CHANNELS-LAST, zero-padded arrays
//     x^ [group][k][b * T + t][Hb + 2][Wb + 2][CiP]      (image at (+1, +1): the spatial halo is part of the array)
//     g^ [group][k][b * T + t][Hb][Wb][CoP]              (Hb, Wb: plane extents rounded up to whole boxes; CiP, CoP: to 32)
// because the reduction runs over POSITIONS: an MFMA K-step needs 32 channels of one position in 32 lanes.  With the channels
// innermost a box of positions x 32 channels arrives by 16-byte LDS-DMA pieces (8 positions x 128 B each) exactly in the
// [position][channel] order the operand reads want -- conflict-free ds_read_b32 with compile-time offsets, no bounds checks
// anywhere (padding is zero), 39 DMA instructions per workgroup and box of 128 positions against 576 MFMAs.  (The direct
// kernel of backward.hip moves dwords -- 224 DMA instructions per wave and box against 216 MFMAs, its limiter -- because
// NCDHW rows of a haloed box start 4 bytes off 16-byte alignment.)
//
// Workgroup = 4 waves, one (32 co x 32 ci) tile of ONE plane k, a range of boxes (split-K over positions); every wave holds
// the 9 (dh, dw) accumulators and takes two rows of each box; two workgroups per CU (78 KB of LDS each, double-buffered
// boxes); at the end the four waves' partial sums are added through LDS in a fixed order and go to
// ws[split][group][k][(dh, dw)][CoP][CiP]; ww_reduce_kernel sums the splits (fixed order: deterministic) and applies A'^T.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

#if defined(__HIP_DEVICE_COMPILE__)  // the builtins exist in the device pass only
using buf_rsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ buf_rsrc make_rsrc(const float* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
// 16 bytes per lane from base + voff + soff (bytes; soff wave-uniform) to dst + 16 * lane
__device__ __forceinline__ void dma_b128(buf_rsrc r, unsigned voff, unsigned soff, float* dst) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
}
#else
struct buf_rsrc {};
__device__ __forceinline__ buf_rsrc make_rsrc(const float*, unsigned) { return {}; }
__device__ __forceinline__ void dma_b128(buf_rsrc, unsigned, unsigned, float*) {}
#endif

__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, k = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// ---------------------------------------------------------------------------------------------------------------------
// Transform passes.  One workgroup = one row of the padded plane x one band tile x 32 channels x 64 columns: it reads the
// tile's input bands of that row (float4 along w), transforms per element, turns the block around in LDS and writes the six
// planes channels-last (128-byte runs), padding included -- every element of the array is written exactly once, nothing is
// cleared beforehand.
// ---------------------------------------------------------------------------------------------------------------------
struct WwTransformArgs {
  const float* seg_x[3];   // the operand: channel concatenation of up to three plain tensors [B][seg_c][N][H][W]
  int seg_c[3];
  int nseg;
  int B, cg, groups, N, H, W, T;
  int CP, Hb, Wb;
  int halo;                // 1: x^ = B^T x (6 input bands, array rows / columns shifted by one); 0: g^ = G' g (4 input bands)
  float* out;
  int wchunks, cchunks;
  float* bias_part;        // g^ pass only: [workgroup of this channel chunk][CP] sums of the operand over the workgroup's elements, or NULL
};

constexpr int TR_W = 64;   // columns per workgroup

struct WwTransformPair {
  WwTransformArgs x, g;    // both operands in one launch: blockIdx.z < zx belongs to x^, the rest to g^
  int zx;
};

__global__ void __launch_bounds__(256) ww_transform_kernel(const WwTransformPair pr) {
  __shared__ float tile[6 * TR_W * 33];
  const int tid = threadIdx.x;
  const bool is_x = (int)blockIdx.z < pr.zx;
  const WwTransformArgs& a = is_x ? pr.x : pr.g;
  // (the grid is the envelope of the two operands' grids: workgroups outside this operand's own have nothing to do)
  if ((int)blockIdx.x >= a.wchunks * a.cchunks || (int)blockIdx.y >= a.Hb + 2 * a.halo) return;
  const int wc = blockIdx.x % a.wchunks, cc = blockIdx.x / a.wchunks;
  const int hp = blockIdx.y;
  int z = is_x ? blockIdx.z : blockIdx.z - pr.zx;
  const int t = z % a.T; z /= a.T;
  const int g = z % a.groups;
  const int b = z / a.groups;
  const int Hp = a.Hb + 2 * a.halo, Wp = a.Wb + 2 * a.halo;
  const int h = hp - a.halo;
  const bool row_ok = h >= 0 && h < a.H;
  const long hw = (long)a.H * a.W;

  // ---- load + transform: thread = (channel c, quad wq), two quads 32 columns apart -----------------------------------
  {
    const int c = tid >> 3, wq = tid & 7;
    const int cl = cc * 32 + c;                         // channel inside the group
    int cs = g * a.cg + cl, seg = 0;
    if (a.nseg > 1 && cs >= a.seg_c[0]) { cs -= a.seg_c[0]; seg = 1; }
    if (seg == 1 && a.nseg > 2 && cs >= a.seg_c[1]) { cs -= a.seg_c[1]; seg = 2; }
    const int segc = seg == 0 ? a.seg_c[0] : (seg == 1 ? a.seg_c[1] : a.seg_c[2]);
    const float* src = (seg == 0 ? a.seg_x[0] : (seg == 1 ? a.seg_x[1] : a.seg_x[2])) + ((long)b * segc + cs) * a.N * hw + (long)h * a.W;
    const bool c_ok = row_ok && cl < a.cg;
    float bsum = 0.f;        // (bias gradient on the side: this pass reads every element of g exactly once)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int wl = it * 32 + wq * 4, w = wc * TR_W + wl;
      float d[6][4];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int n = 4 * t - a.halo + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_ok && w < a.W && n >= 0 && n < a.N && (a.halo || j < 4)) v = *reinterpret_cast<const float4*>(src + (long)n * hw + w);
        d[j][0] = v.x, d[j][1] = v.y, d[j][2] = v.z, d[j][3] = v.w;
        bsum += (v.x + v.y) + (v.z + v.w);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float o[6];
        if (a.halo) {   // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
          const float d0 = d[0][e], d1 = d[1][e], d2 = d[2][e], d3 = d[3][e], d4 = d[4][e], d5 = d[5][e];
          const float a42 = d4 - 4.f * d2, b31 = d3 - 4.f * d1, c42 = d4 - d2, e31 = d3 - d1;
          o[0] = (4.f * d0 + d4) - 5.f * d2;
          o[1] = a42 + b31;
          o[2] = a42 - b31;
          o[3] = c42 + 2.f * e31;
          o[4] = c42 - 2.f * e31;
          o[5] = (4.f * d1 + d5) - 5.f * d3;
        } else {        // G' = [1/4 0 0 0; -1/6 -1/6 -1/6 -1/6; -1/6 1/6 -1/6 1/6; 1/24 1/12 1/6 1/3; 1/24 -1/12 1/6 -1/3; 0 0 0 1]
          const float g0 = d[0][e], g1 = d[1][e], g2 = d[2][e], g3 = d[3][e];
          const float s02 = g0 + g2, s13 = g1 + g3;
          const float ev = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), od = g1 * (1.f / 12.f) + g3 * (1.f / 3.f);
          o[0] = 0.25f * g0;
          o[1] = (s02 + s13) * (-1.f / 6.f);
          o[2] = (s13 - s02) * (1.f / 6.f);
          o[3] = ev + od;
          o[4] = ev - od;
          o[5] = g3;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) tile[(k * TR_W + wl + e) * 33 + c] = o[k];
      }
    }
    if (a.bias_part) {       // the eight lanes of a channel are neighbours: fixed-order butterfly, lane 0 of the eight writes
      bsum += __shfl_xor(bsum, 1, 64);
      bsum += __shfl_xor(bsum, 2, 64);
      bsum += __shfl_xor(bsum, 4, 64);
      const long blk = (((long)b * a.T + t) * (a.Hb + 2 * a.halo) + blockIdx.y) * a.wchunks + wc;     // (per group: the group is in the column)
      if (wq == 0) a.bias_part[blk * a.groups * a.CP + (long)g * a.CP + cl] = bsum;
    }
  }
  __syncthreads();

  // ---- write-out: thread = (four channels c4, column), 32 columns per sweep -------------------------------------------
  const int lo = wc == 0 ? 0 : wc * TR_W + a.halo;                       // this workgroup's columns of the padded row
  const int hi = wc == a.wchunks - 1 ? Wp : (wc + 1) * TR_W + a.halo;
  const int c4 = tid & 7, pi = tid >> 3;
  const long Q = (long)a.B * a.T, q = (long)b * a.T + t;
#pragma unroll 1
  for (int k = 0; k < 6; ++k) {
    float* dst = a.out + ((((long)g * 6 + k) * Q + q) * Hp + hp) * Wp * a.CP + cc * 32 + 4 * c4;
    for (int pc = lo + pi; pc < hi; pc += 32) {
      const int wl = pc - a.halo - wc * TR_W;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (wl >= 0 && wl < TR_W) {     // (columns beyond the image inside the chunk were loaded as zeros)
        const float* s = tile + (k * TR_W + wl) * 33 + 4 * c4;
        v = make_float4(s[0], s[1], s[2], s[3]);
      }
      *reinterpret_cast<float4*>(dst + (long)pc * a.CP) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The GEMM kernel.
// ---------------------------------------------------------------------------------------------------------------------
struct WwArgs {
  const float* xh;
  const float* gh;
  float* ws;
  int groups, CiP, CoP;
  int Q, Hb, Wb;               // planes per (group, k); padded plane extents
  int tiles_co, tiles_ci;
  int splits, bps;             // split-K over boxes: `splits` ranges of `bps` boxes
  int nbh, nbw, nboxes;        // boxes per plane along h, w; boxes per (group, k)
  unsigned total_blocks;
};

template <int BH, int BW>
struct GeoWW {
  static constexpr int POS = BH * BW;                       // positions of a box (K = POS / 2 steps)
  static constexpr int XH = BH + 2, XW = BW + 2, XPOS = XH * XW;
  static constexpr int GP = POS / 8, XP = (XPOS + 7) / 8;   // 16-byte x 64-lane pieces: 8 positions x 32 channels
  static constexpr int GPW = GP / 4, XPW = (XP + 3) / 4;    // ... per wave
  static constexpr int NPW = GPW + XPW;
  static constexpr int G_FLOATS = POS * 32, X_FLOATS = XP * 8 * 32;
  static constexpr int STAGE = G_FLOATS + X_FLOATS;
  static constexpr int RED = 3 * 4 * 1024;                  // the final reduction: 3 taps x 4 waves x (32 x 32)
  static constexpr int LDS = 2 * STAGE > RED ? 2 * STAGE : RED;
  static constexpr int KPR = BW / 2;                        // K-steps per row
  static constexpr int NK = 2 * KPR;                        // ... per wave and box (two rows)
  static_assert(BH == 8 && GP % 4 == 0 && NPW <= NK, "four waves x two rows; one DMA piece per K-step");
};

template <int BH, int BW>
__global__ void __launch_bounds__(256, 2) ww_gemm_kernel(const WwArgs a) {
  using G = GeoWW<BH, BW>;
  constexpr int XW = G::XW, KPR = G::KPR, NK = G::NK;
  __shared__ __attribute__((aligned(16))) float lds[G::LDS];
  float* const st0 = lds;
  float* const st1 = lds + G::STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;

  // tiles fastest: the workgroups that read the same boxes sit side by side (and, after the remap, on one XCD's L2)
  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int ci_t = __builtin_amdgcn_readfirstlane(id % a.tiles_ci); id /= a.tiles_ci;
  const int co_t = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int split = __builtin_amdgcn_readfirstlane(id % a.splits); id /= a.splits;
  const int k = __builtin_amdgcn_readfirstlane(id % 6);
  const int g = __builtin_amdgcn_readfirstlane(id / 6);
  const int Hp = a.Hb + 2, Wp = a.Wb + 2;
  const long gplane = (long)a.Q * a.Hb * a.Wb * a.CoP, xplane = (long)a.Q * Hp * Wp * a.CiP;
  const buf_rsrc rg = make_rsrc(a.gh + ((long)g * 6 + k) * gplane, (unsigned)(gplane * 4));
  const buf_rsrc rx = make_rsrc(a.xh + ((long)g * 6 + k) * xplane, (unsigned)(xplane * 4));

  // per-lane byte offsets of this wave's pieces inside a box (piece j = wv + 4 i: positions 8 j .. 8 j + 7, lane = (position, quad))
  unsigned gv[G::GPW], xv[G::XPW];
#pragma unroll
  for (int i = 0; i < G::GPW; ++i) {
    const int p = (wv + 4 * i) * 8 + (lane >> 3);
    gv[i] = (unsigned)((((p / BW) * a.Wb + p % BW) * a.CoP + co_t * 32 + (lane & 7) * 4) * 4);
  }
#pragma unroll
  for (int i = 0; i < G::XPW; ++i) {
    int e = (wv + 4 * i) * 8 + (lane >> 3);
    if (e >= G::XPOS) e = 0;                  // (the tail of the last piece lands in the stage's padding)
    xv[i] = (unsigned)((((e / XW) * Wp + e % XW) * a.CiP + ci_t * 32 + (lane & 7) * 4) * 4);
  }

  // this workgroup's boxes: bx0 .. bx1 - 1 of the (group, k) array, box = (plane q, hb, wb), wb fastest
  const int bx0 = split * a.bps, bx1 = min(a.nboxes, bx0 + a.bps);
  int nq, nhb, nwb;                           // the box whose pieces are requested next
  {
    int t = bx0;
    nwb = t % a.nbw; t /= a.nbw;
    nhb = t % a.nbh;
    nq = t / a.nbh;
  }
  unsigned gso = 0, xso = 0;
  auto locate = [&]() __attribute__((always_inline)) {     // byte offsets of box (nq, nhb, nwb), then step to the one after
    gso = (unsigned)((((long)nq * a.Hb + nhb * BH) * a.Wb + nwb * BW) * a.CoP * 4);
    xso = (unsigned)((((long)nq * Hp + nhb * BH) * Wp + nwb * BW) * a.CiP * 4);
    if (++nwb == a.nbw) {
      nwb = 0;
      if (++nhb == a.nbh) { nhb = 0; ++nq; }
    }
  };
  auto issue_piece = [&](auto ic, float* st) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < G::GPW) {
      dma_b128(rg, gv[i], gso, st + (wv + 4 * i) * 256);
    } else if constexpr (i < G::NPW) {
      constexpr int ii = i - G::GPW;
      if (G::XP % 4 == 0 || wv + 4 * ii < G::XP) dma_b128(rx, xv[ii], xso, st + G::G_FLOATS + (wv + 4 * ii) * 256);
    }
  };

  // MFMA operands of this wave: rows 2 wv, 2 wv + 1 of the box.  A = g^[position][co = l31], B = x^[position + (dh, dw)][ci = l31],
  // K-step = two neighbouring positions (khalf).
  const int aoff = (2 * wv * BW + khalf) * 32 + l31;
  const int boff = G::G_FLOATS + (2 * wv * XW + khalf) * 32 + l31;

  f32x16 acc[9];    // [dh * 3 + dw]
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  auto mfma_box = [&](const float* st, float* st_next) __attribute__((always_inline)) {
    // step s = (row rl, pair ks): positions 2 ks, 2 ks + 1 of the row.  x^ column c serves (ks, dw) with 2 ks + dw = c: each
    // step fetches two new columns per dh (three at the start of a row); all operands one step ahead.
    float av[2], bc[3][BW + 2];
    auto fetch = [&](auto sc, auto pc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, part = decltype(pc)::value;
      if constexpr (s < NK) {
        constexpr int rl = s / KPR, ks = s % KPR;
        if constexpr (part == 0) {
          av[s & 1] = st[aoff + (rl * BW + 2 * ks) * 32];
        } else {
          constexpr int dh = part - 1;
          if constexpr (ks == 0) bc[dh][0] = st[boff + ((rl + dh) * XW) * 32];
          bc[dh][2 * ks + 1] = st[boff + ((rl + dh) * XW + 2 * ks + 1) * 32];
          bc[dh][2 * ks + 2] = st[boff + ((rl + dh) * XW + 2 * ks + 2) * 32];
        }
      }
    };
    static_for<0, 4>([&](auto pc) __attribute__((always_inline)) { fetch(std::integral_constant<int, 0>{}, pc); });
    static_for<0, NK>([&](auto sc) __attribute__((always_inline)) {
      constexpr int s = decltype(sc)::value, ks = s % KPR;
      static_for<0, 9>([&](auto mc) __attribute__((always_inline)) {
        constexpr int m = decltype(mc)::value, dh = m % 3, dw = m / 3;      // dw = 0 first: its column is the oldest
        acc[dh * 3 + dw] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1], bc[dh][2 * ks + dw], acc[dh * 3 + dw], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (m >= 3 && m < 7) fetch(std::integral_constant<int, s + 1>{}, std::integral_constant<int, m - 3>{});
        if constexpr (m == 7) issue_piece(sc, st_next);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- boxes, double-buffered: the pieces of box i + 1 are requested while the MFMAs of box i run ------------------------
  locate();
  static_for<0, G::NPW>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, st0); });
  for (int bx = bx0; bx < bx1; bx += 2) {
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's pieces of the box have landed ...
    __syncthreads();                         // ... everybody's have, and nobody reads the other stage any more
    if (bx + 1 < bx1) locate();              // (past the last box: the last one once more, into the idle stage)
    mfma_box(st0, st1);
    if (bx + 1 < bx1) {
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
      if (bx + 2 < bx1) locate();
      mfma_box(st1, st0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();                           // no DMA in flight, no operand read pending: the stages become the reduction buffer

  // ---- the four waves' partial sums, added in wave order, three taps per round -------------------------------------------
  float* wsd = a.ws + ((((long)split * a.groups + g) * 6 + k) * 9) * a.CoP * a.CiP + (long)(co_t * 32) * a.CiP + ci_t * 32;
  static_for<0, 3>([&](auto rdc) __attribute__((always_inline)) {
    constexpr int rd = decltype(rdc)::value;
#pragma unroll
    for (int tl = 0; tl < 3; ++tl) {
      float* T = lds + (tl * 4 + wv) * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * khalf) * 32 + l31] = acc[rd * 3 + tl][r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int e = tid + 256 * i, tl = e >> 10, idx = e & 1023;
      const float* T = lds + tl * 4096 + idx;
      const float sum = ((T[0] + T[1024]) + T[2048]) + T[3072];
      wsd[(long)(rd * 3 + tl) * a.CoP * a.CiP + (idx >> 5) * a.CiP + (idx & 31)] = sum;
    }
    __syncthreads();
  });
}

// dw[co][ci][dn][dh][dw] = sum_k A'^T[dn][k] sum_split ws[split][g][k][(dh, dw)][co][ci],  A'^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 1].
// One workgroup = one (output channel, (dh, dw)) x 64 input channels; its four thread groups each add every fourth split
// (six planes = six independent sums in flight), then the four group sums are added in group order: a fixed summation order.
// Workgroups past the weight tiles finish the bias gradient: dbias[c] = bias_scale * sum of the g^ pass's per-workgroup sums
// (32 channels per workgroup, eight slices of the list per channel added in slice order).
__global__ void __launch_bounds__(256) ww_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int splits, int groups,
                                                        int cout_g, int cin_g, int CoP, int CiP, int wblocks,
                                                        const float* __restrict__ bias_part, float* __restrict__ dbias, int nbias,
                                                        float bias_scale) {
  __shared__ float part[4][6][64];
  const int cchunks = (cin_g + 63) / 64;
  int id = blockIdx.x;
  if (id >= wblocks) {
    const int c = (id - wblocks) * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;       // c: index into [groups][CoP]
    float t = 0.f;
    for (int i = sl; i < nbias; i += 8) t += bias_part[(long)i * groups * CoP + c];
    float* red = &part[0][0][0];
    red[sl * 32 + (threadIdx.x & 31)] = t;
    __syncthreads();
    if (sl == 0) {
      for (int i = 1; i < 8; ++i) t += red[i * 32 + threadIdx.x];
      const int g = c / CoP, cl = c % CoP;
      if (cl < cout_g) dbias[g * cout_g + cl] = bias_scale * t;
    }
    return;
  }
  const int tap = id % 9; id /= 9;
  const int cc = id % cchunks; id /= cchunks;
  const int co = id % cout_g;
  const int g = id / cout_g;
  const int cil = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int ci = cc * 64 + cil;
  const long tile = (long)CoP * CiP;
  float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (ci < cin_g) {
    const float* src = ws + (((long)g * 6) * 9 + tap) * tile + (long)co * CiP + ci;
#pragma unroll 4      // (the loads of four splits in flight; the additions stay in split order)
    for (int s = sg; s < splits; s += 4) {
      const float* p = src + (long)s * groups * 54 * tile;
#pragma unroll
      for (int k = 0; k < 6; ++k) m[k] += p[(long)k * 9 * tile];
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) part[sg][k][cil] = m[k];
  __syncthreads();
  if (sg == 0 && ci < cin_g) {
#pragma unroll
    for (int k = 0; k < 6; ++k) m[k] = ((part[0][k][cil] + part[1][k][cil]) + part[2][k][cil]) + part[3][k][cil];
    const float p12 = m[1] + m[2], d12 = m[1] - m[2], p34 = m[3] + m[4], d34 = m[3] - m[4];
    float* dst = dw + (((long)g * cout_g + co) * cin_g + ci) * 27 + tap;
    dst[0] = (m[0] + p12) + p34;
    dst[9] = d12 + 2.f * d34;
    dst[18] = (p12 + 4.f * p34) + m[5];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct WwPlan {
  int T, bw;                 // band tiles; box width (16, or 8 for planes of at most 8 columns)
  int Hb, Wb, CiP, CoP, Q;
  int tiles_co, tiles_ci, nbh, nbw, nboxes, splits, bps;
  bool needs_xp;
  size_t xh_floats, gh_floats, part_floats, bias_floats, xp_floats;
  int nbias;                 // workgroups of the g^ pass per channel chunk
};

bool ww_shape_ok(const tmdiff_conv3d_desc* d) {
  if (!d || d->ksize != 3 || (d->groups != 1 && d->groups != 3) || d->x_bf16) return false;
  if (d->B <= 0 || d->N <= 0 || d->N % 4 || d->N > 16 || d->H <= 0 || d->W <= 0 || d->W % 4) return false;
  if (d->Cin <= 0 || d->Cout <= 0 || d->Cin % d->groups || d->Cout % d->groups || d->nseg < 1 || d->nseg > 3) return false;
  return true;
}

WwPlan ww_plan(const tmdiff_conv3d_desc* d) {
  WwPlan p;
  p.T = d->N / 4;
  p.bw = d->W <= 8 ? 8 : 16;
  p.Hb = (d->H + 7) / 8 * 8;
  p.Wb = (d->W + p.bw - 1) / p.bw * p.bw;
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  p.tiles_ci = (cin_g + 31) / 32; p.tiles_co = (cout_g + 31) / 32;
  p.CiP = p.tiles_ci * 32; p.CoP = p.tiles_co * 32;
  p.Q = d->B * p.T;
  p.nbh = p.Hb / 8; p.nbw = p.Wb / p.bw;
  p.nboxes = p.Q * p.nbh * p.nbw;
  // two workgroups per CU are resident: rounds of 512.  The split count that minimises rounds x (boxes per workgroup + one
  // box's worth of prologue and partial-sum stores); ties go to fewer splits (less to reduce).
  const long units = (long)d->groups * p.tiles_co * p.tiles_ci * 6;
  long best_s = 1;
  double best_cost = 1e30;
  const long smax = p.nboxes < 256 ? p.nboxes : 256;
  for (long s = 1; s <= smax; ++s) {
    const long bps = (p.nboxes + s - 1) / s;
    const long s_eff = (p.nboxes + bps - 1) / bps;
    const long rounds = (units * s_eff + 511) / 512;
    const double cost = (double)rounds * ((double)bps + 1.0);
    if (cost < best_cost - 1e-9) { best_cost = cost; best_s = s_eff; }
  }
  p.bps = (int)((p.nboxes + best_s - 1) / best_s);
  p.splits = (p.nboxes + p.bps - 1) / p.bps;
  p.needs_xp = d->in_shift || d->in_scale || d->in_mask || d->in_act || d->drop_p > 0.f;
  p.xh_floats = (size_t)d->groups * 6 * p.Q * (p.Hb + 2) * (p.Wb + 2) * p.CiP;
  p.gh_floats = (size_t)d->groups * 6 * p.Q * p.Hb * p.Wb * p.CoP;
  p.part_floats = (size_t)p.splits * d->groups * 54 * p.CoP * p.CiP;
  p.nbias = p.Hb * d->B * p.T * ((d->W + TR_W - 1) / TR_W);
  p.bias_floats = (size_t)p.nbias * d->groups * p.CoP;
  p.xp_floats = p.needs_xp ? ((size_t)d->B * d->Cin * d->N * d->H * d->W + 3) / 4 * 4 : 0;
  return p;
}

bool ww_fits(const tmdiff_conv3d_desc* d, const WwPlan& p) {
  // 32-bit byte offsets inside one (group, k) array; grid limits of the transform pass
  const size_t xk = (size_t)p.Q * (p.Hb + 2) * (p.Wb + 2) * p.CiP * 4, gk = (size_t)p.Q * p.Hb * p.Wb * p.CoP * 4;
  return xk < (1ull << 31) && gk < (1ull << 31) && 2L * d->B * d->groups * p.T <= 65535 && p.Hb + 2 <= 65535;
}

WwTransformArgs transform_args(const tmdiff_conv3d_desc* d, const WwPlan& p, const float* const* seg_x, const int* seg_c, int nseg, int C,
                               int CP, int halo, float* out, float* bias_part) {
  WwTransformArgs t;
  for (int i = 0; i < 3; ++i) { t.seg_x[i] = i < nseg ? seg_x[i] : nullptr; t.seg_c[i] = i < nseg ? seg_c[i] : 0; }
  t.nseg = nseg;
  t.B = d->B; t.cg = C / d->groups; t.groups = d->groups; t.N = d->N; t.H = d->H; t.W = d->W; t.T = p.T;
  t.CP = CP; t.Hb = p.Hb; t.Wb = p.Wb; t.halo = halo; t.out = out; t.bias_part = bias_part;
  t.wchunks = (d->W + TR_W - 1) / TR_W; t.cchunks = CP / 32;
  return t;
}

// both transform passes as ONE launch (a weight gradient is four launches; the finetune step has 51 of them)
int launch_transforms(const tmdiff_conv3d_desc* d, const WwPlan& p, const WwTransformArgs& tx, const WwTransformArgs& tg, hipStream_t st) {
  WwTransformPair pr;
  pr.x = tx; pr.g = tg;
  pr.zx = d->B * d->groups * p.T;
  const int gx = tx.wchunks * (tx.cchunks > tg.cchunks ? tx.cchunks : tg.cchunks);
  const dim3 grid((unsigned)gx, (unsigned)(p.Hb + 2), (unsigned)(2 * pr.zx));
  ww_transform_kernel<<<grid, 256, 0, st>>>(pr);
  return tmdiff::check_launch("conv3d_wgrad_wino(transform passes)");
}

}  // namespace

/* 1 if tmdiff_conv3d_wgrad_wino takes the convolution described by d (else: tmdiff_conv3d_wgrad) */
extern "C" int tmdiff_conv3d_wgrad_wino_supported(const tmdiff_conv3d_desc* d) {
  return ww_shape_ok(d) && ww_fits(d, ww_plan(d)) ? 1 : 0;
}

extern "C" size_t tmdiff_conv3d_wgrad_wino_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!ww_shape_ok(d)) return 0;
  const WwPlan p = ww_plan(d);
  return (p.xh_floats + p.gh_floats + p.part_floats + p.bias_floats + p.xp_floats) * sizeof(float);
}

extern "C" int tmdiff_conv3d_wgrad_wino(const tmdiff_conv3d_desc* d, const float* g, float* dw, void* workspace,
                                        tmdiff_stream_t stream) {
  return tmdiff_conv3d_wgrad_wino_bias(d, g, dw, nullptr, workspace, stream);
}

extern "C" int tmdiff_conv3d_wgrad_wino_bias(const tmdiff_conv3d_desc* d, const float* g, float* dw, float* dbias, void* workspace,
                                             tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d && g && dw && workspace, "conv3d_wgrad_wino: NULL pointer");
  if (!ww_shape_ok(d))
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wgrad_wino: fp32 3x3x3, groups 1 or 3, N %% 4 == 0 (at most 16), W %% 4 == 0");
  const WwPlan p = ww_plan(d);
  if (!ww_fits(d, p)) return fail(TMDIFF_E_UNSUPPORTED, "conv3d_wgrad_wino: tensor too large for 32-bit offsets");
  TMDIFF_REQUIRE(aligned16(g) && aligned16(workspace), "conv3d_wgrad_wino: 16-byte aligned gradient / workspace");
  TMDIFF_REQUIRE(!(d->in_mask && d->drop_p > 0.f), "conv3d_wgrad_wino: give either a mask tensor or drop_p, not both");
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    TMDIFF_REQUIRE(d->seg_x[i] && d->seg_c[i] > 0 && aligned16(d->seg_x[i]), "conv3d_wgrad_wino: segment %d is empty / unaligned", i);
    csum += d->seg_c[i];
  }
  TMDIFF_REQUIRE(csum == d->Cin, "conv3d_wgrad_wino: segments hold %d channels, Cin=%d", csum, d->Cin);
  hipStream_t st = as_stream(stream);
  float* xh = static_cast<float*>(workspace);
  float* gh = xh + p.xh_floats;
  float* part = gh + p.gh_floats;
  float* bias_part = part + p.part_floats;
  float* xp = bias_part + p.bias_floats;

  const float* segs[3] = {d->seg_x[0], d->seg_x[1], d->seg_x[2]};
  int segc[3] = {d->seg_c[0], d->seg_c[1], d->seg_c[2]};
  int nseg = d->nseg;
  if (p.needs_xp) {       // prologue / mask / dropout: x' first (the finetune path keeps x' from the forward and never gets here)
    const int rc = launch_prologue_apply(d, xp, st);
    if (rc) return rc;
    segs[0] = xp; segc[0] = d->Cin; nseg = 1;
  }
  // (timing experiments: TMDIFF_WW_PHASES = bit mask of the phases that run -- 1 transform passes, 2 accumulation, 4 reduction;
  //  anything but 7 leaves dw wrong or stale)
  static const int phases = [] {
    const char* e = getenv("TMDIFF_WW_PHASES");
    return e ? atoi(e) : 7;
  }();
  int rc = TMDIFF_OK;
  if (phases & 1) {
    const float* gseg[3] = {g, nullptr, nullptr};
    const int gc[3] = {d->Cout, 0, 0};
    rc = launch_transforms(d, p, transform_args(d, p, segs, segc, nseg, d->Cin, p.CiP, 1, xh, nullptr),
                           transform_args(d, p, gseg, gc, 1, d->Cout, p.CoP, 0, gh, dbias ? bias_part : nullptr), st);
    if (rc) return rc;
  }

  WwArgs a;
  a.xh = xh; a.gh = gh; a.ws = part;
  a.groups = d->groups; a.CiP = p.CiP; a.CoP = p.CoP; a.Q = p.Q; a.Hb = p.Hb; a.Wb = p.Wb;
  a.tiles_co = p.tiles_co; a.tiles_ci = p.tiles_ci; a.splits = p.splits; a.bps = p.bps;
  a.nbh = p.nbh; a.nbw = p.nbw; a.nboxes = p.nboxes;
  const long blocks = (long)d->groups * 6 * p.splits * p.tiles_co * p.tiles_ci;
  TMDIFF_REQUIRE(blocks > 0 && blocks < 0x7fffffffL, "conv3d_wgrad_wino: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  if (phases & 2) {
    if (p.bw == 16)
      ww_gemm_kernel<8, 16><<<(unsigned)blocks, 256, 0, st>>>(a);
    else
      ww_gemm_kernel<8, 8><<<(unsigned)blocks, 256, 0, st>>>(a);
    rc = check_launch("conv3d_wgrad_wino");
    if (rc) return rc;
  }
  if (!(phases & 4)) return TMDIFF_OK;
  const int cout_g = d->Cout / d->groups, cin_g = d->Cin / d->groups;
  const long rblocks = (long)d->groups * cout_g * ((cin_g + 63) / 64) * 9;
  const long bblocks = dbias ? (long)d->groups * p.CoP / 32 : 0;
  ww_reduce_kernel<<<(unsigned)(rblocks + bblocks), 256, 0, st>>>(part, dw, p.splits, d->groups, cout_g, cin_g, p.CoP, p.CiP, (int)rblocks,
                                                                  bias_part, dbias, p.nbias, d->bias_scale);
  return check_launch("conv3d_wgrad_wino(reduce)");
}
