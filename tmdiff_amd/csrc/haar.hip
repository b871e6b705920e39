// 2-D Haar DWT / IDWT butterflies for gfx950.  HBM-bound: every element is read once and
// written once, 16 bytes per lane when the row length allows it.
//
// Reference: DWT_IDWT/DWT_IDWT_layer.py:256-334, :337-430 and DWT_IDWT_Functions.py:47-69,
// :89-112, which build dense [H/2,H] / [W,W/2] filter matrices on every call and run six
// batched GEMMs.  The only non-zero taps are +-s, s = fp32(1/sqrt(2)); the two-stage
// evaluation below (rows, then columns, one multiply by s per stage) keeps the reference's
// rounding order:  L = s*top + s*bot,  LL = s*L[even] + s*L[odd], ...
#include "common.h"

namespace {

constexpr float kS = 0.70710678118654752440f;  // fp32(1/sqrt(2)) == 0x3F3504F3

template <int V>
struct Vec;
template <>
struct Vec<1> {
  using in_t = float2;
  using out_t = float;
};
template <>
struct Vec<4> {
  using in_t = float4;  // two of them per row
  using out_t = float4;
};

__device__ __forceinline__ void butterfly(float a, float b, float c, float d, float& ll, float& lh, float& hl,
                                          float& hh) {
#pragma clang fp contract(off)
  // a=x[2i,2j] b=x[2i,2j+1] c=x[2i+1,2j] d=x[2i+1,2j+1]
  // (plain operators under contract(off): every product and sum is individually rounded, never fused into an FMA, so
  //  every kernel that inlines this gives the same bits -- the __f*_rn helpers of this toolchain are plain operators
  //  defined elsewhere and would still be contracted)
  const float sa = kS * a, sb = kS * b, sc = kS * c, sd = kS * d;
  const float lo0 = sa + sc, lo1 = sb + sd;  // L  = L0 @ X   (low rows)
  const float hi0 = sa - sc, hi1 = sb - sd;  // Hh = H0 @ X   (high rows)
  const float l0 = kS * lo0, l1 = kS * lo1, h0 = kS * hi0, h1 = kS * hi1;
  ll = l0 + l1;
  lh = l0 - l1;
  hl = h0 + h1;
  hh = h0 - h1;
}

__device__ __forceinline__ void inv_butterfly(float ll, float lh, float hl, float hh, float& a, float& b, float& c,
                                              float& d) {
#pragma clang fp contract(off)
  const float s0 = kS * ll, s1 = kS * lh, s2 = kS * hl, s3 = kS * hh;
  const float lo0 = s0 + s1, lo1 = s0 - s1;  // L  = LL @ L1^T + LH @ H1^T
  const float hi0 = s2 + s3, hi1 = s2 - s3;  // Hh = HL @ L1^T + HH @ H1^T
  const float l0 = kS * lo0, l1 = kS * lo1, h0 = kS * hi0, h1 = kS * hi1;
  a = l0 + h0;
  b = l1 + h1;
  c = l0 - h0;
  d = l1 - h1;
}

__device__ __forceinline__ float scaled(float s, float v) {   // s * v as its own rounded product
#pragma clang fp contract(off)
  return s * v;
}

// One thread -> V adjacent output columns of one output row.  grid-stride over all outputs.
// Optional consumer prologue on one output (the LL band of the DWT, the first reconstruction of the IDWT):
//   out = act(out + shift[b, c]) * scale[b, c]   with plane = (b * C + c) * n_per_channel + band-plane index
// -- the convolution that consumes that tensor then reads it as a plain input (no prologue pass, no hand-off).
struct PlanePrologue {
  const float* shift;
  const float* scale;
  int shift_stride, scale_stride;  // row strides in floats (0 = one row broadcast over the batch)
  int C, n_per_channel, act, on;
};

__device__ __forceinline__ void prologue_coefs(const PlanePrologue& p, long plane, float& sh, float& sc) {
  const long bc = plane / p.n_per_channel;
  const int b = (int)(bc / p.C), c = (int)(bc % p.C);
  sh = p.shift ? p.shift[(long)b * p.shift_stride + c] : 0.f;
  sc = p.scale ? p.scale[(long)b * p.scale_stride + c] : 1.f;
}
__device__ __forceinline__ float apply_prologue(int act, float sh, float sc, float v) {   // as the conv kernels' prologue
  float t = v + sh;
  const float ta = tmdiff::silu_f(t);
  return (act ? ta : t) * sc;
}

template <int V>
__global__ void __launch_bounds__(256) dwt2d_kernel(const float* __restrict__ x, float* __restrict__ ll,
                                                    float* __restrict__ lh, float* __restrict__ hl,
                                                    float* __restrict__ hh, long total, int h, int w, float ll_scale,
                                                    float hi_scale, const PlanePrologue pro) {
  const int wv = w / V;  // vector columns per output row
  const int W = 2 * w;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += 256L * gridDim.x) {
    const int j = (int)(i % wv);
    const long rp = i / wv;  // plane * h + row
    const long plane = rp / h;
    const int r = (int)(rp - plane * h);
    const float* top = x + (plane * 2 * h + 2 * r) * W + 2 * V * j;
    const float* bot = top + W;
    float t[2 * V], b[2 * V], o[4][V];
    if constexpr (V == 4) {
      *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(top);
      *reinterpret_cast<float4*>(t + 4) = *reinterpret_cast<const float4*>(top + 4);
      *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(bot);
      *reinterpret_cast<float4*>(b + 4) = *reinterpret_cast<const float4*>(bot + 4);
    } else {
      t[0] = top[0], t[1] = top[1], b[0] = bot[0], b[1] = bot[1];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) butterfly(t[2 * k], t[2 * k + 1], b[2 * k], b[2 * k + 1], o[0][k], o[1][k], o[2][k], o[3][k]);
    const long off = rp * w + V * j;
    float* outs[4] = {ll, lh, hl, hh};
#pragma unroll
    for (int band = 0; band < 4; ++band) {
      if (outs[band] == nullptr) continue;
      const float sc = band == 0 ? ll_scale : hi_scale;
      float v[V];
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = scaled(sc, o[band][k]);
      if (band == 0 && pro.on) {
        float sh, scl;
        prologue_coefs(pro, plane, sh, scl);
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = apply_prologue(pro.act, sh, scl, v[k]);
      }
      if constexpr (V == 4) {
        *reinterpret_cast<float4*>(outs[band] + off) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
        outs[band][off] = v[0];
      }
    }
  }
}

template <int V>
__global__ void __launch_bounds__(256) idwt2d_kernel(const float* __restrict__ ll0, const float* __restrict__ ll1,
                                                     const float* __restrict__ lh, const float* __restrict__ hl,
                                                     const float* __restrict__ hh, float* __restrict__ out0,
                                                     float* __restrict__ out1, long total, int h, int w,
                                                     float in_scale, long hi_ppb, long hi_bstride,
                                                     const PlanePrologue pro) {
  const int wv = w / V;
  const int W = 2 * w;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += 256L * gridDim.x) {
    const int j = (int)(i % wv);
    const long rp = i / wv;
    const long plane = rp / h;
    const int r = (int)(rp - plane * h);
    const long off = rp * w + V * j;
    // high bands may be channel slices of a [B, 3C, N, h, w] tensor: sample stride != planes*h*w
    const long hoff = hi_bstride ? (plane / hi_ppb) * hi_bstride + ((plane % hi_ppb) * h + r) * (long)w + V * j : off;
    float b1[V], b2[V], b3[V];
    if (lh == nullptr) {  // zero high bands (adjoint of an LL-only dwt)
#pragma unroll
      for (int c = 0; c < V; ++c) b1[c] = b2[c] = b3[c] = 0.f;
    } else if constexpr (V == 4) {
      *reinterpret_cast<float4*>(b1) = *reinterpret_cast<const float4*>(lh + hoff);
      *reinterpret_cast<float4*>(b2) = *reinterpret_cast<const float4*>(hl + hoff);
      *reinterpret_cast<float4*>(b3) = *reinterpret_cast<const float4*>(hh + hoff);
    } else {
      b1[0] = lh[hoff], b2[0] = hl[hoff], b3[0] = hh[hoff];
    }
    const float* lls[2] = {ll0, ll1};
    float* outs[2] = {out0, out1};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (lls[k] == nullptr) continue;
      float b0[V], t[2 * V], bt[2 * V];
      if constexpr (V == 4) {
        *reinterpret_cast<float4*>(b0) = *reinterpret_cast<const float4*>(lls[k] + off);
      } else {
        b0[0] = lls[k][off];
      }
#pragma unroll
      for (int c = 0; c < V; ++c)
        inv_butterfly(scaled(in_scale, b0[c]), b1[c], b2[c], b3[c], t[2 * c], t[2 * c + 1], bt[2 * c], bt[2 * c + 1]);
      if (k == 0 && pro.on) {
        float sh, scl;
        prologue_coefs(pro, plane, sh, scl);
#pragma unroll
        for (int c = 0; c < 2 * V; ++c) t[c] = apply_prologue(pro.act, sh, scl, t[c]), bt[c] = apply_prologue(pro.act, sh, scl, bt[c]);
      }
      float* top = outs[k] + (plane * 2 * h + 2 * r) * W + 2 * V * j;
      float* bot = top + W;
      if constexpr (V == 4) {
        *reinterpret_cast<float4*>(top) = *reinterpret_cast<float4*>(t);
        *reinterpret_cast<float4*>(top + 4) = *reinterpret_cast<float4*>(t + 4);
        *reinterpret_cast<float4*>(bot) = *reinterpret_cast<float4*>(bt);
        *reinterpret_cast<float4*>(bot + 4) = *reinterpret_cast<float4*>(bt + 4);
      } else {
        top[0] = t[0], top[1] = t[1], bot[0] = bt[0], bot[1] = bt[1];
      }
    }
  }
}

// ---- bf16-mode producers: the LL band / the first reconstruction written as the packed bf16 units [B][C/8][positions]
// (8 channels = 16 bytes per position) that tmdiff_conv3d_fwd_bf16 takes with x_bf16, consumer prologue applied in fp32
// and rounded (RNE) exactly as its pack pass would -- so that pass disappears.  One thread = one position, 8 channels.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__global__ void __launch_bounds__(256) dwt2d_pack_bf16_kernel(const float* __restrict__ x, uint4* __restrict__ units,
                                                              float* __restrict__ lh, float* __restrict__ hl,
                                                              float* __restrict__ hh, long total, int C, int N, int h,
                                                              int w, float ll_scale, float hi_scale,
                                                              const PlanePrologue pro) {
  const int W2 = 2 * w;
  const long hw = (long)h * w, plane = (long)N * hw;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += 256L * gridDim.x) {
    const int j = (int)(i % w);
    long t = i / w;
    const int r = (int)(t % h); t /= h;
    const int n = (int)(t % N); t /= N;
    const int c8 = (int)(t % (C / 8));
    const int b = (int)(t / (C / 8));
    union { bf16x8 v; uint4 u; } pk;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int c = c8 * 8 + ch;
      const long pl = ((long)b * C + c) * N + n;
      const float* top = x + (pl * 2 * h + 2 * r) * W2 + 2 * j;
      const float2 tt = *reinterpret_cast<const float2*>(top), bb = *reinterpret_cast<const float2*>(top + W2);
      float ll, b1, b2, b3;
      butterfly(tt.x, tt.y, bb.x, bb.y, ll, b1, b2, b3);
      const float sh = pro.shift ? pro.shift[(long)b * pro.shift_stride + c] : 0.f;
      const float sc = pro.scale ? pro.scale[(long)b * pro.scale_stride + c] : 1.f;
      pk.v[ch] = (__bf16)apply_prologue(pro.act, sh, sc, scaled(ll_scale, ll));
      if (lh) {
        const long off = pl * hw + (long)r * w + j;
        lh[off] = scaled(hi_scale, b1), hl[off] = scaled(hi_scale, b2), hh[off] = scaled(hi_scale, b3);
      }
    }
    units[((long)b * (C / 8) + c8) * plane + ((long)n * h + r) * w + j] = pk.u;
  }
}

// (h_up, x_up) = (IDWT(s * hh, bands), IDWT(s * xx, bands)); bands = stacked [B, 3C, N, h, w]; h_up packed with prologue
__global__ void __launch_bounds__(256) idwt2d_pack_bf16_kernel(const float* __restrict__ ll0, const float* __restrict__ ll1,
                                                               const float* __restrict__ bands, uint4* __restrict__ units,
                                                               float* __restrict__ out1, long total, int C, int N, int h,
                                                               int w, float in_scale, const PlanePrologue pro) {
  const int W2 = 2 * w, H2 = 2 * h;
  const long hw = (long)h * w, oplane = (long)N * H2 * W2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += 256L * gridDim.x) {
    const int j = (int)(i % w);
    long t = i / w;
    const int r = (int)(t % h); t /= h;
    const int n = (int)(t % N); t /= N;
    const int c8 = (int)(t % (C / 8));
    const int b = (int)(t / (C / 8));
    union { bf16x8 v; uint4 u; } pk[4];
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const int c = c8 * 8 + ch;
      const long pl = ((long)b * C + c) * N + n;
      const long off = pl * hw + (long)r * w + j;
      const long boff = (((long)b * 3 * C + c) * N + n) * hw + (long)r * w + j;   // band k: + k * C * N * hw
      const float b1 = bands[boff], b2 = bands[boff + (long)C * N * hw], b3 = bands[boff + 2L * C * N * hw];
      float a0, a1, a2, a3, x0, x1, x2, x3;
      inv_butterfly(scaled(in_scale, ll0[off]), b1, b2, b3, a0, a1, a2, a3);
      inv_butterfly(scaled(in_scale, ll1[off]), b1, b2, b3, x0, x1, x2, x3);
      float* top = out1 + (pl * H2 + 2 * r) * W2 + 2 * j;
      *reinterpret_cast<float2*>(top) = make_float2(x0, x1);
      *reinterpret_cast<float2*>(top + W2) = make_float2(x2, x3);
      const float sh = pro.shift ? pro.shift[(long)b * pro.shift_stride + c] : 0.f;
      const float sc = pro.scale ? pro.scale[(long)b * pro.scale_stride + c] : 1.f;
      pk[0].v[ch] = (__bf16)apply_prologue(pro.act, sh, sc, a0);
      pk[1].v[ch] = (__bf16)apply_prologue(pro.act, sh, sc, a1);
      pk[2].v[ch] = (__bf16)apply_prologue(pro.act, sh, sc, a2);
      pk[3].v[ch] = (__bf16)apply_prologue(pro.act, sh, sc, a3);
    }
    uint4* u = units + ((long)b * (C / 8) + c8) * oplane + ((long)n * H2 + 2 * r) * W2 + 2 * j;
    u[0] = pk[0].u, u[1] = pk[1].u, u[W2] = pk[2].u, u[W2 + 1] = pk[3].u;
  }
}

inline int grid_for(long total) {
  long blocks = (total + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 256 * 8 ? 256 * 8 : blocks));  // <= 8 blocks per CU, grid-stride beyond
}

}  // namespace

namespace {
int make_prologue(const tmdiff_plane_prologue* p, int64_t planes, PlanePrologue& q) {
  q = PlanePrologue{nullptr, nullptr, 0, 0, 1, 1, 0, 0};
  if (!p) return TMDIFF_OK;
  TMDIFF_REQUIRE(p->C > 0 && p->n_per_channel > 0 && planes % ((int64_t)p->C * p->n_per_channel) == 0,
                 "haar prologue: planes=%ld is not B * C=%d * n=%d", (long)planes, p->C, p->n_per_channel);
  q.shift = p->shift; q.scale = p->scale; q.C = p->C; q.n_per_channel = p->n_per_channel; q.act = p->act; q.on = 1;
  q.shift_stride = p->shift_stride > 0 ? p->shift_stride : (p->shift_stride < 0 ? 0 : p->C);
  q.scale_stride = p->scale_stride > 0 ? p->scale_stride : (p->scale_stride < 0 ? 0 : p->C);
  return TMDIFF_OK;
}
}  // namespace

extern "C" int tmdiff_haar_dwt2d(const float* x, float* ll, float* lh, float* hl, float* hh, int64_t planes, int32_t H,
                                 int32_t W, float ll_scale, float hi_scale, tmdiff_stream_t stream) {
  return tmdiff_haar_dwt2d_pro(x, ll, lh, hl, hh, planes, H, W, ll_scale, hi_scale, nullptr, stream);
}

extern "C" int tmdiff_haar_dwt2d_pro(const float* x, float* ll, float* lh, float* hl, float* hh, int64_t planes, int32_t H,
                                     int32_t W, float ll_scale, float hi_scale, const tmdiff_plane_prologue* ll_prologue,
                                     tmdiff_stream_t stream) {
  using namespace tmdiff;
  PlanePrologue pro;
  if (int rc = make_prologue(ll_prologue, planes, pro)) return rc;
  TMDIFF_REQUIRE(x && ll, "haar_dwt2d: x and ll must not be NULL");
  TMDIFF_REQUIRE(planes >= 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "haar_dwt2d: H=%d W=%d must be positive and even",
                 H, W);
  if (planes == 0) return TMDIFF_OK;
  const int h = H / 2, w = W / 2;
  const bool vec = (w % 4 == 0) && aligned16(x) && aligned16(ll) && aligned16(lh) && aligned16(hl) && aligned16(hh);
  if (vec) {
    const long total = planes * h * (w / 4);
    dwt2d_kernel<4><<<grid_for(total), 256, 0, as_stream(stream)>>>(x, ll, lh, hl, hh, total, h, w, ll_scale, hi_scale, pro);
  } else {
    const long total = planes * h * w;
    dwt2d_kernel<1><<<grid_for(total), 256, 0, as_stream(stream)>>>(x, ll, lh, hl, hh, total, h, w, ll_scale, hi_scale, pro);
  }
  return check_launch("haar_dwt2d");
}

extern "C" int tmdiff_haar_dwt2d_pack_bf16(const float* x, void* ll_units, float* lh, float* hl, float* hh, int32_t B,
                                           int32_t C, int32_t N, int32_t H, int32_t W, float ll_scale, float hi_scale,
                                           const tmdiff_plane_prologue* ll_prologue, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && ll_units && aligned16(ll_units), "haar_dwt2d_pack_bf16: NULL / unaligned pointer");
  TMDIFF_REQUIRE((lh && hl && hh) || (!lh && !hl && !hh), "haar_dwt2d_pack_bf16: give all three high bands or none");
  TMDIFF_REQUIRE(B >= 0 && C > 0 && C % 8 == 0 && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0,
                 "haar_dwt2d_pack_bf16: C=%d (multiple of 8) H=%d W=%d (even)", C, H, W);
  if (B == 0) return TMDIFF_OK;
  PlanePrologue pro = PlanePrologue{nullptr, nullptr, 0, 0, C, N, 0, 1};
  if (ll_prologue) {
    TMDIFF_REQUIRE(ll_prologue->C == C && ll_prologue->n_per_channel == N, "haar_dwt2d_pack_bf16: prologue C / n mismatch");
    if (int rc = make_prologue(ll_prologue, (int64_t)B * C * N, pro)) return rc;
  }
  const long total = (long)B * (C / 8) * N * (H / 2) * (W / 2);
  dwt2d_pack_bf16_kernel<<<grid_for(total), 256, 0, as_stream(stream)>>>(x, static_cast<uint4*>(ll_units), lh, hl, hh, total, C,
                                                                       N, H / 2, W / 2, ll_scale, hi_scale, pro);
  return check_launch("haar_dwt2d_pack_bf16");
}

extern "C" int tmdiff_haar_idwt2d_pack_bf16(const float* ll0, const float* ll1, const float* stacked_bands, void* out0_units,
                                            float* out1, int32_t B, int32_t C, int32_t N, int32_t h, int32_t w,
                                            float in_scale, const tmdiff_plane_prologue* out0_prologue,
                                            tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(ll0 && ll1 && stacked_bands && out0_units && out1 && aligned16(out0_units),
                 "haar_idwt2d_pack_bf16: NULL / unaligned pointer");
  TMDIFF_REQUIRE(B >= 0 && C > 0 && C % 8 == 0 && N > 0 && h > 0 && w > 0, "haar_idwt2d_pack_bf16: C=%d (multiple of 8)", C);
  if (B == 0) return TMDIFF_OK;
  PlanePrologue pro = PlanePrologue{nullptr, nullptr, 0, 0, C, N, 0, 1};
  if (out0_prologue) {
    TMDIFF_REQUIRE(out0_prologue->C == C && out0_prologue->n_per_channel == N, "haar_idwt2d_pack_bf16: prologue C / n mismatch");
    if (int rc = make_prologue(out0_prologue, (int64_t)B * C * N, pro)) return rc;
  }
  const long total = (long)B * (C / 8) * N * h * w;
  idwt2d_pack_bf16_kernel<<<grid_for(total), 256, 0, as_stream(stream)>>>(ll0, ll1, stacked_bands, static_cast<uint4*>(out0_units),
                                                                        out1, total, C, N, h, w, in_scale, pro);
  return check_launch("haar_idwt2d_pack_bf16");
}

extern "C" int tmdiff_haar_idwt2d(const float* const ll[2], int32_t n_ll, const float* lh, const float* hl,
                                  const float* hh, int64_t hi_planes_per_batch, int64_t hi_batch_stride,
                                  float* const out[2], int64_t planes, int32_t h, int32_t w, float in_scale,
                                  tmdiff_stream_t stream) {
  return tmdiff_haar_idwt2d_pro(ll, n_ll, lh, hl, hh, hi_planes_per_batch, hi_batch_stride, out, planes, h, w, in_scale,
                                nullptr, stream);
}

extern "C" int tmdiff_haar_idwt2d_pro(const float* const ll[2], int32_t n_ll, const float* lh, const float* hl,
                                      const float* hh, int64_t hi_planes_per_batch, int64_t hi_batch_stride,
                                      float* const out[2], int64_t planes, int32_t h, int32_t w, float in_scale,
                                      const tmdiff_plane_prologue* out0_prologue, tmdiff_stream_t stream) {
  using namespace tmdiff;
  PlanePrologue pro;
  if (int rc = make_prologue(out0_prologue, planes, pro)) return rc;
  TMDIFF_REQUIRE(n_ll == 1 || n_ll == 2, "haar_idwt2d: n_ll=%d must be 1 or 2", n_ll);
  TMDIFF_REQUIRE(ll && out && ll[0] && out[0], "haar_idwt2d: NULL band/output");
  TMDIFF_REQUIRE((lh && hl && hh) || (!lh && !hl && !hh), "haar_idwt2d: give all three high bands or none");
  TMDIFF_REQUIRE(n_ll == 1 || (ll[1] && out[1]), "haar_idwt2d: second low band / output is NULL");
  TMDIFF_REQUIRE(planes >= 0 && h > 0 && w > 0, "haar_idwt2d: bad sizes");
  TMDIFF_REQUIRE(hi_batch_stride == 0 || (hi_planes_per_batch > 0 && planes % hi_planes_per_batch == 0 &&
                                          hi_batch_stride >= hi_planes_per_batch * h * w && hi_batch_stride % 4 == 0),
                 "haar_idwt2d: bad high-band slicing (planes_per_batch=%ld stride=%ld)", (long)hi_planes_per_batch,
                 (long)hi_batch_stride);
  if (planes == 0) return TMDIFF_OK;
  const float* ll1 = n_ll == 2 ? ll[1] : nullptr;
  float* out1 = n_ll == 2 ? out[1] : nullptr;
  const bool vec = (w % 4 == 0) && aligned16(ll[0]) && aligned16(ll1) && aligned16(lh) && aligned16(hl) &&
                   aligned16(hh) && aligned16(out[0]) && aligned16(out1);
  if (vec) {
    const long total = planes * h * (w / 4);
    idwt2d_kernel<4><<<grid_for(total), 256, 0, as_stream(stream)>>>(ll[0], ll1, lh, hl, hh, out[0], out1, total, h, w, in_scale,
                                                                     hi_planes_per_batch, hi_batch_stride, pro);
  } else {
    const long total = planes * h * w;
    idwt2d_kernel<1><<<grid_for(total), 256, 0, as_stream(stream)>>>(ll[0], ll1, lh, hl, hh, out[0], out1, total, h, w, in_scale,
                                                                     hi_planes_per_batch, hi_batch_stride, pro);
  }
  return check_launch("haar_idwt2d");
}
