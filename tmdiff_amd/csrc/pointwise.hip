// Bandwidth-bound pointwise members of the WavBEST forward on gfx950:
//   stem  (Cin = 1 -> C0 1x1x1 conv + SiLU, with the PAN/MS condition formed on the fly),
//   head  (SiLU + modulated 1x1x1 conv C0 -> 1),
//   bank-of-Linear projections (embedding MLPs and every Dense() modulation),
//   sinusoidal timestep features.
// Reference: GeneralModel/Hyper_unet_general.py:80-108, :161-173, :260-273, :529-532, :600-609.
#include "common.h"

namespace {

// ---- stem: y[b,co,p] = act(w[co] * x[b,p] + bias[co]),  x = pan - ms  or  xin -----------------
// One thread owns V consecutive positions and walks the output channels: each wave store is
// 64*V*4 contiguous bytes.  The input is read once, the output (Cout x larger) written once.
template <int V>
__global__ void __launch_bounds__(256) stem_kernel(const float* __restrict__ xin, const float* __restrict__ pan,
                                                   const float* __restrict__ ms, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y, int Cout,
                                                   long P, long HW, int act, const float* __restrict__ oscale,
                                                   int oscale_stride) {
  const long pv = (blockIdx.x * 256L + threadIdx.x) * V;
  const int b = blockIdx.y;
  if (pv >= P) return;
  float x[V];
  if (ms) {
    const float* m = ms + b * P + pv;
    const float* q = pan + b * HW + pv % HW;
    if constexpr (V == 4) {
      const float4 mv = *reinterpret_cast<const float4*>(m), qv = *reinterpret_cast<const float4*>(q);
      x[0] = qv.x - mv.x, x[1] = qv.y - mv.y, x[2] = qv.z - mv.z, x[3] = qv.w - mv.w;
    } else {
      x[0] = q[0] - m[0];
    }
  } else {
    if constexpr (V == 4) {
      const float4 v = *reinterpret_cast<const float4*>(xin + b * P + pv);
      x[0] = v.x, x[1] = v.y, x[2] = v.z, x[3] = v.w;
    } else {
      x[0] = xin[b * P + pv];
    }
  }
  for (int co = 0; co < Cout; ++co) {
    const float wc = w[co], bc = bias ? bias[co] : 0.f;
    float o[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      o[k] = __fadd_rn(__fmul_rn(wc, x[k]), bc);
      if (act) o[k] = tmdiff::silu_f(o[k]);
    }
    if (oscale) {  // the consumer's modulation (conv21 of the stem block) applied where the value is produced
      const float sc = oscale[(long)b * oscale_stride + co];
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] *= sc;
    }
    float* dst = y + ((long)b * Cout + co) * P + pv;
    if constexpr (V == 4)
      *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    else
      dst[0] = o[0];
  }
}

// bf16-mode stem: the same values as stem_kernel with out_scale, written as packed bf16 units [B][Cout/8][P] (the input form
// of tmdiff_conv3d_fwd_bf16 with x_bf16) -- the consumer's pack pass disappears.  One thread = one position.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__global__ void __launch_bounds__(256) stem_pack_bf16_kernel(const float* __restrict__ xin, const float* __restrict__ pan,
                                                             const float* __restrict__ ms, const float* __restrict__ w,
                                                             const float* __restrict__ bias, uint4* __restrict__ units,
                                                             int Cout, long P, long HW, int act,
                                                             const float* __restrict__ oscale, int oscale_stride) {
  const long p = blockIdx.x * 256L + threadIdx.x;
  const int b = blockIdx.y;
  if (p >= P) return;
  const float x = ms ? pan[b * HW + p % HW] - ms[b * P + p] : xin[b * P + p];
  for (int c8 = 0; c8 < Cout / 8; ++c8) {
    union { bf16x8 v; uint4 u; } pk;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int co = c8 * 8 + k;
      float o = __fadd_rn(__fmul_rn(w[co], x), bias ? bias[co] : 0.f);
      if (act) o = tmdiff::silu_f(o);
      if (oscale) o *= oscale[(long)b * oscale_stride + co];
      pk.v[k] = (__bf16)o;
    }
    units[((long)b * (Cout / 8) + c8) * P + p] = pk.u;
  }
}

// ---- head: y[b,p] = sum_c (w[c]*scale[b,c]) * silu(x[b,c,p]) ------------------------------------
template <int V>
__global__ void __launch_bounds__(256) head_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ scale, int scale_stride,
                                                   float* __restrict__ y, int C, long P) {
  extern __shared__ float ws[];  // modulated weights of this sample
  const int b = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += 256) ws[c] = scale ? __fmul_rn(w[c], scale[(long)b * scale_stride + c]) : w[c];
  __syncthreads();
  const long pv = (blockIdx.x * 256L + threadIdx.x) * V;
  if (pv >= P) return;
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  const float* src = x + (long)b * C * P + pv;
  for (int c = 0; c < C; ++c) {
    float v[V];
    if constexpr (V == 4) {
      const float4 t = *reinterpret_cast<const float4*>(src + c * P);
      v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    } else {
      v[0] = src[c * P];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = fmaf(ws[c], tmdiff::silu_f(v[k]), acc[k]);
  }
  float* dst = y + (long)b * P + pv;
  if constexpr (V == 4)
    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  else
    dst[0] = acc[0];
}

// ---- bank of linear layers: one wave per output feature, all batch rows --------------------------
// I <= 1024.  The weight row lives in registers (<= 16 per lane); x rows stream from L2.
__global__ void __launch_bounds__(256) linear_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ y, int B,
                                                     int I, int O, int act) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= O) return;
  float wr[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int i = lane + 64 * k;
    wr[k] = i < I ? w[(long)o * I + i] : 0.f;
  }
  const float bo = bias ? bias[o] : 0.f;
  // four batch rows per iteration: their loads and lane reductions overlap (one row at a time was one L2 round trip + six
  // dependent shuffles per row: 28 us for 32 rows); every row's sum keeps its order of additions
  for (int b0 = 0; b0 < B; b0 += 4) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k;
      if (i < I) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (b0 + r < B) s[r] = fmaf(wr[k], x[(long)(b0 + r) * I + i], s[r]);
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[r] += __shfl_xor(s[r], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (b0 + r < B) {
          float v = s[r] + bo;
          if (act) v = tmdiff::silu_f(v);
          y[(long)(b0 + r) * O + o] = v;
        }
    }
  }
}

__global__ void __launch_bounds__(256) gamma_kernel(const float* __restrict__ t, const float* __restrict__ freqs,
                                                    float* __restrict__ emb, int B, int dim) {
  const int half = dim / 2;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= B * dim) return;
  const int b = i / dim, k = i % dim;
  float v = 0.f;  // zero pad when dim is odd
  if (k < half)
    v = cosf(__fmul_rn(t[b], freqs[k]));
  else if (k < 2 * half)
    v = sinf(__fmul_rn(t[b], freqs[k - half]));
  emb[i] = v;
}

}  // namespace

extern "C" int tmdiff_stem_fwd(const float* xin, const float* pan, const float* ms, const float* w, const float* bias,
                               float* y, int32_t B, int32_t Cout, int32_t N, int32_t H, int32_t W, int32_t apply_silu,
                               tmdiff_stream_t stream) {
  return tmdiff_stem_fwd_scaled(xin, pan, ms, w, bias, nullptr, 0, y, B, Cout, N, H, W, apply_silu, stream);
}

extern "C" int tmdiff_stem_fwd_scaled(const float* xin, const float* pan, const float* ms, const float* w, const float* bias,
                                      const float* out_scale, int32_t out_scale_stride, float* y, int32_t B, int32_t Cout,
                                      int32_t N, int32_t H, int32_t W, int32_t apply_silu, tmdiff_stream_t stream) {
  using namespace tmdiff;
  const int oss = out_scale_stride > 0 ? out_scale_stride : (out_scale_stride < 0 ? 0 : Cout);
  TMDIFF_REQUIRE(w && y, "stem_fwd: NULL weights/output");
  TMDIFF_REQUIRE((ms && pan) || (!ms && xin), "stem_fwd: give either (pan, ms) or xin");
  TMDIFF_REQUIRE(B >= 0 && Cout > 0 && N > 0 && H > 0 && W > 0 && B <= 65535, "stem_fwd: bad extents");
  if (B == 0) return TMDIFF_OK;
  const long HW = (long)H * W, P = HW * N;
  const bool vec = HW % 4 == 0 && aligned16(xin) && aligned16(pan) && aligned16(ms) && aligned16(y);
  if (vec) {
    dim3 grid((unsigned)((P / 4 + 255) / 256), B);
    stem_kernel<4><<<grid, 256, 0, as_stream(stream)>>>(xin, pan, ms, w, bias, y, Cout, P, HW, apply_silu, out_scale, oss);
  } else {
    dim3 grid((unsigned)((P + 255) / 256), B);
    stem_kernel<1><<<grid, 256, 0, as_stream(stream)>>>(xin, pan, ms, w, bias, y, Cout, P, HW, apply_silu, out_scale, oss);
  }
  return check_launch("stem_fwd");
}

extern "C" int tmdiff_stem_fwd_pack_bf16(const float* xin, const float* pan, const float* ms, const float* w,
                                         const float* bias, const float* out_scale, int32_t out_scale_stride, void* units,
                                         int32_t B, int32_t Cout, int32_t N, int32_t H, int32_t W, int32_t apply_silu,
                                         tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(w && units && aligned16(units), "stem_fwd_pack_bf16: NULL / unaligned weights / output");
  TMDIFF_REQUIRE((ms && pan) || (!ms && xin), "stem_fwd_pack_bf16: give either (pan, ms) or xin");
  TMDIFF_REQUIRE(B >= 0 && Cout > 0 && Cout % 8 == 0 && N > 0 && H > 0 && W > 0 && B <= 65535, "stem_fwd_pack_bf16: bad extents");
  if (B == 0) return TMDIFF_OK;
  const int oss = out_scale_stride > 0 ? out_scale_stride : (out_scale_stride < 0 ? 0 : Cout);
  const long HW = (long)H * W, P = HW * N;
  stem_pack_bf16_kernel<<<dim3((unsigned)((P + 255) / 256), B), 256, 0, as_stream(stream)>>>(
      xin, pan, ms, w, bias, static_cast<uint4*>(units), Cout, P, HW, apply_silu, out_scale, oss);
  return check_launch("stem_fwd_pack_bf16");
}

extern "C" int tmdiff_head_fwd(const float* x, const float* w, const float* scale, int32_t scale_stride, float* y,
                               int32_t B, int32_t C, int64_t P, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && w && y, "head_fwd: NULL pointer");
  TMDIFF_REQUIRE(B >= 0 && C > 0 && C <= 4096 && P > 0 && B <= 65535, "head_fwd: bad extents B=%d C=%d", B, C);
  if (B == 0) return TMDIFF_OK;
  const int ss = scale_stride > 0 ? scale_stride : (scale_stride < 0 ? 0 : C);
  const bool vec = P % 4 == 0 && aligned16(x) && aligned16(y);
  if (vec) {
    dim3 grid((unsigned)((P / 4 + 255) / 256), B);
    head_kernel<4><<<grid, 256, C * sizeof(float), as_stream(stream)>>>(x, w, scale, ss, y, C, P);
  } else {
    dim3 grid((unsigned)((P + 255) / 256), B);
    head_kernel<1><<<grid, 256, C * sizeof(float), as_stream(stream)>>>(x, w, scale, ss, y, C, P);
  }
  return check_launch("head_fwd");
}

extern "C" int tmdiff_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t I,
                                 int32_t O, int32_t act, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && w && y, "linear_fwd: NULL pointer");
  TMDIFF_REQUIRE(B >= 0 && I > 0 && I <= 1024 && O > 0, "linear_fwd: B=%d I=%d (<=1024) O=%d", B, I, O);
  if (B == 0) return TMDIFF_OK;
  linear_kernel<<<(O + 3) / 4, 256, 0, as_stream(stream)>>>(x, w, bias, y, B, I, O, act);
  return check_launch("linear_fwd");
}

extern "C" int tmdiff_gamma_embedding(const float* t, const float* freqs, float* emb, int32_t B, int32_t dim,
                                      tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(t && freqs && emb, "gamma_embedding: NULL pointer");
  TMDIFF_REQUIRE(B >= 0 && dim >= 2, "gamma_embedding: B=%d dim=%d", B, dim);
  if (B == 0) return TMDIFF_OK;
  gamma_kernel<<<(B * dim + 255) / 256, 256, 0, as_stream(stream)>>>(t, freqs, emb, B, dim);
  return check_launch("gamma_embedding");
}
