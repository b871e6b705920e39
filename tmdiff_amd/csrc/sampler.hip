// Sampler-side elementwise kernels for gfx950 (all HBM-bound, 16 B per lane):
//   DDPM reverse step, DPM-Solver linear combinations, x0 prediction, dynamic thresholding
//   (per-sample |x| quantile by radix select), q_sample, add.
// Reference: GeneralModel/diffusion_general.py:134-138, :192-208, :341-347, :376-378;
//            core/dpm_solver_pytorch.py:302-306, :430-456, :563-927; utils/util.py:135-142.
// Arithmetic uses explicit round-to-nearest mul/add (no FMA contraction) in the reference's
// evaluation order, so a step differs from the PyTorch CPU path only through eps itself.
#include <initializer_list>

#include "common.h"

namespace {

inline int grid_for(long nvec) {
  long blocks = (nvec + 255) / 256;
  return (int)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
}

struct DdpmCoef {
  float c_recip, c_recipm1, coef1, coef2, sigma;
  int clip;
};

__device__ __forceinline__ float ddpm_one(float x, float e, float nz, const DdpmCoef& k) {
  float x0 = __fsub_rn(__fmul_rn(k.c_recip, x), __fmul_rn(k.c_recipm1, e));  // predict_start_from_noise
  if (k.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);                              // clamp_(-1, 1)
  const float mean = __fadd_rn(__fmul_rn(k.coef1, x0), __fmul_rn(k.coef2, x));  // q_posterior
  return __fadd_rn(mean, __fmul_rn(nz, k.sigma));                            // + noise * exp(0.5 logvar)
}

template <int V>
__global__ void __launch_bounds__(256) ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                        const float* __restrict__ noise, const float* __restrict__ ms,
                                                        float* __restrict__ out, float* __restrict__ img, long nvec,
                                                        DdpmCoef k) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += 256L * gridDim.x) {
    float xv[V], ev[V], nv[V], mv[V], ov[V];
    if constexpr (V == 4) {
      *reinterpret_cast<float4*>(xv) = reinterpret_cast<const float4*>(x)[i];
      *reinterpret_cast<float4*>(ev) = reinterpret_cast<const float4*>(eps)[i];
      if (noise) *reinterpret_cast<float4*>(nv) = reinterpret_cast<const float4*>(noise)[i];
      if (img) *reinterpret_cast<float4*>(mv) = reinterpret_cast<const float4*>(ms)[i];
    } else {
      xv[0] = x[i], ev[0] = eps[i];
      if (noise) nv[0] = noise[i];
      if (img) mv[0] = ms[i];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) ov[j] = ddpm_one(xv[j], ev[j], noise ? nv[j] : 0.f, k);
    if constexpr (V == 4) {
      reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<float4*>(ov);
      if (img) reinterpret_cast<float4*>(img)[i] = make_float4(ov[0] + mv[0], ov[1] + mv[1], ov[2] + mv[2], ov[3] + mv[3]);
    } else {
      out[i] = ov[0];
      if (img) img[i] = ov[0] + mv[0];
    }
  }
}

struct AxpbyArgs {
  const float* in[4];
  float coef[4];
  int n_in;
};

template <int V>
__global__ void __launch_bounds__(256) axpby_kernel(AxpbyArgs a, float* __restrict__ out, long nvec) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += 256L * gridDim.x) {
    float acc[V];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k >= a.n_in) break;
      float v[V];
      if constexpr (V == 4)
        *reinterpret_cast<float4*>(v) = reinterpret_cast<const float4*>(a.in[k])[i];
      else
        v[0] = a.in[k][i];
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = k == 0 ? __fmul_rn(a.coef[0], v[j]) : __fadd_rn(acc[j], __fmul_rn(a.coef[k], v[j]));
    }
    if constexpr (V == 4)
      reinterpret_cast<float4*>(out)[i] = *reinterpret_cast<float4*>(acc);
    else
      out[i] = acc[0];
  }
}

// Multi-tensor out_t = ca * a_t + cb * b_t over a list of tensors in ONE launch (the EMA update of 272 parameter
// tensors, utils/EmaUpdater.py:23-38; one launch per tensor cost the host 272 calls per step).  Block = one chunk of at
// most MT_CHUNK elements of one tensor; chunk -> (tensor, offset) through two small device tables.
constexpr int MT_CHUNK = 16384;
__global__ void __launch_bounds__(256) multi_axpby_kernel(const tmdiff_mt_entry* __restrict__ tensors,
                                                          const int32_t* __restrict__ chunk_tensor,
                                                          const int32_t* __restrict__ chunk_index, float ca, float cb) {
  const tmdiff_mt_entry e = tensors[chunk_tensor[blockIdx.x]];
  const long lo = (long)chunk_index[blockIdx.x] * MT_CHUNK;
  const long hi = min(lo + MT_CHUNK, (long)e.n);
  const bool vec = ((reinterpret_cast<uintptr_t>(e.out) | reinterpret_cast<uintptr_t>(e.a) | reinterpret_cast<uintptr_t>(e.b)) & 15u) == 0;
  if (vec) {
    const long hi4 = lo + ((hi - lo) & ~3L);
    for (long i = lo + threadIdx.x * 4L; i < hi4; i += 1024) {
      const float4 x = *reinterpret_cast<const float4*>(e.a + i), y = *reinterpret_cast<const float4*>(e.b + i);
      *reinterpret_cast<float4*>(e.out + i) =
          make_float4(__fadd_rn(__fmul_rn(ca, x.x), __fmul_rn(cb, y.x)), __fadd_rn(__fmul_rn(ca, x.y), __fmul_rn(cb, y.y)),
                      __fadd_rn(__fmul_rn(ca, x.z), __fmul_rn(cb, y.z)), __fadd_rn(__fmul_rn(ca, x.w), __fmul_rn(cb, y.w)));
    }
    for (long i = hi4 + threadIdx.x; i < hi; i += 256) e.out[i] = __fadd_rn(__fmul_rn(ca, e.a[i]), __fmul_rn(cb, e.b[i]));
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) e.out[i] = __fadd_rn(__fmul_rn(ca, e.a[i]), __fmul_rn(cb, e.b[i]));
  }
}

// Multi-tensor AdamW (torch.optim.AdamW semantics: decoupled weight decay, bias-corrected moments, amsgrad off) over a list of
// (parameter, gradient, exp_avg, exp_avg_sq) tensors in ONE launch -- the optimizer step of the finetune loop (reference
// GeneralModel/model.py:30-31, :43: torch.optim.AdamW(lr, weight_decay=1e-4)).  The learning rate and the step count are read
// from DEVICE scalars, so the launch can be recorded into a HIP graph and replayed (torch's own capturable path runs ~25
// multi-tensor / elementwise launches for the same update: 3 ms per step of this network against 0.15 ms here).  Operation order
// as torch's _multi_tensor_adamw: p *= 1 - lr * wd; m = lerp(m, g, 1 - b1); v = b2 v + (1 - b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps), with bc1 = 1 - b1^t, bc2 = 1 - b2^t evaluated in double as the eager optimizer does.
__global__ void __launch_bounds__(256) multi_adamw_kernel(const tmdiff_adamw_entry* __restrict__ tensors,
                                                          const int32_t* __restrict__ chunk_tensor,
                                                          const int32_t* __restrict__ chunk_index, const float* __restrict__ lr_dev,
                                                          const float* __restrict__ step_dev, float beta1, float beta2, float eps,
                                                          float weight_decay) {
  const tmdiff_adamw_entry e = tensors[chunk_tensor[blockIdx.x]];
  const long lo = (long)chunk_index[blockIdx.x] * MT_CHUNK;
  const long hi = min(lo + MT_CHUNK, (long)e.n);
  const float lr = *lr_dev;
  const double t = (double)*step_dev;
  const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
  const float decay = (float)(1.0 - (double)lr * (double)weight_decay);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const float w1 = 1.f - beta1, w2 = 1.f - beta2;
  auto one = [&](float& p, float g, float& m, float& v) __attribute__((always_inline)) {
    p = __fmul_rn(p, decay);
    m = __fadd_rn(m, __fmul_rn(w1, __fsub_rn(g, m)));
    v = __fadd_rn(__fmul_rn(v, beta2), __fmul_rn(__fmul_rn(w2, g), g));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), bc2_sqrt), eps);
    p = __fsub_rn(p, __fmul_rn(step_size, __fdiv_rn(m, denom)));
  };
  const bool vec = ((reinterpret_cast<uintptr_t>(e.p) | reinterpret_cast<uintptr_t>(e.g) | reinterpret_cast<uintptr_t>(e.m) |
                     reinterpret_cast<uintptr_t>(e.v)) & 15u) == 0;
  long i0 = lo;
  if (vec) {
    const long hi4 = lo + ((hi - lo) & ~3L);
    for (long i = lo + threadIdx.x * 4L; i < hi4; i += 1024) {
      float4 p = *reinterpret_cast<float4*>(e.p + i), m = *reinterpret_cast<float4*>(e.m + i), v = *reinterpret_cast<float4*>(e.v + i);
      const float4 g = *reinterpret_cast<const float4*>(e.g + i);
      one(p.x, g.x, m.x, v.x); one(p.y, g.y, m.y, v.y); one(p.z, g.z, m.z, v.z); one(p.w, g.w, m.w, v.w);
      *reinterpret_cast<float4*>(e.p + i) = p;
      *reinterpret_cast<float4*>(e.m + i) = m;
      *reinterpret_cast<float4*>(e.v + i) = v;
    }
    i0 = hi4;
  }
  for (long i = i0 + threadIdx.x; i < hi; i += 256) one(e.p[i], e.g[i], e.m[i], e.v[i]);
}

template <int V>
__global__ void __launch_bounds__(256) x0_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                                 float* __restrict__ x0, long nvec, float alpha, float sigma,
                                                 int is_x_start) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += 256L * gridDim.x) {
    float xv[V], mv[V], ov[V];
    if constexpr (V == 4) {
      *reinterpret_cast<float4*>(xv) = reinterpret_cast<const float4*>(x)[i];
      *reinterpret_cast<float4*>(mv) = reinterpret_cast<const float4*>(m)[i];
    } else {
      xv[0] = x[i], mv[0] = m[i];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      // model_wrapper: x_start output -> noise  (dpm_solver_pytorch.py:304-306)
      const float eps = is_x_start ? __fdiv_rn(__fsub_rn(xv[j], __fmul_rn(alpha, mv[j])), sigma) : mv[j];
      // data_prediction_fn: noise -> x0          (:451-453)
      ov[j] = __fdiv_rn(__fsub_rn(xv[j], __fmul_rn(sigma, eps)), alpha);
    }
    if constexpr (V == 4)
      reinterpret_cast<float4*>(x0)[i] = *reinterpret_cast<float4*>(ov);
    else
      x0[i] = ov[0];
  }
}

template <int V>
__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long nvec, float sign_b) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < nvec; i += 256L * gridDim.x) {
    if constexpr (V == 4) {
      const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i];
      reinterpret_cast<float4*>(out)[i] = make_float4(u.x + sign_b * v.x, u.y + sign_b * v.y, u.z + sign_b * v.z, u.w + sign_b * v.w);
    } else {
      out[i] = a[i] + sign_b * b[i];
    }
  }
}

__global__ void __launch_bounds__(256) q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                       const float* __restrict__ a, float* __restrict__ out,
                                                       long n_per) {
  const int b = blockIdx.y;
  const float ab = a[b];
  const float sb = sqrtf(__fsub_rn(1.f, __fmul_rn(ab, ab)));  // (1 - a**2).sqrt()
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n_per; i += 256L * gridDim.x) {
    const long o = (long)b * n_per + i;
    out[o] = __fadd_rn(__fmul_rn(ab, x0[o]), __fmul_rn(sb, noise[o]));
  }
}

// ---- dynamic thresholding: per-sample k-th order statistic of |x| by 4 x 8-bit radix select ------
// |x| >= 0, so the fp32 bit pattern orders like the value.  Result: s = max(lerp(v[k], v[k+1], frac), max_val)
// as torch.quantile(..., interpolation='linear'), then x = clamp(x, -s, s) / s in place.
// A sample is spread over many workgroups (one workgroup per sample was latency-bound: 0.76 ms for 2 MB):
//   4 histogram launches (8 bits each; a launch first re-derives the bucket chosen so far from the earlier
//   histograms), one launch that counts the elements <= v[k] and finds the next larger value, one that clamps.
// Workspace per sample (QW_STRIDE words, zeroed by the entry point): [0] s, [1] min above (bits), [2..3] count <= v[k]
// (64 bit), [4 + 256*p ...] histogram of pass p.
constexpr int QU = 8;               // loads in flight per thread
constexpr int QW_STRIDE = 4 + 4 * 256;

struct QArgs {
  float* x;
  unsigned* ws;
  long n, k;
  float frac, max_val;
};

// Bucket prefix / rank-in-bucket after `passes` histogram passes.  Called by all 256 threads of a workgroup (a
// one-thread scan of up to 4 x 256 words cost more than the pass itself); per pass a block-wide inclusive scan
// finds the one bin d with excl[d] <= rank < incl[d].
__device__ __forceinline__ void q_prefix(const unsigned* __restrict__ w, int passes, long k, unsigned& prefix,
                                         unsigned long long& rank) {
  __shared__ unsigned wave_tot[4];
  __shared__ unsigned sel_bin, sel_excl;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  prefix = 0;
  rank = (unsigned long long)k;
  for (int p = 0; p < passes; ++p) {
    const unsigned c = w[4 + 256 * p + tid];
    unsigned incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int i = 0; i < wv; ++i) base += wave_tot[i];
    incl += base;
    const unsigned excl = incl - c;
    if ((unsigned long long)excl <= rank && rank < (unsigned long long)incl) { sel_bin = tid; sel_excl = excl; }
    __syncthreads();
    prefix |= sel_bin << (24 - 8 * p);
    rank -= sel_excl;
    __syncthreads();
  }
}

template <class F>
__device__ __forceinline__ void q_for_slice(const float* __restrict__ xs, long n, F&& f) {
  const long stride = (long)gridDim.x * 256 * QU;
  for (long i0 = (long)blockIdx.x * 256 * QU; i0 < n; i0 += stride) {
    unsigned u[QU];
#pragma unroll
    for (int j = 0; j < QU; ++j) {
      const long i = i0 + j * 256L + threadIdx.x;
      u[j] = __float_as_uint(fabsf(xs[i < n ? i : n - 1]));
    }
#pragma unroll
    for (int j = 0; j < QU; ++j) f(u[j], i0 + j * 256L + threadIdx.x < n);
  }
}

__global__ void __launch_bounds__(256) quantile_hist_kernel(const QArgs a, int pass) {
  __shared__ unsigned hist[256];
  unsigned* w = a.ws + (long)blockIdx.y * QW_STRIDE;
  const float* xs = a.x + (long)blockIdx.y * a.n;
  hist[threadIdx.x] = 0;
  unsigned prefix;
  unsigned long long rank;
  q_prefix(w, pass, a.k, prefix, rank);
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const unsigned mask_hi = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
  q_for_slice(xs, a.n, [&](unsigned u, bool live) {
    if (live && (u & mask_hi) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
  });
  __syncthreads();
  if (hist[threadIdx.x]) atomicAdd(&w[4 + 256 * pass + threadIdx.x], hist[threadIdx.x]);
}

__global__ void __launch_bounds__(256) quantile_count_kernel(const QArgs a) {
  __shared__ unsigned sh_min;
  __shared__ unsigned long long sh_cnt;
  unsigned* w = a.ws + (long)blockIdx.y * QW_STRIDE;
  const float* xs = a.x + (long)blockIdx.y * a.n;
  unsigned vlo;  // exact bit pattern of the k-th smallest |x|
  unsigned long long rank;
  q_prefix(w, 4, a.k, vlo, rank);
  if (threadIdx.x == 0) { sh_min = 0x7F800000u; sh_cnt = 0; }
  __syncthreads();
  unsigned long long cnt = 0;
  unsigned vmin = 0x7F800000u;
  q_for_slice(xs, a.n, [&](unsigned u, bool live) {
    if (live) {
      if (u <= vlo) ++cnt; else vmin = u < vmin ? u : vmin;
    }
  });
  atomicAdd(&sh_cnt, cnt);
  atomicMin(&sh_min, vmin);
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(reinterpret_cast<unsigned long long*>(w + 2), sh_cnt);
    atomicMin(&w[1], sh_min);
  }
}

__global__ void __launch_bounds__(256) quantile_apply_kernel(const QArgs a) {
  __shared__ float sh_s;
  unsigned* w = a.ws + (long)blockIdx.y * QW_STRIDE;
  float* xs = a.x + (long)blockIdx.y * a.n;
  unsigned prefix;
  unsigned long long rank;
  q_prefix(w, 4, a.k, prefix, rank);
  if (threadIdx.x == 0) {
    const float vlo = __uint_as_float(prefix);
    // v[k+1] equals v[k] if more copies of it remain, else it is the smallest value above it
    float vhi = vlo;
    const unsigned long long cnt_le = *reinterpret_cast<const unsigned long long*>(w + 2);
    if (a.frac > 0.f && cnt_le < (unsigned long long)a.k + 2) vhi = __uint_as_float(w[1]);
    // at::lerp: w < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
    const float diff = __fsub_rn(vhi, vlo);
    const float qv = a.frac < 0.5f ? __fadd_rn(vlo, __fmul_rn(a.frac, diff))
                                   : __fsub_rn(vhi, __fmul_rn(diff, __fsub_rn(1.f, a.frac)));
    sh_s = fmaxf(qv, a.max_val);
  }
  __syncthreads();
  const float s = sh_s;
  const long stride = (long)gridDim.x * 256 * QU;
  for (long i0 = (long)blockIdx.x * 256 * QU; i0 < a.n; i0 += stride) {
    float v[QU];
#pragma unroll
    for (int j = 0; j < QU; ++j) {
      const long i = i0 + j * 256L + threadIdx.x;
      v[j] = xs[i < a.n ? i : a.n - 1];
    }
#pragma unroll
    for (int j = 0; j < QU; ++j) {
      const long i = i0 + j * 256L + threadIdx.x;
      if (i < a.n) xs[i] = __fdiv_rn(fminf(fmaxf(v[j], -s), s), s);
    }
  }
  // every block has read w[0..3] before any block could overwrite w[0]: s goes to a separate word read by nobody here
  if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<float*>(w)[0] = s;
}

__global__ void __launch_bounds__(256) quantile_init_kernel(unsigned* ws, long words) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < words; i += 256L * gridDim.x)
    ws[i] = (i % QW_STRIDE) == 1 ? 0x7F800000u : 0u;
}

__global__ void __launch_bounds__(256) quantile_gather_s_kernel(const unsigned* ws, float* s_out, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b < B) s_out[b] = reinterpret_cast<const float*>(ws)[(long)b * QW_STRIDE];
}

inline bool all_aligned(std::initializer_list<const void*> ps, long n, int) {
  if (n % 4) return false;
  for (const void* p : ps)
    if (p && !tmdiff::aligned16(p)) return false;
  return true;
}

}  // namespace

extern "C" int tmdiff_ddpm_step(const float* x, const float* eps, const float* noise, const float* ms, float* out,
                                float* img_out, int64_t n, float c_recip, float c_recipm1, float coef1, float coef2,
                                float sigma, int32_t clip, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && eps && out, "ddpm_step: NULL pointer");
  TMDIFF_REQUIRE(n >= 0, "ddpm_step: n=%ld", (long)n);
  TMDIFF_REQUIRE(noise || sigma == 0.f, "ddpm_step: noise is NULL but sigma != 0");
  TMDIFF_REQUIRE(!img_out || ms, "ddpm_step: img_out needs ms");
  if (n == 0) return TMDIFF_OK;
  DdpmCoef k{c_recip, c_recipm1, coef1, coef2, sigma, clip};
  if (all_aligned({x, eps, noise, ms, out, img_out}, n, 0)) {
    ddpm_step_kernel<4><<<grid_for(n / 4), 256, 0, as_stream(stream)>>>(x, eps, noise, ms, out, img_out, n / 4, k);
  } else {
    ddpm_step_kernel<1><<<grid_for(n), 256, 0, as_stream(stream)>>>(x, eps, noise, ms, out, img_out, n, k);
  }
  return check_launch("ddpm_step");
}

extern "C" int tmdiff_axpby(const float* const in[4], const float coef[4], int32_t n_in, float* out, int64_t n,
                            tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(in && coef && out, "axpby: NULL pointer");
  TMDIFF_REQUIRE(n_in >= 1 && n_in <= 4 && n >= 0, "axpby: n_in=%d n=%ld", n_in, (long)n);
  AxpbyArgs a;
  bool al = n % 4 == 0 && aligned16(out);
  for (int k = 0; k < 4; ++k) {
    a.in[k] = k < n_in ? in[k] : nullptr;
    a.coef[k] = k < n_in ? coef[k] : 0.f;
    if (k < n_in) {
      TMDIFF_REQUIRE(in[k] != nullptr, "axpby: input %d is NULL", k);
      al = al && aligned16(in[k]);
    }
  }
  a.n_in = n_in;
  if (n == 0) return TMDIFF_OK;
  if (al)
    axpby_kernel<4><<<grid_for(n / 4), 256, 0, as_stream(stream)>>>(a, out, n / 4);
  else
    axpby_kernel<1><<<grid_for(n), 256, 0, as_stream(stream)>>>(a, out, n);
  return check_launch("axpby");
}

extern "C" int32_t tmdiff_multi_axpby_chunk(void) { return MT_CHUNK; }

extern "C" int tmdiff_multi_axpby(const tmdiff_mt_entry* tensors_dev, const int32_t* chunk_tensor_dev,
                                  const int32_t* chunk_index_dev, int32_t n_chunks, float ca, float cb,
                                  tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(tensors_dev && chunk_tensor_dev && chunk_index_dev && n_chunks >= 0, "multi_axpby: bad arguments");
  if (n_chunks == 0) return TMDIFF_OK;
  multi_axpby_kernel<<<(unsigned)n_chunks, 256, 0, as_stream(stream)>>>(tensors_dev, chunk_tensor_dev, chunk_index_dev, ca, cb);
  return check_launch("multi_axpby");
}

extern "C" int tmdiff_multi_adamw(const tmdiff_adamw_entry* tensors_dev, const int32_t* chunk_tensor_dev,
                                  const int32_t* chunk_index_dev, int32_t n_chunks, const float* lr_dev, const float* step_dev,
                                  float beta1, float beta2, float eps, float weight_decay, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(tensors_dev && chunk_tensor_dev && chunk_index_dev && lr_dev && step_dev && n_chunks >= 0, "multi_adamw: bad arguments");
  TMDIFF_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && eps >= 0.f && weight_decay >= 0.f,
                 "multi_adamw: betas (%g, %g) eps %g weight_decay %g", (double)beta1, (double)beta2, (double)eps, (double)weight_decay);
  if (n_chunks == 0) return TMDIFF_OK;
  multi_adamw_kernel<<<(unsigned)n_chunks, 256, 0, as_stream(stream)>>>(tensors_dev, chunk_tensor_dev, chunk_index_dev, lr_dev,
                                                                         step_dev, beta1, beta2, eps, weight_decay);
  return check_launch("multi_adamw");
}

extern "C" int tmdiff_x0_from_model(const float* x, const float* model_out, float* x0, int64_t n, float alpha,
                                    float sigma, int32_t model_is_x_start, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && model_out && x0 && n >= 0, "x0_from_model: bad arguments");
  if (n == 0) return TMDIFF_OK;
  if (all_aligned({x, model_out, x0}, n, 0))
    x0_kernel<4><<<grid_for(n / 4), 256, 0, as_stream(stream)>>>(x, model_out, x0, n / 4, alpha, sigma, model_is_x_start);
  else
    x0_kernel<1><<<grid_for(n), 256, 0, as_stream(stream)>>>(x, model_out, x0, n, alpha, sigma, model_is_x_start);
  return check_launch("x0_from_model");
}

extern "C" size_t tmdiff_abs_quantile_workspace_bytes(int32_t B, int64_t /*n_per_sample*/) {
  // [0, B) floats: the per-sample thresholds s (readable by the caller); then the select state of every sample
  return B > 0 ? ((size_t)(B + 3) / 4 * 4 + (size_t)B * QW_STRIDE) * sizeof(float) : 0;
}

extern "C" int tmdiff_abs_quantile_clamp(float* x0, int32_t B, int64_t n_per_sample, float q, float max_val,
                                         void* workspace, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x0 && workspace && B >= 0 && n_per_sample > 0, "abs_quantile_clamp: bad arguments");
  TMDIFF_REQUIRE(q >= 0.f && q <= 1.f, "abs_quantile_clamp: q=%f outside [0,1]", q);
  TMDIFF_REQUIRE(B <= 65535, "abs_quantile_clamp: B=%d", B);
  if (B == 0) return TMDIFF_OK;
  // torch.quantile: rank = q * (n - 1) evaluated in the input dtype (fp32); lerp weight = rank - floor(rank)
  const float rank = q * (float)(n_per_sample - 1);
  long k = (long)floorf(rank);
  if (k > n_per_sample - 1) k = n_per_sample - 1;
  float* s_out = reinterpret_cast<float*>(workspace);
  QArgs a{x0, reinterpret_cast<unsigned*>(workspace) + (B + 3) / 4 * 4, n_per_sample, k, rank - (float)k, max_val};
  hipStream_t st = as_stream(stream);
  long per = (n_per_sample + 256 * QU - 1) / (256 * QU);  // workgroups per sample: one batch of loads each, capped
  const long cap = B >= 256 ? 4 : 1024 / B;
  if (per > cap) per = cap;
  const dim3 grid((unsigned)per, (unsigned)B);
  const long words = (long)B * QW_STRIDE;
  quantile_init_kernel<<<(unsigned)((words + 255) / 256), 256, 0, st>>>(a.ws, words);
  for (int pass = 0; pass < 4; ++pass) quantile_hist_kernel<<<grid, 256, 0, st>>>(a, pass);
  quantile_count_kernel<<<grid, 256, 0, st>>>(a);
  quantile_apply_kernel<<<grid, 256, 0, st>>>(a);
  quantile_gather_s_kernel<<<(B + 255) / 256, 256, 0, st>>>(a.ws, s_out, B);
  return check_launch("abs_quantile_clamp");
}

extern "C" int tmdiff_add(const float* a, const float* b, float* out, int64_t n, float sign_b, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(a && b && out && n >= 0, "add: bad arguments");
  if (n == 0) return TMDIFF_OK;
  if (all_aligned({a, b, out}, n, 0))
    add_kernel<4><<<grid_for(n / 4), 256, 0, as_stream(stream)>>>(a, b, out, n / 4, sign_b);
  else
    add_kernel<1><<<grid_for(n), 256, 0, as_stream(stream)>>>(a, b, out, n, sign_b);
  return check_launch("add");
}

extern "C" int tmdiff_q_sample(const float* x0, const float* noise, const float* a, float* out, int32_t B,
                               int64_t n_per_sample, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x0 && noise && a && out && B >= 0 && B <= 65535 && n_per_sample >= 0, "q_sample: bad arguments");
  if (B == 0 || n_per_sample == 0) return TMDIFF_OK;
  long blocks = (n_per_sample + 255) / 256;
  if (blocks > 256) blocks = 256;
  q_sample_kernel<<<dim3((unsigned)blocks, B), 256, 0, as_stream(stream)>>>(x0, noise, a, out, n_per_sample);
  return check_launch("q_sample");
}
