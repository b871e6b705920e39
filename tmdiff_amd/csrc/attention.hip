// Standalone attention operators of core/Attention.py on gfx950.  These modules are
// imported by nothing in the reference; they are built as operators because the north star names them.
//   attn_fwd_kernel : softmax(Q K^T * scale [+ key mask]) V, fp32 MFMA (v_mfma_f32_32x32x2_f32), online softmax
//                     (SpatialSelfAttention :145-157, CrossAttention :186-213)
//   gemm_nt_kernel  : C[M,N] = A[M,K] W[N,K]^T + bias (+ residual), fp32 MFMA -- the token-major Linear layers
//                     (to_q/k/v/out, FeedForward, proj_in/out with use_linear) :172-181, :69-96
//   group_norm, layer_norm, geglu : wave-shuffle reductions / elementwise (:108-109, :279-281, :69-76)
// 1x1 Conv2d projections on [B,C,H,W] reuse tmdiff_conv3d_fwd with ksize 1 (N = 1).
//
// Attention layout trick: the score tile is computed TRANSPOSED, S^T[key, query] = K Q^T, so a lane owns one
// query column: its 16 accumulator registers are 16 keys of that query.  Row statistics (max, sum) are then
// in-lane reductions plus one exchange between the two half-waves, the rescale of the output accumulator
// O^T[d, query] is a per-lane multiply, and P^T feeds the second product O^T += V^T P^T directly from the
// accumulator registers: register r of half-wave h is key (r&3)+8(r>>2)+4h, exactly the K-pair of MFMA step r.
#include <cstdlib>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// ---------------------------------------------------------------------------------------------------------
// attention.  q [BH, Nq, D], k [BH, Nk, D], v [BH, Nk, D] with arbitrary row / head strides (elements).
// One workgroup = 4 waves = 128 queries of one (batch, head); K/V tiles of 32 keys are staged in LDS once per
// workgroup; D <= 256, D % 2 == 0.
// ---------------------------------------------------------------------------------------------------------
struct AttnArgs {
  const float* q; const float* k; const float* v; float* o;
  const unsigned char* mask;  // [B, Nk] key mask (1 = keep) or NULL
  long q_bs, q_hs, q_rs;      // batch / head / row strides of q (elements); same for k, v, o
  long k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs;
  int H, Nq, Nk, D;
  float scale;
};

template <int DT>  // DT = ceil(D / 32): 32-row tiles of the transposed output accumulator
__global__ void __launch_bounds__(256) attn_fwd_kernel(const AttnArgs a) {
  constexpr int DP = DT * 32;       // padded head dim
  constexpr int KS = DP + 1;        // odd LDS row stride: lanes read one row each without bank conflicts
  __shared__ float qs[128 * KS];    // this workgroup's queries   [128][DP]
  __shared__ float ks[32 * KS];     // current key tile            [32][DP]
  __shared__ float vs[32 * KS];     // current value tile          [32][DP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / a.H, hd = bh % a.H;
  const int q0 = blockIdx.x * 128;
  const float* qb = a.q + b * a.q_bs + hd * a.q_hs;
  const float* kb = a.k + b * a.k_bs + hd * a.k_hs;
  const float* vb = a.v + b * a.v_bs + hd * a.v_hs;

  for (int e = tid; e < 128 * DP; e += 256) {
    const int r = e / DP, c = e % DP;
    qs[r * KS + c] = (q0 + r < a.Nq && c < a.D) ? qb[(long)(q0 + r) * a.q_rs + c] * a.scale : 0.f;
  }
  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -3.0e38f, l_run = 0.f;  // running max / sum of this lane's query (both half-waves keep a copy)
  const float* qrow = qs + (wv * 32 + l31) * KS;

  for (int k0 = 0; k0 < a.Nk; k0 += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * DP; e += 256) {
      const int r = e / DP, c = e % DP;
      const bool ok = k0 + r < a.Nk && c < a.D;
      ks[r * KS + c] = ok ? kb[(long)(k0 + r) * a.k_rs + c] : 0.f;
      vs[r * KS + c] = ok ? vb[(long)(k0 + r) * a.v_rs + c] : 0.f;
    }
    __syncthreads();
    // S^T[key, query] = sum_d K[key, d] * (scale Q)[query, d]
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* krow = ks + l31 * KS;
    for (int d = 0; d < DP; d += 2) s = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[d + h], qrow[d + h], s, 0, 0, 0);
    // mask keys past Nk (and masked keys); register r of half h is key (r&3) + 8(r>>2) + 4h
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      bool keep = key < a.Nk;
      if (keep && a.mask) keep = a.mask[(long)b * a.Nk + key] != 0;
      s[r] = keep ? s[r] : -3.4028234e38f;   // masked_fill(-finfo.max), ref :203-204
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_new);
      psum += s[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // O^T[d, query] = alpha * O^T + sum_key V[key, d] * P^T[key, query]; P^T comes straight from s[r]
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
        oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key * KS + t * 32 + l31], s[r], oacc[t], 0, 0, 0);
      }
    }
  }
  // out[query, d] = O^T[d, query] / l; transpose through LDS (reuse the query buffer) for coalesced stores
  __syncthreads();
  const float inv = 1.f / l_run;
  float* ot = qs + wv * 32 * KS;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[l31 * KS + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = oacc[t][r] * inv;
  __syncthreads();
  float* ob = a.o + b * a.o_bs + hd * a.o_hs;
  for (int e = tid; e < 128 * DP; e += 256) {
    const int r = e / DP, c = e % DP;
    if (q0 + r < a.Nq && c < a.D) ob[(long)(q0 + r) * a.o_rs + c] = qs[r * KS + c];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Pipelined variant for head dims 64 and 128 (SpatialSelfAttention at c = 64 / 128, CrossAttention d_head 64): same
// arithmetic and register layout as attn_fwd_kernel, but
//   * K/V tiles go global -> LDS with global_load_lds (one 256-byte row piece per instruction, no registers), double
//     buffered: tile i+1 lands while tile i is multiplied, one barrier per tile instead of a synchronous
//     load - barrier - compute - barrier sequence;
//   * the workgroup's (pre-scaled) query rows are held in registers, so S^T = K Q^T costs one LDS operand read per MFMA
//     instead of two;
//   * the Q staging buffer, the K/V stages and the O transpose buffer share the same LDS (33 / 66 KB: 2+ workgroups per CU).
// ---------------------------------------------------------------------------------------------------------
__device__ __attribute__((unused)) const float kZeroRow[128] = {};

template <int DT>  // D == 32 * DT, DT = 2 or 4
__global__ void __launch_bounds__(256, 2) attn_fwd_dma_kernel(const AttnArgs a) {
  constexpr int D = DT * 32, KS = D + 1, TILE = 32 * KS, RP = D / 64;   // RP = 256-byte pieces per row
  static_assert(DT == 2 || DT == 4, "head dim 64 or 128");
  static_assert(4 * TILE == 128 * KS, "two double-buffered K/V stages alias the 128-row Q / O buffer");
  __shared__ float lds[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y, b = bh / a.H, hd = bh % a.H;
  const int q0 = blockIdx.x * 128;
  const float* qb = a.q + b * a.q_bs + hd * a.q_hs;
  const float* kb = a.k + b * a.k_bs + hd * a.k_hs;
  const float* vb = a.v + b * a.v_bs + hd * a.v_hs;

  // queries: coalesced rows -> LDS -> this lane's row (query wv*32 + l31), elements 2j + h, into registers
  for (int e = tid; e < 128 * D; e += 256) {
    const int r = e / D, c = e % D;
    lds[r * KS + c] = q0 + r < a.Nq ? qb[(long)(q0 + r) * a.q_rs + c] * a.scale : 0.f;
  }
  __syncthreads();
  float qreg[D / 2];
  {
    const float* qrow = lds + (wv * 32 + l31) * KS + h;
#pragma unroll
    for (int j = 0; j < D / 2; ++j) qreg[j] = qrow[2 * j];
  }
  __syncthreads();   // everyone has its queries: the buffer becomes the K/V stages

  // DMA of key tile starting at k0 into stage st: wave w brings rows w, w+4, ... of K and of V
  auto issue = [&](int k0, float* st) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = wv + 4 * i;
      const bool ok = k0 + r < a.Nk;
#pragma unroll
      for (int p = 0; p < RP; ++p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the builtin exists in the device pass only
        const float* ksrc = ok ? kb + (long)(k0 + r) * a.k_rs + p * 64 + lane : kZeroRow + lane;
        const float* vsrc = ok ? vb + (long)(k0 + r) * a.v_rs + p * 64 + lane : kZeroRow + lane;
        __builtin_amdgcn_global_load_lds(ksrc, st + r * KS + p * 64, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(vsrc, st + TILE + r * KS + p * 64, 4, 0, 0);
#else
        (void)ok; (void)kb; (void)vb; (void)st;
#endif
      }
    }
  };

  f32x16 oacc[DT];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[t][r] = 0.f;
  float m_run = -3.0e38f, l_run = 0.f;

  issue(0, lds);
  int it = 0;
  for (int k0 = 0; k0 < a.Nk; k0 += 32, ++it) {
    __syncthreads();   // tile `it` has landed (own pieces: vmcnt(0) in front of the barrier); tile it-1 is consumed by all
    if (k0 + 32 < a.Nk) issue(k0 + 32, lds + ((it + 1) & 1) * 2 * TILE);
    const float* ks = lds + (it & 1) * 2 * TILE;
    const float* vs = ks + TILE;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const float* krow = ks + l31 * KS + h;
#pragma unroll
    for (int j = 0; j < D / 2; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[2 * j], qreg[j], s, 0, 0, 0);
    float mx = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      bool keep = key < a.Nk;
      if (keep && a.mask) keep = a.mask[(long)b * a.Nk + key] != 0;
      s[r] = keep ? s[r] : -3.4028234e38f;   // masked_fill(-finfo.max), ref :203-204
      mx = fmaxf(mx, s[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __expf(s[r] - m_new);
      psum += s[r];
    }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][r] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
        oacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key * KS + t * 32 + l31], s[r], oacc[t], 0, 0, 0);
      }
    }
  }
  // out[query, d] = O^T[d, query] / l; transpose through LDS for coalesced stores
  __syncthreads();
  const float inv = 1.f / l_run;
  float* ot = lds + wv * 32 * KS;
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[l31 * KS + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = oacc[t][r] * inv;
  __syncthreads();
  float* ob = a.o + b * a.o_bs + hd * a.o_hs;
  for (int e = tid; e < 128 * D; e += 256) {
    const int r = e / D, c = e % D;
    if (q0 + r < a.Nq) ob[(long)(q0 + r) * a.o_rs + c] = lds[r * KS + c];
  }
}

// ---------------------------------------------------------------------------------------------------------
// Small-context variant (CrossAttention on the CLIP context: 77 keys, d_head 64 -- core/Attention.py:165-214 with
// context [B, 77, 768]): Nk <= NKR keys, D == 64.  The K and V of one (batch, head) are 2 x 20 KB: they are loaded into LDS
// ONCE per workgroup and stay there while its four waves walk over QPW queries, 32 at a time -- no K/V tile loop, no block
// barrier after the prologue, and the softmax is SINGLE-PASS (all scores of a query are in registers: one max, one sum, no
// running rescale of the output accumulator).  A wave stages its 32 query rows through a wave-private LDS tile (coalesced
// rows in, this lane's row out), computes S^T = K Q^T for the ceil(Nk / 32) key tiles, and skips the P V MFMAs of 8-key
// groups that lie entirely beyond Nk (77 keys: 80 of 96 key slots are multiplied); the next block's query rows are already
// in flight while the MFMAs of the current one run.  LDS: K 20.8 KB (odd row stride) + V 20.5 KB + 4 x 8.3 KB staging = 74.5
// KB at NKR = 80: two workgroups per CU.
// ---------------------------------------------------------------------------------------------------------
template <int NKR>   // key rows held in LDS: 80 (Nk <= 80) or 96
__global__ void __launch_bounds__(256, NKR <= 80 ? 2 : 1) attn_ctx_kernel(const AttnArgs a, const int qpw) {
  constexpr int D = 64, KS = D + 1, NT = (NKR + 31) / 32;
  __shared__ float ks[NKR * KS];
  __shared__ __attribute__((aligned(16))) float vs[NKR * D];
  __shared__ float kbias[NT * 32];          // 0 for a kept key, -FLT_MAX for a masked one / a slot beyond Nk
  __shared__ float stage[4 * 32 * KS];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bh = blockIdx.y, b = bh / a.H, hd = bh % a.H;
  const float* qb = a.q + b * a.q_bs + hd * a.q_hs;
  const float* kb = a.k + b * a.k_bs + hd * a.k_hs;
  const float* vb = a.v + b * a.v_bs + hd * a.v_hs;
  float* ob = a.o + b * a.o_bs + hd * a.o_hs;
  for (int e = tid; e < NKR * D; e += 256) {
    const int r = e >> 6, c = e & 63;
    const bool ok = r < a.Nk;
    ks[r * KS + c] = ok ? kb[(long)r * a.k_rs + c] : 0.f;
    vs[r * D + c] = ok ? vb[(long)r * a.v_rs + c] : 0.f;
  }
  for (int e = tid; e < NT * 32; e += 256)
    kbias[e] = (e < a.Nk && (!a.mask || a.mask[(long)b * a.Nk + e] != 0)) ? 0.f : -3.4028234e38f;   // masked_fill(-finfo.max), ref :203-204
  __syncthreads();
  const int nt = (a.Nk + 31) >> 5;          // key tiles in use (uniform)
  float* st = stage + wv * 32 * KS;         // this wave's staging tile [32 rows][KS]
  const int qbeg = blockIdx.x * qpw + wv * 32, qend = min(a.Nq, (blockIdx.x + 1) * qpw);
  // coalesced rows: float4 i of this lane is row (i * 64 + lane) / 16, columns 4 * ((i * 64 + lane) % 16) ..
  const int lrow = lane >> 4, lc4 = (lane & 15) * 4;
  float4 qn[8];
  auto load_q = [&](int q0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = q0 + 4 * i + lrow;
      qn[i] = row < a.Nq ? *reinterpret_cast<const float4*>(qb + (long)row * a.q_rs + lc4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  if (qbeg < qend) load_q(qbeg);
  for (int q0 = qbeg; q0 < qend; q0 += 128) {
    // ---- this block's queries: staged rows -> this lane's row (query l31), elements 2j + h, pre-scaled ----------------
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float* dst = st + (4 * i + lrow) * KS + lc4;
      dst[0] = qn[i].x, dst[1] = qn[i].y, dst[2] = qn[i].z, dst[3] = qn[i].w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float qreg[D / 2];
#pragma unroll
    for (int j = 0; j < D / 2; ++j) qreg[j] = st[l31 * KS + 2 * j + h] * a.scale;
    if (q0 + 128 < qend) load_q(q0 + 128);          // the next block's rows fly under this block's MFMAs
    // ---- S^T[key, query] for every key tile ---------------------------------------------------------------------------
    f32x16 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
      if (t < nt) {
        const int row = min(32 * t + l31, NKR - 1);          // (slots beyond NKR repeat the last row: masked below)
        const float* krow = ks + row * KS + h;
#pragma unroll
        for (int j = 0; j < D / 2; ++j) s[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(krow[2 * j], qreg[j], s[t], 0, 0, 0);
      }
    }
    // ---- single-pass softmax over this lane's query: register r of tile t, half h is key 32 t + (r&3) + 8 (r>>2) + 4 h ----
    float mx = -3.4028234e38f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float kbv = kbias[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
        s[t][r] = (t < nt && kbv == 0.f) ? s[t][r] : -3.4028234e38f;
        mx = fmaxf(mx, s[t][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool slot = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h < a.Nk;       // (a slot beyond Nk is no key at all)
        s[t][r] = slot ? __expf(s[t][r] - mx) : 0.f;
        psum += s[t][r];
      }
    psum += __shfl_xor(psum, 32, 64);
    // ---- O^T[d, query] = sum_key V[key, d] P^T[key, query] ------------------------------------------------------------
    f32x16 oacc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[tt][r] = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (32 * t + 8 * g < a.Nk) {                        // uniform: whole 8-key groups beyond Nk are skipped
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const int r = 4 * g + r4, key = min(32 * t + r4 + 8 * g + 4 * h, NKR - 1);
              oacc[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(vs[key * D + tt * 32 + l31], s[t][r], oacc[tt], 0, 0, 0);
            }
          }
    }
    // ---- out[query, d] = O^T[d, query] / l: transpose through the staging tile, coalesced float4 rows out -----------------
    const float inv = 1.f / psum;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[l31 * KS + tt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] = oacc[tt][r] * inv;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = q0 + 4 * i + lrow;
      const float* src = st + (4 * i + lrow) * KS + lc4;
      if (row < a.Nq) *reinterpret_cast<float4*>(ob + (long)row * a.o_rs + lc4) = make_float4(src[0], src[1], src[2], src[3]);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------
// C[M,N] = A[M,K] W[N,K]^T (+ bias[N]) (+ residual[M,N]); 64x64 tile per workgroup, each wave a 32x32 block.
// MFMA rows = 32 rows of A (lane holds A[row][k]), cols = 32 output features (lane holds W[feature][k]).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ Wt,
                                                      const float* __restrict__ bias, const float* __restrict__ res,
                                                      float* __restrict__ Cm, int M, int N, int K) {
  constexpr int BK = 32, LS = BK + 1;
  __shared__ float as[64 * LS];
  __shared__ float ws[64 * LS];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();
    for (int e = tid; e < 64 * BK; e += 256) {
      const int r = e / BK, c = e % BK;
      as[r * LS + c] = (m0 + r < M && k0 + c < K) ? A[(long)(m0 + r) * K + k0 + c] : 0.f;
      ws[r * LS + c] = (n0 + r < N && k0 + c < K) ? Wt[(long)(n0 + r) * K + k0 + c] : 0.f;
    }
    __syncthreads();
    const float* ar = as + (wm + l31) * LS;
    const float* wr = ws + (wn + l31) * LS;
#pragma unroll
    for (int k = 0; k < BK; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[k + h], wr[k + h], acc, 0, 0, 0);
  }
  // D layout: col = l31 (feature), row = (r&3)+8(r>>2)+4h (row of A)
  const int n = n0 + wn + l31;
  if (n >= N) return;
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (m < M) {
      float t = acc[r] + bv;
      if (res) t += res[(long)m * N + n];
      Cm[(long)m * N + n] = t;
    }
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// GroupNorm over [B, C, P]: one workgroup per (b, group); two passes (mean, then centred variance).
__global__ void __launch_bounds__(256) group_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, int C,
                                                         long P, int groups, float eps) {
  __shared__ float red[4];
  const int g = blockIdx.x, b = blockIdx.y, cpg = C / groups;
  const long n = (long)cpg * P;
  const float* xs = x + ((long)b * C + g * cpg) * P;
  float* ys = y + ((long)b * C + g * cpg) * P;
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) s += xs[i];
  const float mean = block_sum(s, red) / (float)n;
  float v = 0.f;
  for (long i = threadIdx.x; i < n; i += 256) { const float d = xs[i] - mean; v += d * d; }
  const float rstd = rsqrtf(block_sum(v, red) / (float)n + eps);
  for (long i = threadIdx.x; i < n; i += 256) {
    const int c = g * cpg + (int)(i / P);
    ys[i] = (xs[i] - mean) * rstd * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
  }
}

// LayerNorm over the last dim of [rows, D]: one wave per row.
__global__ void __launch_bounds__(256) layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y,
                                                         long rows, int D, float eps) {
  const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xs = x + row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xs[i];
  const float mean = wave_sum(s) / D;
  float v = 0.f;
  for (int i = lane; i < D; i += 64) { const float d = xs[i] - mean; v += d * d; }
  const float rstd = rsqrtf(wave_sum(v) / D + eps);
  for (int i = lane; i < D; i += 64) y[row * D + i] = (xs[i] - mean) * rstd * gamma[i] + beta[i];
}

// GEGLU: y[r, j] = u[r, j] * gelu(u[r, inner + j]) (exact erf GELU, F.gelu default), u = [rows, 2*inner]
__global__ void __launch_bounds__(256) geglu_kernel(const float* __restrict__ u, float* __restrict__ y, long rows,
                                                    int inner, int gate_only) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= rows * inner) return;
  const long r = i / inner;
  const int j = (int)(i % inner);
  if (gate_only) {  // plain GELU on [rows, inner]
    const float gte = u[i];
    y[i] = 0.5f * gte * (1.f + erff(gte * 0.70710678118654752440f));
    return;
  }
  const float a = u[r * 2 * inner + j], gte = u[r * 2 * inner + inner + j];
  y[i] = a * (0.5f * gte * (1.f + erff(gte * 0.70710678118654752440f)));
}

}  // namespace

extern "C" int tmdiff_attn_fwd(const float* q, const float* k, const float* v, float* out, const unsigned char* key_mask,
                               int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t D, const int64_t q_strides[3],
                               const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3],
                               float scale, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(q && k && v && out && q_strides && k_strides && v_strides && o_strides, "attn_fwd: NULL pointer");
  TMDIFF_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0 && (long)B * H <= 65535, "attn_fwd: bad extents");
  TMDIFF_REQUIRE(D >= 2 && D <= 128 && D % 2 == 0, "attn_fwd: head dim %d (even, <= 128)", D);
  AttnArgs a;
  a.q = q; a.k = k; a.v = v; a.o = out; a.mask = key_mask;
  a.q_bs = q_strides[0]; a.q_hs = q_strides[1]; a.q_rs = q_strides[2];
  a.k_bs = k_strides[0]; a.k_hs = k_strides[1]; a.k_rs = k_strides[2];
  a.v_bs = v_strides[0]; a.v_hs = v_strides[1]; a.v_rs = v_strides[2];
  a.o_bs = o_strides[0]; a.o_hs = o_strides[1]; a.o_rs = o_strides[2];
  a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.scale = scale;
  dim3 grid((Nq + 127) / 128, B * H);
  hipStream_t st = as_stream(stream);
  static const bool no_dma = getenv("TMDIFF_ATTN_SIMPLE") != nullptr;   // experiments: the un-pipelined kernel everywhere
  static const bool no_ctx = getenv("TMDIFF_ATTN_NO_CTX") != nullptr;   // experiments: never the small-context kernel
  // a short context (the 77 CLIP tokens) with d_head 64: K / V resident in LDS, single-pass softmax.  Needs 16-byte aligned
  // query / output rows (float4 row pieces).
  if (!no_dma && !no_ctx && D == 64 && Nk <= 96 && a.q_rs % 4 == 0 && a.o_rs % 4 == 0 && a.q_bs % 4 == 0 && a.q_hs % 4 == 0 &&
      a.o_bs % 4 == 0 && a.o_hs % 4 == 0 && aligned16(q) && aligned16(out)) {
    // queries per workgroup: blocks of 128 (4 waves x 32); enough workgroups to fill the chip twice over, at most 1024 queries each
    int qpw = 128;
    while (qpw < 1024 && (long)((Nq + 2 * qpw - 1) / (2 * qpw)) * B * H >= 1024) qpw *= 2;
    dim3 g((Nq + qpw - 1) / qpw, B * H);
    if (Nk <= 80) attn_ctx_kernel<80><<<g, 256, 0, st>>>(a, qpw);
    else attn_ctx_kernel<96><<<g, 256, 0, st>>>(a, qpw);
    return check_launch("attn_fwd");
  }
  if (!no_dma && (D == 64 || D == 128)) {
    if (D == 64) attn_fwd_dma_kernel<2><<<grid, 256, 0, st>>>(a);
    else attn_fwd_dma_kernel<4><<<grid, 256, 0, st>>>(a);
    return check_launch("attn_fwd");
  }
  switch ((D + 31) / 32) {
    case 1: attn_fwd_kernel<1><<<grid, 256, 0, st>>>(a); break;
    case 2: attn_fwd_kernel<2><<<grid, 256, 0, st>>>(a); break;
    case 3: attn_fwd_kernel<3><<<grid, 256, 0, st>>>(a); break;
    default: attn_fwd_kernel<4><<<grid, 256, 0, st>>>(a); break;
  }
  return check_launch("attn_fwd");
}

extern "C" int tmdiff_gemm_nt(const float* A, const float* Wt, const float* bias, const float* residual, float* Cm,
                              int64_t M, int32_t N, int32_t K, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(A && Wt && Cm && M >= 0 && N > 0 && K > 0 && M < (1L << 31), "gemm_nt: bad arguments");
  if (M == 0) return TMDIFF_OK;
  dim3 grid((N + 63) / 64, (unsigned)((M + 63) / 64));
  gemm_nt_kernel<<<grid, 256, 0, as_stream(stream)>>>(A, Wt, bias, residual, Cm, (int)M, N, K);
  return check_launch("gemm_nt");
}

extern "C" int tmdiff_group_norm(const float* x, const float* gamma, const float* beta, float* y, int32_t B, int32_t C,
                                 int64_t P, int32_t groups, float eps, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && y && B > 0 && B <= 65535 && C > 0 && P > 0 && groups > 0 && C % groups == 0,
                 "group_norm: bad arguments (C=%d groups=%d)", C, groups);
  group_norm_kernel<<<dim3(groups, B), 256, 0, as_stream(stream)>>>(x, gamma, beta, y, C, P, groups, eps);
  return check_launch("group_norm");
}

extern "C" int tmdiff_layer_norm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows,
                                 int32_t D, float eps, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && gamma && beta && y && rows >= 0 && D > 0, "layer_norm: bad arguments");
  if (rows == 0) return TMDIFF_OK;
  layer_norm_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, as_stream(stream)>>>(x, gamma, beta, y, rows, D, eps);
  return check_launch("layer_norm");
}

extern "C" int tmdiff_geglu(const float* u, float* y, int64_t rows, int32_t inner, int32_t gelu_only,
                            tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(u && y && rows >= 0 && inner > 0, "geglu: bad arguments");
  if (rows == 0) return TMDIFF_OK;
  const long n = rows * inner;
  geglu_kernel<<<(unsigned)((n + 255) / 256), 256, 0, as_stream(stream)>>>(u, y, rows, inner, gelu_only);
  return check_launch("geglu");
}
