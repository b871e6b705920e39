// Backward kernels of the finetune path for gfx950.
//
// The reference gets these from ATen autograd (F.conv3d / nn.Conv3d / nn.Linear backward, elementwise
// SiLU / add / mul backward: GeneralModel/Hyper_unet_general.py:51-77, :100-113, :158-273, :334-414).  Here:
//   dL/dx'  : the forward MFMA kernel on g with data-gradient packed weights (conv3d.hip, pack mode 1)
//   dL/dw   : conv3d_wgrad_kernel below (fp32 MFMA; K = batch x positions, split over workgroups, then reduced)
//   dL/dx, dL/dshift, dL/dscale : prologue_bwd_kernel (HBM-bound, one workgroup per (b, c) plane)
//   bias grads, stem / head / linear backward: small HBM-bound reductions.
#include <type_traits>

#include "common.h"

#ifndef TMDIFF_WGRAD_SPAN8
// Eighths of a box's K-steps over which the DMA pieces of the next box are issued (0 = one burst up front).  The LDS-DMA of
// dword pieces is the kernel's limiter (ablations on one box, weighted TFLOP/s: no in-loop DMA 115, a third of the x pieces
// 102, all pieces 88-95): the earlier the pieces go out the earlier they have trickled in -- 8/8: 82, 5/8: 88, 2/8: 93,
// 1/8: 95, burst: 94.
#define TMDIFF_WGRAD_SPAN8 1
#endif
#ifndef TMDIFF_WGRAD_DEBUG
#define TMDIFF_WGRAD_DEBUG 0  // experiment switches (results wrong): 1 = no in-loop DMA, 2 = no per-box barrier, 4 = no operand reads, 8 = a third of the x pieces
#endif

namespace {

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}


using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ bool tmdiff_aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float silu_grad(float t) {  // d/dt [t * sigmoid(t)]
  const float s = 1.0f / (1.0f + __expf(-t));
  return s * (1.0f + t * (1.0f - s));
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------
// wgrad.  GEMM view per (group, tap):  dW[co, ci] = sum_{b,pos} g[b,co,pos] * x'[b,ci,pos+tap]
//   MFMA rows (A) = 32 output channels : lane l holds g[co = l&31][k = l>>5]
//   MFMA cols (B) = 32 input channels  : lane l holds x'[ci = l&31] at position k = l>>5, shifted by the tap
//   K step = 2 positions.
// x' (the prologue output) is formed once per convolution by prologue_apply_kernel (it used to be re-evaluated by
// every channel tile); the wgrad kernel then only copies.  A workgroup owns a 32x32 (co, ci) tile and walks a list
// of 2x8x8 position boxes; per box the g box [32][128] and the haloed x' box [32][4*10*10] are brought into LDS
// (odd row strides: conflict-free column reads) by global_load_lds -- no registers, no ds_write -- into one of two
// stages while the MFMAs of the previous box run out of the other (a box is 64 K-steps x 7 taps = 448 MFMAs per
// wave; one barrier per box; all LDS operand reads have compile-time offsets).
// Eight waves (two per SIMD: while one waits on its operand reads or issues DMA pieces, the other's MFMAs keep the matrix
// pipe busy -- with one wave per SIMD the pipe idled a quarter of the time).  KS=3: wave w accumulates taps w%4, w%4+4, ...
// (7 accumulators, the g operand is shared by all of them) over band w/4 of the box (the box has two band planes);
// KS=1: the eight waves split the positions of the box.  Waves that share a tap add their accumulators through LDS at the
// end (fixed order), so a workgroup contributes one slot to the reduction.
// Partials go to workspace[split][g][tap][co][ci]; wgrad_reduce_kernel sums the splits into PyTorch layout.
// ---------------------------------------------------------------------------------------------------
struct WgradArgs {
  int B, N, H, W, Cin, Cout, cin_g, cout_g, groups;
  const float* xp;  // x' [B, Cin, N, H, W]
  const float* g;
  float* ws;
  float* bias_ws;          // [slot][Cout] partial sums of g over positions (bias gradient), or NULL
  int nbn, nbh, nbw;       // boxes per sample along n, h, w
  int tiles_co, tiles_ci;  // per group
  int splits, boxes_per_split;
  long total_boxes;
};

struct ApplyArgs {
  int B, Cin, nseg;
  int seg_c[3];
  const float* seg_x[3];
  const float* in_shift;
  const float* in_scale;
  int shift_stride, scale_stride;
  const float* in_mask;
  int in_act;
  float* xp;
  long plane;
  uint64_t drop_seed;      // in-kernel dropout (drop_inv > 0): keep mask from (seed, element index)
  const uint64_t* drop_seed_dev;   // ... plus this device word (HIP-graph replays: tmdiff_conv3d_desc.drop_seed_dev)
  uint32_t drop_thresh;
  float drop_inv;
};

// x'[b, c, :] = act(x[b, c, :] + shift[b,c]) * scale[b,c] * mask[b,c,:]  -- one (b, c) plane per blockIdx.y
__global__ void __launch_bounds__(256) prologue_apply_kernel(const ApplyArgs a) {
  const int bc = blockIdx.y, b = bc / a.Cin, c = bc % a.Cin;
  int cs = c, seg = 0;
  if (a.nseg > 1 && cs >= a.seg_c[0]) { cs -= a.seg_c[0]; seg = 1; }
  if (seg == 1 && a.nseg > 2 && cs >= a.seg_c[1]) { cs -= a.seg_c[1]; seg = 2; }
  const int segc = seg == 0 ? a.seg_c[0] : (seg == 1 ? a.seg_c[1] : a.seg_c[2]);
  const float* xs = (seg == 0 ? a.seg_x[0] : (seg == 1 ? a.seg_x[1] : a.seg_x[2])) + ((long)b * segc + cs) * a.plane;
  const float* msk = a.in_mask ? a.in_mask + (long)bc * a.plane : nullptr;
  const float sh = a.in_shift ? a.in_shift[(long)b * a.shift_stride + c] : 0.f;
  const float sc = a.in_scale ? a.in_scale[(long)b * a.scale_stride + c] : 1.f;
  float* dst = a.xp + (long)bc * a.plane;
  const bool drop = a.drop_inv > 0.f;
  const uint64_t dseed = a.drop_seed + (drop && a.drop_seed_dev ? *a.drop_seed_dev : 0ull);
  const uint64_t ebase = (uint64_t)bc * (uint64_t)a.plane;
  if ((a.plane & 3) == 0 && tmdiff_aligned16_dev(xs) && tmdiff_aligned16_dev(dst) && (!msk || tmdiff_aligned16_dev(msk))) {
    for (long i = (blockIdx.x * 256L + threadIdx.x) * 4; i < a.plane; i += 1024L * gridDim.x) {
      const float4 x4 = *reinterpret_cast<const float4*>(xs + i);
      float v[4] = {x4.x, x4.y, x4.z, x4.w};
      float m[4] = {1.f, 1.f, 1.f, 1.f};
      if (msk) {
        const float4 m4 = *reinterpret_cast<const float4*>(msk + i);
        m[0] = m4.x, m[1] = m4.y, m[2] = m4.z, m[3] = m4.w;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float t = v[k] + sh;
        if (a.in_act) t = tmdiff::silu_f(t);
        t *= sc;
        if (msk) t *= m[k];
        if (drop) t *= tmdiff::drop_keep(dseed, ebase + (uint64_t)(i + k), a.drop_thresh, a.drop_inv);
        v[k] = t;
      }
      *reinterpret_cast<float4*>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
    }
    return;
  }
  for (long i = blockIdx.x * 256L + threadIdx.x; i < a.plane; i += 256L * gridDim.x) {
    float v = xs[i] + sh;
    if (a.in_act) v = tmdiff::silu_f(v);
    v *= sc;
    if (msk) v *= msk[i];
    if (drop) v *= tmdiff::drop_keep(dseed, ebase + (uint64_t)i, a.drop_thresh, a.drop_inv);
    dst[i] = v;
  }
}

// One dword per lane by LDS-DMA through a buffer descriptor (as conv3d_wf.hip): base + size in scalar registers, the per-lane address
// one 32-bit byte offset; an offset at or beyond the size -- kOutsideW, or anything at all when the size is 0 (a channel row
// beyond the tensor) -- reads as zero.  A piece then costs its wave no VALU instruction (the pointer form: a compare, a
// 64-bit add and a 64-bit select per piece, ~180 per box against 224 MFMAs).
constexpr unsigned kOutsideW = 0xFFFFFFFCu;
__device__ __forceinline__ void dma_word_buf(const float* base, unsigned bytes, unsigned voff, float* dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000), dst, 4,
                                           voff, 0, 0, 0);
#endif
}

constexpr int WG_WAVES = 8;

template <int KS>
__global__ void __launch_bounds__(64 * WG_WAVES, 1) conv3d_wgrad_kernel(const WgradArgs a) {
  constexpr int TAPS = KS * KS * KS, HALO = KS / 2;
  constexpr int BN = 2, BH = 8, BW = 8, POSB = BN * BH * BW;            // 128 positions per box
  constexpr int HN = BN + 2 * HALO, HH = BH + 2 * HALO, HW = BW + 2 * HALO;
  constexpr int XE = HN * HH * HW;                                      // haloed box elements
  constexpr int XJ = (XE + 63) / 64, GJ = POSB / 64;                    // wave-wide 256-byte pieces per channel row
  constexpr int GS = GJ * 64 + 1, XS = XJ * 64 + 1;                     // odd LDS row strides holding whole pieces
  constexpr int NT = KS == 3 ? 7 : 1;                                   // accumulators per wave
  constexpr int STAGE = 32 * GS + 32 * XS;                              // g box [32][GS], then x' box [32][XS]
  constexpr int ROWS = 32 / WG_WAVES;                                   // channel rows a wave stages
  constexpr int NPIECE = ROWS * (GJ + XJ);                              // pieces per wave per box
  __shared__ float stages[2 * STAGE];   // (one array: the final reduction of the waves' partials reuses all of it)
  float* const st0 = stages;
  float* const st1 = stages + STAGE;

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, khalf = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA destinations / tap offsets stay scalar
  int id = blockIdx.x;
  const int split = id % a.splits; id /= a.splits;
  const int ci_t = id % a.tiles_ci; id /= a.tiles_ci;
  const int co_t = id % a.tiles_co;
  const int g = id / a.tiles_co;
  const int co0 = co_t * 32, ci0 = ci_t * 32;
  const long plane = (long)a.N * a.H * a.W;
  const unsigned pbytes = (unsigned)(plane * 4);

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // Bias gradient on the side: the A operand of every K-step IS g[co = l31][position], so the waves of the first ci tile
  // that load each g element exactly once (KS=3: tap group 0 of either band plane; KS=1: all eight) sum it up as they go.
  const bool do_bias = a.bias_ws != nullptr && ci_t == 0 && (KS == 1 || (wv & 3) == 0);
  float bsum = 0.f;

  const int wt = wv & 3, kh = wv >> 2;   // tap group / K half (band plane of the box) of this wave
  // ---- staging by LDS-DMA: wave w brings in channel rows ROWS*w .. ROWS*w+ROWS-1 of both boxes, one 64-element piece per
  //      instruction (the channel base is wave-uniform; the per-lane position offset is shared by the 8 rows).
  //      Out-of-image positions, layout filler and out-of-range channels read a zero word.
  unsigned goff[GJ], xoff[XJ];  // byte offset of the position inside a channel plane, or kOutsideW
  const float* gbase;      // g  [b][g*cout_g + co0]
  const float* xbase;      // x' [b][g*cin_g + ci0]
  auto locate = [&](long bx) __attribute__((always_inline)) {
    unsigned t = (unsigned)bx;
    const int bw_i = (int)(t % (unsigned)a.nbw); t /= (unsigned)a.nbw;
    const int bh_i = (int)(t % (unsigned)a.nbh); t /= (unsigned)a.nbh;
    const int bn_i = (int)(t % (unsigned)a.nbn);
    const int b = (int)(t / (unsigned)a.nbn);
    const int n0 = bn_i * BN, h0 = bh_i * BH, w0 = bw_i * BW;
#pragma unroll
    for (int j = 0; j < GJ; ++j) {
      const int p = j * 64 + lane;
      const int n = n0 + p / (BH * BW), h = h0 + (p / BW) % BH, w = w0 + p % BW;
      const bool ok = (n < a.N) & (h < a.H) & (w < a.W);
      goff[j] = ok ? (unsigned)(((n * a.H + h) * a.W + w) * 4) : kOutsideW;
    }
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int e = j * 64 + lane;
      const int n = n0 + e / (HH * HW) - HALO, h = h0 + (e / HW) % HH - HALO, w = w0 + e % HW - HALO;
      const bool ok = (e < XE) & ((unsigned)n < (unsigned)a.N) & ((unsigned)h < (unsigned)a.H) & ((unsigned)w < (unsigned)a.W);
      xoff[j] = ok ? (unsigned)(((n * a.H + h) * a.W + w) * 4) : kOutsideW;
    }
    gbase = a.g + ((long)b * a.Cout + g * a.cout_g + co0) * plane;
    xbase = a.xp + ((long)b * a.Cin + g * a.cin_g + ci0) * plane;
  };
  auto issue_piece = [&](auto ic, float* st) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < NPIECE) {
      constexpr int c = i / (GJ + XJ), j = i % (GJ + XJ);
      const int ch = wv * ROWS + c;  // wave-uniform
      if constexpr (j < GJ) {
        // (wave-uniform: the channel row's plane as the descriptor, an empty one for a row beyond the tensor)
        dma_word_buf(gbase + (long)ch * plane, co0 + ch < a.cout_g ? pbytes : 0u, goff[j], st + ch * GS + j * 64);
      } else if constexpr (!((TMDIFF_WGRAD_DEBUG & 8) && ((j - GJ) % 3 != 0))) {   // (8: only every third x piece: results wrong)
        constexpr int jj = j - GJ;
        dma_word_buf(xbase + (long)ch * plane, ci0 + ch < a.cin_g ? pbytes : 0u, xoff[jj], st + 32 * GS + ch * XS + jj * 64);
      }
    }
  };

  // KS=3: this wave's K-steps are the 32 position pairs of band plane kh of the box (BH*BW = 64 positions per plane)
  const int ga_off = l31 * GS + khalf + (KS == 3 ? kh * (BH * BW) : 0);   // A operand of K-step ks: st[ga_off + 2*ks]
  int xj_off[NT];                                   // B operand base with the (wave-uniform) tap offset folded in
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int tap = KS == 3 ? min(wt + 4 * j, TAPS - 1) : 0;
    xj_off[j] = 32 * GS + l31 * XS + khalf + ((tap / (KS * KS)) * HH + (tap / KS) % KS) * HW + tap % KS +
                (KS == 3 ? kh * (HH * HW) : 0);
  }
  // MFMAs of the box in `st`; the pieces of the next box go into `st_next`, spread evenly between the K-steps
  // (a burst of 72 DMA issues would keep the wave off the matrix pipe for a quarter of the box).
  auto mfma_box = [&](const float* st, float* st_next) __attribute__((always_inline)) {
    constexpr int KSTEPS = POSB / 2, PER_WAVE = KS == 3 ? KSTEPS / 2 : KSTEPS / WG_WAVES;
    float av[2], bv[2][NT];
    // operands of K-step k (one workgroup per CU = one wave per SIMD: nobody else hides the LDS latency, so the
    // reads of step k+1 are requested behind the first MFMA of step k)
    auto fetch = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k0 = decltype(kc)::value;
      if constexpr (KS == 3) {
        constexpr int p = 2 * k0;                    // position inside the wave's band plane
        constexpr int ph = p / BW, pw = p % BW;
        av[k0 & 1] = st[ga_off + p];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[k0 & 1][j] = st[xj_off[j] + ph * HW + pw];
      } else {
        const int p = 2 * (wv * PER_WAVE + k0);
        av[k0 & 1] = st[ga_off + p];
        bv[k0 & 1][0] = st[xj_off[0] + p];
      }
      if (do_bias) bsum += av[k0 & 1];
    };
    fetch(std::integral_constant<int, 0>{});
    if constexpr (TMDIFF_WGRAD_SPAN8 == 0)   // experiments: the whole next box requested up front
      static_for<0, NPIECE>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, st_next); });
    static_for<0, PER_WAVE>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k0 = decltype(kc)::value;
      static_for<0, NT>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        // (tap group 3 has no 7th tap: it repeats tap 26 into an accumulator that is never stored)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k0 & 1], bv[k0 & 1][j], acc[j], 0, 0, 0);
        if constexpr (j == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (k0 + 1 < PER_WAVE && !(TMDIFF_WGRAD_DEBUG & 4)) fetch(std::integral_constant<int, k0 + 1>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      // (all pieces go out in the first TMDIFF_WGRAD_SPAN8 eighths of the box, so that they have landed when the barrier comes)
      constexpr int SPAN = PER_WAVE * TMDIFF_WGRAD_SPAN8 / 8 > 0 ? PER_WAVE * TMDIFF_WGRAD_SPAN8 / 8 : 1;
      constexpr int p_lo = k0 < SPAN ? k0 * NPIECE / SPAN : NPIECE, p_hi = k0 < SPAN ? (k0 + 1) * NPIECE / SPAN : NPIECE;
      if constexpr (!(TMDIFF_WGRAD_DEBUG & 1) && TMDIFF_WGRAD_SPAN8 != 0)
        static_for<p_lo, p_hi>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, st_next); });
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  const long box_lo = (long)split * a.boxes_per_split;
  const long box_hi = min(box_lo + a.boxes_per_split, a.total_boxes);
  if (box_lo < box_hi) {
    locate(box_lo);
    static_for<0, NPIECE>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, st0); });
  }
  __syncthreads();
  for (long bx = box_lo; bx < box_hi; bx += 2) {
    // (past the last box the first box is fetched again into the idle stage: valid addresses, nobody reads it, and
    //  the MFMA stream stays free of branches)
    locate(bx + 1 < box_hi ? bx + 1 : box_lo);
    mfma_box(st0, st1);
    if constexpr (!(TMDIFF_WGRAD_DEBUG & 2)) __syncthreads();
    if (bx + 1 < box_hi) {
      locate(bx + 2 < box_hi ? bx + 2 : box_lo);
      mfma_box(st1, st0);
      if constexpr (!(TMDIFF_WGRAD_DEBUG & 2)) __syncthreads();
    }
  }
  // ---- the waves that shared a tap (KS=3: the two band planes; KS=1: all eight) add their accumulators through LDS in a
  //      fixed order, so that the workgroup writes ONE partial per (tap, co, ci) -------------------------------------
  static_assert(2 * STAGE >= (KS == 3 ? 4 : 7) * NT * 16 * 64, "stage buffers hold the partials being combined");
  __shared__ float bred[WG_WAVES][32];
  bsum += __shfl_xor(bsum, 32, 64);           // even + odd positions of the pair
  if (lane < 32) bred[wv][lane] = do_bias ? bsum : 0.f;
  __syncthreads();   // (the last box's LDS reads are done)
  if (a.bias_ws != nullptr && ci_t == 0 && wv == 0 && lane < 32 && co0 + lane < a.cout_g) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WG_WAVES; ++w) t += bred[w][lane];      // fixed order
    a.bias_ws[((long)split * a.groups + g) * a.cout_g + co0 + lane] = t;
  }
  if constexpr (KS == 3) {
    float* xch = stages + (wt * NT) * 16 * 64 + lane;
    if (kh == 1) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(j * 16 + r) * 64] = acc[j][r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] += xch[(j * 16 + r) * 64];
    }
  } else {
    if (wv > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) stages[((wv - 1) * 16 + r) * 64 + lane] = acc[0][r];
    }
    __syncthreads();
    if (wv == 0) {
      for (int w = 0; w < WG_WAVES - 1; ++w)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += stages[(w * 16 + r) * 64 + lane];
    }
  }
  if (KS == 3 ? kh != 0 : wv != 0) return;
  // ---- partial sums -> workspace[split][g][tap][co][ci]; D layout: col = l31 (ci), row = co ----------------------
  const int slot = split;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int tap = KS == 3 ? wt + 4 * j : 0;
    if (tap >= TAPS) continue;
    float* dst = a.ws + (((long)slot * a.groups + g) * TAPS + tap) * a.cout_g * a.cin_g;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * khalf, ci = ci0 + l31;
      if (co < a.cout_g && ci < a.cin_g) dst[(long)co * a.cin_g + ci] = acc[j][r];
    }
  }
}

// dw[g*cout_g + co][ci][tap] = sum_slot ws[slot][g][tap][co][ci]
// A workgroup sums 64 consecutive workspace elements: 4 thread rows take every 4th slot (8 loads in flight each),
// then the rows are combined through LDS in a fixed order (deterministic).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           int slots, int groups, int taps, int cout_g, int cin_g,
                                                           long total, const float* __restrict__ bias_ws,
                                                           float* __restrict__ dbias, float bias_scale) {
  __shared__ float part[4][64];
  if (dbias && blockIdx.x == gridDim.x - 1) {   // bias gradient: sum of the per-workgroup partials, slot order
    const int cout = groups * cout_g;
    for (int c = threadIdx.x; c < cout; c += 256) {
      float t = 0.f;
      for (int k = 0; k < slots; ++k) t += bias_ws[(long)k * cout + c];
      dbias[c] = bias_scale * t;
    }
  }
  const long per_slot = (long)groups * taps * cout_g * cin_g;
  const int col = threadIdx.x & 63, row = threadIdx.x >> 6;
  for (long base = blockIdx.x * 64L; base < total; base += 64L * gridDim.x) {
    const long i = base + col;  // workspace order [g][tap][co][ci] (coalesced reads)
    float s = 0.f;
    if (i < total) {
      int k = row;
      for (; k + 28 < slots; k += 32) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ws[(long)(k + 4 * u) * per_slot + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; k < slots; k += 4) s += ws[(long)k * per_slot + i];
    }
    part[row][col] = s;
    __syncthreads();
    if (row == 0 && i < total) {
      const float t = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
      const int ci = (int)(i % cin_g);
      long r = i / cin_g;
      const int co = (int)(r % cout_g); r /= cout_g;
      const int tap = (int)(r % taps);
      const int g = (int)(r / taps);
      dw[(((long)g * cout_g + co) * cin_g + ci) * taps + tap] = t;
    }
    __syncthreads();
  }
}

// out[c] = scale * sum_{b,p} x[b,c,p].  A channel is spread over gridDim.y slices (one workgroup per channel left
// most of the chip idle: 130 us for 33 MB); every slice adds its partial with one atomic (out is zeroed first).
__global__ void __launch_bounds__(256) zero_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

__global__ void __launch_bounds__(256) channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                                          int C, long P, float scale) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const long per = (P + gridDim.y - 1) / gridDim.y;
  const long lo = blockIdx.y * per, hi = min(lo + per, P);
  float s = 0.f;
  for (int b = 0; b < B; ++b) {
    const float* p = x + ((long)b * C + c) * P;
    long i = lo + threadIdx.x;
    for (; i + 768 < hi; i += 1024) {
      const float v0 = p[i], v1 = p[i + 256], v2 = p[i + 512], v3 = p[i + 768];
      s += (v0 + v1) + (v2 + v3);
    }
    for (; i < hi; i += 256) s += p[i];
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) atomicAdd(&out[c], scale * s);
}

struct PrologueBwdArgs {
  int B, Cin, nseg;
  int seg_c[3];
  const float* seg_x[3];
  float* dx[3];
  int accumulate[3];
  const float* add[3];     // != NULL: dx = add + (the computed gradient) -- another consumer's gradient of the same segment, read here
  const float* in_shift;
  const float* in_scale;
  int shift_stride, scale_stride;
  const float* in_mask;
  int in_act;
  const float* gp;
  float* d_shift;
  float* d_scale;
  long plane;
  uint64_t drop_seed;      // in-kernel dropout, as in ApplyArgs
  const uint64_t* drop_seed_dev;
  uint32_t drop_thresh;
  float drop_inv;
  int slices;              // workgroups per (b, c) plane; > 1: d_shift / d_scale hold [B, Cin, slices] partial sums
};

// `slices` workgroups per (b, c) plane (blockIdx.z); 16-byte accesses when the plane allows.  With slices > 1 the
// per-plane sums d_shift / d_scale are written as [B, Cin, slices] partials and finished by rowsum_kernel (fixed order).
__global__ void __launch_bounds__(256) prologue_bwd_kernel(const PrologueBwdArgs a) {
  __shared__ float red[4];
  const int c = blockIdx.x, b = blockIdx.y, sl = blockIdx.z;
  const float sh = a.in_shift ? a.in_shift[(long)b * a.shift_stride + c] : 0.f;
  const float sc = a.in_scale ? a.in_scale[(long)b * a.scale_stride + c] : 1.f;
  int cs = c, seg = 0;
  if (a.nseg > 1 && cs >= a.seg_c[0]) { cs -= a.seg_c[0]; seg = 1; }
  if (seg == 1 && a.nseg > 2 && cs >= a.seg_c[1]) { cs -= a.seg_c[1]; seg = 2; }
  const int segc = seg == 0 ? a.seg_c[0] : (seg == 1 ? a.seg_c[1] : a.seg_c[2]);
  const float* xs = (seg == 0 ? a.seg_x[0] : (seg == 1 ? a.seg_x[1] : a.seg_x[2])) + ((long)b * segc + cs) * a.plane;
  float* dxs = seg == 0 ? a.dx[0] : (seg == 1 ? a.dx[1] : a.dx[2]);
  const int accum = seg == 0 ? a.accumulate[0] : (seg == 1 ? a.accumulate[1] : a.accumulate[2]);
  const float* adds = seg == 0 ? a.add[0] : (seg == 1 ? a.add[1] : a.add[2]);
  if (dxs) dxs += ((long)b * segc + cs) * a.plane;
  if (adds) adds += ((long)b * segc + cs) * a.plane;
  const float* gp = a.gp + ((long)b * a.Cin + c) * a.plane;
  const float* msk = a.in_mask ? a.in_mask + ((long)b * a.Cin + c) * a.plane : nullptr;
  const bool drop = a.drop_inv > 0.f;
  const uint64_t dseed = a.drop_seed + (drop && a.drop_seed_dev ? *a.drop_seed_dev : 0ull);
  const uint64_t ebase = ((uint64_t)b * a.Cin + c) * (uint64_t)a.plane;
  // this slice's element range (multiples of 4 when vectorised)
  const bool vec = (a.plane & 3) == 0 && tmdiff_aligned16_dev(xs) && tmdiff_aligned16_dev(gp) &&
                   (!dxs || tmdiff_aligned16_dev(dxs)) && (!msk || tmdiff_aligned16_dev(msk)) && (!adds || tmdiff_aligned16_dev(adds));
  const long unit = vec ? 4 : 1;
  const long units = a.plane / unit;
  const long per = (units + a.slices - 1) / a.slices;
  const long lo = sl * per * unit, hi = min((sl + 1) * per, units) * unit;
  float s_sh = 0.f, s_sc = 0.f;
  auto one = [&](float x, float g, float m, long i, float& out) {
    const float t = x + sh;
    const float act = a.in_act ? tmdiff::silu_f(t) : t;
    const float dact = a.in_act ? silu_grad(t) : 1.f;
    float gm = msk ? g * m : g;
    if (drop) gm *= tmdiff::drop_keep(dseed, ebase + (uint64_t)i, a.drop_thresh, a.drop_inv);
    s_sc += gm * act;
    const float dt = gm * sc * dact;
    s_sh += dt;
    out = dt;
  };
  if (vec) {
    for (long i = lo + threadIdx.x * 4L; i < hi; i += 1024) {
      const float4 x4 = *reinterpret_cast<const float4*>(xs + i), g4 = *reinterpret_cast<const float4*>(gp + i);
      float4 m4 = make_float4(1.f, 1.f, 1.f, 1.f), d4 = make_float4(0.f, 0.f, 0.f, 0.f), o4;
      if (msk) m4 = *reinterpret_cast<const float4*>(msk + i);
      if (dxs && accum) d4 = *reinterpret_cast<const float4*>(dxs + i);
      else if (dxs && adds) d4 = *reinterpret_cast<const float4*>(adds + i);
      one(x4.x, g4.x, m4.x, i, o4.x);
      one(x4.y, g4.y, m4.y, i + 1, o4.y);
      one(x4.z, g4.z, m4.z, i + 2, o4.z);
      one(x4.w, g4.w, m4.w, i + 3, o4.w);
      if (dxs) *reinterpret_cast<float4*>(dxs + i) = make_float4(d4.x + o4.x, d4.y + o4.y, d4.z + o4.z, d4.w + o4.w);
    }
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) {
      float o;
      one(xs[i], gp[i], msk ? msk[i] : 1.f, i, o);
      if (dxs) dxs[i] = accum ? dxs[i] + o : (adds ? adds[i] + o : o);
    }
  }
  const long oidx = ((long)b * a.Cin + c) * a.slices + sl;
  if (a.d_shift) {
    const float v = block_sum_256(s_sh, red);
    if (threadIdx.x == 0) a.d_shift[oidx] = v;
  }
  if (a.d_scale) {
    const float v = block_sum_256(s_sc, red);
    if (threadIdx.x == 0) a.d_scale[oidx] = v;
  }
}

// out[r] = sum_k in[r, k] (k ascending: deterministic); finishes the sliced prologue backward
__global__ void __launch_bounds__(256) rowsum_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, int k) {
  const long r = blockIdx.x * 256L + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int j = 0; j < k; ++j) s += in[r * k + j];
  out[r] = s;
}

// stem backward: one workgroup per (co, b): partial sums over the sample's positions
__global__ void __launch_bounds__(256) stem_bwd_kernel(const float* __restrict__ xin, const float* __restrict__ pan,
                                                       const float* __restrict__ ms, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ gy,
                                                       float* __restrict__ dwb, int Cout, long P, long HW) {
  __shared__ float red[4];
  const int co = blockIdx.x, b = blockIdx.y;
  const float wc = w[co], bc = bias ? bias[co] : 0.f;
  const float* g = gy + ((long)b * Cout + co) * P;
  float sw = 0.f, sb = 0.f;
  for (long p = threadIdx.x; p < P; p += 256) {
    const float x = ms ? pan[b * HW + p % HW] - ms[b * P + p] : xin[b * P + p];
    const float u = __fadd_rn(__fmul_rn(wc, x), bc);
    const float gu = g[p] * silu_grad(u);
    sw += gu * x;
    sb += gu;
  }
  const float tw = block_sum_256(sw, red);
  const float tb = block_sum_256(sb, red);
  if (threadIdx.x == 0) {
    dwb[((long)b * Cout + co) * 2 + 0] = tw;
    dwb[((long)b * Cout + co) * 2 + 1] = tb;
  }
}

// stem backward w.r.t. the inputs: dx[b,p] = sum_co gy[b,co,p] * SiLU'(w_co x + b_co) * w_co.  One thread per (b, hw) walks
// the N bands: in the (pan, ms) form x = pan[b,hw] - ms[b,n,hw], so d_ms = -dx and d_pan[b,hw] = sum_n dx[b,n,hw].
__global__ void __launch_bounds__(256) stem_bwd_input_kernel(const float* __restrict__ xin, const float* __restrict__ pan,
                                                             const float* __restrict__ ms, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ gy,
                                                             float* __restrict__ dx, float* __restrict__ dpan, int Cout,
                                                             int N, long HW) {
  const long hw = blockIdx.x * 256L + threadIdx.x;
  const int b = blockIdx.y;
  if (hw >= HW) return;
  const long P = HW * N;
  float acc_pan = 0.f;
  for (int n = 0; n < N; ++n) {
    const long p = n * HW + hw;
    const float x = ms ? pan[b * HW + hw] - ms[b * P + p] : xin[b * P + p];
    float s = 0.f;
    for (int co = 0; co < Cout; ++co) {
      const float wc = w[co];
      const float u = __fadd_rn(__fmul_rn(wc, x), bias ? bias[co] : 0.f);
      s += gy[((long)b * Cout + co) * P + p] * silu_grad(u) * wc;
    }
    acc_pan += s;
    if (dx) dx[b * P + p] = ms ? -s : s;
  }
  if (dpan) dpan[b * HW + hw] = acc_pan;
}

// head backward: one workgroup per (c, b)
__global__ void __launch_bounds__(256) head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ gy,
                                                       float* __restrict__ dx, float* __restrict__ dws, int C, long P) {
  __shared__ float red[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const float wsv = scale ? __fmul_rn(w[c], scale[(long)b * C + c]) : w[c];
  const float* xs = x + ((long)b * C + c) * P;
  float* dxs = dx ? dx + ((long)b * C + c) * P : nullptr;
  const float* g = gy + (long)b * P;
  float s = 0.f;
  for (long p = threadIdx.x; p < P; p += 256) {
    const float v = xs[p], gp = g[p];
    s += gp * tmdiff::silu_f(v);
    if (dxs) dxs[p] = gp * wsv * silu_grad(v);
  }
  const float tot = block_sum_256(s, red);
  if (threadIdx.x == 0 && dws) dws[(long)b * C + c] = tot;
}

// linear backward.  gu[b,o] = gy[b,o] * act'(u[b,o]), u = x @ w^T + bias (recomputed when act != 0).
// kernel 1: gu (B x O), one wave per (o) like the forward.   kernel 2: dx / dw / db from gu.
__global__ void __launch_bounds__(256) linear_gu_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ gy,
                                                        float* __restrict__ gu, int B, int I, int O, int act) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= O) return;
  for (int b = 0; b < B; ++b) {
    float v = gy[(long)b * O + o];
    if (act) {
      float s = 0.f;
      for (int i = lane; i < I; i += 64) s = fmaf(w[(long)o * I + i], x[(long)b * I + i], s);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      v *= silu_grad(s + (bias ? bias[o] : 0.f));
    }
    if (lane == 0) gu[(long)b * O + o] = v;
  }
}

__global__ void __launch_bounds__(256) linear_dw_kernel(const float* __restrict__ x, const float* __restrict__ gu,
                                                        float* __restrict__ dw, float* __restrict__ db, int B, int I,
                                                        int O) {
  const long i = blockIdx.x * 256L + threadIdx.x;  // over O*I
  if (i < (long)O * I) {
    const int o = (int)(i / I), k = (int)(i % I);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s = fmaf(gu[(long)b * O + o], x[(long)b * I + k], s);
    if (dw) dw[i] = s;
  }
  if (db && i < O) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += gu[(long)b * O + i];
    db[i] = s;
  }
}

// dx[b,k] = sum_o gu[b,o] w[o,k].  Workgroup = (b, 64 consecutive k); its 4 waves take every 4th row o (coalesced
// 256-byte reads of w), partials combined through LDS in a fixed order.
__global__ void __launch_bounds__(256) linear_dx_kernel(const float* __restrict__ w, const float* __restrict__ gu,
                                                        float* __restrict__ dx, int B, int I, int O) {
  __shared__ float part[4][64];
  const int col = threadIdx.x & 63, row = threadIdx.x >> 6;
  const int ktiles = (I + 63) / 64;
  const int b = blockIdx.x / ktiles, k = (blockIdx.x % ktiles) * 64 + col;
  float s = 0.f;
  if (k < I) {
    int o = row;
    for (; o + 12 < O; o += 16) {
      const float w0 = w[(long)o * I + k], w1 = w[(long)(o + 4) * I + k], w2 = w[(long)(o + 8) * I + k],
                  w3 = w[(long)(o + 12) * I + k];
      s = fmaf(gu[(long)b * O + o], w0, s);
      s = fmaf(gu[(long)b * O + o + 4], w1, s);
      s = fmaf(gu[(long)b * O + o + 8], w2, s);
      s = fmaf(gu[(long)b * O + o + 12], w3, s);
    }
    for (; o < O; o += 4) s = fmaf(gu[(long)b * O + o], w[(long)o * I + k], s);
  }
  part[row][col] = s;
  __syncthreads();
  if (row == 0 && k < I) dx[(long)b * I + k] = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
}

struct WgradPlan {
  int taps, nbn, nbh, nbw, tiles_co, tiles_ci, splits, boxes_per_split, slots;
  long total_boxes;
  bool needs_xp;        // the prologue output (or the concatenation of the segments) is materialised in the workspace
  size_t partial_floats, bias_floats;
};

inline WgradPlan plan_wgrad(const tmdiff_conv3d_desc* d) {
  WgradPlan p;
  p.taps = d->ksize * d->ksize * d->ksize;
  p.nbn = (d->N + 1) / 2; p.nbh = (d->H + 7) / 8; p.nbw = (d->W + 7) / 8;
  p.total_boxes = (long)d->B * p.nbn * p.nbh * p.nbw;
  const int cout_g = d->Cout / d->groups, cin_g = d->Cin / d->groups;
  p.tiles_co = (cout_g + 31) / 32; p.tiles_ci = (cin_g + 31) / 32;
  const long tiles = (long)d->groups * p.tiles_co * p.tiles_ci;
  // One workgroup is resident per CU (148 KB of LDS), so the launch runs in rounds of 256 workgroups.  Choose the number
  // of K-splits that minimises  rounds x (boxes per workgroup + ~half a box of prologue / partial-sum stores);
  // ties go to the smaller split count (less to reduce).  E.g. 64 tiles x 32 boxes: 4 splits = one round of 8 boxes,
  // where the old "512 workgroups" rule gave 576 workgroups = three rounds of 4.
  long best_s = 1;
  double best_cost = 1e30;
  const long smax = p.total_boxes < 512 ? p.total_boxes : 512;
  for (long s = 1; s <= smax; ++s) {
    const long bps = (p.total_boxes + s - 1) / s;
    const long s_eff = (p.total_boxes + bps - 1) / bps;
    const long rounds = (tiles * s_eff + 255) / 256;
    const double cost = (double)rounds * ((double)bps + 0.5);
    if (cost < best_cost - 1e-9) { best_cost = cost; best_s = s_eff; }
  }
  p.boxes_per_split = (int)((p.total_boxes + best_s - 1) / best_s);
  p.splits = (int)((p.total_boxes + p.boxes_per_split - 1) / p.boxes_per_split);
  p.slots = p.splits;          // the waves of a workgroup combine their partials in LDS
  p.needs_xp = d->nseg > 1 || d->in_shift || d->in_scale || d->in_mask || d->in_act || d->drop_p > 0.f;
  p.partial_floats = ((size_t)p.slots * d->Cout * cin_g * p.taps + 3) / 4 * 4;
  p.bias_floats = ((size_t)p.slots * d->Cout + 3) / 4 * 4;
  return p;
}

}  // namespace

int tmdiff::launch_prologue_apply(const tmdiff_conv3d_desc* d, float* xp, hipStream_t st) {
  TMDIFF_REQUIRE((long)d->B * d->Cin <= 65535, "prologue_apply: B*Cin = %ld exceeds the grid", (long)d->B * d->Cin);
  const long plane = (long)d->N * d->H * d->W;
  ApplyArgs q;
  q.B = d->B; q.Cin = d->Cin; q.nseg = d->nseg;
  for (int i = 0; i < 3; ++i) { q.seg_c[i] = i < d->nseg ? d->seg_c[i] : 0; q.seg_x[i] = i < d->nseg ? d->seg_x[i] : nullptr; }
  q.in_shift = d->in_shift; q.in_scale = d->in_scale; q.in_mask = d->in_mask; q.in_act = d->in_act;
  q.drop_seed = d->drop_seed; q.drop_seed_dev = d->drop_seed_dev; q.drop_thresh = drop_threshold(d->drop_p);
  q.drop_inv = d->drop_p > 0.f ? 1.0f / (1.0f - d->drop_p) : 0.f;
  q.shift_stride = d->in_shift_stride > 0 ? d->in_shift_stride : (d->in_shift_stride < 0 ? 0 : d->Cin);
  q.scale_stride = d->in_scale_stride > 0 ? d->in_scale_stride : (d->in_scale_stride < 0 ? 0 : d->Cin);
  q.xp = xp;
  q.plane = plane;
  long pb = (plane + 1023) / 1024;
  if (pb > 64) pb = 64;
  prologue_apply_kernel<<<dim3((unsigned)pb, (unsigned)(d->B * d->Cin)), 256, 0, st>>>(q);
  return check_launch("prologue_apply");
}

extern "C" int tmdiff_conv3d_prologue_fwd(const tmdiff_conv3d_desc* d, float* xp, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d && xp, "conv3d_prologue_fwd: NULL pointer");
  TMDIFF_REQUIRE(d->B >= 0 && d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0, "conv3d_prologue_fwd: bad extents");
  TMDIFF_REQUIRE(d->nseg >= 1 && d->nseg <= 3, "conv3d_prologue_fwd: nseg=%d", d->nseg);
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    TMDIFF_REQUIRE(d->seg_x[i] && d->seg_c[i] > 0, "conv3d_prologue_fwd: segment %d is empty", i);
    csum += d->seg_c[i];
  }
  TMDIFF_REQUIRE(csum == d->Cin, "conv3d_prologue_fwd: segments hold %d channels, Cin=%d", csum, d->Cin);
  if (d->B == 0) return TMDIFF_OK;
  return launch_prologue_apply(d, xp, as_stream(stream));
}

extern "C" size_t tmdiff_conv3d_wgrad_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!d || d->B <= 0 || d->groups <= 0 || (d->ksize != 1 && d->ksize != 3)) return 0;
  const WgradPlan p = plan_wgrad(d);
  const size_t xp = p.needs_xp ? (size_t)d->B * d->Cin * d->N * d->H * d->W : 0;
  return (p.partial_floats + p.bias_floats + xp) * sizeof(float);
}

extern "C" int tmdiff_conv3d_wgrad(const tmdiff_conv3d_desc* d, const float* g, float* dw, void* workspace,
                                   tmdiff_stream_t stream) {
  return tmdiff_conv3d_wgrad_bias(d, g, dw, nullptr, workspace, stream);
}

extern "C" int tmdiff_conv3d_wgrad_bias(const tmdiff_conv3d_desc* d, const float* g, float* dw, float* dbias,
                                        void* workspace, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d && g && dw, "conv3d_wgrad: NULL pointer");
  TMDIFF_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv3d_wgrad: ksize=%d", d->ksize);
  TMDIFF_REQUIRE(d->groups == 1 || d->groups == 3, "conv3d_wgrad: groups=%d", d->groups);
  TMDIFF_REQUIRE(d->B > 0 && d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 &&
                     d->Cin % d->groups == 0 && d->Cout % d->groups == 0, "conv3d_wgrad: bad extents");
  TMDIFF_REQUIRE(d->nseg >= 1 && d->nseg <= 3, "conv3d_wgrad: nseg=%d", d->nseg);
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    TMDIFF_REQUIRE(d->seg_x[i] && d->seg_c[i] > 0, "conv3d_wgrad: segment %d is empty", i);
    csum += d->seg_c[i];
  }
  TMDIFF_REQUIRE(csum == d->Cin, "conv3d_wgrad: segments hold %d channels, Cin=%d", csum, d->Cin);
  TMDIFF_REQUIRE(workspace != nullptr, "conv3d_wgrad: NULL workspace");
  TMDIFF_REQUIRE(!(d->in_mask && d->drop_p > 0.f), "conv3d_wgrad: give either a mask tensor or drop_p, not both");
  const WgradPlan p = plan_wgrad(d);
  TMDIFF_REQUIRE(p.total_boxes < (1L << 31) && (long)d->N * d->H * d->W * 32 < (1L << 31),
                 "conv3d_wgrad: tensor too large for 32-bit box / offset arithmetic");
  hipStream_t st = as_stream(stream);
  WgradArgs a;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout;
  a.groups = d->groups; a.cin_g = d->Cin / d->groups; a.cout_g = d->Cout / d->groups;
  a.g = g; a.ws = reinterpret_cast<float*>(workspace);
  a.bias_ws = dbias ? a.ws + p.partial_floats : nullptr;
  a.xp = d->seg_x[0];
  if (p.needs_xp) {
    float* xp = a.ws + p.partial_floats + p.bias_floats;
    const int rc = launch_prologue_apply(d, xp, st);
    if (rc) return rc;
    a.xp = xp;
  }
  a.nbn = p.nbn; a.nbh = p.nbh; a.nbw = p.nbw; a.tiles_co = p.tiles_co; a.tiles_ci = p.tiles_ci;
  a.splits = p.splits; a.boxes_per_split = p.boxes_per_split; a.total_boxes = p.total_boxes;
  const long blocks = (long)d->groups * p.tiles_co * p.tiles_ci * p.splits;
  if (d->ksize == 3)
    conv3d_wgrad_kernel<3><<<(unsigned)blocks, 64 * WG_WAVES, 0, st>>>(a);
  else
    conv3d_wgrad_kernel<1><<<(unsigned)blocks, 64 * WG_WAVES, 0, st>>>(a);
  int rc = check_launch("conv3d_wgrad");
  if (rc) return rc;
  const long total = (long)d->Cout * a.cin_g * p.taps;
  long rb = (total + 63) / 64;
  if (rb > 8192) rb = 8192;
  wgrad_reduce_kernel<<<(unsigned)rb, 256, 0, st>>>(a.ws, dw, p.slots, d->groups, p.taps, a.cout_g, a.cin_g, total,
                                                    a.bias_ws, dbias, d->bias_scale);
  return check_launch("conv3d_wgrad(reduce)");
}

extern "C" int tmdiff_channel_sum(const float* x, float* out, int32_t B, int32_t C, int64_t P, float scale,
                                  tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && out && B >= 0 && C > 0 && P > 0, "channel_sum: bad arguments");
  // ~2048 workgroups in total, at least 2048 elements per slice and sample
  long slices = (2048 + C - 1) / C;
  if (slices > (P + 2047) / 2048) slices = (P + 2047) / 2048;
  if (slices < 1) slices = 1;
  zero_kernel<<<(C + 255) / 256, 256, 0, as_stream(stream)>>>(out, C);
  channel_sum_kernel<<<dim3(C, (unsigned)slices), 256, 0, as_stream(stream)>>>(x, out, B, C, P, scale);
  return check_launch("channel_sum");
}

namespace {
// workgroups per (b, c) plane of the prologue backward: enough to fill the chip, at least 2048 elements each
int prologue_bwd_slices(const tmdiff_conv3d_desc* d) {
  const long planes = (long)d->B * d->Cin, plane = (long)d->N * d->H * d->W;
  long s = (1024 + planes - 1) / planes;
  if (s > plane / 2048) s = plane / 2048;
  if (s > 64) s = 64;
  return s < 1 ? 1 : (int)s;
}
}  // namespace

extern "C" size_t tmdiff_conv3d_prologue_bwd_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!d || d->B <= 0 || d->Cin <= 0) return 0;
  const int s = prologue_bwd_slices(d);
  return s > 1 ? (size_t)2 * d->B * d->Cin * s * sizeof(float) : 0;
}

namespace {
int prologue_bwd_impl(const tmdiff_conv3d_desc* d, const float* gp, float* const dx_seg[3], const int32_t accumulate[3],
                      const float* const add_seg[3], float* d_shift, float* d_scale, void* workspace, tmdiff_stream_t stream);
}

extern "C" int tmdiff_conv3d_prologue_bwd_ws(const tmdiff_conv3d_desc* d, const float* gp, float* const dx_seg[3],
                                             const int32_t accumulate[3], float* d_shift, float* d_scale, void* workspace,
                                             tmdiff_stream_t stream) {
  return prologue_bwd_impl(d, gp, dx_seg, accumulate, nullptr, d_shift, d_scale, workspace, stream);
}

/* Three-operand form: dx_seg[i] = add_seg[i] + dL/dx_i (add_seg[i] == NULL: just dL/dx_i) -- the gradient another consumer of the
 * same segment has produced (a ResBlock's identity residual, Hyper_unet_general.py:248) is read here instead of being summed with
 * this one by a launch of its own; add_seg[i] is only read (it may be shared), dx_seg[i] is a tensor of its own. */
extern "C" int tmdiff_conv3d_prologue_bwd_add(const tmdiff_conv3d_desc* d, const float* gp, float* const dx_seg[3],
                                              const float* const add_seg[3], float* d_shift, float* d_scale, void* workspace,
                                              tmdiff_stream_t stream) {
  static const int32_t none[3] = {0, 0, 0};
  TMDIFF_REQUIRE(add_seg != nullptr, "prologue_bwd_add: NULL pointer");
  return prologue_bwd_impl(d, gp, dx_seg, none, add_seg, d_shift, d_scale, workspace, stream);
}

namespace {
int prologue_bwd_impl(const tmdiff_conv3d_desc* d, const float* gp, float* const dx_seg[3], const int32_t accumulate[3],
                      const float* const add_seg[3], float* d_shift, float* d_scale, void* workspace, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d && gp && dx_seg && accumulate, "prologue_bwd: NULL pointer");
  TMDIFF_REQUIRE(d->B > 0 && d->Cin > 0 && d->B <= 65535 && d->nseg >= 1 && d->nseg <= 3, "prologue_bwd: bad extents");
  TMDIFF_REQUIRE(!(d->in_mask && d->drop_p > 0.f), "prologue_bwd: give either a mask tensor or drop_p, not both");
  PrologueBwdArgs a;
  a.B = d->B; a.Cin = d->Cin; a.nseg = d->nseg;
  int csum = 0;
  for (int i = 0; i < 3; ++i) {
    a.seg_c[i] = i < d->nseg ? d->seg_c[i] : 0;
    a.seg_x[i] = i < d->nseg ? d->seg_x[i] : nullptr;
    a.dx[i] = i < d->nseg ? dx_seg[i] : nullptr;
    a.accumulate[i] = i < d->nseg ? accumulate[i] : 0;
    a.add[i] = (add_seg && i < d->nseg) ? add_seg[i] : nullptr;
    if (i < d->nseg) {
      TMDIFF_REQUIRE(d->seg_x[i] != nullptr, "prologue_bwd: segment %d is NULL", i);
      csum += d->seg_c[i];
    }
  }
  TMDIFF_REQUIRE(csum == d->Cin, "prologue_bwd: segments hold %d channels, Cin=%d", csum, d->Cin);
  a.in_shift = d->in_shift; a.in_scale = d->in_scale; a.in_mask = d->in_mask; a.in_act = d->in_act;
  a.shift_stride = d->in_shift_stride > 0 ? d->in_shift_stride : (d->in_shift_stride < 0 ? 0 : d->Cin);
  a.scale_stride = d->in_scale_stride > 0 ? d->in_scale_stride : (d->in_scale_stride < 0 ? 0 : d->Cin);
  a.drop_seed = d->drop_seed; a.drop_seed_dev = d->drop_seed_dev; a.drop_thresh = drop_threshold(d->drop_p);
  a.drop_inv = d->drop_p > 0.f ? 1.0f / (1.0f - d->drop_p) : 0.f;
  a.gp = gp; a.d_shift = d_shift; a.d_scale = d_scale;
  a.plane = (long)d->N * d->H * d->W;
  a.slices = workspace ? prologue_bwd_slices(d) : 1;
  hipStream_t st = as_stream(stream);
  const long planes = (long)d->B * d->Cin;
  if (a.slices > 1) {   // per-slice partial sums go to the workspace, rowsum_kernel finishes them
    float* w = static_cast<float*>(workspace);
    if (d_shift) a.d_shift = w;
    if (d_scale) a.d_scale = w + planes * a.slices;
  }
  prologue_bwd_kernel<<<dim3(d->Cin, d->B, a.slices), 256, 0, st>>>(a);
  int rc = check_launch("conv3d_prologue_bwd");
  if (rc || a.slices == 1) return rc;
  if (d_shift) rowsum_kernel<<<(unsigned)((planes + 255) / 256), 256, 0, st>>>(a.d_shift, d_shift, planes, a.slices);
  if (d_scale) rowsum_kernel<<<(unsigned)((planes + 255) / 256), 256, 0, st>>>(a.d_scale, d_scale, planes, a.slices);
  return check_launch("conv3d_prologue_bwd(rowsum)");
}
}  // namespace

extern "C" int tmdiff_conv3d_prologue_bwd(const tmdiff_conv3d_desc* d, const float* gp, float* const dx_seg[3],
                                          const int32_t accumulate[3], float* d_shift, float* d_scale,
                                          tmdiff_stream_t stream) {
  return tmdiff_conv3d_prologue_bwd_ws(d, gp, dx_seg, accumulate, d_shift, d_scale, nullptr, stream);
}

extern "C" int tmdiff_stem_bwd(const float* xin, const float* pan, const float* ms, const float* w, const float* bias,
                               const float* gy, float* dwb, int32_t B, int32_t Cout, int32_t N, int32_t H, int32_t W,
                               tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(w && gy && dwb, "stem_bwd: NULL pointer");
  TMDIFF_REQUIRE((ms && pan) || (!ms && xin), "stem_bwd: give either (pan, ms) or xin");
  TMDIFF_REQUIRE(B > 0 && B <= 65535 && Cout > 0 && N > 0 && H > 0 && W > 0, "stem_bwd: bad extents");
  const long HW = (long)H * W;
  stem_bwd_kernel<<<dim3(Cout, B), 256, 0, as_stream(stream)>>>(xin, pan, ms, w, bias, gy, dwb, Cout, HW * N, HW);
  return check_launch("stem_bwd");
}

extern "C" int tmdiff_stem_bwd_input(const float* xin, const float* pan, const float* ms, const float* w,
                                     const float* bias, const float* gy, float* dx, float* dpan, int32_t B, int32_t Cout,
                                     int32_t N, int32_t H, int32_t W, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(w && gy && (dx || dpan), "stem_bwd_input: NULL pointer");
  TMDIFF_REQUIRE((ms && pan) || (!ms && xin), "stem_bwd_input: give either (pan, ms) or xin");
  TMDIFF_REQUIRE(ms || !dpan, "stem_bwd_input: d_pan only exists in the (pan, ms) form");
  TMDIFF_REQUIRE(B > 0 && B <= 65535 && Cout > 0 && N > 0 && H > 0 && W > 0, "stem_bwd_input: bad extents");
  const long HW = (long)H * W;
  stem_bwd_input_kernel<<<dim3((unsigned)((HW + 255) / 256), B), 256, 0, as_stream(stream)>>>(xin, pan, ms, w, bias, gy, dx,
                                                                                            dpan, Cout, N, HW);
  return check_launch("stem_bwd_input");
}

extern "C" int tmdiff_head_bwd(const float* x, const float* w, const float* scale, const float* gy, float* dx,
                               float* dws, int32_t B, int32_t C, int64_t P, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && w && gy, "head_bwd: NULL pointer");
  TMDIFF_REQUIRE(B > 0 && B <= 65535 && C > 0 && P > 0, "head_bwd: bad extents");
  head_bwd_kernel<<<dim3(C, B), 256, 0, as_stream(stream)>>>(x, w, scale, gy, dx, dws, C, P);
  return check_launch("head_bwd");
}

extern "C" int tmdiff_linear_bwd(const float* x, const float* w, const float* bias, const float* gy, float* gu_scratch,
                                 float* dx, float* dw, float* db, int32_t B, int32_t I, int32_t O, int32_t act,
                                 tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(x && w && gy, "linear_bwd: NULL pointer");
  TMDIFF_REQUIRE(B > 0 && I > 0 && O > 0, "linear_bwd: bad extents");
  TMDIFF_REQUIRE(!act || gu_scratch, "linear_bwd: act != 0 needs a [B, O] scratch buffer");
  hipStream_t st = as_stream(stream);
  const float* gu = gy;
  if (act) {  // gu = gy * SiLU'(x @ w^T + bias), pre-activation recomputed
    linear_gu_kernel<<<(O + 3) / 4, 256, 0, st>>>(x, w, bias, gy, gu_scratch, B, I, O, act);
    int rc = check_launch("linear_bwd(gu)");
    if (rc) return rc;
    gu = gu_scratch;
  }
  if (dw || db) {
    const long n = (long)O * I;
    linear_dw_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(x, gu, dw, db, B, I, O);
    int rc = check_launch("linear_bwd(dw)");
    if (rc) return rc;
  }
  if (dx) {
    linear_dx_kernel<<<(unsigned)(B * ((I + 63) / 64)), 256, 0, st>>>(w, gu, dx, B, I, O);
    return check_launch("linear_bwd(dx)");
  }
  return TMDIFF_OK;
}
