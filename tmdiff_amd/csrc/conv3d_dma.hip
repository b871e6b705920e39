// conv3d 3x3x3 on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32), "staged" variant.
//
// Same arithmetic, tiling and LDS operand layout as conv3d_mfma_kernel (conv3d.hip); what differs is how a chunk
// of input channels reaches LDS.  There, a workgroup loads it into registers, applies the prologue and writes it
// to LDS in the gaps between MFMAs -- some 60 work items per chunk that cost the MFMA stream ~15 % (its ablation:
// 95 % of the MFMA peak with the loads and the hand-off removed, 83 % with them).  Here
//   1. the prologue output x' = act(x + shift) * scale * mask (and the concatenation of the segments) is formed once
//      per convolution by an elementwise pass (prologue_apply_kernel; skipped when the input is one plain tensor,
//      as for every data-gradient convolution), and
//   2. both operands travel L2/HBM -> LDS with global_load_lds (dword pieces for the haloed input box, 16-byte
//      pieces for the weight slab): no registers, no VALU work, no ds_write.  Out-of-image positions read a zero
//      word in device memory.
// Two LDS stages, one barrier per chunk of KC input channels; the wave's instruction stream is MFMAs, their
// operand ds_reads (compile-time offsets) and ~16 DMA issues per chunk.
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

struct DmaArgs {
  int B, N, H, W;
  int Cin, Cout, cin_g, cout_g, groups;
  const float* xq;  // x' [B, Cin, N, H, W]
  const float* wp;  // packed [g][ci][tap][co] (tmdiff_conv3d_pack_weights)
  const float* bias;
  float bias_scale;
  const float* residual;
  float out_scale;
  float* y;
  float* y2;  // optional second output act2(y + shift2) * scale2 (the consumer's prologue), see the header
  const float* y2_shift;
  const float* y2_scale;
  int y2_shift_stride, y2_scale_stride, y2_act;
  int tiles_n, tiles_h, tiles_w, tiles_co;
  unsigned total_blocks;
  int ksplit, split_chunks;  // split-K: ksplit ranges of split_chunks chunks each (1, cin_g / KC = no split)
  float* part;               // split-K partial outputs [ksplit][B][Cout][plane] (NULL = no split)
  int vec4;                  // W % 4 == 0 and y / y2 / residual 16-byte aligned: dwordx4 epilogue (epilogue.h)
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

template <int KS, int NS, int MSUB, int KC, int TN, int TH, int TW>
struct Geo {
  static constexpr int TAPS = KS * KS * KS, HALO = KS / 2, CO = 32 * MSUB;
  static constexpr int HN = TN + 2 * HALO, HH = TH + 2 * HALO, HW = TW + 2 * HALO;
  static constexpr int TILE_ELEMS = HN * HH * HW;
  static constexpr int LDS_IN = KC * TILE_ELEMS;            // input box, [KC][TILE_ELEMS]
  static constexpr int XP = (LDS_IN + 63) / 64;             // dword pieces (64 floats each)
  static constexpr int X_FLOATS = XP * 64;
  static constexpr int W_UNITS = KC * TAPS * CO / 4;        // weight slab [KC][TAPS][CO] in 16-byte units
  static constexpr int WP = (W_UNITS + 63) / 64;            // 16-byte pieces (64 units each)
  static constexpr int STAGE = X_FLOATS + WP * 256;         // floats per pipeline stage
  static_assert(TN * TH * TW == 4 * NS * 32, "workgroup tile = 4 waves x NS sub-tiles x 32 positions");
  static_assert(KC % 2 == 0, "K step is 2 channels");
};

template <int KS, int NS, int MSUB, int KC, int TN, int TH, int TW>
__global__ void __launch_bounds__(256, 2) conv3d_dma_kernel(const DmaArgs a) {
  using G = Geo<KS, NS, MSUB, KC, TN, TH, TW>;
  constexpr int CO = G::CO;
  constexpr int XK = (G::XP + 3) / 4, WK = (G::WP + 3) / 4;  // pieces per wave
  __shared__ __attribute__((aligned(16))) float st0[G::STAGE];
  __shared__ __attribute__((aligned(16))) float st1[G::STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: DMA destinations stay scalar
  const int l31 = lane & 31, khalf = lane >> 5;

  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int tw_i = __builtin_amdgcn_readfirstlane(id % a.tiles_w); id /= a.tiles_w;
  const int th_i = __builtin_amdgcn_readfirstlane(id % a.tiles_h); id /= a.tiles_h;
  const int tn_i = __builtin_amdgcn_readfirstlane(id % a.tiles_n); id /= a.tiles_n;
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups); id /= a.groups;
  const int b = __builtin_amdgcn_readfirstlane(id % a.B);
  const int split = __builtin_amdgcn_readfirstlane(id / a.B);   // split-K range of this workgroup (outermost index)
  const int n0 = tn_i * TN, h0 = th_i * TH, w0 = tw_i * TW;
  const int co0 = co_tile * CO;
  const long plane = (long)a.N * a.H * a.W;
  const int nchunks = a.split_chunks;                           // chunks of this workgroup: [split * nchunks, +nchunks)

  // ---- DMA sources of this lane (the same for every chunk) -----------------------------------------------------
  // x piece q = wv + 4k covers stage floats q*64 + lane = element (channel kc, box position e) of [KC][TILE_ELEMS]
  int xsrc[XK];  // float offset from the chunk base (< 2^31: checked by the entry point), or -1 = zero word
#pragma unroll
  for (int k = 0; k < XK; ++k) {
    const int f = (wv + 4 * k) * 64 + lane;
    const int kc = f / G::TILE_ELEMS, e = f % G::TILE_ELEMS;
    const int wz = e % G::HW, hz = (e / G::HW) % G::HH, nz = e / (G::HW * G::HH);
    const int n = n0 + nz - G::HALO, h = h0 + hz - G::HALO, w = w0 + wz - G::HALO;
    const bool ok = f < G::LDS_IN && n >= 0 && n < a.N && h >= 0 && h < a.H && w >= 0 && w < a.W;
    xsrc[k] = ok ? kc * (int)plane + (n * a.H + h) * a.W + w : -1;
  }
  // weight piece q covers slab units q*64 + lane: unit u = row (kc*TAPS + tap) * (CO/4) + c4
  int wsrc[WK];  // float offset inside the chunk's rows, or -1
#pragma unroll
  for (int k = 0; k < WK; ++k) {
    const int u = (wv + 4 * k) * 64 + lane;
    wsrc[k] = u < G::W_UNITS ? (u / (CO / 4)) * a.cout_g + (u % (CO / 4)) * 4 : -1;
  }
  const long c_first = (long)split * nchunks * KC;              // first input channel (within the group) of the range
  const float* xg = a.xq + ((long)b * a.Cin + (long)g * a.cin_g + c_first) * plane;
  const float* wg = a.wp + ((long)g * a.cin_g + c_first) * G::TAPS * a.cout_g + co0;
  const float* zero = reinterpret_cast<const float*>(&kZero4);

  // piece i of this wave for chunk c: i < XK = input pieces (dwords), then the weight pieces (16 bytes)
  constexpr int NPIECE = XK + WK;
  auto issue_piece = [&](auto ic, int c, float* st) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < XK) {
      constexpr int k = i;
      const int q = wv + 4 * k;
      if (G::XP % 4 == 0 || q < G::XP) dma_b32(xsrc[k] >= 0 ? xg + (long)c * KC * plane + xsrc[k] : zero, st + q * 64);
    } else if constexpr (i < NPIECE) {
      constexpr int k = i - XK;
      const int q = wv + 4 * k;
      if (G::WP % 4 == 0 || q < G::WP)
        dma_b128(wsrc[k] >= 0 ? wg + (long)c * KC * G::TAPS * a.cout_g + wsrc[k] : zero, st + G::X_FLOATS + q * 256);
    }
  };

  // ---- per-lane operand offsets (floats inside a stage), as in conv3d_mfma_kernel -------------------------------
  int boff[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p = (wv * NS + s) * 32 + l31;
    const int pw = p % TW, ph = (p / TW) % TH, pn = p / (TW * TH);
    boff[s] = (pn * G::HH + ph) * G::HW + pw + khalf * G::TILE_ELEMS;
  }
  const int aoff = G::X_FLOATS + khalf * G::TAPS * CO + l31 * MSUB;  // slab rows hold the tile's channels as [l31][m]

  float bias_v[MSUB];
#pragma unroll
  for (int m = 0; m < MSUB; ++m) bias_v[m] = a.bias ? a.bias[g * a.cout_g + co0 + m * 32 + l31] * a.bias_scale : 0.f;
  float sh2_v[MSUB], sc2_v[MSUB];
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int col = g * a.cout_g + co0 + m * 32 + l31;
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
  constexpr int KSTEPS = (KC / 2) * G::TAPS;
  // While chunk c is multiplied, the pieces of chunk c_next are issued one at a time behind MFMAs, spread evenly over
  // the K-steps (a burst at the start of the chunk keeps the wave away from the matrix pipe for its whole issue time).
  constexpr int PSTRIDE = KSTEPS / NPIECE > 0 ? KSTEPS / NPIECE : 1;
  static_assert(NPIECE <= KSTEPS, "at most one piece per K-step");
  auto mfma_chunk = [&](const float* st, int c_next, float* st_next) __attribute__((always_inline)) {
    float av[2][MSUB], bv[2][NS];
    auto fetch = [&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int kp = ks / G::TAPS, tap = ks % G::TAPS;
      constexpr int dn = tap / (KS * KS), dh = (tap / KS) % KS, dw = tap % KS;
      constexpr int toff = (dn * G::HH + dh) * G::HW + dw;
      const float* ap = st + aoff + (kp * 2 * G::TAPS + tap) * CO;
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
    // the operands of K-step ks+1 are requested right behind the first MFMA of K-step ks (pinned)
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

  static_for<0, NPIECE>([&](auto ic) __attribute__((always_inline)) { issue_piece(ic, 0, st0); });
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    // (past the last chunk the pieces of chunk 0 are fetched again into the idle stage: valid addresses, nobody
    //  reads them, and the MFMA stream stays free of branches)
    mfma_chunk(st0, c + 1 < nchunks ? c + 1 : 0, st1);
    __syncthreads();
    if (c + 1 < nchunks) {
      mfma_chunk(st1, c + 2 < nchunks ? c + 2 : 0, st0);
      __syncthreads();
    }
  }

  if (a.part) {  // split-K: raw partial sums; splitk_reduce_kernel (conv3d.hip) adds them up and applies the epilogue
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int p = (wv * NS + s) * 32 + l31;
        const int n = n0 + p / (TW * TH), h = h0 + (p / TW) % TH, w = w0 + p % TW;
        const bool pok = n < a.N && h < a.H && w < a.W;
        const long sp = pok ? ((long)n * a.H + h) * a.W + w : 0;
        float* dst = a.part + (((long)split * a.B + b) * a.Cout + g * a.cout_g + co0 + m * 32 + 4 * khalf) * plane + sp;
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
  // ---- scalar epilogue: bias, residual, scale; D layout: col = lane&31 (position), row = channel ---------------
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2);
      const float b0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bias_v[m]), row));
      const float b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bias_v[m]), row + 4));
      bias_r[r] = khalf ? b1 : b0;
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int p = (wv * NS + s) * 32 + l31;
      const int n = n0 + p / (TW * TH), h = h0 + (p / TW) % TH, w = w0 + p % TW;
      const bool pok = n < a.N && h < a.H && w < a.W;
      const long sp = pok ? ((long)n * a.H + h) * a.W + w : 0;
      const long obase = ((long)b * a.Cout + g * a.cout_g + co0 + m * 32 + 4 * khalf) * plane + sp;
      float res[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        res[r] = (a.residual && pok) ? a.residual[obase + row * plane] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const float v = (acc[s][m][r] + bias_r[r] + res[r]) * a.out_scale;
        if (pok && a.y) a.y[obase + row * plane] = v;
        acc[s][m][r] = v;
      }
      if (a.y2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2);
          const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sh2_v[m]), row));
          const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sh2_v[m]), row + 4));
          const float c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc2_v[m]), row));
          const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc2_v[m]), row + 4));
          float t = acc[s][m][r] + (khalf ? s1 : s0);
          const float ta = tmdiff::silu_f(t);
          t = (a.y2_act ? ta : t) * (khalf ? c1 : c0);
          if (pok) a.y2[obase + row * plane] = t;
        }
      }
    }
  }
}

template <int KS, int NS, int MSUB, int KC, int TN, int TH, int TW>
int launch(DmaArgs& a, hipStream_t st) {
  constexpr int CO = 32 * MSUB;
  a.tiles_n = (a.N + TN - 1) / TN;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  a.tiles_co = a.cout_g / CO;
  const long blocks = (long)a.ksplit * a.B * a.groups * a.tiles_n * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d_fwd_staged: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  conv3d_dma_kernel<KS, NS, MSUB, KC, TN, TH, TW><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_fwd_staged");
}

bool needs_apply(const tmdiff_conv3d_desc* d) {
  return d->nseg > 1 || d->in_shift || d->in_scale || d->in_mask || d->in_act || d->drop_p > 0.f;
}

// shapes the staged kernel takes (every production layer); others stay on the fused kernel
bool staged_ok(const tmdiff_conv3d_desc* d) {
  if (!d || d->ksize != 3 || (d->groups != 1 && d->groups != 3)) return false;  // 1x1x1 has its own kernel (conv1.hip)
  if (d->Cin <= 0 || d->Cout <= 0 || d->Cin % d->groups || d->Cout % d->groups) return false;
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  return cin_g % 4 == 0 && cout_g % 32 == 0;
}

}  // namespace

extern "C" int tmdiff_conv3d_fwd_staged_supported(const tmdiff_conv3d_desc* d) { return staged_ok(d) ? 1 : 0; }

extern "C" size_t tmdiff_conv3d_fwd_staged_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!staged_ok(d) || d->B <= 0 || !needs_apply(d)) return 0;
  return (size_t)d->B * d->Cin * d->N * d->H * d->W * sizeof(float);
}

extern "C" int tmdiff_conv3d_fwd_staged(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d != nullptr, "conv3d_fwd_staged: NULL descriptor");
  if (!staged_ok(d)) return fail(TMDIFF_E_UNSUPPORTED, "conv3d_fwd_staged: shape not supported (use tmdiff_conv3d_fwd)");
  TMDIFF_REQUIRE(d->B >= 0 && d->N > 0 && d->H > 0 && d->W > 0, "conv3d_fwd_staged: bad extents");
  TMDIFF_REQUIRE(d->nseg >= 1 && d->nseg <= 3, "conv3d_fwd_staged: nseg=%d", d->nseg);
  if (d->B == 0) return TMDIFF_OK;
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    TMDIFF_REQUIRE(d->seg_x[i] != nullptr && d->seg_c[i] > 0, "conv3d_fwd_staged: segment %d is empty", i);
    csum += d->seg_c[i];
  }
  TMDIFF_REQUIRE(csum == d->Cin, "conv3d_fwd_staged: segments hold %d channels, Cin=%d", csum, d->Cin);
  TMDIFF_REQUIRE(d->w_packed && (d->y || d->y2) && aligned16(d->w_packed), "conv3d_fwd_staged: NULL / unaligned weights or output");
  TMDIFF_REQUIRE(!d->y2 || !d->y2_bf16, "conv3d_fwd_staged: a bf16-packed second output needs tmdiff_conv3d_fwd_bf16");
  TMDIFF_REQUIRE(!(d->in_mask && d->drop_p > 0.f) && d->drop_p >= 0.f && d->drop_p < 1.f,
                 "conv3d_fwd_staged: give either a mask tensor or 0 <= drop_p < 1");
  TMDIFF_REQUIRE((long)d->N * d->H * d->W * 8 < (1L << 31), "conv3d_fwd_staged: plane too large");
  hipStream_t st = as_stream(stream);

  DmaArgs a;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cout = d->Cout; a.groups = d->groups;
  a.cin_g = d->Cin / d->groups; a.cout_g = d->Cout / d->groups;
  a.xq = d->seg_x[0];
  if (needs_apply(d)) {
    TMDIFF_REQUIRE(workspace != nullptr && aligned16(workspace), "conv3d_fwd_staged: this convolution needs its workspace");
    float* xp = static_cast<float*>(workspace);
    const int rc = launch_prologue_apply(d, xp, st);
    if (rc) return rc;
    a.xq = xp;
  }
  a.wp = d->w_packed; a.bias = d->bias; a.bias_scale = d->bias_scale;
  a.residual = d->residual; a.out_scale = d->out_scale; a.y = d->y;
  a.y2 = d->y2; a.y2_shift = d->y2_shift; a.y2_scale = d->y2_scale; a.y2_act = d->y2_act;
  a.y2_shift_stride = d->y2_shift_stride > 0 ? d->y2_shift_stride : (d->y2_shift_stride < 0 ? 0 : d->Cout);
  a.y2_scale_stride = d->y2_scale_stride > 0 ? d->y2_scale_stride : (d->y2_scale_stride < 0 ? 0 : d->Cout);
  a.vec4 = tmdiff::epilogue_vec_ok(d);

  // tile configuration and split-K factor: plan_conv3 (conv3d.hip), the same rule as tmdiff_conv3d_fwd
  Conv3Plan plan = plan_conv3(d);
  a.ksplit = 1; a.split_chunks = a.cin_g / 4; a.part = nullptr;
  const size_t need = (size_t)plan.ksplit * d->B * d->Cout * d->N * d->H * d->W * sizeof(float);
  if (plan.ksplit > 1 && d->splitk_ws && (size_t)d->splitk_ws_bytes >= need && aligned16(d->splitk_ws)) {
    a.ksplit = plan.ksplit; a.split_chunks = a.cin_g / 4 / plan.ksplit; a.part = static_cast<float*>(d->splitk_ws);
  }
  int rc;
  switch (plan.tile) {
    case 0: rc = launch<3, 1, 2, 4, 2, 8, 8>(a, st); break;
    case 1: rc = launch<3, 2, 2, 4, 4, 8, 8>(a, st); break;
    case 2: rc = launch<3, 4, 1, 4, 4, 8, 16>(a, st); break;
    default: rc = launch<3, 2, 1, 4, 4, 8, 8>(a, st); break;
  }
  if (rc || !a.part) return rc;
  SplitKReduceArgs r{a.part, a.ksplit, d->B, d->Cout, (long)d->N * d->H * d->W, d->bias, d->bias_scale, d->residual,
                     d->out_scale, d->y, d->y2, d->y2_shift, d->y2_scale, a.y2_shift_stride, a.y2_scale_stride, d->y2_act};
  return launch_splitk_reduce(r, st);
}
