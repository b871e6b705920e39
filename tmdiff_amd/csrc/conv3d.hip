// conv3d 3x3x3 / 1x1x1 as an implicit GEMM on the fp32 matrix cores of gfx950
// (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD == the 157 TFLOP/s fp32 peak).
//
// Replaces nn.Conv3d / F.conv3d / modulated_conv3d of GeneralModel/Hyper_unet_general.py
// (:51-77, :161-164, :224-231, :260, :344-361) plus the elementwise ops around them; see
// include/tmdiff_hip.h for the fused prologue / epilogue contract.
//
// GEMM view per (batch b, group g):   D[co, pos] = sum_{ci,tap} Wp[ci,tap,co] * X'[ci, pos+tap]
//   MFMA rows  (A operand) = 32 output channels   -> lane l holds Wp[.., co = l&31] for k = l>>5
//   MFMA cols  (B operand) = 32 output positions  -> lane l holds X'[k = l>>5][pos = l&31]
//   K step = 2 input channels at one tap.
// A workgroup (256 threads = 4 waves) owns a TN x TH x TW box of output positions (256 of
// them) and CO output channels.  Per chunk of KC input channels it stages in LDS
//   - the haloed input box [KC][TN+2][TH+2][TW+2] with the prologue applied once per element
//     (zero padding stays exactly zero), and
//   - the weight slab [KC][taps][CO] (a contiguous row range of the packed weights),
// then every wave issues taps*KC/2 K-steps over its 2 position sub-tiles x CO/32 channel
// sub-tiles.  All LDS operand reads are ds_read_b32 with compile-time offsets; the fp32 MFMA
// needs only two operand dwords per 64 cycles, so LDS bandwidth is not the limiter -- the
// MFMA pipe is, and several workgroups per CU overlap one's staging with another's MFMAs.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "epilogue.h"

#ifndef TMDIFF_CONV_DEBUG
#define TMDIFF_CONV_DEBUG 0  // experiment switches: 1 = no in-loop global loads, 2 = no hand-off items (results wrong)
#endif

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E) -- indices must be constants so that
// the register arrays below are addressed statically (a runtime index would send them to scratch).
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

struct ConvArgs {
  int B, N, H, W;
  int Cin, Cout;          // totals
  int cin_g, cout_g;      // per group
  int groups;
  int nseg;
  int seg_c[3];
  const float* seg_x[3];
  const float* wp;        // packed [g][ci][tap][co]
  const float* bias;
  float bias_scale;
  const float* in_shift;
  const float* in_scale;
  int shift_stride, scale_stride;
  const float* in_mask;
  int in_act;
  const float* residual;
  float out_scale;
  float* y;
  float* y2;              // optional second output act2(y + shift2) * scale2 (the consumer's prologue), see the header
  const float* y2_shift;
  const float* y2_scale;
  int y2_shift_stride, y2_scale_stride, y2_act;
  int tiles_n, tiles_h, tiles_w, tiles_co;  // tiles_co per group
  int w_vec4;                               // cout_g % 4 == 0 -> 16-byte weight loads
  unsigned total_blocks;
  uint64_t drop_seed;                       // in-kernel dropout (drop_inv > 0; MASK instantiations): common.h drop_keep
  const uint64_t* drop_seed_dev;            // ... plus this device word (tmdiff_conv3d_desc.drop_seed_dev), or NULL
  uint32_t drop_thresh;
  float drop_inv;
  int ksplit, split_ch;                     // split-K: ksplit ranges of split_ch input channels (1, cin_g = no split)
  float* part;                              // split-K partial outputs [ksplit][B][Cout][plane] (NULL = no split)
  int vec4;                                 // W % 4 == 0, y / y2 / residual 16-byte aligned: dwordx4 epilogue (epilogue.h)
};

// Tile geometry.  A workgroup = 4 waves; wave w owns NS position sub-tiles (32 positions each) x MSUB
// channel sub-tiles (32 channels each): NS*MSUB fp32 32x32 accumulators (16 AGPRs each).
template <int KS, int NS, int MSUB, int KC, int TN, int TH, int TW>
struct Geo {
  static constexpr int TAPS = KS * KS * KS;
  static constexpr int HALO = KS / 2;
  static constexpr int CO = 32 * MSUB;
  static constexpr int HN = TN + 2 * HALO, HH = TH + 2 * HALO, HW = TW + 2 * HALO;
  static constexpr int TILE_ELEMS = HN * HH * HW;
  static constexpr int POS = TN * TH * TW;
  static constexpr int EPT = (TILE_ELEMS + 255) / 256;         // staged input elements per thread per channel
  static constexpr int W4 = (KC * TAPS * CO / 4 + 255) / 256;  // staged weight float4s per thread per chunk
  static constexpr int LDS_IN = KC * TILE_ELEMS;
  static constexpr int LDS_W = KC * TAPS * CO;
  static constexpr int DUMMY = (LDS_IN + LDS_W + 3) / 4 * 4;   // 16-B sink for lanes that have nothing to stage
  static constexpr int STAGE = DUMMY + 4;                      // floats per pipeline stage (16-B aligned)
  static_assert(POS == 4 * NS * 32, "workgroup tile = 4 waves x NS sub-tiles x 32 positions");
  static_assert(KC % 2 == 0, "K step is 2 channels");
  static_assert(LDS_IN % 4 == 0, "weight slab must start 16-B aligned");
};

// XCD-aware block id: blocks b and b+8 share an XCD (round-robin dispatch), so hand each XCD a
// contiguous run of logical tiles -- neighbouring tiles (same input box, other channel tile;
// adjacent boxes sharing halo lines) then hit in that XCD's L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, k = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Software pipeline (one barrier per chunk, two LDS stages):
//   issue global loads of chunk c+1 into registers -> MFMA loop over chunk c (stage c&1) -> apply the
//   prologue to the prefetched registers and write them to stage (c+1)&1 -> barrier.
// The loads fly under ~100-200 MFMAs per wave; the only serial part per chunk is the register->LDS
// hand-off, which the other workgroup on the CU covers.
// FAST: cin_g % KC == 0, cout_g % CO == 0 and 16-byte-loadable weight rows (every production layer):
// no per-channel / per-column bounds logic in the loop.
template <int KS, int NS, int MSUB, int KC, int TN, int TH, int TW, bool FAST, bool MASK>
__global__ void __launch_bounds__(256, 2) conv3d_mfma_kernel(const ConvArgs a) {
  using G = Geo<KS, NS, MSUB, KC, TN, TH, TW>;
  constexpr int CO = G::CO;
  __shared__ __attribute__((aligned(16))) float lds[2 * G::STAGE];

  uint64_t dseed = a.drop_seed;
  if constexpr (MASK) {
    if (a.drop_seed_dev) dseed += *a.drop_seed_dev;      // (per-step part of the dropout seed: HIP-graph replays)
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- which tile -------------------------------------------------------------------------
  // (integer division runs on the VALU; readfirstlane moves the wave-uniform results back to SGPRs)
  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int co_tile = __builtin_amdgcn_readfirstlane(id % a.tiles_co); id /= a.tiles_co;
  const int tw_i = __builtin_amdgcn_readfirstlane(id % a.tiles_w); id /= a.tiles_w;
  const int th_i = __builtin_amdgcn_readfirstlane(id % a.tiles_h); id /= a.tiles_h;
  const int tn_i = __builtin_amdgcn_readfirstlane(id % a.tiles_n); id /= a.tiles_n;
  const int g = __builtin_amdgcn_readfirstlane(id % a.groups); id /= a.groups;
  const int b = __builtin_amdgcn_readfirstlane(id % a.B);
  const int split = __builtin_amdgcn_readfirstlane(id / a.B);   // split-K range of this workgroup (outermost index)
  const int c_begin = split * a.split_ch, c_end = c_begin + a.split_ch;
  const int n0 = tn_i * TN, h0 = th_i * TH, w0 = tw_i * TW;
  const int co0 = co_tile * CO;  // within group
  const long plane = (long)a.N * a.H * a.W;

  // ---- per-thread staging pattern (same for every input channel) -----------------------------
  int goff[G::EPT];
  bool gok[G::EPT];
#pragma unroll
  for (int i = 0; i < G::EPT; ++i) {
    const int e = tid + 256 * i;
    const int wz = e % G::HW, hz = (e / G::HW) % G::HH, nz = e / (G::HW * G::HH);
    const int n = n0 + nz - G::HALO, h = h0 + hz - G::HALO, w = w0 + wz - G::HALO;
    gok[i] = (e < G::TILE_ELEMS) && n >= 0 && n < a.N && h >= 0 && h < a.H && w >= 0 && w < a.W;
    goff[i] = gok[i] ? (n * a.H + h) * a.W + w : 0;
  }
  // weight slab float4 idx = tid + 256*j  ->  row = idx / (CO/4) = wrow0 + j*WSTEP, col = wcol (256 % (CO/4) == 0)
  constexpr int WSTEP = 256 / (CO / 4);
  const int wrow0 = tid / (CO / 4), wcol = (tid % (CO / 4)) * 4;

  // ---- per-lane operand offsets ---------------------------------------------------------------
  int boff[NS];  // float index of this lane's position inside the haloed box, tap (0,0,0)
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int p = (wv * NS + s) * 32 + l31;
    const int pw = p % TW, ph = (p / TW) % TH, pn = p / (TW * TH);
    boff[s] = (pn * G::HH + ph) * G::HW + pw + khalf * G::TILE_ELEMS;
  }
  // A operand: the slab row holds the block's CO channels as [l31][m] (see pack_weights), so one
  // ds_read_b32/b64/b128 fetches this lane's weight for all MSUB channel sub-tiles.
  const int aoff = G::LDS_IN + khalf * G::TAPS * CO + l31 * MSUB;

  f32x16 acc[NS][MSUB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int m = 0; m < MSUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  const float* wp_g = a.wp + (long)g * a.cin_g * G::TAPS * a.cout_g;

  float bias_v[MSUB];  // epilogue bias, fetched now so that its latency is long gone when it is needed
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int col = min(co0 + m * 32 + l31, a.cout_g - 1);
    const float* bp = a.bias ? a.bias + g * a.cout_g + col : a.wp;  // always a valid address
    const float raw = *bp;
    bias_v[m] = a.bias ? a.bias_scale * raw : 0.f;
  }
  float sh2_v[MSUB], sc2_v[MSUB];  // second-output shift / scale of channel co0 + m*32 + l31 (same readlane scheme)
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    const int col = g * a.cout_g + min(co0 + m * 32 + l31, a.cout_g - 1);
    sh2_v[m] = (a.y2 && a.y2_shift) ? a.y2_shift[(long)b * a.y2_shift_stride + col] : 0.f;
    sc2_v[m] = (a.y2 && a.y2_scale) ? a.y2_scale[(long)b * a.y2_scale_stride + col] : 1.f;
  }

  // prefetch registers of the chunk in flight
  float xr[KC][G::EPT];
  float wr[G::W4][4];  // (scalars, not float4[]: an array of vectors is left in scratch by SROA here)
  float ssv = 0.f;     // lane l < KC: shift of channel cn+l; KC <= l < 2KC: scale of channel cn+l-KC
  bool cval[KC];

  // Software pipeline, one barrier per chunk, two LDS stages.  While the K-steps of chunk `it` run from stage
  // it&1, chunk it+1 is brought in behind them.  The unit of interleaving is ONE MFMA: the fp32 32x32x2 MFMA
  // holds the matrix pipe for 64 cycles, in which the wave can issue ~a dozen other instructions for free, so
  // every MFMA is followed by a "slot" that carries at most a small work item:
  //   slot 0 of a K-step        : LDS operand fetch for the NEXT K-step,
  //   early aux slots           : the global loads of chunk it+1 (registers xr / wr / ssv), one per slot,
  //   late aux slots            : the register -> LDS hand-off (prologue math, then ds_write) to stage (it+1)&1.
  // sched_barrier(0) pins that order (hipcc otherwise sinks the loads next to their uses and bunches the
  // hand-off after the MFMAs).  Every load is unconditional (out-of-range elements read a clamped, valid address
  // and are zeroed by a select when staged): a branch around a load would force vmcnt(0) waits.  Per-channel
  // shift / scale come through ONE vector load + v_readlane, not scalar loads: SMEM shares lgkmcnt with LDS.
  constexpr int MF = NS * MSUB;                      // MFMAs per K-step
  constexpr int KSTEPS = (KC / 2) * G::TAPS;
  constexpr int AUX = KSTEPS * (MF - 1);             // aux slots per chunk
  constexpr int NL = 1 + KC + KC * G::EPT + G::W4;   // load items: ssv, KC source pointers, inputs, weights
  constexpr int NP = 2 * KC * G::EPT + G::W4;        // hand-off items: (math, write) per input element, weights
  constexpr int LSPAN = AUX / 2 < NL ? (AUX / 2 > 0 ? AUX / 2 : 1) : NL;
  constexpr int PSPAN = AUX - AUX / 2 < NP ? (AUX - AUX / 2 > 0 ? AUX - AUX / 2 : 1) : NP;
  static_assert(MF >= 2, "need at least one aux slot per K-step");

  const float* srcp[KC];
  auto load_item = [&](auto qc, int cn) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::value;
    if constexpr (q == 0) {
      // (the loaded value is NOT touched here -- no select on it -- or the wave would sit in vmcnt(0) for the full
      //  memory latency in the middle of the MFMA stream; absent shift / scale are substituted at the point of use)
      const bool is_shift = lane < KC;
      const float* sp = is_shift ? a.in_shift : a.in_scale;
      const long row = (long)b * (is_shift ? a.shift_stride : a.scale_stride);
      const int cl = min(cn + (lane % KC), a.cin_g - 1);
      const float* sp2 = (sp && lane < 2 * KC) ? sp + row + g * a.cin_g + cl : a.wp;  // always a valid address
      ssv = *sp2;
    } else if constexpr (q <= KC) {
      constexpr int ci = q - 1;
      const int cl = cn + ci;  // channel within group
      cval[ci] = FAST || cl < a.cin_g;
      const int cs = g * a.cin_g + (cval[ci] ? cl : 0);  // channel within the concatenated input (clamped)
      // which concat segment holds it: selects between COMPUTED addresses (a select between the kernel-argument
      // fields themselves becomes a dependent s_load + lgkmcnt(0) inside the MFMA stream)
      const int c0 = a.seg_c[0], c1 = a.seg_c[1], c2 = a.seg_c[2];
      const float* e0 = a.seg_x[0] + ((long)b * c0 + cs) * plane;
      const float* e1 = a.seg_x[1] + ((long)b * c1 + (cs - c0)) * plane;
      const float* e2 = a.seg_x[2] + ((long)b * c2 + (cs - c0 - c1)) * plane;
      srcp[ci] = cs < c0 ? e0 : (cs < c0 + c1 ? e1 : e2);
    } else if constexpr (q <= KC + KC * G::EPT) {
      constexpr int ci = (q - 1 - KC) / G::EPT, i = (q - 1 - KC) % G::EPT;
      xr[ci][i] = srcp[ci][goff[i]];
    } else {
      constexpr int j = q - 1 - KC - KC * G::EPT;
      const int row = min(wrow0 + j * WSTEP, KC * G::TAPS - 1);
      if constexpr (FAST) {
        const float4 v = *reinterpret_cast<const float4*>(wp_g + ((long)cn * G::TAPS + row) * a.cout_g + co0 + wcol);
        wr[j][0] = v.x, wr[j][1] = v.y, wr[j][2] = v.z, wr[j][3] = v.w;
      } else {
        const int cl = cn + row / G::TAPS;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cl < a.cin_g) {
          const float* p = wp_g + ((long)cn * G::TAPS + row) * a.cout_g + co0 + wcol;
          if (a.w_vec4 && co0 + wcol + 3 < a.cout_g) {
            v = *reinterpret_cast<const float4*>(p);
          } else {
            if (co0 + wcol + 0 < a.cout_g) v.x = p[0];
            if (co0 + wcol + 1 < a.cout_g) v.y = p[1];
            if (co0 + wcol + 2 < a.cout_g) v.z = p[2];
            if (co0 + wcol + 3 < a.cout_g) v.w = p[3];
          }
        }
        wr[j][0] = v.x, wr[j][1] = v.y, wr[j][2] = v.z, wr[j][3] = v.w;
      }
    }
  };

  // hand-off item `pc` of the chunk starting at channel cn -> LDS stage st
  auto stage_item = [&](auto pcc, float* st, int cn) __attribute__((always_inline)) {
    constexpr int pc = decltype(pcc)::value;
    if constexpr (pc < 2 * KC * G::EPT) {
      constexpr int ci = (pc / 2) / G::EPT, i = (pc / 2) % G::EPT;
      const int e = tid + 256 * i;
      if constexpr (pc % 2 == 0) {  // prologue math, in place in the prefetch register
        const float sh = a.in_shift ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ssv), ci)) : 0.f;
        const float sc = a.in_scale ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ssv), KC + ci)) : 1.f;
        float t = xr[ci][i] + sh;
        const float ta = tmdiff::silu_f(t);
        t = (a.in_act ? ta : t) * sc;
        if constexpr (MASK)  // dropout (training only): a mask tensor read here (not prefetched), or the counter-based hash
          if (cval[ci]) {
            const long ei = ((long)b * a.Cin + g * a.cin_g + cn + ci) * plane + goff[i];
            t *= a.in_mask ? a.in_mask[ei] : tmdiff::drop_keep(dseed, (uint64_t)ei, a.drop_thresh, a.drop_inv);
          }
        xr[ci][i] = t;
      } else {
        // halo / out-of-range elements are exactly zero: the conv pads the ACTIVATED tensor.  Lanes past the end
        // of the box write the stage's sink word instead of branching (a branch would split the MFMA block).
        st[e < G::TILE_ELEMS ? ci * G::TILE_ELEMS + e : G::DUMMY] = (gok[i] && cval[ci]) ? xr[ci][i] : 0.f;
      }
    } else {
      constexpr int j = pc - 2 * KC * G::EPT;
      const int row = wrow0 + j * WSTEP;
      *reinterpret_cast<float4*>(st + (row < KC * G::TAPS ? G::LDS_IN + row * CO + wcol : G::DUMMY)) =
          make_float4(wr[j][0], wr[j][1], wr[j][2], wr[j][3]);
    }
  };

  // prologue: chunk 0 -> stage 0
  static_for<0, NL>([&](auto qc) __attribute__((always_inline)) { load_item(qc, c_begin); });
  static_for<0, NP>([&](auto pcc) __attribute__((always_inline)) { stage_item(pcc, lds, c_begin); });
  __syncthreads();

  for (int it = 0;; ++it) {
    const int cn = c_begin + (it + 1) * KC;  // first channel of the chunk brought in during this iteration
    const bool more = cn < c_end;
    const int cn_ld = more ? cn : c_begin;  // last chunk: re-load the first one (valid addresses, result unused) -> no branches
    const float* st = lds + (it & 1) * G::STAGE;
    float* st_next = lds + ((it + 1) & 1) * G::STAGE;

    auto fetch = [&](auto ksc, float (&av)[MSUB], float (&bv)[NS]) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int kp = ks / G::TAPS, tap = ks % G::TAPS;
      constexpr int dn = tap / (KS * KS), dh = (tap / KS) % KS, dw = tap % KS;
      constexpr int toff = (dn * G::HH + dh) * G::HW + dw;
      const float* ap = st + aoff + (kp * 2 * G::TAPS + tap) * CO;
      if constexpr (MSUB == 1) {
        av[0] = ap[0];
      } else if constexpr (MSUB == 2) {
        const float2 t2 = *reinterpret_cast<const float2*>(ap);
        av[0] = t2.x, av[1] = t2.y;
      } else {
        const float4 t4 = *reinterpret_cast<const float4*>(ap);
        av[0] = t4.x, av[1] = t4.y, av[2] = t4.z, av[3] = t4.w;
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) bv[s] = st[boff[s] + kp * 2 * G::TILE_ELEMS + toff];
    };
    float av[2][MSUB], bv[2][NS];
    fetch(std::integral_constant<int, 0>{}, av[0], bv[0]);
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, KSTEPS>([&](auto ksc) __attribute__((always_inline)) {
      constexpr int ks = decltype(ksc)::value;
      static_for<0, MF>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        constexpr int s = j / MSUB, m = j % MSUB;
        acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][m], bv[ks & 1][s], acc[s][m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j == 0) {
          if constexpr (ks + 1 < KSTEPS) fetch(std::integral_constant<int, ks + 1>{}, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
        } else {
          constexpr int ax = ks * (MF - 1) + (j - 1);  // aux slot index
          // load items q with floor(q*LSPAN/NL) == ax, hand-off items pc with AUX-PSPAN+floor(pc*PSPAN/NP) == ax
          if constexpr (ax < LSPAN && !(TMDIFF_CONV_DEBUG & 1)) {
              static_for<(ax * NL + LSPAN - 1) / LSPAN, ((ax + 1) * NL + LSPAN - 1) / LSPAN>(
                  [&](auto qc) __attribute__((always_inline)) { load_item(qc, cn_ld); });
          }
          if constexpr (ax >= AUX - PSPAN && !(TMDIFF_CONV_DEBUG & 2)) {
            constexpr int ps = ax - (AUX - PSPAN);
              static_for<(ps * NP + PSPAN - 1) / PSPAN, ((ps + 1) * NP + PSPAN - 1) / PSPAN>(
                  [&](auto pcc) __attribute__((always_inline)) { stage_item(pcc, st_next, cn_ld); });
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if (!more) break;
    __syncthreads();
  }

  if (a.part) {  // split-K: raw partial sums; splitk_reduce_kernel adds them up and applies the epilogue
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

  if constexpr (FAST) {
    if (a.vec4) {
      static_assert(sizeof(lds) >= 4 * 4096, "the epilogue borrows 4 KB of LDS per wave");
      __syncthreads();   // (the chunk loop leaves without a barrier: the other waves may still be reading the stage)
      tmdiff::epilogue_vec<NS, MSUB, TN, TH, TW>(a, acc, bias_v, sh2_v, sc2_v, b, g, co0, n0, h0, w0, wv, lane, plane,
                                                 lds + wv * 1024);
      return;
    }
  }
  // ---- scalar epilogue: bias, residual, scale; D layout: col = lane&31 (position), row = channel --------
  // Loads first (bias rows, then all residual elements of a sub-tile), then the stores: no load->store chains.
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    // lane l31 holds the bias of channel co0 + m*32 + l31 (one coalesced load, issued before the main loop);
    // accumulator register r of a lane needs row (r&3) + 8*(r>>2) + 4*khalf of it -> v_readlane + select.
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
        const bool ok = pok && (FAST || co0 + m * 32 + 4 * khalf + row < a.cout_g);
        res[r] = (a.residual && ok) ? a.residual[obase + row * plane] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const bool ok = pok && (FAST || co0 + m * 32 + 4 * khalf + row < a.cout_g);
        const float v = (acc[s][m][r] + bias_r[r] + res[r]) * a.out_scale;
        if (ok && a.y) a.y[obase + row * plane] = v;
        acc[s][m][r] = v;
      }
      if (a.y2) {  // (wave-uniform) the consumer's prologue on the finished values
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2);
          const bool ok = pok && (FAST || co0 + m * 32 + 4 * khalf + row < a.cout_g);
          const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sh2_v[m]), row));
          const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sh2_v[m]), row + 4));
          const float c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc2_v[m]), row));
          const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sc2_v[m]), row + 4));
          float t = acc[s][m][r] + (khalf ? s1 : s0);
          const float ta = tmdiff::silu_f(t);
          t = (a.y2_act ? ta : t) * (khalf ? c1 : c0);
          if (ok) a.y2[obase + row * plane] = t;
        }
      }
    }
  }
}

// Column order inside a packed row.  When the (output) channel count is a multiple of 64 the forward kernel
// runs 64-channel tiles with two 32-channel MFMA sub-tiles per wave; channel c of a tile is then stored at
// (c % 32) * 2 + c / 32 so that one 8-byte LDS read returns a lane's weight for both sub-tiles.
__host__ __device__ inline int packed_col(int c, int n_out) {
  if (n_out % 64) return c;
  const int t = c / 64, r = c % 64;
  return t * 64 + (r % 32) * 2 + r / 32;
}

// packed[g][ci][tap][col(co)] <- w[g*cout_g + co][ci][tap]               (mode 0, forward)
// packed[g][co][taps-1-tap][col(ci)] <- w[g*cout_g + co][ci][tap]        (mode 1, data gradient: roles swapped)
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ w, float* __restrict__ packed,
                                                           int cout_g, int cin_g, int taps, int groups, int mode,
                                                           long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += 256L * gridDim.x) {
    // i enumerates the source tensor [g][co][ci][tap]
    const int tap = (int)(i % taps);
    long r = i / taps;
    const int ci = (int)(r % cin_g); r /= cin_g;
    const int co = (int)(r % cout_g);
    const int g = (int)(r / cout_g);
    long dst;
    if (mode == 0)
      dst = (((long)g * cin_g + ci) * taps + tap) * cout_g + packed_col(co, cout_g);
    else
      dst = (((long)g * cout_g + co) * taps + (taps - 1 - tap)) * cin_g + packed_col(ci, cin_g);
    packed[dst] = w[i];
  }
}

// ---- split-K reduction: y = (sum_s part[s] + bias_scale*bias + residual) * out_scale, optional y2; V floats per thread ----
template <int V>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const tmdiff::SplitKReduceArgs r) {
  const long per_row = (r.plane + V - 1) / V;
  const long i = blockIdx.x * 256L + threadIdx.x;
  const long row = i / per_row;                     // b * Cout + co
  if (row >= (long)r.B * r.Cout) return;
  const long p = (i % per_row) * V;
  const int co = (int)(row % r.Cout), b = (int)(row / r.Cout);
  const long off = row * r.plane + p;
  const long sstride = (long)r.B * r.Cout * r.plane;
  float v[V];
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(r.part + off);
    v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    for (int s = 1; s < r.ksplit; ++s) {             // fixed order: deterministic
      const float4 u = *reinterpret_cast<const float4*>(r.part + s * sstride + off);
      v[0] += u.x, v[1] += u.y, v[2] += u.z, v[3] += u.w;
    }
  } else {
    v[0] = r.part[off];
    for (int s = 1; s < r.ksplit; ++s) v[0] += r.part[s * sstride + off];
  }
  const float bias = r.bias ? r.bias[co] * r.bias_scale : 0.f;
  float res[V];
#pragma unroll
  for (int k = 0; k < V; ++k) res[k] = 0.f;
  if (r.residual) {
    if constexpr (V == 4) {
      const float4 t = *reinterpret_cast<const float4*>(r.residual + off);
      res[0] = t.x, res[1] = t.y, res[2] = t.z, res[3] = t.w;
    } else {
      res[0] = r.residual[off];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) v[k] = (v[k] + bias + res[k]) * r.out_scale;   // same association as the in-kernel epilogue
  if (r.y) {
    if constexpr (V == 4)
      *reinterpret_cast<float4*>(r.y + off) = make_float4(v[0], v[1], v[2], v[3]);
    else
      r.y[off] = v[0];
  }
  if (r.y2) {
    const float sh = r.y2_shift ? r.y2_shift[(long)b * r.y2_shift_stride + co] : 0.f;
    const float sc = r.y2_scale ? r.y2_scale[(long)b * r.y2_scale_stride + co] : 1.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float t = v[k] + sh;
      const float ta = tmdiff::silu_f(t);
      v[k] = (r.y2_act ? ta : t) * sc;
    }
    if constexpr (V == 4)
      *reinterpret_cast<float4*>(r.y2 + off) = make_float4(v[0], v[1], v[2], v[3]);
    else
      r.y2[off] = v[0];
  }
}

// Multi-tensor form: every convolution weight of a network, forward AND data-gradient packing, in ONE launch (a training
// step re-packs all ~70 weights twice: 140 launches of 8 us were 1.1 ms of a 42 ms step).  Both packings are transposes of
// the source [co][ci][tap], so each goes through an LDS tile shaped for ITS destination's contiguous axis:
//   type A (forward packing  [ci][tap][co]):        tile = 32 co x 8 ci  x taps, written as 128-byte runs over co;
//   type B (data-gradient    [co][taps-1-tap][ci]): tile = 8 co  x 64 ci x taps, written as 256-byte runs over ci.
// (A single pass with scattered 4-byte writes took 0.72 ms for the 31 M weights of the ch 32-256 network.)
constexpr int PK_A_CO = 32, PK_A_CI = 8, PK_B_CO = 8, PK_B_CI = 64, PK_MAXTAPS = 27;
__global__ void __launch_bounds__(256) pack_weights_multi_kernel(const tmdiff_pack_entry* __restrict__ entries,
                                                                 const int32_t* __restrict__ chunk_tensor,
                                                                 const int32_t* __restrict__ chunk_index) {
  __shared__ float tile[PK_B_CO * (PK_B_CI * PK_MAXTAPS + 1)];   // also holds a type-A tile: 32 x (8*27 + 1)
  const tmdiff_pack_entry e = entries[chunk_tensor[blockIdx.x]];
  const int taps = e.ksize * e.ksize * e.ksize, cout_g = e.Cout / e.groups, cin_g = e.Cin / e.groups;
  const int raw = chunk_index[blockIdx.x];
  const bool type_b = (raw >> 30) & 1;
  int id = raw & 0x3FFFFFFF;
  const int tid = threadIdx.x;
  if (!type_b) {
    const int nci = (cin_g + PK_A_CI - 1) / PK_A_CI, nco = (cout_g + PK_A_CO - 1) / PK_A_CO;
    const int cib = id % nci; id /= nci;
    const int cob = id % nco;
    const int g = id / nco;
    const int ci0 = cib * PK_A_CI, co0 = cob * PK_A_CO;
    const int ncil = min(PK_A_CI, cin_g - ci0), run = ncil * taps, LS = PK_A_CI * PK_MAXTAPS + 1;
    for (int i = tid; i < PK_A_CO * run; i += 256) {      // source: per co a contiguous run of ncil*taps floats
      const int co = i / run, k = i % run;
      if (co0 + co < cout_g) tile[co * LS + k] = e.w[(((long)g * cout_g + co0 + co) * cin_g + ci0) * taps + k];
    }
    __syncthreads();
    if (e.packed_fwd) {
      const int co = tid % PK_A_CO;
      if (co0 + co < cout_g)
        for (int k = tid / PK_A_CO; k < run; k += 256 / PK_A_CO)     // k = ci_local * taps + tap
          e.packed_fwd[(((long)g * cin_g + ci0) * taps + k) * cout_g + packed_col(co0 + co, cout_g)] = tile[co * LS + k];
    }
  } else {
    const int nci = (cin_g + PK_B_CI - 1) / PK_B_CI, nco = (cout_g + PK_B_CO - 1) / PK_B_CO;
    const int cib = id % nci; id /= nci;
    const int cob = id % nco;
    const int g = id / nco;
    const int ci0 = cib * PK_B_CI, co0 = cob * PK_B_CO;
    const int ncil = min(PK_B_CI, cin_g - ci0), run = ncil * taps, LS = PK_B_CI * PK_MAXTAPS + 1;
    for (int i = tid; i < PK_B_CO * run; i += 256) {
      const int co = i / run, k = i % run;
      if (co0 + co < cout_g) tile[co * LS + k] = e.w[(((long)g * cout_g + co0 + co) * cin_g + ci0) * taps + k];
    }
    __syncthreads();
    if (e.packed_dgrad) {
      const int ci = tid % PK_B_CI;
      if (ci < ncil)
        for (int q = tid / PK_B_CI; q < PK_B_CO * taps; q += 256 / PK_B_CI) {   // q = co_local * taps + tap
          const int co = q / taps, tap = q % taps;
          if (co0 + co < cout_g)
            e.packed_dgrad[(((long)g * cout_g + co0 + co) * taps + (taps - 1 - tap)) * cin_g + packed_col(ci0 + ci, cin_g)] =
                tile[co * LS + ci * taps + tap];
        }
    }
  }
}

template <int KS, int NS, int MSUB, int KC, int TN, int TH, int TW>
int launch(ConvArgs& a, hipStream_t st) {
  constexpr int CO = 32 * MSUB;
  a.tiles_n = (a.N + TN - 1) / TN;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  a.tiles_co = (a.cout_g + CO - 1) / CO;
  const long blocks = (long)a.ksplit * a.B * a.groups * a.tiles_n * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  const bool fast = a.cin_g % KC == 0 && a.cout_g % CO == 0 && a.w_vec4;
  const bool masked = a.in_mask || a.drop_inv > 0.f;
  if (masked && fast)  // training (dropout) path
    conv3d_mfma_kernel<KS, NS, MSUB, KC, TN, TH, TW, true, true><<<(unsigned)blocks, 256, 0, st>>>(a);
  else if (masked)
    conv3d_mfma_kernel<KS, NS, MSUB, KC, TN, TH, TW, false, true><<<(unsigned)blocks, 256, 0, st>>>(a);
  else if (fast)
    conv3d_mfma_kernel<KS, NS, MSUB, KC, TN, TH, TW, true, false><<<(unsigned)blocks, 256, 0, st>>>(a);
  else
    conv3d_mfma_kernel<KS, NS, MSUB, KC, TN, TH, TW, false, false><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_fwd");
}

}  // namespace

namespace tmdiff {

int launch_splitk_reduce(const SplitKReduceArgs& r, hipStream_t st) {
  const bool vec = r.plane % 4 == 0 && aligned16(r.part) && aligned16(r.y) && aligned16(r.y2) && aligned16(r.residual);
  const long per_row = vec ? r.plane / 4 : r.plane;
  const long blocks = ((long)r.B * r.Cout * per_row + 255) / 256;
  if (vec)
    splitk_reduce_kernel<4><<<(unsigned)blocks, 256, 0, st>>>(r);
  else
    splitk_reduce_kernel<1><<<(unsigned)blocks, 256, 0, st>>>(r);
  return check_launch("conv3d split-K reduce");
}

// Split target: once a launch has fewer workgroups than this, its input channels are divided until it has at least that
// many (the chip holds 256 CUs x 2 workgroups of these kernels).  TMDIFF_SPLITK=<n> overrides (0 = never split).
static long splitk_target() {
  static const long t = [] {
    const char* e = getenv("TMDIFF_SPLITK");
    return e ? atol(e) : 384L;
  }();
  return t;
}

bool epilogue_vec_ok(const tmdiff_conv3d_desc* d) {
  static const bool on = [] {
    const char* e = getenv("TMDIFF_EPILOGUE_VEC");   // experiments: "0" = the scalar epilogue everywhere
    return !(e && e[0] == '0');
  }();
  // (the dwordx4 epilogue addresses up to 128 channels of a sample through one descriptor of 32-bit offsets: planes <= 2^23)
  return on && d->W % 4 == 0 && aligned16(d->y) && aligned16(d->residual) && (d->y2_bf16 || aligned16(d->y2)) &&
         (long)d->N * d->H * d->W <= (1L << 23);
}

Conv3Plan plan_conv3(const tmdiff_conv3d_desc* d) {
  Conv3Plan p{3, 0, 1};
  const int cin_g = d->Cin / d->groups, cout_g = d->Cout / d->groups;
  const long boxes48 = (long)d->B * d->groups * ((d->N + 3) / 4) * ((d->H + 7) / 8) * ((d->W + 7) / 8);
  const bool c64 = cout_g % 64 == 0;
  const long wg256 = boxes48 * ((cout_g + 63) / 64);
  // Layers whose 256-position grid would leave CUs idle or badly quantised (the 8x8x8 level: 128-384 workgroups) use
  // 128-position tiles; 32-channel tiles take 512 positions when that still gives every CU a workgroup.
  static const long small_limit = getenv("TMDIFF_SMALLGRID") ? atol(getenv("TMDIFF_SMALLGRID")) : 2 * 256;  // experiments
  if (c64 && wg256 < small_limit && d->N > 2) {
    p.tile = 0;
    p.blocks = (long)d->B * d->groups * ((d->N + 1) / 2) * ((d->H + 7) / 8) * ((d->W + 7) / 8) * (cout_g / 64);
  } else if (c64) {
    p.tile = 1;
    p.blocks = wg256;
  } else {
    const long wg512 = (long)d->B * d->groups * ((d->N + 3) / 4) * ((d->H + 7) / 8) * ((d->W + 15) / 16) * ((cout_g + 31) / 32);
    if (d->W >= 16 && wg512 >= 256) {
      p.tile = 2;
      p.blocks = wg512;
    } else {
      p.tile = 3;
      p.blocks = boxes48 * ((cout_g + 31) / 32);
    }
  }
  // split-K: only exact shapes (whole chunks, whole channel tiles), only when the caller lent a workspace
  const long target = splitk_target();
  if (d->ksize != 3 || target <= 0 || p.blocks >= target || cin_g % 4 || cout_g % 32) return p;
  const int nchunks = cin_g / 4;
  int best = 1;
  for (int s = 2; s <= nchunks / 2; ++s) {            // at least two chunks per range
    if (nchunks % s) continue;
    best = s;
    if (p.blocks * s >= target) break;
  }
  p.ksplit = best;
  return p;
}

}  // namespace tmdiff

extern "C" size_t tmdiff_conv3d_fwd_splitk_workspace_bytes(const tmdiff_conv3d_desc* d) {
  if (!d || d->ksize != 3 || d->B <= 0 || d->groups <= 0 || d->Cin % d->groups || d->Cout % d->groups) return 0;
  const tmdiff::Conv3Plan p = tmdiff::plan_conv3(d);
  return p.ksplit > 1 ? (size_t)p.ksplit * d->B * d->Cout * d->N * d->H * d->W * sizeof(float) : 0;
}

extern "C" int tmdiff_conv3d_pack_weights(const float* w, float* packed, int32_t Cout, int32_t Cin, int32_t ksize,
                                          int32_t groups, int32_t mode, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(w && packed, "pack_weights: NULL pointer");
  TMDIFF_REQUIRE(ksize == 1 || ksize == 3, "pack_weights: ksize=%d (1 or 3)", ksize);
  TMDIFF_REQUIRE(groups >= 1 && Cout > 0 && Cin > 0 && Cout % groups == 0 && Cin % groups == 0,
                 "pack_weights: Cout=%d Cin=%d groups=%d", Cout, Cin, groups);
  TMDIFF_REQUIRE(mode == 0 || mode == 1, "pack_weights: mode=%d", mode);
  const int taps = ksize * ksize * ksize;
  const long total = (long)Cout * (Cin / groups) * taps;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  pack_weights_kernel<<<(int)blocks, 256, 0, as_stream(stream)>>>(w, packed, Cout / groups, Cin / groups, taps, groups,
                                                                 mode, total);
  return check_launch("conv3d_pack_weights");
}

// number of workgroups (chunks) a weight of this shape needs: type-A tiles first, then type-B tiles
extern "C" int32_t tmdiff_conv3d_pack_weights_multi_chunks(int32_t Cout, int32_t Cin, int32_t groups, int32_t* n_type_a) {
  if (groups <= 0 || Cout % groups || Cin % groups) return 0;
  const int cout_g = Cout / groups, cin_g = Cin / groups;
  const int na = groups * ((cout_g + PK_A_CO - 1) / PK_A_CO) * ((cin_g + PK_A_CI - 1) / PK_A_CI);
  const int nb = groups * ((cout_g + PK_B_CO - 1) / PK_B_CO) * ((cin_g + PK_B_CI - 1) / PK_B_CI);
  if (n_type_a) *n_type_a = na;
  return na + nb;
}

extern "C" int tmdiff_conv3d_pack_weights_multi(const tmdiff_pack_entry* entries_dev, const int32_t* chunk_tensor_dev,
                                                const int32_t* chunk_index_dev, int32_t n_chunks, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(entries_dev && chunk_tensor_dev && chunk_index_dev && n_chunks >= 0, "pack_weights_multi: bad arguments");
  if (n_chunks == 0) return TMDIFF_OK;
  pack_weights_multi_kernel<<<(unsigned)n_chunks, 256, 0, as_stream(stream)>>>(entries_dev, chunk_tensor_dev, chunk_index_dev);
  return check_launch("conv3d_pack_weights_multi");
}

extern "C" int tmdiff_conv3d_fwd(const tmdiff_conv3d_desc* d, tmdiff_stream_t stream) {
  using namespace tmdiff;
  TMDIFF_REQUIRE(d != nullptr, "conv3d_fwd: NULL descriptor");
  TMDIFF_REQUIRE(d->B >= 0 && d->N > 0 && d->H > 0 && d->W > 0, "conv3d_fwd: bad extents B=%d N=%d H=%d W=%d", d->B,
                 d->N, d->H, d->W);
  TMDIFF_REQUIRE(d->ksize == 1 || d->ksize == 3, "conv3d_fwd: ksize=%d (1 or 3)", d->ksize);
  TMDIFF_REQUIRE(d->groups == 1 || d->groups == 3, "conv3d_fwd: groups=%d (1 or 3)", d->groups);
  TMDIFF_REQUIRE(d->Cin > 0 && d->Cout > 0 && d->Cin % d->groups == 0 && d->Cout % d->groups == 0,
                 "conv3d_fwd: Cin=%d Cout=%d groups=%d", d->Cin, d->Cout, d->groups);
  TMDIFF_REQUIRE(d->nseg >= 1 && d->nseg <= 3, "conv3d_fwd: nseg=%d", d->nseg);
  if (d->B == 0) return TMDIFF_OK;  // empty batch: nothing to read or write
  int csum = 0;
  for (int i = 0; i < d->nseg; ++i) {
    TMDIFF_REQUIRE(d->seg_x[i] != nullptr && d->seg_c[i] > 0, "conv3d_fwd: segment %d is empty", i);
    csum += d->seg_c[i];
  }
  TMDIFF_REQUIRE(csum == d->Cin, "conv3d_fwd: segments hold %d channels, Cin=%d", csum, d->Cin);
  if (d->groups == 3)
    TMDIFF_REQUIRE(d->nseg == 1 || (d->nseg == 3 && d->seg_c[0] == d->seg_c[1] && d->seg_c[1] == d->seg_c[2]),
                   "conv3d_fwd: groups=3 wants 1 segment or 3 equal ones");
  TMDIFF_REQUIRE(d->w_packed && (d->y || d->y2), "conv3d_fwd: NULL weights/output");
  TMDIFF_REQUIRE(!d->y2 || !d->y2_bf16, "conv3d_fwd: a bf16-packed second output needs tmdiff_conv3d_fwd_bf16");
  TMDIFF_REQUIRE((long)d->N * d->H * d->W < (1L << 31), "conv3d_fwd: plane too large for 32-bit offsets");

  if (d->ksize == 1) {  // bandwidth kernel for the shapes it takes (every production 1x1x1 layer)
    const int rc = conv1_fp32_try(d, as_stream(stream));
    if (rc != TMDIFF_E_UNSUPPORTED) return rc;
  }
  if (d->xp_out)
    return fail(TMDIFF_E_UNSUPPORTED, "conv3d_fwd: only the 1x1x1 bandwidth kernel writes the by-product xp_out (fp32, ksize 1, no mask / "
                                      "dropout / second output, Cin/g and every segment %% 16 == 0, Cout/g %% 32 == 0)");

  ConvArgs a;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cout = d->Cout; a.groups = d->groups;
  a.cin_g = d->Cin / d->groups; a.cout_g = d->Cout / d->groups;
  a.nseg = d->nseg;
  for (int i = 0; i < 3; ++i) {  // unused segments: never selected (huge channel count), but with a valid pointer
    a.seg_c[i] = i < d->nseg ? d->seg_c[i] : (1 << 28);
    a.seg_x[i] = i < d->nseg ? d->seg_x[i] : d->seg_x[0];
  }
  a.wp = d->w_packed; a.bias = d->bias; a.bias_scale = d->bias_scale;
  a.in_shift = d->in_shift; a.in_scale = d->in_scale; a.in_mask = d->in_mask; a.in_act = d->in_act;
  TMDIFF_REQUIRE(!(d->in_mask && d->drop_p > 0.f), "conv3d_fwd: give either a mask tensor or drop_p, not both");
  TMDIFF_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f, "conv3d_fwd: drop_p=%g", (double)d->drop_p);
  a.drop_seed = d->drop_seed; a.drop_seed_dev = d->drop_seed_dev; a.drop_thresh = drop_threshold(d->drop_p);
  a.drop_inv = d->drop_p > 0.f ? 1.0f / (1.0f - d->drop_p) : 0.f;
  a.shift_stride = d->in_shift_stride > 0 ? d->in_shift_stride : (d->in_shift_stride < 0 ? 0 : d->Cin);
  a.scale_stride = d->in_scale_stride > 0 ? d->in_scale_stride : (d->in_scale_stride < 0 ? 0 : d->Cin);
  a.residual = d->residual; a.out_scale = d->out_scale; a.y = d->y;
  a.y2 = d->y2; a.y2_shift = d->y2_shift; a.y2_scale = d->y2_scale; a.y2_act = d->y2_act;
  a.y2_shift_stride = d->y2_shift_stride > 0 ? d->y2_shift_stride : (d->y2_shift_stride < 0 ? 0 : d->Cout);
  a.y2_scale_stride = d->y2_scale_stride > 0 ? d->y2_scale_stride : (d->y2_scale_stride < 0 ? 0 : d->Cout);
  a.w_vec4 = (a.cout_g % 4 == 0) && aligned16(d->w_packed);
  a.vec4 = epilogue_vec_ok(d);
  hipStream_t st = as_stream(stream);

  a.ksplit = 1; a.split_ch = a.cin_g; a.part = nullptr;
  if (d->ksize == 3) {  // tile configuration and split-K factor: plan_conv3 (shared with the staged kernel)
    Conv3Plan plan = plan_conv3(d);
    const size_t need = (size_t)plan.ksplit * d->B * d->Cout * d->N * d->H * d->W * sizeof(float);
    if (plan.ksplit > 1 && d->splitk_ws && (size_t)d->splitk_ws_bytes >= need && aligned16(d->splitk_ws)) {
      a.ksplit = plan.ksplit; a.split_ch = a.cin_g / plan.ksplit; a.part = static_cast<float*>(d->splitk_ws);
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
  const bool c64 = a.cout_g % 64 == 0;
  if (c64) return launch<1, 2, 2, 8, 4, 8, 8>(a, st);
  return d->W >= 16 ? launch<1, 4, 1, 8, 4, 8, 16>(a, st) : launch<1, 2, 1, 8, 4, 8, 8>(a, st);
}
