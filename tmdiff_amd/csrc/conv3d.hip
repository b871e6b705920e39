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
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

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
  int tiles_n, tiles_h, tiles_w, tiles_co;  // tiles_co per group
  int w_vec4;                               // cout_g % 4 == 0 -> 16-byte weight loads
  unsigned total_blocks;
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
  static constexpr int STAGE = (LDS_IN + LDS_W + 3) / 4 * 4;   // floats per pipeline stage (16-B aligned)
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
template <int KS, int NS, int MSUB, int KC, int TN, int TH, int TW, bool FAST>
__global__ void __launch_bounds__(256, 2) conv3d_mfma_kernel(const ConvArgs a) {
  using G = Geo<KS, NS, MSUB, KC, TN, TH, TW>;
  constexpr int CO = G::CO;
  __shared__ __attribute__((aligned(16))) float lds[2 * G::STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, khalf = lane >> 5;

  // ---- which tile -------------------------------------------------------------------------
  unsigned id = xcd_remap(blockIdx.x, a.total_blocks);
  const int co_tile = id % a.tiles_co; id /= a.tiles_co;
  const int tw_i = id % a.tiles_w; id /= a.tiles_w;
  const int th_i = id % a.tiles_h; id /= a.tiles_h;
  const int tn_i = id % a.tiles_n; id /= a.tiles_n;
  const int g = id % a.groups;
  const int b = id / a.groups;
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

  // prefetch registers of the chunk in flight
  float xr[KC][G::EPT];
  float wr[G::W4][4];  // (scalars, not float4[]: an array of vectors is left in scratch by SROA here)
  float shr[KC], scr[KC];
  bool cval[KC];

  // Software pipeline, one barrier per chunk, two LDS stages.  Iteration `it`:
  //   1. issue the global loads of chunk it+1 (registers xr / wr; they fly under the MFMAs),
  //   2. run the K-steps of chunk it from stage it&1, fetching the LDS operands one K-step ahead,
  //   3. hand chunk it+1 to stage (it+1)&1 in NPIECE small pieces (prologue math + ds_write) that are
  //      slotted between the K-steps of the second half of step 2, i.e. issued in the shadow of MFMAs,
  //   4. barrier.
  // All loads are unconditional (out-of-range elements read a clamped, valid address and are zeroed by a
  // select when staged): a branch around a load would split the MFMA basic block and force vmcnt(0) waits.
  constexpr int NPIECE = KC * G::EPT + G::W4;
  constexpr int KSTEPS = (KC / 2) * G::TAPS;
  constexpr int PIECE0 = KSTEPS > NPIECE ? KSTEPS - NPIECE : 0;  // first K-step that carries a piece

  auto issue_loads = [&](int cn) {
#pragma unroll
    for (int ci = 0; ci < KC; ++ci) {
      const int cl = cn + ci;  // channel within group
      cval[ci] = FAST || cl < a.cin_g;
      const int cg = g * a.cin_g + (cval[ci] ? cl : 0);  // channel within the concatenated input (clamped)
      shr[ci] = a.in_shift ? a.in_shift[(long)b * a.shift_stride + cg] : 0.f;
      scr[ci] = a.in_scale ? a.in_scale[(long)b * a.scale_stride + cg] : 1.f;
      int cs = cg, segc = a.seg_c[0];  // which concat segment holds channel cg (no dynamic kernarg indexing)
      const float* base = a.seg_x[0];
      if (a.nseg > 1 && cs >= segc) {
        cs -= segc; base = a.seg_x[1]; segc = a.seg_c[1];
        if (a.nseg > 2 && cs >= segc) { cs -= segc; base = a.seg_x[2]; segc = a.seg_c[2]; }
      }
      const float* src = base + ((long)b * segc + cs) * plane;
#pragma unroll
      for (int i = 0; i < G::EPT; ++i) xr[ci][i] = src[goff[i]];
    }
#pragma unroll
    for (int j = 0; j < G::W4; ++j) {
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

  // piece `pc` of the register -> LDS hand-off of the chunk starting at channel cn
  auto stage_piece = [&](int pc, float* st, int cn) {
    if (pc < KC * G::EPT) {
      const int ci = pc / G::EPT, i = pc % G::EPT;
      const int e = tid + 256 * i;
      if (e < G::TILE_ELEMS) {
        float t = xr[ci][i] + shr[ci];
        if (a.in_act) t = tmdiff::silu_f(t);
        t *= scr[ci];
        if (a.in_mask && cval[ci])  // dropout mask (training only): read here, not prefetched
          t *= a.in_mask[((long)b * a.Cin + g * a.cin_g + cn + ci) * plane + goff[i]];
        // halo / out-of-range elements are exactly zero: the conv pads the ACTIVATED tensor
        st[ci * G::TILE_ELEMS + e] = (gok[i] && cval[ci]) ? t : 0.f;
      }
    } else {
      const int j = pc - KC * G::EPT;
      const int row = wrow0 + j * WSTEP;
      if (row < KC * G::TAPS)
        *reinterpret_cast<float4*>(st + G::LDS_IN + row * CO + wcol) = make_float4(wr[j][0], wr[j][1], wr[j][2], wr[j][3]);
    }
  };

  // prologue: chunk 0 -> stage 0
  issue_loads(0);
#pragma unroll
  for (int pc = 0; pc < NPIECE; ++pc) stage_piece(pc, lds, 0);
  __syncthreads();

  for (int it = 0;; ++it) {
    const int cn = (it + 1) * KC;  // first channel of the chunk to prefetch
    const bool more = cn < a.cin_g;
    if (more) issue_loads(cn);
    const float* st = lds + (it & 1) * G::STAGE;
    float* st_next = lds + ((it + 1) & 1) * G::STAGE;

    auto fetch = [&](int ks, float (&av)[MSUB], float (&bv)[NS]) {
      const int kp = ks / G::TAPS, tap = ks % G::TAPS;
      const int dn = tap / (KS * KS), dh = (tap / KS) % KS, dw = tap % KS;
      const int toff = (dn * G::HH + dh) * G::HW + dw;
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
    fetch(0, av[0], bv[0]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks + 1 < KSTEPS) fetch(ks + 1, av[(ks + 1) & 1], bv[(ks + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this step's MFMAs (hipcc sinks it otherwise)
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int m = 0; m < MSUB; ++m)
          acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ks & 1][m], bv[ks & 1][s], acc[s][m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // hand-off pieces ride behind the MFMAs just issued (on the last chunk they stage stale registers into the
      // unused stage: harmless, and it keeps this block free of branches)
      if (ks >= PIECE0) {
#pragma unroll
        for (int pc = (ks - PIECE0) * NPIECE / (KSTEPS - PIECE0); pc < (ks + 1 - PIECE0) * NPIECE / (KSTEPS - PIECE0); ++pc)
          stage_piece(pc, st_next, cn);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!more) break;
    __syncthreads();
  }

  // ---- epilogue: bias, residual, scale; D layout: col = lane&31 (position), row = channel --------
  // Loads first (bias rows, then all residual elements of a sub-tile), then the stores: no load->store chains.
#pragma unroll
  for (int m = 0; m < MSUB; ++m) {
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;  // channel within group
      const int cglob = g * a.cout_g + min(col, a.cout_g - 1);
      bias_r[r] = a.bias ? a.bias_scale * a.bias[cglob] : 0.f;
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
        if (ok) a.y[obase + row * plane] = (acc[s][m][r] + bias_r[r] + res[r]) * a.out_scale;
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

template <int KS, int NS, int MSUB, int KC, int TN, int TH, int TW>
int launch(ConvArgs& a, hipStream_t st) {
  constexpr int CO = 32 * MSUB;
  a.tiles_n = (a.N + TN - 1) / TN;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  a.tiles_co = (a.cout_g + CO - 1) / CO;
  const long blocks = (long)a.B * a.groups * a.tiles_n * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  const bool fast = a.cin_g % KC == 0 && a.cout_g % CO == 0 && a.w_vec4;
  if (fast)
    conv3d_mfma_kernel<KS, NS, MSUB, KC, TN, TH, TW, true><<<(unsigned)blocks, 256, 0, st>>>(a);
  else
    conv3d_mfma_kernel<KS, NS, MSUB, KC, TN, TH, TW, false><<<(unsigned)blocks, 256, 0, st>>>(a);
  return tmdiff::check_launch("conv3d_fwd");
}

}  // namespace

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
  TMDIFF_REQUIRE(d->w_packed && d->y, "conv3d_fwd: NULL weights/output");
  TMDIFF_REQUIRE((long)d->N * d->H * d->W < (1L << 31), "conv3d_fwd: plane too large for 32-bit offsets");

  ConvArgs a;
  a.B = d->B; a.N = d->N; a.H = d->H; a.W = d->W;
  a.Cin = d->Cin; a.Cout = d->Cout; a.groups = d->groups;
  a.cin_g = d->Cin / d->groups; a.cout_g = d->Cout / d->groups;
  a.nseg = d->nseg;
  for (int i = 0; i < 3; ++i) { a.seg_c[i] = i < d->nseg ? d->seg_c[i] : 0; a.seg_x[i] = i < d->nseg ? d->seg_x[i] : nullptr; }
  a.wp = d->w_packed; a.bias = d->bias; a.bias_scale = d->bias_scale;
  a.in_shift = d->in_shift; a.in_scale = d->in_scale; a.in_mask = d->in_mask; a.in_act = d->in_act;
  a.shift_stride = d->in_shift_stride > 0 ? d->in_shift_stride : (d->in_shift_stride < 0 ? 0 : d->Cin);
  a.scale_stride = d->in_scale_stride > 0 ? d->in_scale_stride : (d->in_scale_stride < 0 ? 0 : d->Cin);
  a.residual = d->residual; a.out_scale = d->out_scale; a.y = d->y;
  a.w_vec4 = (a.cout_g % 4 == 0) && aligned16(d->w_packed);
  hipStream_t st = as_stream(stream);

  // Tile choice.  <NS, MSUB>: 2x2 = 256 positions x 64 channels when the channel count is a multiple of 64
  // (the packed rows are then sub-tile interleaved, see packed_col); otherwise 32-channel tiles, 512 positions
  // (4x1) on planes at least 16 wide, else 256 positions (2x1).
  const bool c64 = a.cout_g % 64 == 0;
  if (d->ksize == 3) {
    if (c64) return launch<3, 2, 2, 4, 4, 8, 8>(a, st);
    return d->W >= 16 ? launch<3, 4, 1, 4, 4, 8, 16>(a, st) : launch<3, 2, 1, 4, 4, 8, 8>(a, st);
  }
  if (c64) return launch<1, 2, 2, 8, 4, 8, 8>(a, st);
  return d->W >= 16 ? launch<1, 4, 1, 8, 4, 8, 16>(a, st) : launch<1, 2, 1, 8, 4, 8, 8>(a, st);
}
