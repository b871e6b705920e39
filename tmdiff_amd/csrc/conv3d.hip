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

template <int KS, int CO, int KC, int TN, int TH, int TW>
struct Geo {
  static constexpr int TAPS = KS * KS * KS;
  static constexpr int HALO = KS / 2;
  static constexpr int HN = TN + 2 * HALO, HH = TH + 2 * HALO, HW = TW + 2 * HALO;
  static constexpr int TILE_ELEMS = HN * HH * HW;
  static constexpr int POS = TN * TH * TW;
  static constexpr int EPT = (TILE_ELEMS + 255) / 256;  // staged elements per thread per channel
  static constexpr int LDS_IN = KC * TILE_ELEMS;
  static constexpr int LDS_W = KC * TAPS * CO;
  static constexpr int MSUB = CO / 32;
  static_assert(POS == 256, "a workgroup tile is 256 positions (4 waves x 2 sub-tiles x 32)");
  static_assert(KC % 2 == 0 && CO % 32 == 0, "K step is 2 channels; channel sub-tiles are 32 wide");
};

// XCD-aware block id: blocks b and b+8 share an XCD (round-robin dispatch), so hand each XCD a
// contiguous run of logical tiles -- neighbouring tiles (same input box, other channel tile;
// adjacent boxes sharing halo lines) then hit in that XCD's L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, k = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

template <int KS, int CO, int KC, int TN, int TH, int TW>
__global__ void __launch_bounds__(256) conv3d_mfma_kernel(const ConvArgs a) {
  using G = Geo<KS, CO, KC, TN, TH, TW>;
  __shared__ __attribute__((aligned(16))) float lds[G::LDS_IN + G::LDS_W];
  float* lds_in = lds;
  float* lds_w = lds + G::LDS_IN;

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

  // ---- per-lane operand offsets ---------------------------------------------------------------
  int boff[2];  // float index of this lane's position inside the haloed box, tap (0,0,0)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int p = (wv * 2 + s) * 32 + l31;
    const int pw = p % TW, ph = (p / TW) % TH, pn = p / (TW * TH);
    boff[s] = (pn * G::HH + ph) * G::HW + pw + khalf * G::TILE_ELEMS;
  }
  const int aoff = khalf * G::TAPS * CO + l31;

  f32x16 acc[2][G::MSUB];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int m = 0; m < G::MSUB; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][m][r] = 0.f;

  const float* wp_g = a.wp + (long)g * a.cin_g * G::TAPS * a.cout_g;

  for (int c0 = 0; c0 < a.cin_g; c0 += KC) {
    if (c0) __syncthreads();  // everyone is done reading the previous chunk
    // ---- stage the haloed input box, prologue applied once per element -------------------------
#pragma unroll 1
    for (int ci = 0; ci < KC; ++ci) {
      const int cl = c0 + ci;           // channel within group
      const bool cvalid = cl < a.cin_g;
      int cg = g * a.cin_g + cl;        // channel within the concatenated input
      const float* src = nullptr;
      const float* msk = nullptr;
      float sh = 0.f, sc = 1.f;
      if (cvalid) {
        if (a.in_shift) sh = a.in_shift[(long)b * a.shift_stride + cg];
        if (a.in_scale) sc = a.in_scale[(long)b * a.scale_stride + cg];
        if (a.in_mask) msk = a.in_mask + ((long)b * a.Cin + cg) * plane;
        int cs = cg, segc = a.seg_c[0];  // which concat segment holds channel cg (no dynamic kernarg indexing)
        const float* base = a.seg_x[0];
        if (a.nseg > 1 && cs >= segc) {
          cs -= segc; base = a.seg_x[1]; segc = a.seg_c[1];
          if (a.nseg > 2 && cs >= segc) { cs -= segc; base = a.seg_x[2]; segc = a.seg_c[2]; }
        }
        src = base + ((long)b * segc + cs) * plane;
      }
      float v[G::EPT];
#pragma unroll
      for (int i = 0; i < G::EPT; ++i) v[i] = (cvalid && gok[i]) ? src[goff[i]] : 0.f;
#pragma unroll
      for (int i = 0; i < G::EPT; ++i) {
        const int e = tid + 256 * i;
        if (e < G::TILE_ELEMS) {
          float t = 0.f;
          if (cvalid && gok[i]) {
            t = v[i] + sh;
            if (a.in_act) t = tmdiff::silu_f(t);
            t *= sc;
            if (msk) t *= msk[goff[i]];
          }
          lds_in[ci * G::TILE_ELEMS + e] = t;
        }
      }
    }
    // ---- stage the weight slab: rows [c0*TAPS, (c0+KC)*TAPS) x columns [co0, co0+CO) -------------
    {
      constexpr int ROWS = KC * G::TAPS;
      constexpr int V4 = CO / 4;
      for (int idx = tid; idx < ROWS * V4; idx += 256) {
        const int row = idx / V4, c4 = (idx % V4) * 4;
        const int cl = c0 + row / G::TAPS;
        const long grow = (long)c0 * G::TAPS + row;
        float4 wv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cl < a.cin_g) {
          const float* p = wp_g + grow * a.cout_g + co0 + c4;
          if (a.w_vec4 && co0 + c4 + 3 < a.cout_g) {
            wv4 = *reinterpret_cast<const float4*>(p);
          } else {
            if (co0 + c4 + 0 < a.cout_g) wv4.x = p[0];
            if (co0 + c4 + 1 < a.cout_g) wv4.y = p[1];
            if (co0 + c4 + 2 < a.cout_g) wv4.z = p[2];
            if (co0 + c4 + 3 < a.cout_g) wv4.w = p[3];
          }
        }
        *reinterpret_cast<float4*>(lds_w + row * CO + c4) = wv4;
      }
    }
    __syncthreads();
    // ---- K loop over the chunk: KC/2 channel pairs x taps --------------------------------------
#pragma unroll
    for (int kp = 0; kp < KC / 2; ++kp) {
#pragma unroll
      for (int tap = 0; tap < G::TAPS; ++tap) {
        const int dn = tap / (KS * KS), dh = (tap / KS) % KS, dw = tap % KS;
        const int toff = (dn * G::HH + dh) * G::HW + dw;
        float av[G::MSUB], bv[2];
#pragma unroll
        for (int m = 0; m < G::MSUB; ++m) av[m] = lds_w[aoff + (kp * 2 * G::TAPS + tap) * CO + m * 32];
#pragma unroll
        for (int s = 0; s < 2; ++s) bv[s] = lds_in[boff[s] + kp * 2 * G::TILE_ELEMS + toff];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int m = 0; m < G::MSUB; ++m)
            acc[s][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[s], acc[s][m], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: bias, residual, scale; D layout: col = lane&31 (position), row = channel --------
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int p = (wv * 2 + s) * 32 + l31;
    const int n = n0 + p / (TW * TH), h = h0 + (p / TW) % TH, w = w0 + p % TW;
    if (n >= a.N || h >= a.H || w >= a.W) continue;
    const long sp = ((long)n * a.H + h) * a.W + w;
#pragma unroll
    for (int m = 0; m < G::MSUB; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int col = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;  // channel within group
        if (col >= a.cout_g) continue;
        const int cglob = g * a.cout_g + col;
        const long o = ((long)b * a.Cout + cglob) * plane + sp;
        float t = acc[s][m][r];
        if (a.bias) t += a.bias_scale * a.bias[cglob];
        if (a.residual) t += a.residual[o];
        a.y[o] = t * a.out_scale;
      }
    }
  }
}

// packed[g][ci][tap][co] <- w[g*cout_g + co][ci][tap]              (mode 0, forward)
// packed[g][co][taps-1-tap][ci] <- w[g*cout_g + co][ci][tap]       (mode 1, data gradient: roles swapped)
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
      dst = (((long)g * cin_g + ci) * taps + tap) * cout_g + co;
    else
      dst = (((long)g * cout_g + co) * taps + (taps - 1 - tap)) * cin_g + ci;
    packed[dst] = w[i];
  }
}

template <int KS, int CO, int KC, int TN, int TH, int TW>
int launch(ConvArgs& a, hipStream_t st) {
  a.tiles_n = (a.N + TN - 1) / TN;
  a.tiles_h = (a.H + TH - 1) / TH;
  a.tiles_w = (a.W + TW - 1) / TW;
  a.tiles_co = (a.cout_g + CO - 1) / CO;
  const long blocks = (long)a.B * a.groups * a.tiles_n * a.tiles_h * a.tiles_w * a.tiles_co;
  if (blocks <= 0 || blocks > 0x7fffffffL) return tmdiff::fail(TMDIFF_E_INVALID, "conv3d: grid of %ld blocks", blocks);
  a.total_blocks = (unsigned)blocks;
  conv3d_mfma_kernel<KS, CO, KC, TN, TH, TW><<<(unsigned)blocks, 256, 0, st>>>(a);
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

  // Channel-tile choice: 64-wide tiles halve the staging per MFMA; fall back to 32 when the
  // layer has few channels or too few tiles to fill 256 CUs several times over.
  const long sp_tiles = (long)d->B * d->groups * ((d->N + 3) / 4) * ((d->H + 7) / 8) * ((d->W + 7) / 8);
  const bool wide = a.cout_g > 32 && sp_tiles * ((a.cout_g + 63) / 64) >= 2048;
  if (d->ksize == 3) {
    return wide ? launch<3, 64, 4, 4, 8, 8>(a, st) : launch<3, 32, 4, 4, 8, 8>(a, st);
  }
  return wide ? launch<1, 64, 32, 4, 8, 8>(a, st) : launch<1, 32, 32, 4, 8, 8>(a, st);
}
