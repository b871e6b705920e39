/* tmdiff_hip.h -- C ABI of libtmdiff_hip.so: the MI355X (gfx950) kernels of the TMDiff
 * denoising hot path.
 *
 * The reference (codgodtao/TMDiff) has no FFI of its own: its hot path runs through
 * PyTorch ATen.  Each entry point below therefore replaces the ATen call sequence of one
 * reference site (file:line given per function).  Conventions:
 *   - plain pointers + sizes + a hipStream_t passed as void*; no torch types;
 *   - every function returns int: 0 = TMDIFF_OK, negative = TMDIFF_E_*; nothing throws
 *     across the boundary; tmdiff_last_error_string() describes the last failure of the
 *     calling thread;
 *   - the library never allocates or frees device memory and never synchronises: the
 *     caller owns inputs, outputs and workspaces (so calls are HIP-graph capturable);
 *   - all tensors are dense fp32, 5-D activations are [B, C, N, H, W] (N = spectral
 *     bands = the conv "depth" axis), row-major, 16-byte aligned base pointers.
 */
#ifndef TMDIFF_HIP_H
#define TMDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TMDIFF_ABI_VERSION 6

#define TMDIFF_OK 0
#define TMDIFF_E_INVALID (-1)     /* bad argument / shape */
#define TMDIFF_E_UNSUPPORTED (-2) /* valid but not implemented configuration */
#define TMDIFF_E_LAUNCH (-3)      /* HIP reported an error at launch */

typedef void* tmdiff_stream_t; /* hipStream_t */

int tmdiff_version(void);
const char* tmdiff_last_error_string(void);

/* ------------------------------------------------------------------------------------
 * conv3d, kernel 3x3x3 (pad 1) or 1x1x1 (pad 0), stride 1, groups 1 or 3.
 * Replaces nn.Conv3d / F.conv3d and modulated_conv3d
 * (GeneralModel/Hyper_unet_general.py:51-77, :161-164, :224-231, :260, :344-361) together
 * with the elementwise ops the reference runs around them:
 *   prologue  x' = act(x + in_shift[b,c]) * in_scale[b,c] * in_mask[b,c,n,h,w]
 *             (temb shift :239, :401; Swish :242, :245, :371, :402; modulation :69 folded
 *             from the weights onto the input channels; Dropout mask :243, :246, :403)
 *   epilogue  y = (conv(x') + bias_scale*bias[co] + residual) * out_scale
 *             (bias; skip additions :249, :408; the 2x of convH_0 :381 via bias_scale)
 * The input may be given as up to three channel segments that the reference would
 * torch.cat first (Hyper_unet_general.py:631-634); for groups == 3 segment g is group g's
 * input (the cat of the three wavelet high bands at :381).
 * ------------------------------------------------------------------------------------ */
typedef struct tmdiff_conv3d_desc {
  int32_t B, N, H, W;
  int32_t Cin, Cout;   /* totals over all groups */
  int32_t groups;      /* 1 or 3 */
  int32_t ksize;       /* 1 or 3 */
  int32_t nseg;        /* 1..3 input segments */
  int32_t seg_c[3];    /* channels per segment, sum == Cin (groups==3: each Cin/3) */
  const float* seg_x[3];
  const float* w_packed; /* from tmdiff_conv3d_pack_weights (or _bf16 for tmdiff_conv3d_fwd_bf16) */
  const float* bias;     /* [Cout] or NULL */
  float bias_scale;
  const float* in_shift; /* [B, Cin] (row stride in_shift_stride floats) or NULL */
  const float* in_scale; /* [B, Cin] (row stride in_scale_stride floats) or NULL */
  int32_t in_shift_stride; /* 0 = dense rows (Cin); > 0 = row stride in floats (a layer's slice of a bank of */
  int32_t in_scale_stride; /* projections); -1 = one row broadcast over the batch (one prompt for all samples) */
  const float* in_mask;  /* [B, Cin, N, H, W] multiplicative mask (dropout) or NULL */
  int32_t in_act;        /* 0 = identity, 1 = SiLU */
  const float* residual; /* [B, Cout, N, H, W] or NULL */
  float out_scale;
  float* y;              /* [B, Cout, N, H, W]; may be NULL when only y2 is wanted */
  /* Optional second output: the consumer's prologue applied to this convolution's result,
   *   y2 = act2(y + y2_shift[b,co]) * y2_scale[b,co]
   * so that the consumer reads a plain tensor (e.g. conv20 -> conv21 of a ResBlock, Hyper_unet_general.py:244-248:
   * conv21's SiLU and text modulation are applied where conv20's result is produced).  y2_bf16 == 0: fp32
   * [B, Cout, N, H, W]; != 0: bf16 units of 8 channels [B][Cout/8][N*H*W] as tmdiff_conv3d_fwd_bf16 packs them
   * (only the bf16 entry point writes that form; Cout/groups a multiple of 32).  Strides as in_shift_stride. */
  float* y2;
  const float* y2_shift;
  const float* y2_scale;
  int32_t y2_shift_stride, y2_scale_stride;
  int32_t y2_act;
  int32_t y2_bf16;
  /* != 0: seg_x[0] is not fp32 but the bf16 units [B][Cin/8][N*H*W] a producer wrote as its y2 (prologue already
   * applied): nseg 1, no shift / scale / act / mask.  Only tmdiff_conv3d_fwd_bf16 (3x3x3) accepts it; it then skips
   * its pack pass and needs no workspace. */
  int32_t x_bf16;
  /* Optional split-K workspace (fp32 3x3x3 entry points tmdiff_conv3d_fwd / _fwd_staged only).  A launch whose grid
   * would leave most of the 256 CUs idle (single images: B = 1 at every level of the UNet; the 8x8 / 16x16 levels at
   * small batches) is split over the input channels: `ksplit` workgroups per output tile accumulate disjoint channel
   * ranges into partial outputs [ksplit][B][Cout][N*H*W] here, and a second kernel sums them in a fixed order and
   * applies the epilogue (deterministic; the fused and the staged entry point split identically, so they still agree
   * bit for bit).  NULL = never split.  Size: tmdiff_conv3d_fwd_splitk_workspace_bytes(d) (0 = this launch does not
   * split); 16-byte aligned. */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
  /* In-kernel dropout (training; nn.Dropout(p) at Hyper_unet_general.py:230, :243-246, :349, :403): drop_p > 0 multiplies
   * the prologue output by a Bernoulli(1 - p) keep mask scaled by 1 / (1 - p) that is a pure function of
   * (drop_seed, element index ((b * Cin + c) * N*H*W + pos)) -- a counter-based hash (splitmix64 finaliser), so the
   * forward, tmdiff_conv3d_wgrad and tmdiff_conv3d_prologue_bwd regenerate the same mask from the same descriptor and no
   * mask tensor exists.  in_mask (a caller-supplied mask tensor, parity runs) must then be NULL.  fp32 entry points only. */
  uint64_t drop_seed;
  float drop_p;
  /* Optional DEVICE word added to drop_seed when the kernel starts (NULL: none).  A launch recorded into a HIP graph is
   * replayed with the arguments it was captured with; with the per-step part of the seed in device memory (one 8-byte word
   * the step bumps before the graph runs) every replay still draws a fresh mask -- and forward, weight gradient and
   * prologue backward of one step still agree, since all three read the same word.  (ABI v6) */
  const uint64_t* drop_seed_dev;
  /* Optional folded "residual convolution" (tmdiff_conv3d_wf_fwd only; ABI v6): y = (conv(x') + bias + rc_w^T rc_x) * out_scale,
   * i.e. the 1x1x1 res_conv of a ResBlock (Hyper_unet_general.py:231, :248) on the block's RAW input rc_x [B, rc_cin, N, H, W],
   * accumulated on the matrix pipe where its consumer conv21 would have added it -- no launch of its own, its result never
   * written and read back.  rc_w = the weight [Cout, rc_cin, 1, 1, 1] as PyTorch holds it (contiguous, NOT packed); the
   * caller adds res_conv's bias into `bias`.  Needs groups 1, rc_cin % 32 == 0 (at most 512), residual NULL and planes wider
   * than 8 columns; else TMDIFF_E_UNSUPPORTED.  (A grid that splits its input channels adds it to the partial sums of range 0.) */
  const float* rc_x;
  const float* rc_w;
  int32_t rc_cin;
  /* Optional third output (tmdiff_conv3d_wf_fwd only; ABI v6): the HALVED LL BAND of y, (a + b + c + d) / 4 over every 2 x 2 pixel
   * block, [B, Cout, N, H/2, W/2] -- all a down block reads of its ResBlock's raw output (its Conv_2 path,
   * Hyper_unet_general.py:374, :390, :396), so that neither y nor an LL-only DWT pass over it is needed.  Needs y == NULL and
   * y2 != NULL, 8 bands, even H, W % 4 == 0, planes wider than 8 columns, an unsplit grid; else TMDIFF_E_UNSUPPORTED. */
  float* y_ll;
  /* ... and with y_hi[0..2] != NULL (then y2 == NULL): the WHOLE Haar transform of y instead of y (tmdiff_conv3d_wf_fwd only; ABI v6) --
   * y_hi = LH, HL, HH = (a - b + c - d) / 2, (a + b - c - d) / 2, (a - b - c + d) / 2 of every 2 x 2 block (a b / c d), and y_ll = the
   * halved LL band passed through the second output's prologue, act2(LL / 2 + y2_shift) * y2_scale: what a down block whose high
   * bands are kept makes of its Conv_0 output (DWT, then Conv_1's prologue on the LL band; Hyper_unet_general.py:388-396,
   * DWT_IDWT_Functions.py:47-57) without a full-resolution tensor or a transform pass.  Same shape conditions as y_ll. */
  float* y_hi[3];
  /* Optional by-product of a 1x1x1 convolution on the bandwidth kernel (tmdiff_conv3d_fwd, ksize 1; ABI v6): xp_out [B, Cin, N, H, W]
   * receives act(x + xp_shift[b, c]) of every element of the (possibly segmented) input -- the prologue output that ANOTHER
   * convolution of the same input wants (a ResBlock's conv20 beside its res_conv, Hyper_unet_general.py:243-248: both read the
   * concatenated block input, one raw, one through SiLU(x + Dense(e))), written by the kernel that loads every element anyway.
   * xp_shift: [B, Cin] (row stride xp_shift_stride as in_shift_stride) or NULL; xp_act != 0: SiLU.  Taken by the 16-byte form of
   * the kernel (planes of multiples of 4 positions, 16-byte aligned tensors, at least 512 tiles of 512 positions; any prologue of
   * its own except an activation) and by the small-grid form (fewer than 512 tiles of 256 positions, at least 128 input channels
   * per group, a raw input: no in_shift / in_scale / in_act); a sample of fewer than 2^30 elements.  tmdiff_conv3d_fwd_xp_supported()
   * answers for a descriptor; any other launch: TMDIFF_E_UNSUPPORTED. */
  float* xp_out;
  const float* xp_shift;
  int32_t xp_shift_stride, xp_act;
  /* != 0 (tmdiff_conv3d_wf_fwd only, even H, W % 4 == 0, a grid that does not split its input channels): y2 is written in
   * "space to depth" form [B, 4 Cout, N, H/2, W/2], channel 4 co + 2 ph + pw holding y2[co][n][2i + ph][2j + pw] -- the input
   * form of tmdiff_conv3d_wfll_fwd (the down blocks' Conv_0 + LL band, Hyper_unet_general.py:371-372, :389, :396). */
  int32_t y2_s2d;
} tmdiff_conv3d_desc;

/* w [Cout, Cin/groups, k, k, k] (PyTorch layout) -> packed [g][ci][tap][co] used by the
 * kernels.  mode 0: forward weights.  mode 1: weights of the data-gradient convolution
 * (taps flipped, ci/co swapped): conv3d_fwd(dy, packed_mode1) == dL/dx'.
 * packed must hold Cout*Cin/groups*k^3 floats. */
int tmdiff_conv3d_pack_weights(const float* w, float* packed, int32_t Cout, int32_t Cin, int32_t ksize,
                               int32_t groups, int32_t mode, tmdiff_stream_t stream);
/* The same for a whole list of weight tensors in ONE launch, both packings at once (a training step re-packs every weight
 * for its forward and its data-gradient convolution after each optimizer update).  `entries_dev` is a DEVICE array; workgroup
 * k works on entry chunk_tensor_dev[k]; chunk_index_dev[k] = tile number within that entry, with bit 30 set for the tiles of
 * the data-gradient packing: an entry needs tmdiff_conv3d_pack_weights_multi_chunks(...) workgroups, the first *n_type_a of
 * them forward-packing tiles (numbered 0..), the rest data-gradient tiles (numbered 0.. | 1<<30).  ksize 1 or 3.  Either
 * destination may be NULL. */
typedef struct tmdiff_pack_entry {
  const float* w;       /* [Cout, Cin/groups, k, k, k] */
  float* packed_fwd;    /* mode 0 */
  float* packed_dgrad;  /* mode 1 */
  int32_t Cout, Cin, ksize, groups;
} tmdiff_pack_entry;
int32_t tmdiff_conv3d_pack_weights_multi_chunks(int32_t Cout, int32_t Cin, int32_t groups, int32_t* n_type_a);
int tmdiff_conv3d_pack_weights_multi(const tmdiff_pack_entry* entries_dev, const int32_t* chunk_tensor_dev,
                                     const int32_t* chunk_index_dev, int32_t n_chunks, tmdiff_stream_t stream);
int tmdiff_conv3d_fwd(const tmdiff_conv3d_desc* d, tmdiff_stream_t stream);
/* 1 when tmdiff_conv3d_fwd writes the by-product d->xp_out for this descriptor (else it would fail with TMDIFF_E_UNSUPPORTED) */
int tmdiff_conv3d_fwd_xp_supported(const tmdiff_conv3d_desc* d);
/* bytes of d->splitk_ws this convolution would use (0: its grid fills the chip, or the shape is not split) */
size_t tmdiff_conv3d_fwd_splitk_workspace_bytes(const tmdiff_conv3d_desc* d);

/* ---- staged variant of tmdiff_conv3d_fwd (same arithmetic, exact fp32, bit-identical results) ----------------
 * Two launches instead of one: the prologue output x' (and the concatenation of the segments) is written once into
 * `workspace` ([B, Cin, N, H, W] fp32, tmdiff_conv3d_fwd_staged_workspace_bytes(d) bytes; 0 and NULL allowed when
 * the input is a single tensor without shift / scale / act / mask), then a convolution whose operands go
 * L2/HBM -> LDS directly (global_load_lds) so that its instruction stream is MFMAs and operand reads only.
 * The kernel itself is 3-6 % faster than the fused one; with the prologue pass it wins where that pass is absent or
 * amortised (plain inputs such as every data-gradient convolution, >= 128 output channels).  3x3x3 only; shapes it does
 * not take (Cin/groups not a multiple of 4, Cout/groups not a multiple of 32, ksize 1) return TMDIFF_E_UNSUPPORTED --
 * use tmdiff_conv3d_fwd. */
int tmdiff_conv3d_fwd_staged_supported(const tmdiff_conv3d_desc* d);
size_t tmdiff_conv3d_fwd_staged_workspace_bytes(const tmdiff_conv3d_desc* d);
int tmdiff_conv3d_fwd_staged(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream);

/* ---- 3x3x3 convolution followed by the scaled Haar LL band, as one strided convolution (exact fp32) -----------
 * Replaces the pair  h = Conv_0(x') ; hLL = dwt(h).LL * ll_scale  of WaveletUPorDown(down=True) where the high bands are
 * not used (reference GeneralModel/Hyper_unet_general.py:371-372, :389, :396; the main branch of WavBEST.forward).  The LL
 * band is linear and local, so the pair equals a convolution with a 3x4x4 kernel and stride (1,2,2) whose weights are
 * sums of the 3x3x3 ones: 48 instead of 4 x 27 multiply-adds per output, same numbers up to fp32 summation order.
 *   d         : the descriptor of the 3x3x3 convolution (N, H, W = INPUT extents, H and W even; one plain fp32 input
 *               without shift / scale / act / mask; groups 1; Cin % 2 == 0, Cout % 64 == 0; bias as for the convolution);
 *               y / y2 / residual are [B, Cout, N, H/2, W/2] and follow the usual epilogue
 *               out = (LL(conv(x) + bias) * ll_scale + residual) * out_scale,  y2 = act2(out + shift2) * scale2.
 *   w_packed  : from tmdiff_conv3d_ll_pack_weights(w [Cout, Cin, 3, 3, 3], ..., ll_scale)
 *               (tmdiff_conv3d_ll_packed_bytes bytes; 0 = shape not supported).
 * Shapes it does not take return TMDIFF_E_UNSUPPORTED (tmdiff_conv3d_ll_supported says so beforehand): run the
 * convolution and tmdiff_haar_dwt2d instead.
 * Small grids are split over the input channels like tmdiff_conv3d_fwd when d->splitk_ws lends
 * tmdiff_conv3d_ll_splitk_workspace_bytes(d) bytes (0: the grid fills the chip). */
int tmdiff_conv3d_ll_supported(const tmdiff_conv3d_desc* d);
size_t tmdiff_conv3d_ll_splitk_workspace_bytes(const tmdiff_conv3d_desc* d);
size_t tmdiff_conv3d_ll_packed_bytes(int32_t Cout, int32_t Cin);
int tmdiff_conv3d_ll_pack_weights(const float* w, float* packed, int32_t Cout, int32_t Cin, float ll_scale,
                                  tmdiff_stream_t stream);
int tmdiff_conv3d_ll_fwd(const tmdiff_conv3d_desc* d, float ll_scale, tmdiff_stream_t stream);

/* ---- 3x3x3 convolution, Winograd F(4,3) / F(2,3) along the band axis (exact-fp32 matrix cores) --------------------
 * 2x (N % 4 == 0: six transformed planes per four bands) or 1.5x (N % 2 == 0: four planes per two bands) fewer multiply-adds.
 * Same descriptor and epilogue as tmdiff_conv3d_fwd (fp32, groups 1 or 3, N even, W % 4 == 0, Cin/groups % 2 == 0,
 * Cout/groups % 32 == 0, no mask tensor (d->drop_p in-kernel dropout is taken); segments and the shift / scale / act
 * prologue are taken).  `workspace` receives the transformed input (tmdiff_conv3d_wino_workspace_bytes(d) = 1.5-2x the
 * input bytes plus a border); d->w_packed comes from tmdiff_conv3d_wino_pack_weights with planes =
 * tmdiff_conv3d_wino_planes(N) (tmdiff_conv3d_wino_packed_bytes bytes).  Results agree with tmdiff_conv3d_fwd to 1e-6
 * relative L2 (F(4,3)) / 4e-7 (F(2,3)). */
int tmdiff_conv3d_wino_supported(const tmdiff_conv3d_desc* d);
int32_t tmdiff_conv3d_wino_planes(int32_t N);                     /* 6, 4, or 0 (odd N) */
int64_t tmdiff_conv3d_wino_blocks(const tmdiff_conv3d_desc* d);   /* workgroups of its grid (no split-K: keep small grids on tmdiff_conv3d_fwd) */
size_t tmdiff_conv3d_wino_workspace_bytes(const tmdiff_conv3d_desc* d);
size_t tmdiff_conv3d_wino_packed_bytes(int32_t Cout, int32_t Cin, int32_t groups, int32_t planes);
/* mode bit 0 clear: w = [Cout, Cin/groups, 3,3,3] of this convolution; set (data gradient): w is the FORWARD convolution's
 * weight [Cin, Cout/groups, 3,3,3] and this convolution (Cin -> Cout) is its transpose with mirrored taps, as
 * tmdiff_conv3d_pack_weights(mode 1).  mode bit 1 (value 2): natural column order (tmdiff_conv3d_wf_fwd) instead of the
 * interleaved 64-channel tiles of tmdiff_conv3d_wino_fwd */
int tmdiff_conv3d_wino_pack_weights(const float* w, float* packed, int32_t Cout, int32_t Cin, int32_t groups,
                                    int32_t mode, int32_t planes, tmdiff_stream_t stream);
/* Multi-tensor form, ONE launch for every (weight, mode) of a network (the finetune step re-packs the Winograd weights of its
 * ~50 convolutions in both forms after every optimizer step).  `entries_dev` is a DEVICE array; workgroup k packs weight rows
 * [chunk_index_dev[k] * tmdiff_conv3d_wino_pack_weights_multi_chunk(), ...) of entry chunk_tensor_dev[k] (a row = the 27 taps of one
 * (output, input channel) pair: numel(w) / 27 rows per entry).  Cout / Cin / groups are
 * those of the weight tensor w = [Cout, Cin/groups, 3,3,3]; mode / planes as tmdiff_conv3d_wino_pack_weights. */
typedef struct tmdiff_wino_pack_entry {
  const float* w;
  float* packed;
  int32_t Cout, Cin, groups, mode, planes, reserved;
} tmdiff_wino_pack_entry;
int32_t tmdiff_conv3d_wino_pack_weights_multi_chunk(void);
int tmdiff_conv3d_wino_pack_weights_multi(const tmdiff_wino_pack_entry* entries_dev, const int32_t* chunk_tensor_dev,
                                          const int32_t* chunk_index_dev, int32_t n_chunks, tmdiff_stream_t stream);
int tmdiff_conv3d_wino_fwd(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream);
/* stage 1: the input-transform pass alone; stage 2: the convolution alone on a workspace that holds it; 0: both */
int tmdiff_conv3d_wino_fwd_stage(const tmdiff_conv3d_desc* d, void* workspace, int32_t stage, tmdiff_stream_t stream);
/* ... and with the prologue output x' [B, Cin, N, H, W] (in-kernel dropout d->drop_p included) also written to xp_out by the
 * transform pass: the finetune path keeps it for the weight gradient */
int tmdiff_conv3d_wino_fwd_xp(const tmdiff_conv3d_desc* d, void* workspace, int32_t stage, float* xp_out, tmdiff_stream_t stream);
/* ... and with the transform chosen by the caller (planes 6 or 4 = what the weights were packed for; 0 = automatic): F(2,3)
 * has twice the tiles along the bands, so it still fills the chip where F(4,3) would not */
int tmdiff_conv3d_wino_fwd_planes(const tmdiff_conv3d_desc* d, void* workspace, int32_t stage, float* xp_out, int32_t planes,
                                  tmdiff_stream_t stream);

/* ---- 3x3x3 convolution, Winograd F(4,3) along the band axis with the input transform INSIDE the kernel -----------
 * (csrc/conv3d_wf.hip).  The whole band axis lies in one workgroup (N = 8: two tiles, N = 4: one), the kernel reads the plain
 * convolution input and forms v = B^T d in LDS between its MFMAs: no transformed copy of the input in HBM, no transform
 * pass.  Same descriptor and epilogue as tmdiff_conv3d_fwd (fp32, groups 1 or 3, N = 8 or 4, W % 4 == 0, Cin/groups % 2 == 0,
 * Cout/groups % 32 == 0, no mask tensor).  An input that is ONE plain tensor (no prologue / dropout: every convolution whose
 * producer applied the consumer's prologue, every data-gradient convolution) -- or, for groups = 3, three plain segments of
 * Cin/3 channels, one per group -- needs no workspace; otherwise `workspace`
 * (tmdiff_conv3d_wf_workspace_bytes(d) = the input's bytes) receives the prologue output x' first (one elementwise pass,
 * as tmdiff_conv3d_fwd_staged).  d->w_packed = tmdiff_conv3d_wino_pack_weights(..., mode | 2, planes 6): the same
 * transformed weights in natural column order.  Replaces, for the Python reference, the F.conv3d / nn.Conv3d calls of
 * GeneralModel/Hyper_unet_general.py:51-77, :161-164, :224-227, :344-361 on 8- and 4-band tensors. */
int tmdiff_conv3d_wf_supported(const tmdiff_conv3d_desc* d);
int64_t tmdiff_conv3d_wf_blocks(const tmdiff_conv3d_desc* d);     /* workgroups of its grid, split-K included */
/* small grids (single images, the deep levels of a small batch) split the input channels over workgroups when d->splitk_ws
 * lends tmdiff_conv3d_wf_splitk_workspace_bytes(d) bytes (0: the grid needs no split); splitk_reduce_kernel sums the ranges
 * in a fixed order and applies the epilogue, as for tmdiff_conv3d_fwd */
size_t tmdiff_conv3d_wf_splitk_workspace_bytes(const tmdiff_conv3d_desc* d);
size_t tmdiff_conv3d_wf_workspace_bytes(const tmdiff_conv3d_desc* d);
/* The launch plan for a convolution of these extents without a descriptor (host-side routing): returns the split-K factor
 * (1 = none) and writes the number of output tiles (workgroups = tiles x factor); 0 = shape not taken.  llm != 0: the composed
 * Conv_0 + LL mode of tmdiff_conv3d_wfll_fwd (Cin, H, W those of the space-to-depth tensor). */
int32_t tmdiff_conv3d_wf_plan(int32_t B, int32_t Cin, int32_t Cout, int32_t N, int32_t H, int32_t W, int32_t groups, int32_t llm,
                              int64_t* tiles);
int tmdiff_conv3d_wf_fwd(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream);
/* `Conv_0` (3x3x3) of a down block and the halved Haar LL band of its output as ONE convolution (as tmdiff_conv3d_ll_fwd: same
 * descriptor -- H, W the INPUT extents, outputs at half of them, bias multiplied by 2 * ll_scale) WITH Winograd F(4,3) along the
 * bands on top: the LL band acts on (h, w) only, so the composed 3x4x4 stride-2 kernel keeps its three band taps -- 16 x 6 / 4 =
 * 24 multiply-adds per output instead of 48 (and 108 for the convolution + DWT pair).  seg_x[0] is the producer's SPACE-TO-DEPTH
 * second output (y2_s2d above: [B, 4 Cin, N, H/2, W/2]); on it the composed kernel is a stride-1 convolution with 2 x 2 of the
 * 3 x 3 taps per virtual channel, which the kernel of tmdiff_conv3d_wf_fwd runs with a 24-step K loop.  8- and 4-band tensors, W % 8 == 0,
 * Cout % 32 == 0; weights from tmdiff_conv3d_wfll_pack_weights (Cin x 96 x Cout floats). */
int tmdiff_conv3d_wfll_supported(const tmdiff_conv3d_desc* d);
size_t tmdiff_conv3d_wfll_packed_bytes(int32_t Cout, int32_t Cin);
int tmdiff_conv3d_wfll_pack_weights(const float* w, float* packed, int32_t Cout, int32_t Cin, float ll_scale,
                                    tmdiff_stream_t stream);
size_t tmdiff_conv3d_wfll_splitk_workspace_bytes(const tmdiff_conv3d_desc* d);
int tmdiff_conv3d_wfll_fwd(const tmdiff_conv3d_desc* d, float ll_scale, tmdiff_stream_t stream);

/* ---- bf16 compute / fp32 accumulate (WorldView-3 inference) --------------------------
 * Same descriptor and fused prologue / epilogue as tmdiff_conv3d_fwd; activations, bias, residual and output stay
 * fp32 in memory.  The prologue result x' and the weights are rounded to bf16 (round to nearest even), products
 * are accumulated in fp32 on v_mfma_f32_32x32x16_bf16.  d->w_packed must come from tmdiff_conv3d_pack_weights_bf16
 * (tmdiff_conv3d_packed_bf16_bytes bytes; 0 = shape not supported).
 * Supported: Cin/groups a multiple of 8 (ksize 3) or 16 (ksize 1), every segment a multiple of 8 channels,
 * Cout/groups a multiple of 32, no in_mask; anything else returns TMDIFF_E_UNSUPPORTED and the caller keeps using tmdiff_conv3d_fwd (forward only:
 * training runs in fp32).
 * workspace NULL: one fused kernel (prologue evaluated while staging, per channel tile).  workspace of
 * tmdiff_conv3d_bf16_workspace_bytes(d) bytes: two kernels -- the prologue output is packed to bf16 once
 * ([B][Cin/8][N*H*W] units of 8 channels), then a convolution whose operands go HBM -> LDS directly
 * (global_load_lds) -- the better choice when several channel tiles share the input.  Same results bit for bit.
 * ksize 1 is a bandwidth kernel without LDS (operands built in registers); it ignores the workspace. */
size_t tmdiff_conv3d_packed_bf16_bytes(int32_t Cout, int32_t Cin, int32_t ksize, int32_t groups);
int tmdiff_conv3d_pack_weights_bf16(const float* w, void* packed, int32_t Cout, int32_t Cin, int32_t ksize,
                                    int32_t groups, tmdiff_stream_t stream);
size_t tmdiff_conv3d_bf16_workspace_bytes(const tmdiff_conv3d_desc* d);
int tmdiff_conv3d_fwd_bf16(const tmdiff_conv3d_desc* d, void* workspace, tmdiff_stream_t stream);

/* ---- backward of the fused convolution (finetune path) ----------------------------------
 * With x' = prologue(x) and y = (conv(x', w) + bias_scale*bias + residual) * out_scale, and g = dL/dy * out_scale:
 *   dL/dresidual = g;  dL/dbias = bias_scale * sum_{b,pos} g      -> tmdiff_channel_sum
 *   dL/dx'       = conv3d_fwd(g, pack_weights(w, mode 1))          (same kernel, roles of Cin/Cout swapped)
 *   dL/dw        = tmdiff_conv3d_wgrad                             (fp32 MFMA, split over the batch/boxes)
 *   dL/dx, dL/dshift, dL/dscale = tmdiff_conv3d_prologue_bwd(dL/dx', x, ...)
 * ATen computes these inside autograd for F.conv3d / nn.Conv3d (Hyper_unet_general.py:74, :244, :372...). */

/* x'[B, Cin, N, H, W] = act(cat(segments) + shift) * scale * mask (or in-kernel dropout): the prologue of the convolution
 * described by d, on its own (the pass tmdiff_conv3d_fwd_staged and tmdiff_conv3d_wgrad run internally).  The finetune path
 * uses it where the convolution itself is tmdiff_conv3d_ll_fwd, which takes a plain input; x' is then kept for the weight
 * gradient. */
int tmdiff_conv3d_prologue_fwd(const tmdiff_conv3d_desc* d, float* xp, tmdiff_stream_t stream);

/* dw [Cout, Cin/groups, k,k,k] (PyTorch layout) = sum_{b,pos} g[b,co,pos] * x'[b,ci,pos+tap]; x' is formed from
 * the segments / shift / scale / mask / act of `d` exactly as in the forward (d->y, residual, bias, w_packed are
 * ignored).  `g` is [B, Cout, N, H, W].  workspace: tmdiff_conv3d_wgrad_workspace_bytes(d) bytes (partial sums). */
size_t tmdiff_conv3d_wgrad_workspace_bytes(const tmdiff_conv3d_desc* d);
int tmdiff_conv3d_wgrad(const tmdiff_conv3d_desc* d, const float* g, float* dw, void* workspace,
                        tmdiff_stream_t stream);
/* The same, and dbias[Cout] = d->bias_scale * sum_{b,pos} g[b,co,pos] on the side (the kernel reads every g element anyway;
 * this replaces a tmdiff_channel_sum pass over g).  dbias NULL = tmdiff_conv3d_wgrad. */
int tmdiff_conv3d_wgrad_bias(const tmdiff_conv3d_desc* d, const float* g, float* dw, float* dbias, void* workspace,
                             tmdiff_stream_t stream);

/* The same weight gradient in the Winograd domain along the band axis (csrc/wgrad_wino.hip): F(3,4), the transpose of the
 * forward's F(4,3) -- dw = A'^T[(G' g) (.) (B^T x')], half the multiply-adds.  Both operands are first transformed into
 * channels-last zero-padded arrays by one pass each (inside the workspace), the accumulation over positions runs on the matrix
 * pipe per Winograd plane, a reduction kernel sums the split partials in a fixed order and applies A'^T.  Takes fp32 3x3x3,
 * groups 1 or 3, N % 4 == 0 (at most 16 bands), W % 4 == 0, any channel counts; _supported() says so (else: tmdiff_conv3d_wgrad;
 * ATen's conv3d backward: Hyper_unet_general.py:74, :244, :372).  Same arguments and result layout as tmdiff_conv3d_wgrad. */
int tmdiff_conv3d_wgrad_wino_supported(const tmdiff_conv3d_desc* d);
size_t tmdiff_conv3d_wgrad_wino_workspace_bytes(const tmdiff_conv3d_desc* d);
int tmdiff_conv3d_wgrad_wino(const tmdiff_conv3d_desc* d, const float* g, float* dw, void* workspace,
                             tmdiff_stream_t stream);
/* ... and dbias[Cout] = d->bias_scale * sum_{b,pos} g on the side (the pass that transforms g sums it up as it reads; fixed
 * summation order).  dbias NULL = tmdiff_conv3d_wgrad_wino. */
int tmdiff_conv3d_wgrad_wino_bias(const tmdiff_conv3d_desc* d, const float* g, float* dw, float* dbias, void* workspace,
                                  tmdiff_stream_t stream);

/* out[c] = scale * sum_{b, p} x[b, c, p]   (x is [B, C, P]); bias gradients. */
int tmdiff_channel_sum(const float* x, float* out, int32_t B, int32_t C, int64_t P, float scale,
                       tmdiff_stream_t stream);

/* Backward of the prologue x' = act(x + shift[b,c]) * scale[b,c] * mask: given gp = dL/dx' [B,Cin,N,H,W] and the
 * forward descriptor `d` (segments, shift, scale, mask, act), writes dL/dx into dx_seg[i] (same segmenting as
 * d->seg_x; NULL = not needed; accumulate[i] != 0 adds to the existing contents) and the per-(b,c) reductions
 * d_shift[B,Cin], d_scale[B,Cin] (NULL = not needed, dense rows). */
int tmdiff_conv3d_prologue_bwd(const tmdiff_conv3d_desc* d, const float* gp, float* const dx_seg[3],
                               const int32_t accumulate[3], float* d_shift, float* d_scale, tmdiff_stream_t stream);
/* Same, with a workspace of tmdiff_conv3d_prologue_bwd_workspace_bytes(d) bytes (0 = not needed): planes are then cut
 * into several workgroups each when B*Cin alone cannot fill the chip (the full-resolution 32-channel layers at small
 * batches); the per-plane sums are finished in a fixed order (deterministic). */
size_t tmdiff_conv3d_prologue_bwd_workspace_bytes(const tmdiff_conv3d_desc* d);
int tmdiff_conv3d_prologue_bwd_ws(const tmdiff_conv3d_desc* d, const float* gp, float* const dx_seg[3],
                                  const int32_t accumulate[3], float* d_shift, float* d_scale, void* workspace,
                                  tmdiff_stream_t stream);
/* Three-operand form: dx_seg[i] = add_seg[i] + dL/dx_i (add_seg[i] == NULL: just dL/dx_i) -- the gradient ANOTHER consumer of the
 * same segment has produced (a ResBlock's identity residual, Hyper_unet_general.py:248) is read here instead of being summed with
 * this one by a launch of its own.  add_seg[i] is only read; dx_seg[i] is a tensor of its own. */
int tmdiff_conv3d_prologue_bwd_add(const tmdiff_conv3d_desc* d, const float* gp, float* const dx_seg[3],
                                   const float* const add_seg[3], float* d_shift, float* d_scale, void* workspace,
                                   tmdiff_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Stem / head pointwise convolutions (bandwidth kernels):
 *  stem: y[b,co,p] = SiLU(w[co]*x[b,p] + bias[co])            AdaptionModulateBEST.conv20 + act (:169-170)
 *        x is given as pan[b,1,H,W] broadcast over N minus ms[b,N,H,W] when `ms` != NULL
 *        (WavBEST.forward :605-608), else x = xin[b,N,H,W] (to3D(x_t) :609).
 *  head: y[b,p] = sum_c w[c]*scale[b,c]*SiLU(x[b,c,p])         FinalBlock act + modulated conv24 (:270-272)
 * ------------------------------------------------------------------------------------ */
int tmdiff_stem_fwd(const float* xin, const float* pan, const float* ms, const float* w, const float* bias, float* y,
                    int32_t B, int32_t Cout, int32_t N, int32_t H, int32_t W, int32_t apply_silu,
                    tmdiff_stream_t stream);
/* The same with the consumer's per-(b, co) modulation folded in: y *= out_scale[b, co] (row stride as in_scale_stride) --
 * conv21 of the stem block (Hyper_unet_general.py:171) then reads a plain tensor. */
int tmdiff_stem_fwd_scaled(const float* xin, const float* pan, const float* ms, const float* w, const float* bias,
                           const float* out_scale, int32_t out_scale_stride, float* y, int32_t B, int32_t Cout, int32_t N,
                           int32_t H, int32_t W, int32_t apply_silu, tmdiff_stream_t stream);
int tmdiff_stem_fwd_pack_bf16(const float* xin, const float* pan, const float* ms, const float* w, const float* bias,
                              const float* out_scale, int32_t out_scale_stride, void* units, int32_t B, int32_t Cout, int32_t N,
                              int32_t H, int32_t W, int32_t apply_silu, tmdiff_stream_t stream);
int tmdiff_head_fwd(const float* x, const float* w, const float* scale, int32_t scale_stride, float* y, int32_t B,
                    int32_t C, int64_t P, tmdiff_stream_t stream); /* scale_stride as in_scale_stride above */
/* stem backward: with u = w[co]*x + bias[co], y = SiLU(u): dwb[b, co, 0] = sum_p gy*SiLU'(u)*x and
 * dwb[b, co, 1] = sum_p gy*SiLU'(u) (per-sample partials [B, Cout, 2]; the caller sums over b).
 * head backward: dx[b,c,p] = gy[b,p]*w[c]*scale[b,c]*SiLU'(x); dws[b,c] = sum_p gy[b,p]*SiLU(x[b,c,p])
 * (dL/d(w[c]*scale[b,c])); scale rows dense [B,C] or NULL (= 1). */
int tmdiff_stem_bwd(const float* xin, const float* pan, const float* ms, const float* w, const float* bias,
                    const float* gy, float* dwb, int32_t B, int32_t Cout, int32_t N, int32_t H, int32_t W,
                    tmdiff_stream_t stream);
/* stem backward w.r.t. its inputs (WavBEST.forward is differentiable in x_t / PAN / MS in the reference, :605-609):
 * xin form: dx[B,N,H,W] = sum_co gy*SiLU'(u)*w;  (pan, ms) form: dx = d_ms = -that, dpan[B,H,W] = sum over the N bands.
 * Either output may be NULL. */
int tmdiff_stem_bwd_input(const float* xin, const float* pan, const float* ms, const float* w, const float* bias,
                          const float* gy, float* dx, float* dpan, int32_t B, int32_t Cout, int32_t N, int32_t H,
                          int32_t W, tmdiff_stream_t stream);
int tmdiff_head_bwd(const float* x, const float* w, const float* scale, const float* gy, float* dx, float* dws,
                    int32_t B, int32_t C, int64_t P, tmdiff_stream_t stream);

/* ------------------------------------------------------------------------------------
 * 2-D Haar DWT / IDWT on the (H, W) axes of [B*C*N, H, W] planes.
 * Replaces DWT_2D / IDWT_2D (DWT_IDWT/DWT_IDWT_layer.py:256-334, :337-430) and
 * DWTFunction_2D / IDWTFunction_2D (DWT_IDWT/DWT_IDWT_Functions.py:47-69, :89-112).
 *  dwt:  ll = ll_scale * LL(x); lh/hl/hh = hi_scale * {LH,HL,HH}(x) (NULL = band not needed)
 *        (the /2 of WaveletUPorDown :396 is ll_scale = 0.5).
 *  idwt: out_k = IDWT(in_scale * ll_k, lh, hl, hh) for k < n_ll low bands sharing the same
 *        high bands (the two iwt calls with 2*h and 2*x at :383-386 are one launch).  The high
 *        bands may be channel slices of one [B, 3C, N, h, w] tensor (the convH_0 output, :381-384):
 *        hi_planes_per_batch = C*N planes per sample, hi_batch_stride = floats between samples
 *        (0 = dense [planes, h, w] bands).  lh/hl/hh may all be NULL = zero high bands (the adjoint of an
 *        LL-only dwt).
 * The adjoint of dwt is idwt and vice versa (orthonormal transform), which is how the
 * backward passes are served.
 * ------------------------------------------------------------------------------------ */
/* Producer-side prologue: the DWT's LL band / the IDWT's first reconstruction can be written with the CONSUMER convolution's
 * prologue already applied,  out = act(out + shift[b, c]) * scale[b, c]  (plane = (b * C + c) * n_per_channel + n), so that the
 * consumer (Conv_1 of a wavelet block, Hyper_unet_general.py:400-406) reads a plain tensor: same bits as applying the
 * prologue on the consumer side.  Strides as tmdiff_conv3d_desc::in_shift_stride (0 dense = C, > 0 floats, -1 broadcast row). */
typedef struct tmdiff_plane_prologue {
  const float* shift; /* [B, C] or NULL */
  const float* scale; /* [B, C] or NULL */
  int32_t shift_stride, scale_stride;
  int32_t C, n_per_channel;
  int32_t act; /* 0 identity, 1 SiLU */
} tmdiff_plane_prologue;
int tmdiff_haar_dwt2d_pro(const float* x, float* ll, float* lh, float* hl, float* hh, int64_t planes, int32_t H, int32_t W,
                          float ll_scale, float hi_scale, const tmdiff_plane_prologue* ll_prologue, tmdiff_stream_t stream);
int tmdiff_haar_idwt2d_pro(const float* const ll[2], int32_t n_ll, const float* lh, const float* hl, const float* hh,
                           int64_t hi_planes_per_batch, int64_t hi_batch_stride, float* const out[2], int64_t planes,
                           int32_t h, int32_t w, float in_scale, const tmdiff_plane_prologue* out0_prologue,
                           tmdiff_stream_t stream);
/* bf16-mode producers: the LL band / first reconstruction (x, ll0, ll1: [B, C, N, ., .] fp32; stacked_bands [B, 3C, N, h, w])
 * written as the packed bf16 units [B][C/8][N*h*w] of 8 channels (16 bytes) that tmdiff_conv3d_fwd_bf16 reads with x_bf16 = 1,
 * prologue applied in fp32 and rounded to nearest even exactly as that entry point's own pack pass does.  C % 8 == 0. */
int tmdiff_haar_dwt2d_pack_bf16(const float* x, void* ll_units, float* lh, float* hl, float* hh, int32_t B, int32_t C,
                                int32_t N, int32_t H, int32_t W, float ll_scale, float hi_scale,
                                const tmdiff_plane_prologue* ll_prologue, tmdiff_stream_t stream);
int tmdiff_haar_idwt2d_pack_bf16(const float* ll0, const float* ll1, const float* stacked_bands, void* out0_units, float* out1,
                                 int32_t B, int32_t C, int32_t N, int32_t h, int32_t w, float in_scale,
                                 const tmdiff_plane_prologue* out0_prologue, tmdiff_stream_t stream);
int tmdiff_haar_dwt2d(const float* x, float* ll, float* lh, float* hl, float* hh, int64_t planes, int32_t H,
                      int32_t W, float ll_scale, float hi_scale, tmdiff_stream_t stream);
int tmdiff_haar_idwt2d(const float* const ll[2], int32_t n_ll, const float* lh, const float* hl, const float* hh,
                       int64_t hi_planes_per_batch, int64_t hi_batch_stride, float* const out[2], int64_t planes,
                       int32_t h, int32_t w, float in_scale, tmdiff_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Small dense layers: y[b,o] = act(sum_i x[b,i]*w[o,i] + bias[o]); nn.Linear (+Swish) of the
 * embedding MLPs and every Dense() modulation projection (Hyper_unet_general.py:100-108,
 * :529-532).  One launch serves a whole bank of projections: w is [O_total, I].
 * gamma_embedding (:80-97): emb[b, :half] = cos(t*f), emb[b, half:2*half] = sin(t*f), zero pad if dim is
 * odd; freqs[half] is the host-computed fp32 table exp(-ln(1e4)*k/half) (a 1-ulp difference in f
 * would be amplified by t ~ 1000, so the table is not recomputed on the device).
 * ------------------------------------------------------------------------------------ */
int tmdiff_linear_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t I, int32_t O,
                      int32_t act, tmdiff_stream_t stream);
int tmdiff_gamma_embedding(const float* t, const float* freqs, float* emb, int32_t B, int32_t dim,
                           tmdiff_stream_t stream);
/* backward of y = act(x @ w^T + bias): gu = gy * act'(u) with the pre-activation u recomputed into gu_scratch
 * [B, O] (only needed when act != 0); dx[b,i] = sum_o gu[b,o] w[o,i]; dw[o,i] = sum_b gu[b,o] x[b,i];
 * db[o] = sum_b gu[b,o].  Any of dx / dw / db may be NULL. */
int tmdiff_linear_bwd(const float* x, const float* w, const float* bias, const float* gy, float* gu_scratch,
                      float* dx, float* dw, float* db, int32_t B, int32_t I, int32_t O, int32_t act,
                      tmdiff_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Sampler elementwise updates.
 *  ddpm_step: one reverse step of GeneralDiffusion.p_sample (diffusion_general.py:203-208 with
 *    :376-378, :192-194, :134-138):  x0 = clamp(c_recip*x - c_recipm1*eps, -1, 1);
 *    out = coef1*x0 + coef2*x + sigma*noise   (noise may be NULL when sigma == 0)
 *    and optionally img_out = out + ms (res2img, utils/util.py:135-137).
 *  axpby: out = sum_{k<n} coef[k] * in[k], n <= 4: every DPM-Solver update
 *    (core/dpm_solver_pytorch.py:563-927).
 *  x0_from_model: DPM-Solver++ data prediction for an x_start-parameterised network
 *    (dpm_solver_pytorch.py:302-306 then :447-456): eps = (x - alpha*out)/sigma;
 *    x0 = (x - sigma*eps)/alpha.
 *  abs_quantile_clamp: dynamic thresholding (:430-439): per sample s = max(quantile(|x0|, q), max_val)
 *    (linear interpolation between order statistics, as torch.quantile), x0 = clamp(x0,-s,s)/s.
 *    workspace: tmdiff_abs_quantile_workspace_bytes(B, n) bytes (select state; on return its first B floats hold
 *    the per-sample thresholds s).  Runs as a few launches (histogram passes spread over many workgroups).
 * ------------------------------------------------------------------------------------ */
int tmdiff_ddpm_step(const float* x, const float* eps, const float* noise, const float* ms, float* out,
                     float* img_out, int64_t n, float c_recip, float c_recipm1, float coef1, float coef2,
                     float sigma, int32_t clip, tmdiff_stream_t stream);
int tmdiff_axpby(const float* const in[4], const float coef[4], int32_t n_in, float* out, int64_t n,
                 tmdiff_stream_t stream);
/* Multi-tensor form, ONE launch for a whole parameter list: out_t = ca * a_t + cb * b_t for every entry (the EMA update of
 * utils/EmaUpdater.py:23-38: out = a = EMA weights, b = live weights).  `tensors_dev` is a DEVICE array of entries;
 * the launch has one workgroup per chunk of tmdiff_multi_axpby_chunk() elements: chunk k works on elements
 * [chunk_index_dev[k] * chunk, ...) of tensor chunk_tensor_dev[k] (both DEVICE int32 arrays of n_chunks entries). */
typedef struct tmdiff_mt_entry {
  float* out;
  const float* a;
  const float* b;
  int64_t n;
} tmdiff_mt_entry;
int32_t tmdiff_multi_axpby_chunk(void);
int tmdiff_multi_axpby(const tmdiff_mt_entry* tensors_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_index_dev,
                       int32_t n_chunks, float ca, float cb, tmdiff_stream_t stream);
/* AdamW step (torch.optim.AdamW semantics, amsgrad off; reference GeneralModel/model.py:30-31, :43) of a whole list of parameter
 * tensors in ONE launch, chunked like tmdiff_multi_axpby (same chunk size, same two device tables).  lr_dev / step_dev are DEVICE
 * scalars (the learning rate; the step count t of THIS update, already incremented, as a float), so a recorded launch can be
 * replayed from a HIP graph.  In place: p, m (exp_avg), v (exp_avg_sq). */
typedef struct tmdiff_adamw_entry {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} tmdiff_adamw_entry;
int tmdiff_multi_adamw(const tmdiff_adamw_entry* tensors_dev, const int32_t* chunk_tensor_dev, const int32_t* chunk_index_dev,
                       int32_t n_chunks, const float* lr_dev, const float* step_dev, float beta1, float beta2, float eps,
                       float weight_decay, tmdiff_stream_t stream);
int tmdiff_x0_from_model(const float* x, const float* model_out, float* x0, int64_t n, float alpha, float sigma,
                         int32_t model_is_x_start, tmdiff_stream_t stream);
size_t tmdiff_abs_quantile_workspace_bytes(int32_t B, int64_t n_per_sample);
int tmdiff_abs_quantile_clamp(float* x0, int32_t B, int64_t n_per_sample, float q, float max_val, void* workspace,
                              tmdiff_stream_t stream);
/* out = a + b (res2img / img2res with sign), q_sample: out = a[b]*x0 + sqrt(1-a[b]^2)*noise. */
int tmdiff_add(const float* a, const float* b, float* out, int64_t n, float sign_b, tmdiff_stream_t stream);
int tmdiff_q_sample(const float* x0, const float* noise, const float* a, float* out, int32_t B, int64_t n_per_sample,
                    tmdiff_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Standalone attention operators of core/Attention.py (imported by nothing in the reference; built because the
 * north star names them).  All fp32.
 *  attn_fwd : out = softmax(q k^T * scale [key mask]) v per (batch, head); fp32 MFMA, online softmax.
 *             q/k/v/out are addressed as base + b*strides[0] + head*strides[1] + row*strides[2] + d (elements),
 *             which covers both the '(b h) n d' split of CrossAttention (:186) and the channel-major
 *             [B, C, HW] tensors of SpatialSelfAttention (:143-153, via a transposed view prepared by the caller).
 *             key_mask [B, Nk] bytes (1 = keep) or NULL; head dim D even, <= 128.
 *  gemm_nt  : C[M,N] = A[M,K] W[N,K]^T + bias[N] + residual[M,N]   (nn.Linear on token-major activations)
 *  group_norm (:108-109, eps 1e-6, affine), layer_norm (:279-281), geglu / gelu (:69-76, :84-87).
 * ------------------------------------------------------------------------------------ */
int tmdiff_attn_fwd(const float* q, const float* k, const float* v, float* out, const unsigned char* key_mask,
                    int32_t B, int32_t H, int32_t Nq, int32_t Nk, int32_t D, const int64_t q_strides[3],
                    const int64_t k_strides[3], const int64_t v_strides[3], const int64_t o_strides[3], float scale,
                    tmdiff_stream_t stream);
int tmdiff_gemm_nt(const float* A, const float* Wt, const float* bias, const float* residual, float* C, int64_t M,
                   int32_t N, int32_t K, tmdiff_stream_t stream);
int tmdiff_group_norm(const float* x, const float* gamma, const float* beta, float* y, int32_t B, int32_t C,
                      int64_t P, int32_t groups, float eps, tmdiff_stream_t stream);
int tmdiff_layer_norm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t D,
                      float eps, tmdiff_stream_t stream);
int tmdiff_geglu(const float* u, float* y, int64_t rows, int32_t inner, int32_t gelu_only, tmdiff_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Per-operator names (SURVEY 8b "minimum exports").  Thin, argument-checked fronts of the entry points above for
 * callers that bind one symbol per ATen call they replace:
 *  conv3d_k{3,1}_fwd   = tmdiff_conv3d_fwd with d->ksize required to be 3 / 1            (F.conv3d / nn.Conv3d)
 *  conv3d_k{3,1}_dgrad = the same kernel on mode-1 packed weights: d->seg_x[0] = dL/dy, d->y = dL/dx'
 *  conv3d_k{3,1}_wgrad = tmdiff_conv3d_wgrad
 *  haar_dwt2d_fwd / haar_idwt2d_fwd = the transforms; *_bwd = their adjoints
 *    dwt2d_bwd : dx = IDWT(ll_scale*g_ll, hi_scale*(g_lh, g_hl, g_hh))  (DWTFunction_2D.backward, DWT_IDWT_Functions.py:60-69);
 *                the three high-band gradients must be all NULL (LL-only forward) or all given, and then
 *                hi_scale must be 1 (the hot path only ever scales the LL band, Hyper_unet_general.py:396).
 *    idwt2d_bwd: g_ll = in_scale * LL(g_out), (g_lh, g_hl, g_hh) = high bands of DWT(g_out)               (IDWTFunction_2D.backward, :101-112)
 *  dpm_axpby{2,3,4}: out = sum_k coef_k * in_k with exactly 2 / 3 / 4 terms (dpm_solver_pytorch.py:563-927).
 * ------------------------------------------------------------------------------------ */
int tmdiff_conv3d_k3_fwd(const tmdiff_conv3d_desc* d, tmdiff_stream_t stream);
int tmdiff_conv3d_k3_dgrad(const tmdiff_conv3d_desc* d, tmdiff_stream_t stream);
int tmdiff_conv3d_k3_wgrad(const tmdiff_conv3d_desc* d, const float* g, float* dw, void* workspace,
                           tmdiff_stream_t stream);
int tmdiff_conv3d_k1_fwd(const tmdiff_conv3d_desc* d, tmdiff_stream_t stream);
int tmdiff_conv3d_k1_dgrad(const tmdiff_conv3d_desc* d, tmdiff_stream_t stream);
int tmdiff_conv3d_k1_wgrad(const tmdiff_conv3d_desc* d, const float* g, float* dw, void* workspace,
                           tmdiff_stream_t stream);
int tmdiff_haar_dwt2d_fwd(const float* x, float* ll, float* lh, float* hl, float* hh, int64_t planes, int32_t H,
                          int32_t W, float ll_scale, float hi_scale, tmdiff_stream_t stream);
int tmdiff_haar_dwt2d_bwd(const float* g_ll, const float* g_lh, const float* g_hl, const float* g_hh, float* dx,
                          int64_t planes, int32_t H, int32_t W, float ll_scale, float hi_scale,
                          tmdiff_stream_t stream);
int tmdiff_haar_idwt2d_fwd(const float* ll, const float* lh, const float* hl, const float* hh, float* out,
                           int64_t planes, int32_t h, int32_t w, float in_scale, tmdiff_stream_t stream);
int tmdiff_haar_idwt2d_bwd(const float* g_out, float* g_ll, float* g_lh, float* g_hl, float* g_hh, int64_t planes,
                           int32_t h, int32_t w, float in_scale, tmdiff_stream_t stream);
int tmdiff_dpm_axpby2(const float* x0, float c0, const float* x1, float c1, float* out, int64_t n,
                      tmdiff_stream_t stream);
int tmdiff_dpm_axpby3(const float* x0, float c0, const float* x1, float c1, const float* x2, float c2, float* out,
                      int64_t n, tmdiff_stream_t stream);
int tmdiff_dpm_axpby4(const float* x0, float c0, const float* x1, float c1, const float* x2, float c2,
                      const float* x3, float c3, float* out, int64_t n, tmdiff_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TMDIFF_HIP_H */
