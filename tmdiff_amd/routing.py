"""Which kernel family runs each 3x3x3 convolution: the ONE place the decision is taken (host logic only, no GPU work).

A *family* is a C entry point of libtmdiff_hip.so with the kernel behind it:

  family    C entry point                   kernel (csrc/)                                       executes per output
  --------  ------------------------------  ---------------------------------------------------  -------------------
  wf        tmdiff_conv3d_wf_fwd            conv3d_wf_kernel<2,8,16> (8 bands) / <1,16,16> (4)   13.5 multiply-adds / ci
  wf_pair   tmdiff_conv3d_wf_fwd            conv3d_wf_kernel<2,8,16,PAIR> (8 bands x 8 columns)  13.5
  wfll      tmdiff_conv3d_wfll_fwd          conv3d_wf_kernel<..., LLM> (Conv_0 + LL composed)    24 per quarter-res output
  ll        tmdiff_conv3d_ll_fwd            conv3d_ll_kernel (Conv_0 + LL composed, direct)      48 per quarter-res output
  staged    tmdiff_conv3d_fwd_staged        (prologue_apply_kernel +) conv3d_dma_kernel          27
  fused     tmdiff_conv3d_fwd               conv3d_mfma_kernel (prologue applied while staging)  27
  bf16      tmdiff_conv3d_fwd_bf16          (pack_x_bf16_kernel +) conv3d_bf16_dma_kernel        27, bf16 operands
  wino4/2   tmdiff_conv3d_wino_fwd_planes   wino_input_kernel + conv3d_wino_kernel               13.5 / 18 -- tmdiff_amd.fallback

`conv3_family` decides from the extents alone (plus what the input looks like), so the decision for every layer of every
BASELINE configuration can be tabulated without a GPU: `unet_conv3_layers` enumerates the 3x3x3 convolutions of one WavBEST
forward (reference GeneralModel/Hyper_unet_general.py:600-636 -- 51 of its 73 convolution calls), `unet_table` routes them.
`tools/routing_table.py` writes the table to profiles/ and tests/test_host_logic.py asserts it: every family of the product
path is reached by some BASELINE configuration, and the `fallback` families (band counts other than 4 / 8) by none.

The rules, in order (fp32):
  1. Winograd F(4,3) along the bands with the transform inside the kernel (wf): 8- or 4-band tensors, W % 4 == 0, Cin/g even,
     Cout/g % 32 == 0, no mask tensor; the plan of the kernel itself (tmdiff_conv3d_wf_plan: tiles, split-K factor) must reach
     `config.wino_min_blocks` workgroups and the plane must fill `config.wf_min_fill` of its 8 x 16 / 16 x 16 tiles (8 bands x
     8 columns: two images per tile, pair mode).
  2. other even band counts whose grid is large enough: transform pass + Winograd kernel (wino4 / wino2, tmdiff_amd.fallback).
  3. the direct kernels, whose split-K fills the chip on small grids: staged (operands by LDS-DMA; needs Cin/g % 4 == 0 and
     Cout/g % 32 == 0) where the input is plain or the prologue pass is amortised (Cout/g >= 128, Cin/g >= 384, dropout / mask,
     a kept x'), else fused.
Batch-size note (README): the same sample can take different families at B = 1 and B = 32 (rule 1's grid threshold), hence
different fp32 summation orders; tests/test_gpu_configs.py bounds the difference at 1e-5.
"""
import collections
import ctypes as C

from ._lib import lib

PRODUCT_FAMILIES = ("wf", "wf_pair", "wfll", "ll", "staged", "fused", "bf16")
FALLBACK_FAMILIES = ("wino4", "wino2")


def _config():
    from . import ops
    return ops.config


def wino_weight_ok(cout, cin, ksize=3, groups=1):
    """Weight shapes the Winograd kernels (conv3d_wf, fallback.conv3d_wino) take."""
    return (ksize == 3 and groups in (1, 3) and cin % groups == 0 and cout % groups == 0 and (cin // groups) % 2 == 0 and
            (cout // groups) % 32 == 0)


_WF_ROUTES = {}      # (the plan is a pure function of the extents and the switches: one library call per distinct shape)


def wf_route(b, cin, cout, n, h, w, groups=1, masked=False, llm=False):
    """(taken, split): whether conv3d_wf runs a convolution of these extents and into how many ranges it splits the input
    channels (1 = no split-K).  llm: the composed Conv_0 + LL mode (cin, h, w those of the space-to-depth tensor)."""
    cfg = _config()
    key = (b, cin, cout, n, h, w, groups, masked, llm, cfg.key())
    r = _WF_ROUTES.get(key)
    if r is None:
        if len(_WF_ROUTES) > 4096:
            _WF_ROUTES.clear()
        r = _WF_ROUTES[key] = _wf_route(cfg, b, cin, cout, n, h, w, groups, masked, llm)
    return r


def _wf_route(cfg, b, cin, cout, n, h, w, groups, masked, llm):
    if cin % groups or cout % groups or masked or not cfg.wf:
        return False, 1
    tiles = C.c_int64(0)
    split = lib.tmdiff_conv3d_wf_plan(b, cin, cout, n, h, w, groups, 1 if llm else 0, C.byref(tiles))
    if split == 0:
        return False, 1
    if not cfg.wf_splitk:
        split = 1
    th = 8 if n == 8 else 16
    pair = n == 8 and w == 8          # two images side by side in one 8 x 16 tile
    if pair and not cfg.wf_pair:
        return False, 1
    fill = (h * w) / float(((h + th - 1) // th) * th * (8 if pair else ((w + 15) // 16) * 16))
    if pair:
        fill *= b / (2.0 * ((b + 1) // 2))        # (an odd batch leaves the last pair's second half empty)
    return bool(tiles.value * split >= cfg.wino_min_blocks and fill >= cfg.wf_min_fill), split


def wfll_route(b, cin, cout, n, h, w):
    """True when conv3d_wf's composed-LL mode takes Conv_0 + LL of a [b, cin, n, h, w] input (h, w: full resolution)."""
    if not (_config().wfll and n in (4, 8) and h % 2 == 0 and w % 8 == 0 and cout % 32 == 0):
        return False
    return wf_route(b, 4 * cin, cout, n, h // 2, w // 2, llm=True)[0]


def wino_plan(b, cout, n, h, w, groups=1):
    """(planes, workgroups) of the transform-pass Winograd kernel for these extents; planes 0 = not taken."""
    planes = lib.tmdiff_conv3d_wino_planes(int(n))
    if not planes or w % 4:
        return 0, 0
    cg = cout // groups
    per_tile = b * groups * ((h + 7) // 8) * (((w + 7) // 8) * (cg // 64) if cg % 64 == 0 else ((w + 15) // 16) * (cg // 32))
    blocks = lambda p: per_tile * ((n // (p - 2) + 1) // 2)          # (= tmdiff_conv3d_wino_blocks)
    mn = _config().wino_min_blocks
    if planes == 6 and blocks(6) < mn <= blocks(4):
        planes = 4                   # F(2,3) has twice the tiles along the bands: it still fills the chip here
    return (planes, blocks(planes)) if blocks(planes) >= mn else (0, blocks(planes))


def conv3_family(b, cin, cout, n, h, w, groups=1, plain=True, masked=False, dropout=False, keep_xp=False, math="fp32"):
    """The family that runs a 3x3x3 convolution [b, cin, n, h, w] -> cout.  plain: the input is one tensor (or the three
    tensors of a grouped convolution's three groups) with no prologue; masked: a dropout MASK TENSOR multiplies the input
    (parity runs); dropout: in-kernel dropout; keep_xp: the caller wants the prologue output x' kept (finetune)."""
    cfg = _config()
    if math == "bf16":
        return "bf16"
    if cfg.winograd and not masked and wino_weight_ok(cout, cin, 3, groups):
        if wf_route(b, cin, cout, n, h, w, groups, masked)[0]:
            return "wf_pair" if (n == 8 and w == 8) else "wf"
        planes, _ = wino_plan(b, cout, n, h, w, groups)
        if planes:
            return "wino4" if planes == 6 else "wino2"
    return direct_family(cin, cout, groups, plain, masked, dropout, keep_xp)


def direct_family(cin, cout, groups=1, plain=True, masked=False, dropout=False, keep_xp=False):
    """"staged" or "fused" for a 3x3x3 convolution on the direct kernels.  Measured (tools/bench_conv.py, B = 32): the staged
    kernel itself is 3-6 % faster than the fused one, but its prologue pass costs 8 B per input element -- a net win when the
    input needs no pass (one plain tensor: every data-gradient convolution), is shared by >= 2 channel tiles (Cout/g >= 128),
    is wide (Cin/g >= 384), or carries a dropout mask (the fused kernel reads the mask inside its MFMA stream); a kept x'
    (finetune) forces it.  Both give the same bits (tests/test_gpu_kernels.py::test_conv3d_staged_equals_fused)."""
    cin_g, cout_g = cin // groups, cout // groups
    staged_ok = groups in (1, 3) and cin % groups == 0 and cout % groups == 0 and cin_g % 4 == 0 and cout_g % 32 == 0
    want = {"0": False, "1": True}.get(_config().fp32_staged, plain or cout_g >= 128 or cin_g >= 384 or masked or dropout)
    return "staged" if staged_ok and (want or keep_xp) else "fused"


def k1_side_xp(b, seg_c, cout, n, h, w, groups=1):
    """True when a 1x1x1 convolution of the segments seg_c (channels each) runs on a form of the bandwidth kernel that can also
    write the prologue output of its input (make_conv_desc side_xp=; csrc/conv1.hip conv1_fp32_try): segments and Cin / groups
    of multiples of 16 channels, Cout / groups % 32 == 0, planes of multiples of 4 positions, and either at least 512 tiles of
    512 positions (the 16-byte kernel) or a small grid of at least 128 input channels per group (the kernel that splits the
    channels over its waves); contiguous torch tensors are 16-byte aligned.  tmdiff_conv3d_fwd_xp_supported is the library's own answer for a filled descriptor."""
    cin, plane = sum(seg_c), n * h * w
    if cin % groups or cout % groups or any(c % 16 for c in seg_c) or (cin // groups) % 16 or (cout // groups) % 32 or plane % 4:
        return False
    cout_g = cout // groups
    co_tiles = cout_g // 64 if cout_g % 64 == 0 else cout_g // 32
    vec = b * groups * ((plane + 511) // 512) * co_tiles >= 512                                   # the 16-byte kernel
    small = b * groups * ((plane + 255) // 256) * co_tiles < 512 and (cin // groups) // 16 >= 8    # channels over the waves
    return (vec or small) and plane * 16 < (1 << 31) and cin * plane < (1 << 30)


def ll_family(b, cin, cout, n, h, w, producer_s2d=True):
    """Conv_0 + halved LL band of a main-branch down block on a [b, cin, n, h, w] input: "wfll" (the producer hands over its
    second output in space-to-depth form), "ll" (composed, direct), or None (convolution + LL-only DWT)."""
    cfg = _config()
    if not (cfg.ll_compose and cin % 2 == 0 and cout % 64 == 0 and h % 2 == 0 and w % 2 == 0):
        return None
    if cfg.winograd and producer_s2d and wfll_route(b, cin, cout, n, h, w):
        return "wfll"
    return "ll"


# ---- the network's 3x3x3 convolutions ------------------------------------------------------------------------------------
Layer = collections.namedtuple("Layer", "name cin cout groups h w plain kind")      # kind: "conv" | "conv0_ll"


def unet_conv3_layers(channels, h, w):
    """The 51 3x3x3 convolutions of one WavBEST inference forward at level-0 planes h x w, in the fused (default) graph:
    every convolution reads its producer's second output (plain = ONE plain tensor) except the four three-segment conv20s of
    the up path (prologue pass = concatenation) and convH_0, whose three segments are its three groups' inputs (read in place
    by conv3d_wf, no pass)."""
    c = list(channels)
    lv = [(h >> k, w >> k) for k in range(4)]
    out = []
    add = lambda name, ci, co, k, g=1, plain=True, kind="conv": out.append(Layer(name, ci, co, g, lv[k][0], lv[k][1], plain, kind))
    for stem in ("conv1", "conv2"):
        add(stem + ".conv21", c[0], c[0], 0)
    for suffix, main in (("_1", False), ("", True)):
        for k in range(3):
            blk = f"down{k + 1}{suffix}"
            add(blk + ".conv20.conv20", c[k], c[k + 1], k)
            add(blk + ".conv20.conv21", c[k + 1], c[k + 1], k)
            add(blk + ".down.Conv_0", c[k + 1], c[k + 1], k, kind="conv0_ll" if main else "conv")
            add(blk + ".down.Conv_1", c[k + 1], c[k + 1], k + 1)
    add("middle1.conv20", c[3], c[3], 3)
    add("middle1.conv21", c[3], c[3], 3)
    for k, upn in ((3, "up1"), (2, "up2"), (1, "up3")):
        add(upn + ".conv20.conv20", 3 * c[k], c[k - 1], k, plain=False)
        add(upn + ".conv20.conv21", c[k - 1], c[k - 1], k)
        add(upn + ".up1.Conv_0", c[k - 1], c[k - 1], k)
        add(upn + ".up1.convH_0.0", 3 * c[k], 3 * c[k - 1], k, g=3, plain=False)      # (three tensors: its three groups' inputs)
        add(upn + ".up1.Conv_1", c[k - 1], c[k - 1], k - 1)
    add("final.conv20.conv20", 3 * c[0], c[0], 0, plain=False)
    add("final.conv20.conv21", c[0], c[0], 0)
    for k in (1, 2, 3):
        add(f"final.conv2{k}.conv20", c[0], c[0], 0)
        add(f"final.conv2{k}.conv21", c[0], c[0], 0)
    return out


def unet_table(channels, b, n, h, w, math="fp32"):
    """[(Layer, family)] of one inference forward of a batch of b tiles with n bands."""
    rows = []
    for L in unet_conv3_layers(channels, h, w):
        if L.kind == "conv0_ll" and math == "fp32":
            # (the producer -- the ResBlock's conv21 in front -- writes the space-to-depth form only from an unsplit wf launch)
            takes, split = wf_route(b, L.cin, L.cin, n, L.h, L.w)
            fam = ll_family(b, L.cin, L.cout, n, L.h, L.w, producer_s2d=takes and split == 1)
            if fam is not None:
                rows.append((L, fam))
                continue
        bf = math == "bf16" and L.cin // L.groups % 8 == 0 and L.cout // L.groups % 32 == 0 and (L.plain or L.cin // 3 % 8 == 0)
        rows.append((L, conv3_family(b, L.cin, L.cout, n, L.h, L.w, L.groups, plain=L.plain, math="bf16" if bf else "fp32")))
    return rows


# BASELINE.json configs as (label, channels, batch per GPU, bands, plane) -- B in {1, 8, 32}, N in {4, 8}, 64^2 / 256^2
FULL, WIDE = [32, 64, 128, 256], [64, 128, 256, 512]
BASELINE_CASES = (
    ("configs[0] single tile, T=50", FULL, 1, 8, 64, "fp32"),
    ("configs[1] batch 32 (benchmark)", FULL, 32, 8, 64, "fp32"),
    ("configs[2] WV-3 256x256, ch 64-512, fp32", WIDE, 1, 8, 256, "fp32"),
    ("configs[2] WV-3 256x256, ch 64-512, bf16", WIDE, 1, 8, 256, "bf16"),
    ("configs[3] finetune local batch 8 (forward graph)", FULL, 8, 8, 64, "fp32"),
    ("configs[4] GF-2 tiles, batch 32", FULL, 32, 4, 64, "fp32"),
    ("configs[4] WV-3 tiles, batch 32", FULL, 32, 8, 64, "fp32"),
    ("configs[4] GF-2 tiles, batch 8", FULL, 8, 4, 64, "fp32"),
    ("configs[4] GF-2 256x256 tiles, batch 1", FULL, 1, 4, 256, "fp32"),
)
# ... and the widths the reference's constructor defaults to (WavBEST(channels=None) -> [16, 32, 64, 128],
# Hyper_unet_general.py:524-527) / the reference-generated TINY fixtures use: output channels that are not multiples of 32 run on
# the general-shape direct kernel ("fused"), which no BASELINE width needs
OTHER_CASES = (
    ("reference default widths, batch 32", [16, 32, 64, 128], 32, 8, 64, "fp32"),
    ("TINY fixture widths, batch 2", [4, 8, 16, 32], 2, 8, 16, "fp32"),
)
