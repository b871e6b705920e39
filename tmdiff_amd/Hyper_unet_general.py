"""WavBEST denoising UNet on MI355X: the reference's module tree, HIP kernels underneath.

Mirrors the API of the reference's GeneralModel/Hyper_unet_general.py -- ``WavBEST(channels,
embed_dim, inter_dim)`` and ``forward(x_t, t_input, PAN, MS, prompt)`` (:523-636) -- and keeps
its parameter names and shapes, so a reference checkpoint loads (``clip_text_model.*`` keys are
ignored).  The nn.Conv3d / nn.Linear objects below are parameter containers only: no ATen
convolution or matmul runs in ``forward``.  What runs instead (see include/tmdiff_hip.h):

  * every 3x3x3 / 1x1x1 / grouped conv  -> tmdiff_conv3d_fwd (fp32 MFMA implicit GEMM) with the
    SiLU, timestep shift, text modulation, bias, residual add fused in; ``torch.cat`` inputs are
    read from their three sources in place;
  * Haar DWT / IDWT                      -> butterfly kernels, /2 and x2 folded in, LL-only where
    the reference throws the high bands away (:390, :394, :620-627);
  * embedding MLPs and all Dense()s      -> two bank-of-linear launches per step;
  * the condition branch (conv1, down*_1, convH_0 of the skips, every Dense(prompt)) does not
    depend on x_t or t (:611-618, flag=True) and is computed once per image set and reused
    by all sampling steps (``begin_condition_cache`` / ``end_condition_cache``).

Text conditioning: the reference encodes five fixed paragraphs with a frozen CLIP text model at
construction (:561-598).  CLIP weights are not part of this package; ``text_embeddings`` (a
dict prompt -> [1,768] tensor, or a path to a torch-saved dict) injects the pooled vectors,
defaulting to fixed seeded vectors (``synthetic_text_embeddings``).
"""
import math

import torch
import torch.nn as nn

from . import ops

PROMPTS = ("QB", "WV3", "GF2", "WV2", "WV4")
# (behaviour switches: ops.config)

def synthetic_text_embeddings(seed=1234):
    """prompt -> randn(1,768) from a CPU generator seeded ``seed + index``."""
    out = {}
    for k, name in enumerate(PROMPTS):
        g = torch.Generator(device="cpu").manual_seed(seed + k)
        out[name] = torch.randn(1, 768, generator=g)
    return out


class Swish(nn.Module):
    """Kept for module-tree parity (it has no parameters); SiLU is fused into the conv prologues."""

    def forward(self, x):  # pragma: no cover - not on the HIP path
        raise RuntimeError("Swish is fused into the HIP kernels; it is never called as a module")


# ---- block modules -------------------------------------------------------------------------------------------------
# Same constructors, parameter names and ``forward`` signatures as the reference blocks (:158-273, :334-414).  Their
# ``forward`` is the differentiable path through the autograd-wrapped HIP operators (tmdiff_amd.autograd) and is what
# ``WavBEST.forward_train`` is assembled from (``run`` = the same block with its Dense() projections already evaluated,
# so the network can evaluate all of them in two launches); the fused no-grad inference path of ``WavBEST.forward``
# reads the parameters directly and never calls these.

_MASK_FN = None


def set_dropout_mask_fn(fn):
    """Parity hook: ``fn(shape) -> CPU/GPU float tensor`` supplies every dropout mask (already scaled by 1/(1-p)), in
    the order the reference's nn.Dropout modules are called; ``None`` restores the device generator."""
    global _MASK_FN
    _MASK_FN = fn


def _drop_mask(module, shape, p, device):
    """Dropout of a convolution input: None (eval), a mask tensor from the parity hook, or a (seed, p) spec for the
    in-kernel generator.  The seed is drawn from torch's CPU generator (so torch.manual_seed reproduces a run and
    per-rank seeds give per-rank masks) -- a host-side draw, no device work and no mask tensor in HBM."""
    if not module.training or p <= 0.0:
        return None
    if _MASK_FN is not None:
        return _MASK_FN(tuple(shape)).to(device=device, dtype=torch.float32).contiguous()
    from .autograd import DropSpec
    return DropSpec(int(torch.randint(0, 2 ** 62, (1,)).item()), p)


def _lin(x, layer):
    from . import autograd as A
    return A.linear(x, layer.weight, layer.bias)


def _need_gpu(t):
    if t.device.type != "cuda":
        raise RuntimeError("tmdiff_amd blocks run on the HIP kernels only: move the module and its inputs to a GPU")


class Dense(nn.Module):
    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.dense = nn.Linear(input_dim, output_dim)

    def forward(self, x):
        return _lin(x, self.dense)[..., None, None]


class AdaptionModulateBEST(nn.Module):
    def __init__(self, channel_in, channel_out, embed_dim):
        super().__init__()
        self.conv20 = nn.Conv3d(channel_in, channel_out, 1)
        self.conv21 = nn.Conv3d(channel_out, channel_out, 3, padding=1)
        self.act = Swish()
        self.dense2 = Dense(embed_dim, channel_out)

    def run(self, scale, **inp):
        """inp: xin=[B,N,H,W], or pan=[B,1,H,W] and ms=[B,N,H,W] (the condition PAN*1 - MS formed in the kernel)."""
        from . import autograd as A
        a0 = A.stem(self.conv20.weight, self.conv20.bias, **inp)
        return A.conv3d([a0], self.conv21.weight, None, scale=scale)

    def forward(self, h, embed, context):
        _need_gpu(h)
        if h.shape[1] != 1:
            raise ValueError("AdaptionModulateBEST takes a one-channel input [B, 1, N, H, W]")
        return self.run(_lin(context, self.dense2.dense), xin=h[:, 0].contiguous())


class ResBlockModulateBEST(nn.Module):
    def __init__(self, channel_in, channel_out, embed_dim, flag=False):
        super().__init__()
        self.conv20 = nn.Conv3d(channel_in, channel_out, 3, padding=1)
        self.conv21 = nn.Conv3d(channel_out, channel_out, 3, padding=1)
        self.dense1 = Dense(embed_dim, channel_in)
        self.dense2 = Dense(embed_dim, channel_out)
        self.dropout = nn.Dropout(0.2)
        self.res_conv = nn.Conv3d(channel_in, channel_out, 1) if channel_in != channel_out else nn.Identity()
        self.act = Swish()
        self.flag = flag

    def run(self, segs, shift, scale):
        """segs: the channel segments of the input (a torch.cat the reference materialises, :631-634)."""
        from . import autograd as A
        b, _, n, h, w = segs[0].shape
        cin = sum(s.shape[1] for s in segs)
        dev = segs[0].device
        m20 = _drop_mask(self, (b, cin, n, h, w), self.dropout.p, dev)
        if isinstance(self.res_conv, nn.Conv3d) and ops.config.train_fused_resblock:      # one autograd node: the two gradients of every input segment meet in a kernel
            m21 = _drop_mask(self, (b, self.conv20.out_channels, n, h, w), self.dropout.p, dev)
            return A.resblock_rc(segs, self.conv20.weight, self.conv20.bias, self.conv21.weight, self.res_conv.weight,
                                 self.res_conv.bias, None if self.flag else shift, scale, m20, m21)
        if not isinstance(self.res_conv, nn.Conv3d) and len(segs) == 1 and ops.config.train_fused_resblock:
            m21 = _drop_mask(self, (b, self.conv20.out_channels, n, h, w), self.dropout.p, dev)
            return A.resblock_id(segs[0], self.conv20.weight, self.conv20.bias, self.conv21.weight, None if self.flag else shift, scale, m20, m21)
        t1 = A.conv3d(segs, self.conv20.weight, self.conv20.bias, shift=None if self.flag else shift, act=True, mask=m20)
        res = A.conv3d(segs, self.res_conv.weight, self.res_conv.bias) if isinstance(self.res_conv, nn.Conv3d) else segs[0]
        return A.conv3d([t1], self.conv21.weight, None, scale=scale, act=True,
                        mask=_drop_mask(self, t1.shape, self.dropout.p, dev), residual=res)

    def forward(self, x, embed, prompt):
        _need_gpu(x)
        return self.run([x], None if self.flag else _lin(embed, self.dense1.dense), _lin(prompt, self.dense2.dense))


class WaveletUPorDown(nn.Module):
    def __init__(self, act=None, in_ch=None, out_ch=None, temb_dim=None, up=False, down=False, flag=False,
                 dropout=0.2, skip_rescale=False, zemb_dim=None, init_scale=0.0, hi_in_ch=None):
        super().__init__()
        out_ch = out_ch if out_ch else in_ch
        if skip_rescale or out_ch != in_ch:
            raise NotImplementedError("WavBEST only instantiates skip_rescale=False, out_ch == in_ch")
        self.up, self.down, self.flag = up, down, flag
        self.in_ch = self.out_ch = in_ch
        self.Conv_0 = nn.Conv3d(in_ch, out_ch, 3, padding=1)
        self.Dense_0 = nn.Linear(temb_dim, out_ch)
        with torch.no_grad():            # DDPM "default_init": fan-avg uniform (ref :417-454), zero bias
            bound = math.sqrt(3.0 * 1.0 / ((temb_dim + out_ch) / 2))
            self.Dense_0.weight.uniform_(-bound, bound)
            self.Dense_0.bias.zero_()
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = nn.Conv3d(out_ch, out_ch, 3, padding=1)
        self.Conv_2 = nn.Conv3d(in_ch, out_ch, 1)
        if up:
            self.convH_0 = nn.Sequential(nn.Conv3d(hi_in_ch * 3, out_ch * 3, 3, padding=1, groups=3))
        self.dense1 = Dense(zemb_dim, in_ch)
        self.dense2 = Dense(zemb_dim, in_ch)      # exists in the reference (:366), unused by its forward

    def run(self, x, shift, scale, skipH=None, want_high=True):
        from . import autograd as A
        # down, high bands dropped: Conv_0 + halved LL band as one strided convolution (autograd._ConvLL)
        ll = (self.down and not want_high and ops.config.ll_compose and
              ops.ll_conv_supported(self.Conv_0.out_channels, self.Conv_0.in_channels, 3, 1) and
              x.shape[3] % 2 == 0 and x.shape[4] % 2 == 0)
        hh = None if ll else A.conv3d([x], self.Conv_0.weight, self.Conv_0.bias, act=True)
        # down: Conv_2 commutes with the halved LL band (see WavBEST._down), so it runs after it, on a quarter of the positions
        xx = None if self.down and ops.config.conv2_after_ll else A.conv3d([x], self.Conv_2.weight, self.Conv_2.bias)
        hH = None
        if self.up:
            ch = self.convH_0[0]
            bands = A.conv3d(list(skipH), ch.weight, ch.bias, bias_scale=2.0, groups=3)      # convH_0(cat/2)*2
            h_in, x_in = A.haar_idwt2d_pair(hh, xx, bands, in_scale=2.0)
        elif self.down:
            if ll:
                h_in, lh, hl, hhh = A.conv3d_ll(x, self.Conv_0.weight, self.Conv_0.bias, 0.5), None, None, None
            else:
                h_in, lh, hl, hhh = A.haar_dwt2d(hh, want_high=want_high, ll_scale=0.5)
            if xx is None:
                x_in = A.conv3d([A.haar_dwt2d(x, want_high=False, ll_scale=0.5)[0]], self.Conv_2.weight, self.Conv_2.bias)
            else:
                x_in = A.haar_dwt2d(xx, want_high=False, ll_scale=0.5)[0]
            hH = (lh, hl, hhh)
        else:
            raise NotImplementedError("WavBEST only instantiates up or down wavelet blocks")
        out = A.conv3d([h_in], self.Conv_1.weight, None, shift=None if self.flag else shift, scale=scale, act=True,
                       mask=_drop_mask(self, h_in.shape, self.Dropout_0.p, h_in.device), residual=x_in)
        return (out, hH) if self.down else out

    def forward(self, x, temb=None, zemb=None, skipH=None):
        _need_gpu(x)
        return self.run(x, None if self.flag else _lin(temb, self.Dense_0), _lin(zemb, self.dense1.dense), skipH)


class ResblockDownOneModulateBEST(nn.Module):
    def __init__(self, channel_in, channel_out, embed_dim, flag=False):
        super().__init__()
        self.conv20 = ResBlockModulateBEST(channel_in, channel_out, embed_dim, flag)
        self.down = WaveletUPorDown(in_ch=channel_out, temb_dim=embed_dim, zemb_dim=embed_dim, down=True, flag=flag)

    def forward(self, x, embed, prompt):
        return self.down(self.conv20(x, embed, prompt), embed, prompt)


class ResblockUpOneModulateBEST(nn.Module):
    def __init__(self, channel_in, channel_out, embed_dim):
        super().__init__()
        self.up1 = WaveletUPorDown(in_ch=channel_out, temb_dim=embed_dim, zemb_dim=embed_dim, up=True,
                                   hi_in_ch=channel_in)
        self.conv20 = ResBlockModulateBEST(channel_in * 3, channel_out, embed_dim)

    def forward(self, x, embed, skipH, prompt):
        return self.up1(self.conv20(x, embed, prompt), embed, prompt, skipH)


class FinalBlockModulateBEST(nn.Module):
    def __init__(self, channel_in, channel_out, embed_dim):
        super().__init__()
        self.conv20 = ResBlockModulateBEST(channel_in * 3, channel_in, embed_dim)
        self.conv21 = ResBlockModulateBEST(channel_in, channel_in, embed_dim)
        self.conv22 = ResBlockModulateBEST(channel_in, channel_in, embed_dim)
        self.conv23 = ResBlockModulateBEST(channel_in, channel_in, embed_dim)
        self.conv24 = nn.Conv3d(channel_in, channel_out, 1)
        self.dense2 = Dense(embed_dim, channel_in)
        self.act = Swish()

    def forward(self, x, embed, prompt):
        from . import autograd as A
        if self.conv24.out_channels != 1:
            raise NotImplementedError("the head kernel produces one output channel (WavBEST: channel_out = 1)")
        h = x
        for blk in (self.conv20, self.conv21, self.conv22, self.conv23):
            h = blk(h, embed, prompt)
        return A.head(h, self.conv24.weight, _lin(prompt, self.dense2.dense)).unsqueeze(1)


class _Bank:
    """A set of Linear layers over the same input evaluated by ONE tmdiff_linear_fwd launch.
    ``run(x)`` returns the [rows, total] output; ``slot(out, name)`` gives the device pointer of
    that layer's first output feature plus the row stride the conv prologue needs."""

    def __init__(self, layers, device):
        self.offsets, off = {}, 0
        for n, lin in layers:
            self.offsets[n] = off
            off += lin.out_features
        self.total = off
        self.weight = torch.cat([lin.weight.detach().to(device, torch.float32) for _, lin in layers]).contiguous()
        self.bias = torch.cat([lin.bias.detach().to(device, torch.float32) for _, lin in layers]).contiguous()

    def run(self, x):
        return ops.linear(x, self.weight, self.bias, act=False)

    def slot(self, out, name):
        stride = -1 if out.shape[0] == 1 else self.total       # one row: broadcast over the batch
        return out.data_ptr() + 4 * self.offsets[name], stride


class WavBEST(nn.Module):
    def __init__(self, channels=None, embed_dim=128, inter_dim=32, text_embeddings=None):
        super().__init__()
        if channels is None:
            channels = [16, 32, 64, 128]
        c, e = list(channels), embed_dim
        self.channels = c
        self.inter_dim = inter_dim
        self.embed = nn.Sequential(nn.Linear(inter_dim, e), Swish(), nn.Linear(e, e))
        self.embed2 = nn.Sequential(nn.Linear(768, e * 4), Swish(), nn.Linear(e * 4, e * 4), Swish(),
                                    nn.Linear(e * 4, e))
        self.conv1 = AdaptionModulateBEST(1, c[0], e)
        self.conv2 = AdaptionModulateBEST(1, c[0], e)
        self.down1 = ResblockDownOneModulateBEST(c[0], c[1], e)
        self.down2 = ResblockDownOneModulateBEST(c[1], c[2], e)
        self.down3 = ResblockDownOneModulateBEST(c[2], c[3], e)
        self.down1_1 = ResblockDownOneModulateBEST(c[0], c[1], e, flag=True)
        self.down2_1 = ResblockDownOneModulateBEST(c[1], c[2], e, flag=True)
        self.down3_1 = ResblockDownOneModulateBEST(c[2], c[3], e, flag=True)
        self.middle1 = ResBlockModulateBEST(c[3], c[3], e)
        self.up1 = ResblockUpOneModulateBEST(c[3], c[2], e)
        self.up2 = ResblockUpOneModulateBEST(c[2], c[1], e)
        self.up3 = ResblockUpOneModulateBEST(c[1], c[0], e)
        self.final = FinalBlockModulateBEST(c[0], 1, e)
        self.act = Swish()
        if isinstance(text_embeddings, str):
            text_embeddings = torch.load(text_embeddings, map_location="cpu")
        self.text_embeddings = {k: v.detach().float().reshape(1, 768) for k, v in
                                (text_embeddings or synthetic_text_embeddings()).items()}
        # fp32 frequency table of gamma_embedding (ref :89-92), computed on the host exactly like the reference
        half = inter_dim // 2
        self._freqs_cpu = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
        self.compute_dtype = "fp32"   # "bf16": bf16 operands / fp32 accumulation in the 3x3x3 convs (inference only)
        self._prep = None        # packed weights / banks, keyed on parameter versions
        self._cond = None        # pinned step-invariant tensors of the current image set

    # ---- reference-shaped helpers ---------------------------------------------------------------
    def get_prompt(self, prompt):
        """Sensor paragraph for a prompt name, ``None`` for an unknown one (ref :574-585)."""
        from .prompts import PROMPT_TEXT
        return PROMPT_TEXT.get(prompt)

    def get_embeding(self, prompt):
        return self.text_embeddings.get(prompt)

    def load_state_dict(self, state_dict, strict=True, **kw):
        state_dict = {k: v for k, v in state_dict.items() if not k.startswith("clip_text_model.")}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def set_compute_dtype(self, dtype):
        """"fp32" (default, exact-fp32 MFMA) or "bf16": the 3x3x3 convolutions of the no-grad forward round their
        operands to bf16 and accumulate in fp32 ("bf16 compute / fp32 accumulate").
        Activations, the embedding MLPs, wavelets, 1x1x1 convs, the sampler and all of training stay fp32."""
        dtype = {"float32": "fp32", "bfloat16": "bf16", torch.float32: "fp32", torch.bfloat16: "bf16"}.get(dtype, dtype)
        if dtype not in ("fp32", "bf16"):
            raise ValueError(f"compute dtype must be 'fp32' or 'bf16', got {dtype!r}")
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._prep = None
            self._cond = None
        return self

    # ---- weight-dependent preparation (redone when any parameter changes) --------------------------
    def _prepare(self):
        params = list(self.parameters())
        key = (tuple(p._version for p in params), tuple(p.data_ptr() for p in params), self.compute_dtype, ops.config.key())
        if self._prep is not None and self._prep["key"] == key:
            return self._prep
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("tmdiff_amd.WavBEST runs on the HIP kernels only: move the module to a GPU (.cuda())")
        prep = {"key": key, "w": {}, "w_ll": {}, "w_wfll": {}, "w_wino": {}, "bf16": set(), "freqs": self._freqs_cpu.to(dev)}
        for name, m in self.named_modules():
            if isinstance(m, nn.Conv3d) and m.in_channels > 1 and m.out_channels > 1:
                w = m.weight.detach().float().contiguous()
                # every multi-segment input of the network splits into equal segments (3 x c): check their size too
                seg = [m.in_channels // 3] if m.in_channels % 3 == 0 else None
                if self.compute_dtype == "bf16" and ops.bf16_conv_supported(m.out_channels, m.in_channels,
                                                                            m.kernel_size[0], m.groups, seg):
                    prep["w"][name] = ops.pack_conv_weight_bf16(w, groups=m.groups)
                    prep["bf16"].add(name)
                else:
                    prep["w"][name] = ops.pack_conv_weight(w, groups=m.groups)
                    # exact-fp32 mode: the 3x3x3 convolutions also in the Winograd F(2,3)-along-n form (csrc/conv3d_wino.hip:
                    # 1.5x fewer multiply-adds; taken when the band count is even and the grid fills the chip)
                    if ops.config.winograd and ops.wino_conv_supported(m.out_channels, m.in_channels, m.kernel_size[0], m.groups):
                        prep["w_wino"][name] = {}      # packed on first use, per plane count of the transform (the band count decides)
        # Conv_0 of the main branch's down blocks is followed by an LL-only DWT (its high bands are dropped): the pair runs
        # as one strided convolution on composed weights (csrc/conv3d_ll.hip), exact-fp32 mode only
        if self.compute_dtype == "fp32" and ops.config.ll_compose:
            for blk in ("down1", "down2", "down3"):
                m = self.get_submodule(blk + ".down.Conv_0")
                if ops.ll_conv_supported(m.out_channels, m.in_channels, m.kernel_size[0], m.groups):
                    prep["w_ll"][blk + ".down.Conv_0"] = ops.pack_conv_weight_ll(m.weight.detach().float().contiguous(), 0.5)
                    # ... and with Winograd along the bands on top (conv3d_wf's composed-LL mode), for the launches whose producer
                    # can hand over its second output in space-to-depth form (_ll_s2d)
                    if ops.config.winograd and ops.config.wfll and m.out_channels % 32 == 0:
                        prep["w_wfll"][blk + ".down.Conv_0"] = ops.pack_conv_weight_wfll(m.weight.detach().float().contiguous(), 0.5)
        shift, scale = [], []
        for name, m in self.named_modules():
            if isinstance(m, ResBlockModulateBEST):
                scale.append((name + ".dense2", m.dense2.dense))
                if not m.flag:
                    shift.append((name + ".dense1", m.dense1.dense))
            elif isinstance(m, WaveletUPorDown):
                scale.append((name + ".dense1", m.dense1.dense))
                if not m.flag:
                    shift.append((name + ".Dense_0", m.Dense_0))
            elif isinstance(m, (AdaptionModulateBEST, FinalBlockModulateBEST)):
                scale.append((name + ".dense2", m.dense2.dense))
        prep["shift_bank"] = _Bank(shift, dev)
        prep["scale_bank"] = _Bank(scale, dev)
        self._prep = prep
        self._cond = None
        return prep

    # ---- fused building blocks -------------------------------------------------------------------
    def _conv(self, P, name, segs, use_bias=True, bias_scale=1.0, bias=None, **kw):
        """One convolution of the fused inference graph on the kernel family tmdiff_amd.routing picks for its extents.
        bias: another module's bias instead of this one's (a folded res_conv's)."""
        m = self.get_submodule(name)
        if bias is None:
            bias = m.bias.detach() if (use_bias and m.bias is not None) else None
        math = "bf16" if name in P["bf16"] else "fp32"
        if m.kernel_size[0] != 3:
            return ops.conv3d(segs, P["w"][name], m.out_channels, m.kernel_size[0], groups=m.groups, math=math, bias=bias,
                              bias_scale=bias_scale, **kw)
        ww = P["w_wino"].get(name)        # the Winograd forms of this weight, packed on first use (None: shape not taken)

        def packed(planes):               # transform-pass kernels (tmdiff_amd.fallback): per plane count of the transform
            if planes not in ww:
                ww[planes] = ops.pack_conv_weight_wino(m.weight.detach().float().contiguous(), groups=m.groups, planes=planes)
            return ww[planes]

        def packed_wf():                  # conv3d_wf: F(4,3), natural column order (32-channel tiles)
            if "wf" not in ww:
                ww["wf"] = ops.pack_conv_weight_wino(m.weight.detach().float().contiguous(), groups=m.groups, mode=2, planes=6)
            return ww["wf"]

        weights = ops.ConvWeights(lambda: P["w"][name], packed_wf if ww is not None else None, packed if ww is not None else None)
        return ops.conv3d_auto(segs, weights, m.out_channels, groups=m.groups, math=math, bias=bias, bias_scale=bias_scale, **kw)

    @staticmethod
    def _shift(P, S, name):
        ptr, stride = P["shift_bank"].slot(S["shift"], name)
        return {"in_shift": ptr, "shift_stride": stride}

    @staticmethod
    def _scale(P, S, name):
        ptr, stride = P["scale_bank"].slot(S["scale"], name)
        return {"in_scale": ptr, "scale_stride": stride}

    # Producer-side prologues (fp32 inference): wherever a tensor has ONE convolution as its consumer, the kernel that
    # produces it also writes the consumer's prologue output act(y + shift) * scale (a second output of a convolution's
    # epilogue, the LL band of a DWT, the first reconstruction of an IDWT, the stem), so the consumer is a plain-input
    # convolution: it runs on the staged kernel (operands by LDS-DMA, matrix pipe busy 0.85-0.89 at 2.3 GHz against
    # 0.74-0.83 at 2.1 GHz for the kernel that applies the prologue while staging) without a prologue pass.  Same bits
    # as the consumer-side prologue.  `pre` = such a tensor for a block's first convolution; `emit` = the prologue spec
    # of the block's consumer.  Multi-segment consumers (the up path's conv20) keep their own prologue pass.
    def _spec(self, P, S, shift=None, scale=None, act=True):
        d = {"act": act}
        if shift is not None:
            d["shift"], d["shift_stride"] = P["shift_bank"].slot(S["shift"], shift)
        if scale is not None:
            d["scale"], d["scale_stride"] = P["scale_bank"].slot(S["scale"], scale)
        return d

    def _resblock(self, P, S, name, segs, flag, pre=None, emit=None, want_ll=False):
        """ResBlockModulateBEST (ref :237-249): conv20 with fused (shift,) SiLU; optional 1x1x1
        res_conv; conv21 with fused SiLU + text modulation + residual add.  Returns (y, y2): y2 = `emit` applied to y
        (None without emit).  want_ll (the block in front of a down block, whose raw output is read by nothing but the LL
        band of the Conv_2 path): where conv21's launch can (`_emit_ll`), it writes LL(y) / 2 instead of y -- returns
        (None, y2, y_ll); else (y, y2, None)."""
        rb = self.get_submodule(name)
        sh = {} if flag else self._shift(P, S, name + ".dense1")
        # res_conv (1x1x1, where the channel count changes) folded into conv21's epilogue where conv21 runs on conv3d_wf unsplit:
        # W1^T x is accumulated by the matrix pipe into conv21's output blocks -- no launch of its own, no residual tensor
        # written and read back (19 -> 13 1x1x1 launches per step at the benchmark batch)
        rc = side = None
        if isinstance(rb.res_conv, nn.Conv3d):
            rc = self._fold_res_conv(P, name, segs)
            if rc is None and pre is None and self._side_xp(P, name, segs):
                # a segmented input (an up block's concat): res_conv's launch reads every element anyway and also writes
                # conv20's prologue output SiLU(x + shift) -- conv20 then reads one plain tensor, no prologue pass
                side = ops.scratch_like(segs, "side_xp")
                res = self._conv(P, name + ".res_conv", segs, side_xp=dict(
                    out=side, shift=sh.get("in_shift"), shift_stride=sh.get("shift_stride", 0), act=True))
            else:
                res = None if rc is not None else self._conv(P, name + ".res_conv", segs)
        else:
            res = segs[0]
        rckw = {} if rc is None else {"res_conv": rc, "bias": rb.res_conv.bias.detach()}
        sc = self._scale(P, S, name + ".dense2")
        # conv20's result feeds conv21 only: its epilogue applies conv21's prologue (SiLU, text modulation) and conv21
        # reads that directly -- a plain fp32 tensor for the staged kernel, or the packed bf16 units in the bf16 mode
        # (no prologue / pack pass in between; same bits either way)
        both16 = name + ".conv20" in P["bf16"] and name + ".conv21" in P["bf16"]
        kw = {} if emit is None else {"emit": emit}
        if ops.config.epilogue_fuse and (both16 or (name + ".conv20" not in P["bf16"] and name + ".conv21" not in P["bf16"])):
            mid = dict(act=True, scale=sc["in_scale"], scale_stride=sc["scale_stride"])
            if side is not None:    # (res_conv's launch wrote conv20's prologue output: a plain-input convolution)
                t1p = self._conv(P, name + ".conv20", [side], keep_y=False, emit=mid)
            elif pre is not None:   # (bf16 mode: `pre` is the packed bf16 form a bf16 producer wrote)
                t1p = self._conv(P, name + ".conv20", [pre], keep_y=False, emit=mid,
                                 x_bf16_shape=tuple(segs[0].shape[2:]) if pre.dtype == torch.int16 else None)
            else:
                t1p = self._conv(P, name + ".conv20", segs, in_act=True, keep_y=False, emit=mid, **sh)
            shape = tuple(segs[0].shape[2:]) if both16 else None
            if want_ll and emit is not None and self._emit_ll(P, name, segs[0]):
                y2, yll = self._conv(P, name + ".conv21", [t1p], use_bias=False, residual=res, keep_y=False,
                                     emit=dict(emit, ll=True), **rckw)
                return None, y2, yll
            out = self._conv(P, name + ".conv21", [t1p], use_bias=False, residual=res, x_bf16_shape=shape, **rckw, **kw)
        else:
            assert pre is None
            t1 = self._conv(P, name + ".conv20", segs, in_act=True, **sh)
            out = self._conv(P, name + ".conv21", [t1], use_bias=False, in_act=True, residual=res, **sc, **rckw, **kw)
        out = out if emit is not None else (out, None)
        return (*out, None) if want_ll else out

    def _emit_ll(self, P, name, x, conv=".conv21", switch="emit_ll"):
        """True when the convolution `name + conv` (a ResBlock's conv21; a down block's Conv_0 for the whole Haar transform) can
        write the halved LL band / the Haar transform of its output instead of the output itself (desc.y_ll / y_hi): fp32,
        8 bands, even H, planes of at least 16 columns, the convolution on conv3d_wf without splitting its input channels, and
        the down block's Conv_2 after the LL band (ops.config.conv2_after_ll).  x: a tensor of the convolution's input extents."""
        cfg = ops.config
        if not (getattr(cfg, switch) and cfg.conv2_after_ll and cfg.epilogue_fuse) or (name + conv) in P["bf16"]:
            return False
        m = self.get_submodule(name + conv)
        b, _, n, h, w = x.shape
        if P["w_wino"].get(name + conv) is None or n != 8 or w == 8 or h % 2 or w % 4 or m.groups != 1:
            return False
        from . import routing
        return (routing.conv3_family(b, m.in_channels, m.out_channels, n, h, w, 1, plain=False) == "wf" and
                routing.wf_route(b, m.in_channels, m.out_channels, n, h, w)[1] == 1)

    def _side_xp(self, P, name, segs):
        """True when res_conv of the ResBlock `name` can write conv20's prologue output on the side (make_conv_desc side_xp=):
        fp32 inference with the conv20 -> conv21 epilogue fusion, a segmented input, conv20 on conv3d_wf (whose prologue would
        otherwise be a pass of its own), res_conv on the 16-byte bandwidth kernel."""
        cfg = ops.config
        if not (cfg.side_xp and cfg.epilogue_fuse) or len(segs) < 2 or any((name + k) in P["bf16"] for k in (".conv20", ".conv21", ".res_conv")):
            return False
        m = self.get_submodule(name + ".conv20")
        b, _, n, h, w = segs[0].shape
        from . import routing
        if P["w_wino"].get(name + ".conv20") is None or m.groups != 1:
            return False
        if routing.conv3_family(b, m.in_channels, m.out_channels, n, h, w, 1, plain=False) not in ("wf", "wf_pair"):
            return False
        return routing.k1_side_xp(b, [s.shape[1] for s in segs], self.get_submodule(name + ".res_conv").out_channels, n, h, w)

    def _fold_res_conv(self, P, name, segs, k1=".res_conv", k3=".conv21"):
        """(x, the 1x1x1 weight, Cx) when the 1x1x1 convolution `name + k1` -- a ResBlock's res_conv, a down block's Conv_2 --
        whose result is only ever the residual of the 3x3x3 convolution `name + k3` can ride in that convolution's epilogue
        (make_conv_desc res_conv=), else None: fp32, one input tensor x of a multiple of 32 channels at the consumer's plane
        size, the consumer on conv3d_wf (not its pair mode; a split-K grid adds it to the partial sums of its first range)."""
        if not ops.config.fuse_res_conv or len(segs) != 1 or (name + k3) in P["bf16"] or (name + k1) in P["bf16"]:
            return None
        m3, x = self.get_submodule(name + k3), segs[0]
        b, cx, n, h, w = x.shape
        if P["w_wino"].get(name + k3) is None or cx % 32 or cx > 512 or m3.groups != 1 or (n == 8 and w == 8):
            return None
        from . import routing
        if routing.conv3_family(b, m3.in_channels, m3.out_channels, n, h, w, 1, plain=False) != "wf":
            return None
        return x, self.get_submodule(name + k1).weight.detach(), cx

    def _ll_s2d(self, P, blk, h):
        """True when the main branch's down block `blk` runs Conv_0 + LL as conv3d_wf_ll: its weights exist, the ResBlock in
        front ends in a conv3d_wf launch that does not split its input channels (only that epilogue writes the space-to-depth
        form), and the composed convolution's own grid is taken by the kernel.  h: the ResBlock's input."""
        if not ops.config.epilogue_fuse or P["w_wfll"].get(blk + ".down.Conv_0") is None:
            return False
        c21, c0 = blk + ".conv20.conv21", self.get_submodule(blk + ".down.Conv_0")
        if P["w_wino"].get(c21) is None or c21 in P["bf16"] or (blk + ".conv20.conv20") in P["bf16"]:
            return False
        m21 = self.get_submodule(c21)
        b, _, n, hh, ww = h.shape
        takes, split = ops.wf_route(b, m21.in_channels, m21.out_channels, n, hh, ww, m21.groups)
        return bool(takes and split == 1 and m21.out_channels == c0.in_channels and
                    ops.wfll_route(b, c0.in_channels, c0.out_channels, n, hh, ww))

    def _conv0(self, P, name, x, pre):
        """Conv_0 of a wavelet block on SiLU(x): from the producer's second output when there is one."""
        if pre is None:
            return self._conv(P, name + ".Conv_0", [x], in_act=True)
        return self._conv(P, name + ".Conv_0", [pre], x_bf16_shape=tuple(x.shape[2:]) if pre.dtype == torch.int16 else None)

    def _down(self, P, S, name, x, flag, want_high, pre=None, emit=None, fuse=False, pre_s2d=False, xq=None):
        """WaveletUPorDown(down=True) (ref :369-414); /2 folded into the DWT, LL-only when the
        caller drops the high bands.  Returns (out, out2, bands).  pre_s2d: `pre` is in space-to-depth form (_ll_s2d).
        xq: LL(x) / 2 as the producer already wrote it (_resblock(want_ll=True)); x itself may then be None."""
        w_ll = None if want_high or pre is None or pre.dtype != torch.float32 else P["w_wfll" if pre_s2d else "w_ll"].get(name + ".Conv_0")
        assert not pre_s2d or w_ll is not None
        conv_ll = ops.conv3d_wf_ll if pre_s2d else ops.conv3d_ll
        # high bands kept (condition branch): where Conv_0 runs on conv3d_wf unsplit, its epilogue writes the Haar transform of its
        # output -- LL through Conv_1's prologue, LH, HL, HH -- instead of the output: no full-resolution tensor, no DWT pass
        dwt4 = None
        if (w_ll is None and fuse and want_high and pre is not None and pre.dtype == torch.float32 and
                self._emit_ll(P, name, pre, conv=".Conv_0", switch="emit_dwt")):
            pro = self._spec(P, S, shift=None if flag else name + ".Dense_0", scale=name + ".dense1")
            dwt4 = self._conv(P, name + ".Conv_0", [pre], keep_y=False, emit=dict(pro, dwt=True))
        hh = None if (w_ll is not None or dwt4 is not None) else self._conv0(P, name, x, pre)
        # The reference runs the 1x1x1 Conv_2 at full resolution and keeps the halved LL band of its output (:390, :396).
        # Both are linear and act on different axes (channels / the 2x2 pixel block), and the halved LL band of a
        # constant is that constant, so LL(Conv_2(x)) / 2 == Conv_2(LL(x) / 2): the convolution runs on a quarter of
        # the positions and the full-resolution intermediate is never written (same value up to fp32 summation order).
        rc2 = {}
        if ops.config.conv2_after_ll:
            if xq is None:
                xq = ops.haar_dwt2d(x, want_high=False, ll_scale=0.5)[0]
            # ... and where Conv_1 runs on conv3d_wf unsplit, Conv_2 rides in its epilogue like a ResBlock's res_conv
            rc = self._fold_res_conv(P, name, [xq], k1=".Conv_2", k3=".Conv_1")
            if rc is not None:
                xll, rc2 = None, {"res_conv": rc, "bias": self.get_submodule(name + ".Conv_2").bias.detach()}
            else:
                xll = self._conv(P, name + ".Conv_2", [xq])
        else:
            xll = ops.haar_dwt2d(self._conv(P, name + ".Conv_2", [x]), want_high=False, ll_scale=0.5)[0]
        kw = dict(rc2) if emit is None else dict(rc2, emit=emit)
        if w_ll is not None:
            # Conv_0 and the halved LL band of its output as ONE strided convolution (only the LL band is used here); with
            # `fuse` its epilogue applies Conv_1's prologue, as the DWT does below
            c0 = self.get_submodule(name + ".Conv_0")
            if fuse:
                pro = self._spec(P, S, shift=None if flag else name + ".Dense_0", scale=name + ".dense1")
                hll = conv_ll(pre, w_ll, c0.out_channels, 0.5, bias=c0.bias.detach(), emit=pro, keep_y=False)
                out = self._conv(P, name + ".Conv_1", [hll], use_bias=False, residual=xll, **kw)
            else:
                hll = conv_ll(pre, w_ll, c0.out_channels, 0.5, bias=c0.bias.detach())
                sh = {} if flag else self._shift(P, S, name + ".Dense_0")
                out = self._conv(P, name + ".Conv_1", [hll], use_bias=False, in_act=True, residual=xll,
                                 **self._scale(P, S, name + ".dense1"), **sh, **kw)
            out, out2 = out if emit is not None else (out, None)
            return out, out2, (None, None, None)
        if dwt4 is not None:
            hll, lh, hl, hhh = dwt4
            out = self._conv(P, name + ".Conv_1", [hll], use_bias=False, residual=xll, **kw)
        elif fuse:   # Conv_1's prologue (shift, SiLU, text modulation) is applied to the LL band where the DWT writes it
            pro = self._spec(P, S, shift=None if flag else name + ".Dense_0", scale=name + ".dense1")
            p16 = name + ".Conv_1" in P["bf16"]          # bf16 mode: ... as the packed bf16 units the convolution reads
            hll, lh, hl, hhh = ops.haar_dwt2d(hh, want_high=want_high, ll_scale=0.5, ll_prologue=pro, pack_bf16=p16)
            out = self._conv(P, name + ".Conv_1", [hll], use_bias=False, residual=xll,
                             x_bf16_shape=tuple(xll.shape[2:]) if p16 else None, **kw)
        else:
            hll, lh, hl, hhh = ops.haar_dwt2d(hh, want_high=want_high, ll_scale=0.5)
            sh = {} if flag else self._shift(P, S, name + ".Dense_0")
            out = self._conv(P, name + ".Conv_1", [hll], use_bias=False, in_act=True, residual=xll,
                             **self._scale(P, S, name + ".dense1"), **sh, **kw)
        out, out2 = out if emit is not None else (out, None)
        return out, out2, (lh, hl, hhh)

    def _up(self, P, S, name, x, bands, pre=None, fuse=False):
        """WaveletUPorDown(up=True) (ref :379-386, :398-408); ``bands`` = convH_0 output
        [B, 3C, N, h, w], step-invariant and cached with the condition branch."""
        hh = self._conv0(P, name, x, pre)
        xx = self._conv(P, name + ".Conv_2", [x])
        if fuse:   # Conv_1's prologue applied to the h reconstruction where the IDWT writes it
            pro = self._spec(P, S, shift=name + ".Dense_0", scale=name + ".dense1")
            p16 = name + ".Conv_1" in P["bf16"]
            h_up, x_up = ops.haar_idwt2d([hh, xx], None, None, None, in_scale=2.0, stacked_bands=bands, out0_prologue=pro,
                                         pack_bf16=p16)
            return self._conv(P, name + ".Conv_1", [h_up], use_bias=False, residual=x_up,
                              x_bf16_shape=tuple(x_up.shape[2:]) if p16 else None)
        h_up, x_up = ops.haar_idwt2d([hh, xx], None, None, None, in_scale=2.0, stacked_bands=bands)
        return self._conv(P, name + ".Conv_1", [h_up], use_bias=False, in_act=True, residual=x_up,
                          **self._shift(P, S, name + ".Dense_0"), **self._scale(P, S, name + ".dense1"))

    def _producer_fuse(self, P):
        """(conv -> conv edges, wavelet / stem producers): the first in both compute modes (in the bf16 mode the second
        output is the packed bf16 form, and only if every MFMA convolution of the network runs on the bf16 kernels), the
        second likewise (fp32 tensors, or the packed bf16 units from the *_pack_bf16 variants of the DWT / IDWT / stem)."""
        if not (ops.config.producer_fuse and ops.config.epilogue_fuse):
            return False, False
        if self.compute_dtype == "fp32":
            return True, True
        all16 = len(P["bf16"]) == len(P["w"])
        return all16, all16

    # ---- condition branch (independent of x_t and t) -------------------------------------------------
    def _prompt_rows(self, prompt, batch, device):
        if isinstance(prompt, (list, tuple)):            # per-sample prompts (mixed-satellite batches)
            if len(prompt) != batch:
                raise ValueError("per-sample prompt list must have one entry per batch element")
            rows = [self.get_embeding(p) for p in prompt]
        else:
            rows = [self.get_embeding(prompt)]
        if any(r is None for r in rows):                  # same exception type as the reference (:602)
            raise AttributeError(f"unknown prompt {prompt!r}: 'NoneType' object has no attribute 'repeat'")
        # device copies of the (fixed) embedding vectors are kept: a per-call host -> device copy of pageable memory
        # makes the host wait for the stream to drain
        key = (str(device), tuple(prompt) if isinstance(prompt, (list, tuple)) else prompt)
        cache = self.__dict__.setdefault("_emb_dev", {})
        hit = cache.get(key)
        if hit is None or hit[0] is not self.text_embeddings:
            if len(cache) > 64:
                cache.clear()
            hit = cache[key] = (self.text_embeddings, torch.cat(rows).to(device))
        return hit[1]

    def _condition(self, P, PAN, MS, prompt):
        b = MS.shape[0]
        pe = self._prompt_rows(prompt, b, MS.device)
        lin = lambda i, x: ops.linear(x, self.embed2[i].weight.detach(), self.embed2[i].bias.detach(), act=True)
        pemb = lin(4, lin(2, lin(0, pe)))                 # act(embed2(prompt)) -- one row per distinct prompt row
        S = {"scale": P["scale_bank"].run(pemb)}
        c0 = self.channels[0]
        fuse_c, fuse = self._producer_fuse(P)
        spec = lambda **k: self._spec(P, S, **k) if fuse_c else None
        w0, b0 = self.conv1.conv20.weight.detach().reshape(-1), self.conv1.conv20.bias.detach()
        if fuse:    # the stem writes conv21's modulated input; conv21's epilogue writes SiLU(h) for down1_1.conv20 (flag: no shift)
            osc, oss = P["scale_bank"].slot(S["scale"], "conv1.dense2")
            p16 = "conv1.conv21" in P["bf16"]
            a0 = ops.stem(w0, b0, c0, pan=PAN.contiguous(), ms=MS.contiguous(), out_scale=osc, out_scale_stride=oss,
                          pack_bf16=p16)
            h, hp = self._conv(P, "conv1.conv21", [a0], use_bias=False, emit=spec(),
                               x_bf16_shape=(MS.shape[1], MS.shape[2], MS.shape[3]) if p16 else None)
        else:
            a0 = ops.stem(w0, b0, c0, pan=PAN.contiguous(), ms=MS.contiguous())
            out = self._conv(P, "conv1.conv21", [a0], use_bias=False, **self._scale(P, S, "conv1.dense2"),
                             **({"emit": spec()} if fuse_c else {}))
            h, hp = out if fuse_c else (out, None)
        cond = {"h0": h, "pan": PAN, "ms": MS, "prompt": prompt, "scale": S["scale"]}
        for lvl, (dn, upn) in enumerate((("down1_1", "up3"), ("down2_1", "up2"), ("down3_1", "up1")), start=1):
            h, ha, hq = self._resblock(P, S, dn + ".conv20", [h], flag=True, pre=hp, emit=spec(), want_ll=True)
            # (the last level's output only feeds the up path's three-segment conv20: nothing to emit)
            h, hp, skip = self._down(P, S, dn + ".down", h, flag=True, want_high=True, pre=ha,
                                     emit=spec() if lvl < 3 else None, fuse=fuse, xq=hq)
            cond[f"h{lvl}"] = h
            # convH_0(cat(skipH)/2)*2 == grouped conv of the three bands + 2*bias (exact: powers of two)
            cond[f"bands{lvl}"] = self._conv(P, upn + ".up1.convH_0.0", list(skip), bias_scale=2.0)
        return cond

    @staticmethod
    def _tensor_key(t):
        return (t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()))

    def begin_condition_cache(self, PAN, MS, prompt):
        """Compute the x_t/t-independent half of the network once; later ``forward`` calls given the same PAN / MS
        storage (same pointer, shape and version counter: an in-place update invalidates it) and prompt reuse it until
        ``end_condition_cache``."""
        with torch.no_grad():
            self._cond = self._condition(self._prepare(), PAN, MS, prompt)
        self._cond["key"] = (self._tensor_key(PAN), self._tensor_key(MS), prompt)

    def end_condition_cache(self):
        self._cond = None

    def invalidate_prepared(self):
        """Drop the packed weights / projection banks / condition cache.  ``_prepare`` notices parameter updates that
        bump ``Parameter._version`` (optimizers, ``load_state_dict``); call this after writing parameters through
        ``.data`` or a raw kernel (the EMA update does)."""
        self._prep = None
        self._cond = None
        self.__dict__.pop("_train_pack", None)       # the training path's packed weights (ops.PackedWeights)
        self.__dict__.pop("_train_pack_wino", None)

    _DERIVED = ("_train_pack", "_train_pack_wino", "_emb_dev", "_freqs_dev", "_side_stream")

    def __getstate__(self):
        """copy.deepcopy / pickle: everything derived from the parameters' ADDRESSES stays behind (packed weights and their
        device tables of raw pointers, projection banks, the condition cache, device copies of constants) -- the copy
        rebuilds them from its own tensors on first use.  ADVICE r3: a deep copy of a network that had already trained kept
        a table pointing at the source network's weights."""
        state = {k: v for k, v in self.__dict__.items() if k not in self._DERIVED}
        state["_prep"] = state["_cond"] = None
        return state

    # ---- forward ------------------------------------------------------------------------------------
    def forward(self, x_t, t_input, PAN=None, MS=None, prompt=None):
        """Reference ``WavBEST.forward`` (:600-636).  Under ``torch.no_grad()`` (sampling) this is the fused
        inference path; with grad mode on and anything that requires grad (a parameter or an input) it is the
        differentiable path (``forward_train``), as the reference's forward is."""
        if torch.is_grad_enabled() and (any(p.requires_grad for p in self.parameters()) or any(
                torch.is_tensor(t) and t.requires_grad for t in (x_t, PAN, MS))):
            return self.forward_train(x_t, t_input, PAN, MS, prompt)
        with torch.no_grad():
            return self._forward_infer(x_t, t_input, PAN, MS, prompt)

    def _forward_infer(self, x_t, t_input, PAN, MS, prompt):
        P = self._prepare()
        cond = self._cond
        if not (cond is not None and cond.get("key") == (self._tensor_key(PAN), self._tensor_key(MS), prompt)):
            cond = self._condition(P, PAN, MS, prompt)
        b = x_t.shape[0]
        t = t_input.reshape(-1).to(device=x_t.device, dtype=torch.float32)
        t = (t if t.numel() == b else t.expand(b)).contiguous()
        g = ops.gamma_embedding(t, P["freqs"], self.inter_dim)
        e = ops.linear(g, self.embed[0].weight.detach(), self.embed[0].bias.detach(), act=True)
        temb = ops.linear(e, self.embed[2].weight.detach(), self.embed[2].bias.detach(), act=True)
        S = {"shift": P["shift_bank"].run(temb), "scale": cond["scale"]}

        fuse_c, fuse = self._producer_fuse(P)
        spec = lambda **k: self._spec(P, S, **k) if fuse_c else None
        w0, b0 = self.conv2.conv20.weight.detach().reshape(-1), self.conv2.conv20.bias.detach()
        if fuse:
            osc, oss = P["scale_bank"].slot(S["scale"], "conv2.dense2")
            p16 = "conv2.conv21" in P["bf16"]
            a0 = ops.stem(w0, b0, self.channels[0], xin=x_t.contiguous(), out_scale=osc, out_scale_stride=oss, pack_bf16=p16)
            h, hp = self._conv(P, "conv2.conv21", [a0], use_bias=False, emit=spec(shift="down1.conv20.dense1"),
                               x_bf16_shape=tuple(x_t.shape[1:]) if p16 else None)
        else:
            a0 = ops.stem(w0, b0, self.channels[0], xin=x_t.contiguous())
            out = self._conv(P, "conv2.conv21", [a0], use_bias=False, **self._scale(P, S, "conv2.dense2"),
                             **({"emit": spec(shift="down1.conv20.dense1")} if fuse_c else {}))
            h, hp = out if fuse_c else (out, None)
        hs = [h]
        for dn, nxt in (("down1", "down2.conv20"), ("down2", "down3.conv20"), ("down3", "middle1")):
            # Conv_0 + LL with Winograd on top: the ResBlock hands its second output over in space-to-depth form
            sp = spec()
            s2d = sp is not None and self._ll_s2d(P, dn, h)
            h, ha, hq = self._resblock(P, S, dn + ".conv20", [h], flag=False, pre=hp, emit=dict(sp, s2d=True) if s2d else sp,
                                       want_ll=True)
            h, hp, _ = self._down(P, S, dn + ".down", h, flag=False, want_high=False, pre=ha,
                                  emit=spec(shift=nxt + ".dense1"), fuse=fuse, pre_s2d=s2d, xq=hq)
            hs.append(h)
        h, _ = self._resblock(P, S, "middle1", [hs[3]], flag=False, pre=hp)
        for lvl, upn in ((3, "up1"), (2, "up2"), (1, "up3")):
            h, ha = self._resblock(P, S, upn + ".conv20", [h, cond[f"h{lvl}"], hs[lvl]], flag=False, emit=spec())
            h = self._up(P, S, upn + ".up1", h, cond[f"bands{lvl}"], pre=ha, fuse=fuse)
        h, hp = self._resblock(P, S, "final.conv20", [h, cond["h0"], hs[0]], flag=False, emit=spec(shift="final.conv21.dense1"))
        for k in (1, 2, 3):
            h, hp = self._resblock(P, S, f"final.conv2{k}", [h], flag=False, pre=hp,
                                   emit=spec(shift=f"final.conv2{k + 1}.dense1") if k < 3 else None)
        sc = self._scale(P, S, "final.dense2")
        return ops.head(h, self.final.conv24.weight.detach().reshape(-1), sc["in_scale"], sc["scale_stride"])

    # ---- training path (finetune): same graph through the autograd-wrapped HIP ops ----------------------------
    def forward_train(self, x_t, t_input, PAN=None, MS=None, prompt=None):
        """Differentiable forward (dropout active when ``self.training``); used by
        ``GeneralDiffusion.p_losses_dynamic``.  Same graph as ``forward``, nothing cached."""
        from . import autograd as A
        dev = x_t.device
        if dev.type != "cuda":
            raise RuntimeError("tmdiff_amd.WavBEST runs on the HIP kernels only: move the module to a GPU (.cuda())")
        b = x_t.shape[0]
        # every convolution weight, forward and data-gradient packing, in one launch per step (re-done when a weight's
        # version changed, i.e. after the optimizer step)
        pk = self.__dict__.get("_train_pack")
        if pk is None or pk.convs[0][0].device != dev:
            pk = self.__dict__["_train_pack"] = ops.PackedWeights(
                [(m.weight, m.groups) for m in self.modules()
                 if isinstance(m, nn.Conv3d) and m.in_channels > 1 and m.out_channels > 1])
        ops.PACKED = pk.refresh()
        # ... and the Winograd (conv3d_wf) forms of the 3x3x3 weights, forward and data gradient, in one more
        wk = self.__dict__.get("_train_pack_wino")
        if wk is None:     # (owner = this network's weights: a foreign weight is packed on the spot, never registered)
            wk = self.__dict__["_train_pack_wino"] = ops.WinoPackedWeights(owner=[w for w, _ in pk.convs])
        ops.WINO_PACKED = wk.refresh() if ops.config.wino_multipack else None
        lin = lambda seq, i, x, act: A.linear(x, seq[i].weight, seq[i].bias, act=act)
        pe = self._prompt_rows(prompt, b, dev)
        pemb = lin(self.embed2, 4, lin(self.embed2, 2, lin(self.embed2, 0, pe, True), True), True)
        t = t_input.reshape(-1).to(device=dev, dtype=torch.float32)
        t = (t if t.numel() == b else t.expand(b)).contiguous()
        fr = self.__dict__.get("_freqs_dev")
        if fr is None or fr.device != dev:
            fr = self.__dict__["_freqs_dev"] = self._freqs_cpu.to(dev)
        g = ops.gamma_embedding(t, fr, self.inter_dim)
        temb = lin(self.embed, 2, lin(self.embed, 0, g, True), True)

        shift_layers, scale_layers = [], []
        for name, m in self.named_modules():
            if isinstance(m, ResBlockModulateBEST):
                scale_layers.append((name + ".dense2", m.dense2.dense))
                if not m.flag:
                    shift_layers.append((name + ".dense1", m.dense1.dense))
            elif isinstance(m, WaveletUPorDown):
                scale_layers.append((name + ".dense1", m.dense1.dense))
                if not m.flag:
                    shift_layers.append((name + ".Dense_0", m.Dense_0))
            elif isinstance(m, (AdaptionModulateBEST, FinalBlockModulateBEST)):
                scale_layers.append((name + ".dense2", m.dense2.dense))

        def bank(x, layers):     # one launch for all projections of x; autograd splits the gradient back
            out = A.linear(x, torch.cat([l.weight for _, l in layers]), torch.cat([l.bias for _, l in layers]))
            parts = A.split_cols(out, [l.out_features for _, l in layers])
            return {n_: p for (n_, _), p in zip(layers, parts)}

        # (the prompt embedding is one row for the whole batch: expanded HERE, so that every scale block is a [B, C] table and the
        #  gradient's sum over the batch happens once, in this expand's backward, not once per convolution)
        shifts, scales = bank(temb, shift_layers), bank(pemb.expand(temb.shape[0], -1) if pemb.shape[0] == 1 else pemb, scale_layers)
        sh = lambda name: shifts.get(name)          # flag=True blocks have no shift entry (and ignore it)

        def resblock(name, segs):
            return self.get_submodule(name).run(segs, sh(name + ".dense1"), scales[name + ".dense2"])

        def down(name, x, want_high):
            return self.get_submodule(name).run(x, sh(name + ".Dense_0"), scales[name + ".dense1"], want_high=want_high)

        # The condition branch (conv1, down*_1: a third of the forward) and the main branch's down path are independent until
        # the up path.  At the finetune batch most of their launches fill half of the CU slots or exactly one round, so set-up,
        # epilogue and tail of every launch are exposed; on two streams the workgroups of one branch fill what the other leaves
        # idle (ops.config.train_two_streams; at B = 32 every launch fills the chip and this gains nothing -- DESIGN.md 3).
        # Autograd runs each backward node on its forward's stream, so the two backward passes overlap as well.
        cur = torch.cuda.current_stream(dev)
        side = None
        if ops.config.train_two_streams:
            side = self.__dict__.get("_side_stream")
            if side is None or side.device != dev:
                side = self.__dict__["_side_stream"] = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
        cond, skips = {}, {}
        with torch.cuda.stream(side if side is not None else cur):
            cond[0] = self.conv1.run(scales["conv1.dense2"], pan=PAN.contiguous(), ms=MS.contiguous())
            h = cond[0]
            for lvl, dn in enumerate(("down1_1", "down2_1", "down3_1"), start=1):
                h = resblock(dn + ".conv20", [h])
                h, skips[lvl] = down(dn + ".down", h, True)
                cond[lvl] = h
        hs = [self.conv2.run(scales["conv2.dense2"], xin=x_t.contiguous())]
        h = hs[0]
        for dn in ("down1", "down2", "down3"):
            h = resblock(dn + ".conv20", [h])
            h, _ = down(dn + ".down", h, False)
            hs.append(h)
        h = resblock("middle1", [hs[3]])
        if side is not None:          # join: the up path reads the condition branch's tensors on the current stream
            cur.wait_stream(side)
            for t in list(cond.values()) + [b_ for bands in skips.values() for b_ in bands if b_ is not None]:
                t.record_stream(cur)
        for lvl, upn in ((3, "up1"), (2, "up2"), (1, "up3")):
            h = resblock(upn + ".conv20", [h, cond[lvl], hs[lvl]])
            h = self.get_submodule(upn + ".up1").run(h, sh(upn + ".up1.Dense_0"), scales[upn + ".up1.dense1"],
                                                     skipH=skips[lvl])
        h = resblock("final.conv20", [h, cond[0], hs[0]])
        for k in (1, 2, 3):
            h = resblock(f"final.conv2{k}", [h])
        return A.head(h, self.final.conv24.weight, scales["final.dense2"])
