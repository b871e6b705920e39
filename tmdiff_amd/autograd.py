"""torch.autograd.Function wrappers of the fused HIP ops: the finetune (training) path of WavBEST.

Forward and backward both run in libtmdiff_hip.so; autograd only wires the graph ("each HIP op is
wrapped in a torch.autograd.Function with explicit backward").  Reference semantics being differentiated:
GeneralModel/Hyper_unet_general.py:51-77 (modulated conv), :237-249 (ResBlock), :369-414 (wavelet block),
DWT_IDWT/DWT_IDWT_Functions.py:60-69, :104-112 (hand-written DWT/IDWT backward = the transposed transform).
"""
import ctypes as C

import torch

from . import ops
from ._lib import check, lib


# (switches: ops.config -- winograd, wgrad_bias (43.4 vs 43.2 ms per finetune step with the bias gradient inside the direct
#  weight-gradient kernel: the saved pass is paid back inside the MFMA stream, so the separate pass stays the default),
#  train_ll_wino, wgrad_wino_bias)


class DropSpec(tuple):
    """(seed, p): dropout of a convolution's prologue output evaluated inside the kernels from a counter-based hash of
    (seed, element index) -- forward, weight gradient and prologue backward regenerate the same mask; no mask tensor."""
    __slots__ = ()

    def __new__(cls, seed, p):
        return super().__new__(cls, (int(seed), float(p)))


def _wf_weights(w, groups, mode):
    """conv3d_wf's packing of w (mode 2: forward, 3: data gradient): from the network's one-launch multi-tensor packing
    (WavBEST.forward_train refreshes it once per step; a pair it has not seen yet is packed on the spot and joins it)."""
    if ops.WINO_PACKED is not None:
        return ops.WINO_PACKED.get(w, groups, mode)
    return ops.pack_conv_weight_wino(w, groups, mode=mode, planes=6)


def _rows(t, what):
    """make_conv_desc arguments for a per-(sample, channel) prologue table [B, C] that may be a column block of a projection
    bank (rows `stride(0)` floats apart, as _SplitCols hands them out): pointer + row stride, no contiguous copy."""
    if t is None:
        return {}
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
        raise ValueError(f"{what}: need a float32 [B, C] tensor on the GPU with unit column stride")
    return {"in_" + what: t.data_ptr(), what + "_stride": t.stride(0)}


def _conv_forward(meta, weight, bias, shift, scale, residual, mask, segs, need_w):
    """y = (conv3d(act(cat(segs) + shift) * scale * mask, w) + bias_scale*bias + residual) * out_scale, and what its backward needs:
    (y, state) with state = dict(meta, drop, w, shift, scale, mask, xp, segs, wp_dgrad, has_bias, has_res)."""
    act, groups, bias_scale, out_scale = meta
    segs = [s.contiguous() for s in segs]
    w = weight.contiguous()
    cout, ksize = w.shape[0], w.shape[2]
    pre = ops.PACKED.lookup(w) if ops.PACKED is not None else None     # packed once per step for the whole network
    wp = pre[0] if pre is not None else ops.pack_conv_weight(w, groups=groups, mode=0)
    # mask: None, a tensor (caller-supplied dropout mask, parity runs) or a DropSpec (seed, p): in-kernel dropout
    drop = mask if isinstance(mask, DropSpec) else None
    mask = None if drop is not None else mask
    # a 3x3x3 convolution with dropout runs as prologue pass + staged kernel anyway: let the pass write x' into a
    # tensor of its own and keep it -- the weight gradient then needs no prologue pass of its own
    xp = None
    if ksize == 3 and (drop is not None or mask is not None) and need_w:
        b, _, n, h, wd = segs[0].shape
        cin = sum(s.shape[1] for s in segs)
        if cin % groups == 0 and (cin // groups) % 4 == 0 and (cout // groups) % 32 == 0:   # shapes the staged kernel takes
            xp = torch.empty(b, cin, n, h, wd, device=segs[0].device, dtype=torch.float32)
    kw = dict(bias=bias, bias_scale=bias_scale, in_act=act, in_mask=mask, drop=drop, residual=residual, out_scale=out_scale,
              xp_out=xp, **_rows(shift, "shift"), **_rows(scale, "scale"))
    cin = sum(s.shape[1] for s in segs)
    # 3x3x3 convolutions (in-kernel dropout included): the family tmdiff_amd.routing picks -- Winograd along the bands where
    # its grid fills the chip (conv3d_wf's prologue pass writes x' where the weight gradient will read it), else direct
    if ksize == 3:
        wino_ok = ops.config.winograd and ops.wino_conv_supported(cout, cin, ksize, groups)
        weights = ops.ConvWeights(lambda: wp, (lambda: _wf_weights(w, groups, 2)) if wino_ok else None,
                                  (lambda planes: ops.pack_conv_weight_wino(w, groups, planes=planes)) if wino_ok else None)
        y = ops.conv3d_auto(segs, weights, cout, groups=groups, **kw)
    else:
        y = ops.conv3d(segs, wp, cout, ksize, groups=groups, **kw)
    state = dict(meta=meta, drop=drop, w=w, shift=shift, scale=scale, mask=mask, xp=xp, segs=segs,
                 wp_dgrad=pre[1] if pre is not None else None, has_bias=bias is not None, has_res=residual is not None)
    return y, state


def _conv_backward(st, gy, need_w, need_b, need_shift, need_scale, need_res, need_segs, outs=None, accumulate=None, adds=None):
    """Gradients of _conv_forward: (d_w, d_bias, d_shift, d_scale, d_res, d_segs).  outs / accumulate: tensors the input
    gradients are ADDED to instead of fresh ones (the other consumer's gradient of the same segments: no sum pass afterwards);
    adds: tensors that are only READ and added into fresh outputs (a gradient that may be shared: the identity residual's)."""
    act, groups, bias_scale, out_scale = st["meta"]
    w, shift, scale, mask, xp, segs = st["w"], st["shift"], st["scale"], st["mask"], st["xp"], st["segs"]
    has_bias, has_shift, has_scale = st["has_bias"], shift is not None, scale is not None
    g = gy.contiguous()
    if out_scale != 1.0:
        g = ops.axpby([g], [out_scale])
    cout, cin_g, ksize = w.shape[0], w.shape[1], w.shape[2]
    cin = cin_g * groups
    d_bias = None
    d_res = g if (st["has_res"] and need_res) else None
    # descriptor of the forward prologue (what x' was): used by wgrad and by the prologue backward
    desc = ops.make_conv_desc(segs, 0, cout, ksize, g, groups=groups, in_act=act, in_mask=mask, drop=st["drop"],
                              **_rows(shift, "shift"), **_rows(scale, "scale"))
    if need_w:   # x' kept by the forward: a plain single-tensor input, no prologue pass inside the weight gradient
        desc_w = ops.make_conv_desc([xp], 0, cout, ksize, g, groups=groups) if xp is not None else desc
        if has_bias and need_b and (ops.config.wgrad_bias or (ops.config.wgrad_wino_bias and ops.wgrad_wino_takes(desc_w))):   # the bias gradient rides along in the weight-gradient kernel
            desc_w.bias_scale = bias_scale
            d_w, d_bias = ops.conv3d_wgrad(desc_w, g, tuple(w.shape), want_bias=True)
        else:
            d_w = ops.conv3d_wgrad(desc_w, g, tuple(w.shape))
    else:
        d_w = None
    if has_bias and need_b and d_bias is None:
        d_bias = ops.channel_sum(g, bias_scale)
    need_x = any(need_segs) or (has_shift and need_shift) or (has_scale and need_scale)
    d_shift = d_scale = None
    d_segs = [None] * len(segs)
    if need_x:
        wp_t = st["wp_dgrad"] if st["wp_dgrad"] is not None else ops.pack_conv_weight(w, groups=groups, mode=1)
        if ksize == 3:     # the data gradient is a 3x3x3 convolution too (a plain input: no pass at all)
            wino_ok = ops.config.winograd and ops.wino_conv_supported(cin, cout, ksize, groups)
            weights = ops.ConvWeights(lambda: wp_t, (lambda: _wf_weights(w, groups, 3)) if wino_ok else None,
                                      (lambda planes: ops.pack_conv_weight_wino(w, groups, mode=1, planes=planes)) if wino_ok else None)
            gp = ops.conv3d_auto([g], weights, cin, groups=groups)                # dL/dx'
        else:
            gp = ops.conv3d([g], wp_t, cin, ksize, groups=groups)
        plain = not (act or has_shift or has_scale or mask is not None or st["drop"] is not None)
        if plain and len(segs) == 1 and outs is None:      # x' IS x: dL/dx' is the input gradient (no pass to copy it)
            return d_w, d_bias, None, None, d_res, [gp]
        if outs is None:
            outs = [torch.empty_like(s_) if need_segs[i] else None for i, s_ in enumerate(segs)]
            accumulate = [False] * len(segs)
        d_shift, d_scale = ops.conv3d_prologue_bwd(desc, gp, outs, accumulate, has_shift and need_shift, has_scale and need_scale,
                                                   add_segs=adds)
        d_segs = outs
    return d_w, d_bias, d_shift, d_scale, d_res, d_segs


_STATE_TENSORS = ("w", "shift", "scale", "mask", "xp")


def _stash(ctx, prefix, st, tensors):
    """Move a _conv_forward state's tensors into the list save_for_backward will get; the rest stays on ctx."""
    idx = {}
    for k in _STATE_TENSORS:
        if st[k] is not None:
            idx[k] = len(tensors)
            tensors.append(st[k])
    idx["segs"] = [len(tensors) + i for i in range(len(st["segs"]))]
    tensors.extend(st["segs"])
    setattr(ctx, prefix, (idx, {k: v for k, v in st.items() if k not in _STATE_TENSORS and k != "segs"}))


def _unstash(ctx, prefix, saved):
    idx, rest = getattr(ctx, prefix)
    st = dict(rest)
    for k in _STATE_TENSORS:
        st[k] = saved[idx[k]] if k in idx else None
    st["segs"] = [saved[i] for i in idx["segs"]]
    return st


class _FusedConv3d(torch.autograd.Function):
    """y = (conv3d(act(cat(segs) + shift) * scale * mask, w) + bias_scale*bias + residual) * out_scale"""

    @staticmethod
    def forward(ctx, meta, weight, bias, shift, scale, residual, mask, *segs):
        y, st = _conv_forward(meta, weight, bias, shift, scale, residual, mask, segs, ctx.needs_input_grad[1])
        tensors = []
        _stash(ctx, "st", st, tensors)
        ctx.save_for_backward(*tensors)
        return y

    @staticmethod
    def backward(ctx, gy):
        st = _unstash(ctx, "st", ctx.saved_tensors)
        need = ctx.needs_input_grad  # (meta, weight, bias, shift, scale, residual, mask, *segs)
        d_w, d_bias, d_shift, d_scale, d_res, d_segs = _conv_backward(st, gy, need[1], need[2], need[3], need[4], need[5], need[7:])
        return (None, d_w, d_bias, d_shift, d_scale, d_res, None, *d_segs)


class _ResBlockRC(torch.autograd.Function):
    """ResBlockModulateBEST with a res_conv (reference Hyper_unet_general.py:237-249) as ONE node:
    y = conv21(act(t1) * scale * mask21) + res_conv(x), t1 = conv20(act(x + shift) * mask20) + b20.  Same launches as the three
    _FusedConv3d nodes it replaces, except in the backward: every input segment has two consumers (conv20 and res_conv), and
    autograd would sum their two gradients with a launch per segment -- here conv20's prologue backward ADDS into the tensors
    res_conv's data gradient has just written (fresh tensors that nothing else holds)."""

    @staticmethod
    def forward(ctx, metas, w20, b20, w21, wrc, brc, shift, scale, mask20, mask21, *segs):
        m20, m21, mrc = metas
        need = ctx.needs_input_grad
        t1, s20 = _conv_forward(m20, w20, b20, shift, None, None, mask20, segs, need[1])
        res, src = _conv_forward(mrc, wrc, brc, None, None, None, None, segs, need[4])
        y, s21 = _conv_forward(m21, w21, None, None, scale, res, mask21, [t1], need[3])
        tensors = []
        for prefix, st in (("s20", s20), ("src", src), ("s21", s21)):
            _stash(ctx, prefix, st, tensors)
        ctx.save_for_backward(*tensors)
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        s20, src, s21 = (_unstash(ctx, k, saved) for k in ("s20", "src", "s21"))
        need = ctx.needs_input_grad  # (metas, w20, b20, w21, wrc, brc, shift, scale, mask20, mask21, *segs)
        need_segs = list(need[10:])
        any_x = any(need_segs) or need[6]
        d_w21, _, _, d_scale, d_res, (d_t1,) = _conv_backward(s21, gy, need[3], False, False, need[7], True, [True])
        d_wrc, d_brc, _, _, _, d_segs = _conv_backward(src, d_res, need[4], need[5], False, False, False, need_segs)
        if all(need_segs):      # conv20's input gradients are added to res_conv's (no sum pass)
            d_w20, d_b20, d_shift, _, _, d_segs = _conv_backward(s20, d_t1, need[1], need[2], need[6], False, False, need_segs,
                                                                 outs=d_segs, accumulate=[True] * len(d_segs))
        else:
            d_w20, d_b20, d_shift, _, _, d2 = _conv_backward(s20, d_t1, need[1], need[2], need[6], False, False, need_segs)
            d_segs = [a if b_ is None else (b_ if a is None else a + b_) for a, b_ in zip(d_segs, d2)] if any_x else d_segs
        return (None, d_w20, d_b20, d_w21, d_wrc, d_brc, d_shift, d_scale, None, None, *d_segs)


class _ResBlockId(torch.autograd.Function):
    """ResBlockModulateBEST without a res_conv (channel_in == channel_out; reference Hyper_unet_general.py:248: the block's input is
    the residual) as ONE node: y = conv21(act(t1) * scale * mask21) + x, t1 = conv20(act(x + shift) * mask20) + b20.  In the backward
    x has two gradients -- the incoming one itself (the residual) and conv20's -- and conv20's prologue backward writes their SUM
    (three-operand form: the incoming gradient is only read, it may be shared)."""

    @staticmethod
    def forward(ctx, metas, w20, b20, w21, shift, scale, mask20, mask21, x):
        m20, m21 = metas
        need = ctx.needs_input_grad
        t1, s20 = _conv_forward(m20, w20, b20, shift, None, None, mask20, [x], need[1])
        y, s21 = _conv_forward(m21, w21, None, None, scale, s20["segs"][0], mask21, [t1], need[3])
        tensors = []
        for prefix, st in (("s20", s20), ("s21", s21)):
            _stash(ctx, prefix, st, tensors)
        ctx.save_for_backward(*tensors)
        return y

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        s20, s21 = (_unstash(ctx, k, saved) for k in ("s20", "s21"))
        need = ctx.needs_input_grad  # (metas, w20, b20, w21, shift, scale, mask20, mask21, x)
        d_w21, _, _, d_scale, d_res, (d_t1,) = _conv_backward(s21, gy, need[3], False, False, need[5], True, [True])
        d_w20, d_b20, d_shift, _, _, (d_x,) = _conv_backward(s20, d_t1, need[1], need[2], need[4], False, False, [True],
                                                             adds=[d_res] if need[8] else None)
        return (None, d_w20, d_b20, d_w21, d_shift, d_scale, None, None, d_x if need[8] else None)


def resblock_id(x, w20, b20, w21, shift, scale, mask20, mask21):
    """The differentiable ResBlock without a res_conv as one autograd node (see _ResBlockId)."""
    b = x.shape[0]
    metas = ((True, 1, 1.0, 1.0), (True, 1, 1.0, 1.0))
    return _ResBlockId.apply(metas, w20, b20, w21, _fix_rows(shift, b), _fix_rows(scale, b), mask20, mask21, x)


def resblock_rc(segs, w20, b20, w21, wrc, brc, shift, scale, mask20, mask21):
    """The differentiable ResBlock with a res_conv as one autograd node (see _ResBlockRC); arguments as for three conv3d() calls."""
    b = segs[0].shape[0]
    metas = ((True, 1, 1.0, 1.0), (True, 1, 1.0, 1.0), (False, 1, 1.0, 1.0))
    return _ResBlockRC.apply(metas, w20, b20, w21, wrc, brc, _fix_rows(shift, b), _fix_rows(scale, b), mask20, mask21, *segs)


def _fix_rows(t, b):
    # (a column block of a projection bank stays the view it is: the kernels take a row stride; only a table that is not [B, C]
    #  with unit column stride -- one row broadcast over the batch -- is materialised)
    if t is None:
        return None
    if t.shape[0] != b:
        t = t.expand(b, t.shape[1])
    return t if (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]) else t.contiguous()


def conv3d(segs, weight, bias=None, bias_scale=1.0, shift=None, scale=None, act=False, mask=None, residual=None,
           groups=1, out_scale=1.0):
    """Differentiable fused convolution; shift / scale are dense [B, Cin] tensors."""
    b = segs[0].shape[0]
    fix = lambda t: _fix_rows(t, b)
    return _FusedConv3d.apply((bool(act), int(groups), float(bias_scale), float(out_scale)), weight, bias, fix(shift),
                              fix(scale), residual, mask, *segs)


class _ConvLL(torch.autograd.Function):
    """o = LL(conv3d(act(x), w) + bias) * ll_scale -- Conv_0 of a down block whose high bands are dropped, followed by the
    halved LL band (WaveletUPorDown, reference Hyper_unet_general.py:371-372, :389, :396).  Forward: the prologue pass (its
    output x' is kept for the weight gradient) and ONE strided convolution on composed weights (ops.conv3d_ll: 48 instead
    of 4 x 27 multiply-adds per output).  Backward: the adjoint of the LL band spreads the gradient back to full
    resolution, then the data / weight / bias / prologue gradients of the 3x3x3 convolution as in _FusedConv3d."""

    @staticmethod
    def forward(ctx, ll_scale, weight, bias, x):
        w, x = weight.contiguous(), x.contiguous()
        cout = w.shape[0]
        d = ops.Conv3dDesc()
        b, cin, n, h, wd = x.shape
        d.B, d.N, d.H, d.W, d.Cin, d.Cout, d.groups, d.ksize, d.nseg = b, n, h, wd, cin, cout, 1, 3, 1
        d.seg_c[0], d.seg_x[0], d.in_act = cin, x.data_ptr(), 1
        xp = ops.conv3d_prologue(d, tuple(x.shape))
        if ops.config.train_ll_wino and ops.wfll_route(b, cin, cout, n, h, wd):
            # with Winograd along the bands on top (conv3d_wf's composed-LL mode): x' once more in space-to-depth form (a copy:
            # the weight gradient keeps reading the plain x'; 28.37 -> 28.19 ms per finetune step)
            xs = xp.view(b, cin, n, h // 2, 2, wd // 2, 2).permute(0, 1, 4, 6, 2, 3, 5).reshape(b, 4 * cin, n, h // 2, wd // 2)
            y = ops.conv3d_wf_ll(xs, ops.pack_conv_weight_wfll(w, ll_scale), cout, ll_scale, bias=bias)
        else:
            y = ops.conv3d_ll(xp, ops.pack_conv_weight_ll(w, ll_scale), cout, ll_scale, bias=bias)
        ctx.ll_scale = ll_scale
        ctx.has_bias = bias is not None
        ctx.save_for_backward(w, x, xp)
        return y

    @staticmethod
    def backward(ctx, go):
        w, x, xp = ctx.saved_tensors
        need = ctx.needs_input_grad          # (ll_scale, weight, bias, x)
        go = go.contiguous()
        cout, cin = w.shape[0], w.shape[1]
        g = ops.haar_idwt2d([go], None, None, None, in_scale=ctx.ll_scale)[0]      # adjoint of the scaled LL band
        d_w = ops.conv3d_wgrad(ops.make_conv_desc([xp], 0, cout, 3, g), g, tuple(w.shape)) if need[1] else None
        d_b = ops.channel_sum(go, 2.0 * ctx.ll_scale) if (ctx.has_bias and need[2]) else None
        d_x = None
        if need[3]:
            pre = ops.PACKED.lookup(w) if ops.PACKED is not None else None
            wp_t = pre[1] if pre is not None else ops.pack_conv_weight(w, mode=1)
            # dL/dx': a 3x3x3 convolution of the up-sampled gradient -- Winograd along the bands (13.5 multiply-adds per
            # element; the transposed form of the composed strided convolution would take 12)
            gp = ops.conv3d_auto([g], ops.ConvWeights(lambda: wp_t, (lambda: _wf_weights(w, 1, 3)) if ops.config.winograd else None),
                                 cin)
            d_x = torch.empty_like(x)
            ops.conv3d_prologue_bwd(ops.make_conv_desc([x], 0, cout, 3, g, in_act=True), gp, [d_x], [False], False, False)
        return None, d_w, d_b, d_x


def conv3d_ll(x, weight, bias=None, ll_scale=0.5):
    """Differentiable LL(conv3d(SiLU(x), w) + bias) * ll_scale (see _ConvLL); shapes per ops.ll_conv_supported."""
    return _ConvLL.apply(float(ll_scale), weight, bias, x)


class _HaarDWT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, want_high, ll_scale):
        ctx.want_high, ctx.ll_scale = want_high, ll_scale
        outs = ops.haar_dwt2d(x.contiguous(), want_high=want_high, ll_scale=ll_scale)
        return tuple(o for o in outs if o is not None)

    @staticmethod
    def backward(ctx, *grads):
        g_ll = grads[0].contiguous()
        if ctx.want_high and any(g is not None for g in grads[1:]):
            hi = [g.contiguous() if g is not None else torch.zeros_like(g_ll) for g in grads[1:]]
            dx = ops.haar_idwt2d([g_ll], *hi, in_scale=ctx.ll_scale)[0]
        else:
            dx = ops.haar_idwt2d([g_ll], None, None, None, in_scale=ctx.ll_scale)[0]
        return dx, None, None


def haar_dwt2d(x, want_high=True, ll_scale=1.0):
    outs = _HaarDWT.apply(x, want_high, ll_scale)
    return tuple(outs) + (None,) * (4 - len(outs))


class _HaarIDWT2(torch.autograd.Function):
    """(h_up, x_up) = (IDWT(s*h, bands), IDWT(s*x, bands)); bands = stacked [B, 3C, N, h, w]"""

    @staticmethod
    def forward(ctx, h, x, bands, in_scale):
        ctx.in_scale = in_scale
        return tuple(ops.haar_idwt2d([h.contiguous(), x.contiguous()], None, None, None, in_scale=in_scale,
                                     stacked_bands=bands.contiguous()))

    @staticmethod
    def backward(ctx, g_h, g_x):
        ah = ops.haar_dwt2d(g_h.contiguous(), want_high=True, ll_scale=ctx.in_scale)
        ax = ops.haar_dwt2d(g_x.contiguous(), want_high=True, ll_scale=ctx.in_scale)
        d_bands = torch.cat([ops.add(ah[k], ax[k]) for k in (1, 2, 3)], dim=1)
        return ah[0], ax[0], d_bands, None


def haar_idwt2d_pair(h, x, bands, in_scale=1.0):
    return _HaarIDWT2.apply(h, x, bands, in_scale)


class _Stem(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight, bias, xin, pan, ms):
        c0 = weight.shape[0]
        w = weight.reshape(-1).contiguous()
        y = ops.stem(w, bias, c0, xin=xin, pan=pan, ms=ms)
        ctx.save_for_backward(w, bias, *(t for t in (xin, pan, ms) if t is not None))
        ctx.mode = xin is not None
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        w, bias, *ins = ctx.saved_tensors
        kw = {"xin": ins[0]} if ctx.mode else {"pan": ins[0], "ms": ins[1]}
        gy = gy.contiguous()
        need = ctx.needs_input_grad          # (weight, bias, xin, pan, ms)
        d_w = d_b = None
        if need[0] or need[1]:
            dwb = ops.stem_bwd(w, bias, gy, **kw).sum(0)       # [C0, 2], tiny
            d_w, d_b = dwb[:, 0].reshape(ctx.wshape), dwb[:, 1].contiguous()
        d_xin = d_pan = d_ms = None
        if ctx.mode and need[2]:
            d_xin, _ = ops.stem_bwd_input(w, bias, gy, **kw)
        elif not ctx.mode and (need[3] or need[4]):
            d_ms, d_pan = ops.stem_bwd_input(w, bias, gy, need_x=need[4], need_pan=need[3], **kw)
        return d_w, d_b, d_xin, d_pan, d_ms


def stem(weight, bias, xin=None, pan=None, ms=None):
    return _Stem.apply(weight, bias, xin, pan, ms)


class _Head(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, scale):
        w = weight.reshape(-1).contiguous()
        x = x.contiguous()
        scale = scale.expand(x.shape[0], scale.shape[1]).contiguous()
        ctx.save_for_backward(x, w, scale)
        ctx.wshape = weight.shape
        return ops.head(x, w, scale)

    @staticmethod
    def backward(ctx, gy):
        x, w, scale = ctx.saved_tensors
        dx, dws = ops.head_bwd(x, w, scale, gy.contiguous(), need_dx=ctx.needs_input_grad[0])
        d_w = (dws * scale).sum(0).reshape(ctx.wshape)     # [B, C] elementwise on tiny tensors
        d_scale = dws * w[None, :]
        return dx, d_w, d_scale


def head(x, weight, scale):
    return _Head.apply(x, weight, scale)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x, weight = x.contiguous(), weight.contiguous()
        ctx.act = act
        ctx.save_for_backward(x, weight, bias)
        return ops.linear(x, weight, bias, act=act)

    @staticmethod
    def backward(ctx, gy):
        x, w, bias = ctx.saved_tensors
        n = ctx.needs_input_grad
        dx, dw, db = ops.linear_bwd(x, w, bias, gy.contiguous(), act=ctx.act, need_dx=n[0], need_dw=n[1],
                                    need_db=n[2] and bias is not None)
        return dx, dw, db, None


def linear(x, weight, bias=None, act=False):
    return _Linear.apply(x, weight, bias, act)


class _SplitCols(torch.autograd.Function):
    """[B, sum(sizes)] -> the [B, size_k] column blocks (views).  Plain slicing would do, but its backward builds a zero-filled
    full-width tensor per slice and adds them up (three tiny kernels per Dense() projection, ~180 launches per training
    step for the two projection banks); here the backward is one concatenation."""

    @staticmethod
    def forward(ctx, out, sizes):
        ctx.sizes = sizes
        return torch.split(out, sizes, dim=1)       # views: their consumers take a row stride (autograd._rows)

    @staticmethod
    def backward(ctx, *grads):
        return torch.cat([g.contiguous() for g in grads], dim=1), None      # (materialised: unused blocks arrive as zeros)


def split_cols(out, sizes):
    return _SplitCols.apply(out, tuple(int(s) for s in sizes))
