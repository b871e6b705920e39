"""The attention operators of the reference's core/Attention.py on the HIP kernels (inference forward).

``core/Attention.py`` is imported by nothing in TMDiff and WavBEST contains no attention layer;
these classes exist because BASELINE's north star names them.  Module tree and parameter names follow the
reference (GEGLU :69-76, FeedForward :79-96, SpatialSelfAttention :112-162, CrossAttention :165-214,
BasicTransformerBlock :266-296, SpatialTransformer :299-362), so its state_dicts load; ``NIN`` / ``AttnBlockpp`` are
the DDPM++ attention block that GeneralModel/Hyper_unet_general.py defines (:471-515) and never instantiates.  Only the vanilla softmax
path exists (the reference falls back to it when xformers is absent, :31-35, :267-274).  nn.Linear / nn.Conv2d /
norm modules are parameter containers: the arithmetic is tmdiff_gemm_nt, tmdiff_conv3d_fwd (1x1), tmdiff_attn_fwd,
tmdiff_group_norm, tmdiff_layer_norm, tmdiff_geglu.
"""
import torch
from torch import nn

from . import ops


def _conv1x1(conv, x, residual=None):
    """nn.Conv2d(kernel 1) on [B, C, H, W] through the conv3d MFMA kernel (N = 1)."""
    b, c, h, w = x.shape
    wt = conv.weight.detach().reshape(conv.out_channels, c, 1, 1, 1).contiguous()
    wp = ops.pack_conv_weight(wt)
    res = None if residual is None else residual.reshape(b, conv.out_channels, 1, h, w)
    y = ops.conv3d([x.reshape(b, c, 1, h, w)], wp, conv.out_channels, 1,
                   bias=conv.bias.detach() if conv.bias is not None else None, residual=res)
    return y.reshape(b, conv.out_channels, h, w)


def _linear(lin, x, residual=None):
    return ops.gemm_nt(x.contiguous(), lin.weight.detach(), lin.bias.detach() if lin.bias is not None else None,
                       residual)


def Normalize(in_channels):
    return nn.GroupNorm(num_groups=32, num_channels=in_channels, eps=1e-6, affine=True)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    @torch.no_grad()
    def forward(self, x):
        return ops.geglu(_linear(self.proj, x))


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        inner = int(dim * mult)
        first = GEGLU(dim, inner) if glu else nn.Sequential(nn.Linear(dim, inner), nn.GELU())
        self.net = nn.Sequential(first, nn.Dropout(dropout), nn.Linear(inner, dim_out or dim))

    @torch.no_grad()
    def forward(self, x, residual=None):
        first = self.net[0]
        h = first(x) if isinstance(first, GEGLU) else ops.geglu(_linear(first[0], x), gelu_only=True)
        return _linear(self.net[2], h, residual)


class SpatialSelfAttention(nn.Module):
    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    @torch.no_grad()
    def forward(self, x):
        b, c, h, w = x.shape
        x = x.contiguous()
        y = ops.group_norm(x, self.norm.weight.detach(), self.norm.bias.detach(), 32, self.norm.eps)
        tok = lambda t: t.reshape(b, c, h * w).transpose(1, 2).contiguous()         # [B, HW, C] token-major
        q, k, v = tok(_conv1x1(self.q, y)), tok(_conv1x1(self.k, y)), tok(_conv1x1(self.v, y))
        o = ops.attention(q, k, v, float(int(c) ** -0.5), heads=1)                  # [B, HW, C]
        o = o.transpose(1, 2).reshape(b, c, h, w).contiguous()
        return _conv1x1(self.proj_out, o, residual=x)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = context_dim or query_dim
        self.scale, self.heads = dim_head ** -0.5, heads
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))

    @torch.no_grad()
    def forward(self, x, context=None, mask=None, residual=None):
        ctx = x if context is None else context
        q, k, v = _linear(self.to_q, x), _linear(self.to_k, ctx), _linear(self.to_v, ctx)
        if mask is not None:
            mask = mask.reshape(mask.shape[0], -1)
        o = ops.attention(q, k, v, float(self.scale), heads=self.heads, key_mask=mask)
        return _linear(self.to_out[0], o, residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        self.disable_self_attn = disable_self_attn
        self.attn1 = CrossAttention(dim, context_dim if disable_self_attn else None, n_heads, d_head, dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head, dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    @torch.no_grad()
    def forward(self, x, context=None):
        ln = lambda m, t: ops.layer_norm(t, m.weight.detach(), m.bias.detach(), m.eps)
        x = x.contiguous()
        x = self.attn1(ln(self.norm1, x), context=context if self.disable_self_attn else None, residual=x)
        x = self.attn2(ln(self.norm2, x), context=context, residual=x)
        return self.ff(ln(self.norm3, x), residual=x)


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None, disable_self_attn=False,
                 use_linear=False, use_checkpoint=True):
        super().__init__()
        if context_dim is not None and not isinstance(context_dim, list):
            context_dim = [context_dim]
        inner = n_heads * d_head
        self.in_channels, self.use_linear = in_channels, use_linear
        self.norm = Normalize(in_channels)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            BasicTransformerBlock(inner, n_heads, d_head, dropout, context_dim[d], disable_self_attn=disable_self_attn)
            for d in range(depth))
        self.proj_out = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(inner, in_channels, 1)
        for p in self.proj_out.parameters():        # zero_module (:99-105)
            p.detach().zero_()

    @torch.no_grad()
    def forward(self, x, context=None):
        ctx = context if isinstance(context, list) else [context]
        b, c, h, w = x.shape
        x = x.contiguous()
        y = ops.group_norm(x, self.norm.weight.detach(), self.norm.bias.detach(), 32, self.norm.eps)
        if not self.use_linear:
            y = _conv1x1(self.proj_in, y)
        y = y.reshape(b, y.shape[1], h * w).transpose(1, 2).contiguous()
        if self.use_linear:
            y = _linear(self.proj_in, y)
        for i, blk in enumerate(self.transformer_blocks):
            y = blk(y, context=ctx[i])
        if self.use_linear:
            y = _linear(self.proj_out, y)
        y = y.transpose(1, 2).reshape(b, -1, h, w).contiguous()
        if not self.use_linear:
            return _conv1x1(self.proj_out, y, residual=x)
        return ops.add(y, x)


class NIN(nn.Module):
    """y[b, :, h, w] = x[b, :, h, w] @ W + b (Hyper_unet_general.py:471-480); W is [in, out]."""

    def __init__(self, in_dim, num_units, init_scale=0.1):
        super().__init__()
        w = torch.empty(in_dim, num_units)
        # DDPM default_init (ref :417-454): fan-avg uniform variance scaling
        bound = (3.0 * max(init_scale, 1e-10) / ((in_dim + num_units) / 2.0)) ** 0.5
        self.W = nn.Parameter(w.uniform_(-bound, bound))
        self.b = nn.Parameter(torch.zeros(num_units))


class AttnBlockpp(nn.Module):
    """Hyper_unet_general.py:483-515 on the HIP kernels.  ``channels`` is the FOLDED channel count C*N (the block
    folds 'b c n h w -> b (c n) h w' before its GroupNorm); the softmax scale uses the unfolded C, as there."""

    def __init__(self, channels, skip_rescale=True, init_scale=0.0):
        super().__init__()
        self.GroupNorm_0 = nn.GroupNorm(num_groups=min(channels // 4, 32), num_channels=channels, eps=1e-6)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=init_scale)
        self.skip_rescale = skip_rescale

    @torch.no_grad()
    def forward(self, x):
        b, c, n, h, w = x.shape
        cf = c * n
        xf = x.contiguous().reshape(b, cf, h, w)
        gn = self.GroupNorm_0
        t = ops.group_norm(xf, gn.weight.detach(), gn.bias.detach(), gn.num_groups, gn.eps)
        tok = t.reshape(b, cf, h * w).transpose(1, 2).contiguous()                        # [B, HW, CF] token-major
        nin = lambda m, a, res=None: ops.gemm_nt(a, m.W.detach().t().contiguous(), m.b.detach(), res)
        q, k, v = nin(self.NIN_0, tok), nin(self.NIN_1, tok), nin(self.NIN_2, tok)
        o = ops.attention(q, k, v, float(int(c) ** -0.5), heads=1)                        # [B, HW, CF]
        x_tok = xf.reshape(b, cf, h * w).transpose(1, 2).contiguous()
        y = nin(self.NIN_3, o, x_tok)                                                     # x + NIN_3(h), token-major
        if self.skip_rescale:
            y = ops.axpby([y], [2.0 ** -0.5])
        return y.transpose(1, 2).reshape(b, c, n, h, w).contiguous()
