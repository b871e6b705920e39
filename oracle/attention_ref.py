"""Oracle (test infrastructure): the attention operators of core/Attention.py, CPU fp32.

    GEGLU :69-76   FeedForward :79-96   SpatialSelfAttention :112-162
    CrossAttention :165-214   BasicTransformerBlock :266-296   SpatialTransformer :299-362

These modules are imported by nothing in the reference; they are in
scope as standalone operators because the north star names them.  Parameter names
follow the reference so ``fill_weights_`` produces identical weights on both sides.
Only the vanilla softmax path is restated (xformers is not installed; the reference
falls back to ``CrossAttention`` in that case, :31-35, :267-274).
"""
import torch
import torch.nn.functional as F
from torch import nn


def group_norm32(c):
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class GEGLU(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.proj = nn.Linear(din, dout * 2)

    def forward(self, x):
        a, gate = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, glu=False, dropout=0.0):
        super().__init__()
        inner = int(dim * mult)
        first = GEGLU(dim, inner) if glu else nn.Sequential(nn.Linear(dim, inner), nn.GELU())
        self.net = nn.Sequential(first, nn.Dropout(dropout), nn.Linear(inner, dim_out or dim))

    def forward(self, x):
        return self.net(x)


class SpatialSelfAttention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.in_channels = c
        self.norm = group_norm32(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward(self, x):
        b, c, h, w = x.shape
        y = self.norm(x)
        q = self.q(y).reshape(b, c, h * w).transpose(1, 2)          # [b, hw, c]
        k = self.k(y).reshape(b, c, h * w)                          # [b, c, hw]
        v = self.v(y).reshape(b, c, h * w)
        p = torch.softmax(torch.bmm(q, k) * (int(c) ** -0.5), dim=2)  # [b, i, j]
        o = torch.bmm(v, p.transpose(1, 2)).reshape(b, c, h, w)      # o[c,i] = sum_j v[c,j] p[i,j]
        return x + self.proj_out(o)


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

    def forward(self, x, context=None, mask=None):
        h = self.heads
        ctx = x if context is None else context
        split = lambda t: t.reshape(t.shape[0], t.shape[1], h, -1).permute(0, 2, 1, 3)   # [b, h, n, d]
        q, k, v = split(self.to_q(x)), split(self.to_k(ctx)), split(self.to_v(ctx))
        sim = torch.matmul(q.float(), k.float().transpose(-1, -2)) * self.scale
        if mask is not None:
            m = mask.reshape(mask.shape[0], 1, 1, -1)
            sim = sim.masked_fill(~m, -torch.finfo(sim.dtype).max)
        out = torch.matmul(sim.softmax(dim=-1), v)                                        # [b, h, n, d]
        out = out.permute(0, 2, 1, 3).reshape(x.shape[0], x.shape[1], -1)
        return self.to_out(out)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, n_heads, d_head, dropout=0.0, context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        self.disable_self_attn = disable_self_attn
        self.attn1 = CrossAttention(dim, context_dim if disable_self_attn else None, n_heads, d_head, dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=gated_ff)
        self.attn2 = CrossAttention(dim, context_dim, n_heads, d_head, dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def forward(self, x, context=None):
        x = self.attn1(self.norm1(x), context=context if self.disable_self_attn else None) + x
        x = self.attn2(self.norm2(x), context=context) + x
        return self.ff(self.norm3(x)) + x


class SpatialTransformer(nn.Module):
    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0.0, context_dim=None,
                 disable_self_attn=False, use_linear=False, use_checkpoint=True):
        super().__init__()
        if context_dim is not None and not isinstance(context_dim, list):
            context_dim = [context_dim]
        inner = n_heads * d_head
        self.in_channels, self.use_linear = in_channels, use_linear
        self.norm = group_norm32(in_channels)
        self.proj_in = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList(
            BasicTransformerBlock(inner, n_heads, d_head, dropout, context_dim[d], disable_self_attn=disable_self_attn)
            for d in range(depth))
        self.proj_out = nn.Linear(in_channels, inner) if use_linear else nn.Conv2d(inner, in_channels, 1)
        for p in self.proj_out.parameters():          # zero_module (:99-105)
            p.detach().zero_()

    def forward(self, x, context=None):
        ctx = context if isinstance(context, list) else [context]
        b, c, h, w = x.shape
        y = self.norm(x)
        if not self.use_linear:
            y = self.proj_in(y)
        y = y.reshape(b, y.shape[1], h * w).transpose(1, 2).contiguous()
        if self.use_linear:
            y = self.proj_in(y)
        for i, blk in enumerate(self.transformer_blocks):
            y = blk(y, context=ctx[i])
        if self.use_linear:
            y = self.proj_out(y)
        y = y.transpose(1, 2).reshape(b, -1, h, w).contiguous()
        if not self.use_linear:
            y = self.proj_out(y)
        return y + x


# ---- DDPM++ attention block of GeneralModel/Hyper_unet_general.py:471-515 (never instantiated by WavBEST) ----------
class NIN(nn.Module):
    """1x1 "network in network": y[b, :, h, w] = x[b, :, h, w] @ W + b (ref :471-480)."""

    def __init__(self, in_dim, num_units):
        super().__init__()
        self.W = nn.Parameter(torch.zeros(in_dim, num_units))
        self.b = nn.Parameter(torch.zeros(num_units))

    def forward(self, x):
        return torch.einsum("bchw,cu->buhw", x, self.W) + self.b[None, :, None, None]


class AttnBlockpp(nn.Module):
    """ref :483-515.  `channels` is the folded count C*N: the block folds [B,C,N,H,W] -> [B,C*N,H,W], normalises with
    min(channels/4, 32) groups (eps 1e-6), attends over the H*W positions with scale C^-1/2 (the UNFOLDED C), and
    returns (x + h)/sqrt(2) when skip_rescale."""

    def __init__(self, channels, skip_rescale=True):
        super().__init__()
        self.GroupNorm_0 = nn.GroupNorm(min(channels // 4, 32), channels, eps=1e-6)
        self.NIN_0, self.NIN_1, self.NIN_2, self.NIN_3 = (NIN(channels, channels) for _ in range(4))
        self.skip_rescale = skip_rescale

    def forward(self, x):
        b, c, n, h, w = x.shape
        xf = x.reshape(b, c * n, h, w)
        t = self.GroupNorm_0(xf)
        q, k, v = self.NIN_0(t), self.NIN_1(t), self.NIN_2(t)
        att = torch.einsum("bcp,bcq->bpq", q.flatten(2), k.flatten(2)) * (int(c) ** -0.5)
        att = att.softmax(-1)
        o = torch.einsum("bpq,bcq->bcp", att, v.flatten(2)).reshape(b, c * n, h, w)
        o = self.NIN_3(o)
        y = (xf + o).reshape(b, c, n, h, w)
        return y / (2.0 ** 0.5) if self.skip_rescale else y
