"""Oracle (test infrastructure): WavBEST UNet restated in plain PyTorch, CPU fp32.

Follows GeneralModel/Hyper_unet_general.py of the reference:
    modulated_conv3d       :51-77     gamma_embedding        :80-97
    Dense / Swish          :100-113   AdaptionModulateBEST   :158-173
    ResblockDownOne...     :176-196   ResblockUpOne...       :199-217
    ResBlockModulateBEST   :220-249   FinalBlockModulateBEST :252-273
    WaveletUPorDown        :334-414   WavBEST                :523-636
The module tree and parameter names are the reference's, so a reference
``state_dict`` (minus ``clip_text_model.*``) loads with ``strict=True``.

Differences that are deliberate and documented in DESIGN.md:
  * the frozen CLIP text encoder (core/clip.py, weights not available offline) is
    replaced by an injectable ``text_embeddings: dict[str, Tensor[1,768]]``;
  * no hard-coded ``.to("cuda")`` (:602) -- tensors stay where the inputs are;
  * Haar DWT/IDWT use the closed form in ``haar_ref.py`` instead of per-call dense
    matrices (identical taps and evaluation order).
Dropout(0.2) layers exist (reference :230, :349) and are active in ``train()``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .haar_ref import haar_dwt2d, haar_idwt2d

PROMPTS = ("QB", "WV3", "GF2", "WV2", "WV4")


def silu(x):
    return x * torch.sigmoid(x)


def modconv3d(x, w, s, padding):
    """Per-sample input-channel weight modulation, no bias, no demodulation (ref :51-77).

    x [B,I,N,H,W], w [O,I,k,k,k], s [B,I,1,1] -> [B,O,N,H,W]; evaluated like the
    reference as one grouped convolution over the batch.
    """
    b = x.shape[0]
    o, i = w.shape[:2]
    wm = w.unsqueeze(0) * s.unsqueeze(1).unsqueeze(5)          # [B,O,I,k,k,k]
    y = F.conv3d(x.reshape(1, b * i, *x.shape[2:]), wm.reshape(b * o, i, *w.shape[2:]),
                 None, 1, padding, 1, groups=b)
    return y.reshape(b, o, *y.shape[2:])


def timestep_features(t, dim, max_period=10000):
    """Sinusoidal embedding [cos | sin] of possibly fractional t (ref :80-97)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


class Dense(nn.Module):
    def __init__(self, din, dout):
        super().__init__()
        self.dense = nn.Linear(din, dout)

    def forward(self, v):
        return self.dense(v)[..., None, None]


class Swish(nn.Module):
    def forward(self, x):
        return silu(x)


class AdaptionModulateBEST(nn.Module):
    def __init__(self, cin, cout, edim):
        super().__init__()
        self.conv20 = nn.Conv3d(cin, cout, 1)
        self.conv21 = nn.Conv3d(cout, cout, 3, padding=1)
        self.act = Swish()
        self.dense2 = Dense(edim, cout)

    def forward(self, h, temb, pemb):
        h = silu(self.conv20(h))
        return modconv3d(h, self.conv21.weight, self.dense2(pemb), 1)


class ResBlockModulateBEST(nn.Module):
    def __init__(self, cin, cout, edim, flag=False):
        super().__init__()
        self.conv20 = nn.Conv3d(cin, cout, 3, padding=1)
        self.conv21 = nn.Conv3d(cout, cout, 3, padding=1)
        self.dense1 = Dense(edim, cin)
        self.dense2 = Dense(edim, cout)
        self.dropout = nn.Dropout(0.2)
        self.res_conv = nn.Conv3d(cin, cout, 1) if cin != cout else nn.Identity()
        self.act = Swish()
        self.flag = flag

    def forward(self, x, temb, pemb):
        h = x if self.flag else x + self.dense1(temb).unsqueeze(-1)
        h = self.conv20(self.dropout(silu(h)))
        h = self.dropout(silu(h))
        h = modconv3d(h, self.conv21.weight, self.dense2(pemb), 1)
        return h + self.res_conv(x)


class WaveletUPorDown(nn.Module):
    def __init__(self, in_ch, temb_dim, zemb_dim, up=False, down=False, flag=False, hi_in_ch=None, dropout=0.2):
        super().__init__()
        out_ch = in_ch
        self.up, self.down, self.flag = up, down, flag
        self.Conv_0 = nn.Conv3d(in_ch, out_ch, 3, padding=1)
        self.Dense_0 = nn.Linear(temb_dim, out_ch)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = nn.Conv3d(out_ch, out_ch, 3, padding=1)
        self.Conv_2 = nn.Conv3d(in_ch, out_ch, 1)
        if up:
            self.convH_0 = nn.Sequential(nn.Conv3d(hi_in_ch * 3, out_ch * 3, 3, padding=1, groups=3))
        self.dense1 = Dense(zemb_dim, in_ch)
        self.dense2 = Dense(zemb_dim, in_ch)     # present in the reference (:366), never used

    def forward(self, x, temb, pemb, skipH=None):
        b, c, n, hh, ww = x.shape
        h = self.Conv_0(silu(x))
        x = self.Conv_2(x)
        fold = lambda v: v.reshape(b, -1, v.shape[-2], v.shape[-1])                 # 'b c n h w -> b (c n) h w'
        unfold = lambda v, ch: v.reshape(b, ch, n, v.shape[-2], v.shape[-1])
        h, x = fold(h), fold(x)
        hH = None
        if self.up:
            d = h.shape[1]
            sk = fold(self.convH_0(torch.cat(skipH, dim=1) / 2.) * 2.)
            bands = (sk[:, :d], sk[:, d:2 * d], sk[:, 2 * d:])
            h = haar_idwt2d(2. * h, *bands)
            x = haar_idwt2d(2. * x, *bands)
        elif self.down:
            h, hlh, hhl, hhh = haar_dwt2d(h)
            x = haar_dwt2d(x)[0]
            hH = (unfold(hlh, c), unfold(hhl, c), unfold(hhh, c))
            h, x = h / 2., x / 2.
        h, x = unfold(h, c), unfold(x, c)
        if not self.flag:
            h = h + self.Dense_0(temb)[:, :, None, None, None]
        h = self.Dropout_0(silu(h))
        h = modconv3d(h, self.Conv_1.weight, self.dense1(pemb), 1)
        out = x + h
        return (out, hH) if self.down else out


class ResblockDownOneModulateBEST(nn.Module):
    def __init__(self, cin, cout, edim, flag=False):
        super().__init__()
        self.conv20 = ResBlockModulateBEST(cin, cout, edim, flag)
        self.down = WaveletUPorDown(cout, edim, edim, down=True, flag=flag)

    def forward(self, x, temb, pemb):
        return self.down(self.conv20(x, temb, pemb), temb, pemb)


class ResblockUpOneModulateBEST(nn.Module):
    def __init__(self, cin, cout, edim):
        super().__init__()
        self.up1 = WaveletUPorDown(cout, edim, edim, up=True, hi_in_ch=cin)
        self.conv20 = ResBlockModulateBEST(cin * 3, cout, edim)

    def forward(self, x, temb, skipH, pemb):
        return self.up1(self.conv20(x, temb, pemb), temb, pemb, skipH)


class FinalBlockModulateBEST(nn.Module):
    def __init__(self, cin, cout, edim):
        super().__init__()
        self.conv20 = ResBlockModulateBEST(cin * 3, cin, edim)
        self.conv21 = ResBlockModulateBEST(cin, cin, edim)
        self.conv22 = ResBlockModulateBEST(cin, cin, edim)
        self.conv23 = ResBlockModulateBEST(cin, cin, edim)
        self.conv24 = nn.Conv3d(cin, cout, 1)
        self.dense2 = Dense(edim, cin)
        self.act = Swish()

    def forward(self, x, temb, pemb):
        h = x
        for blk in (self.conv20, self.conv21, self.conv22, self.conv23):
            h = blk(h, temb, pemb)
        return modconv3d(silu(h), self.conv24.weight, self.dense2(pemb), 0)


def synthetic_text_embeddings(seed=1234):
    """Fixed synthetic 768-d 'pooled CLIP' vectors, one per prompt name.

    The real embeddings need CLIP weights that are not available offline;
    BASELINE config 1 asks for a fixed text embedding.  Prompt p gets
    randn(1,768) from a CPU generator seeded ``seed + index(p)``.
    """
    out = {}
    for k, name in enumerate(PROMPTS):
        g = torch.Generator(device="cpu").manual_seed(seed + k)
        out[name] = torch.randn(1, 768, generator=g)
    return out


class WavBESTRef(nn.Module):
    """Reference-shaped UNet (ref :523-636) with injectable text embeddings."""

    def __init__(self, channels=None, embed_dim=128, inter_dim=32, text_embeddings=None):
        super().__init__()
        c = channels if channels is not None else [16, 32, 64, 128]
        e = embed_dim
        self.inter_dim = inter_dim
        self.embed = nn.Sequential(nn.Linear(inter_dim, e), Swish(), nn.Linear(e, e))
        self.embed2 = nn.Sequential(nn.Linear(768, 4 * e), Swish(), nn.Linear(4 * e, 4 * e), Swish(),
                                    nn.Linear(4 * e, e))
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
        self.text_embeddings = dict(text_embeddings) if text_embeddings is not None else synthetic_text_embeddings()

    def get_embeding(self, prompt):
        return self.text_embeddings.get(prompt)      # unknown prompt -> None -> AttributeError below, as ref :602

    def forward(self, x_t, t_input, PAN=None, MS=None, prompt=None):
        t = t_input.view(-1)
        pe = self.get_embeding(prompt).repeat((x_t.shape[0], 1)).to(x_t.device)
        pemb = silu(self.embed2(pe))
        temb = silu(self.embed(timestep_features(t, self.inter_dim)))
        cond = (PAN.repeat(1, MS.shape[1], 1, 1) - MS).unsqueeze(1)
        x = x_t.unsqueeze(1)

        h0c = self.conv1(cond, temb, pemb)
        h1c, s1 = self.down1_1(h0c, temb, pemb)
        h2c, s2 = self.down2_1(h1c, temb, pemb)
        h3c, s3 = self.down3_1(h2c, temb, pemb)

        h0 = self.conv2(x, temb, pemb)
        h1, _ = self.down1(h0, temb, pemb)
        h2, _ = self.down2(h1, temb, pemb)
        h3, _ = self.down3(h2, temb, pemb)

        h = self.middle1(h3, temb, pemb)
        h = self.up1(torch.cat([h, h3c, h3], dim=1), temb, s3, pemb)
        h = self.up2(torch.cat([h, h2c, h2], dim=1), temb, s2, pemb)
        h = self.up3(torch.cat([h, h1c, h1], dim=1), temb, s1, pemb)
        h = self.final(torch.cat([h, h0c, h0], dim=1), temb, pemb)
        return h.squeeze(1)


def fill_weights_(module: nn.Module, seed: int = 0):
    """Deterministic key-hashed filler: every fp32 matrix/kernel of ``state_dict`` gets
    N(0, 1/fan_in) values, biases N(0, 0.05^2) -- except the ``Dense`` projections
    (``dense{1,2}.dense.bias``, the modulation scales / shifts) which get mean 1 so that
    the modulated convolutions keep O(1) activations -- from a generator seeded by
    crc32(key) ^ seed, so the reference and the build agree without weight files."""
    import zlib
    with torch.no_grad():
        for key, ten in module.state_dict().items():
            if not ten.is_floating_point() or "clip_text_model" in key:
                continue
            g = torch.Generator(device="cpu").manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
            if ten.dim() >= 2:
                fan_in = ten[0].numel()
                ten.copy_(torch.randn(ten.shape, generator=g) * math.sqrt(1.0 / fan_in))
            elif key.endswith("bias"):
                mean = 1.0 if key.endswith(".dense.bias") else 0.0
                ten.copy_(mean + torch.randn(ten.shape, generator=g) * 0.05)
            elif key.endswith(".b"):                     # NIN bias of AttnBlockpp
                ten.copy_(torch.randn(ten.shape, generator=g) * 0.05)
            elif key.endswith("weight"):                 # 1-D norm scales (attention ops only)
                ten.copy_(1.0 + torch.randn(ten.shape, generator=g) * 0.1)
    return module
