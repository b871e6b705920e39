"""Thin tensor-level wrappers over the C ABI (include/tmdiff_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every FLOP of
these ops runs in libtmdiff_hip.so.  All wrappers require CUDA(HIP) fp32 contiguous
tensors and raise otherwise -- there is no eager fallback.
"""
import ctypes as C
import os

import torch

from . import _lib, routing
from ._lib import Conv3dDesc, check, lib



class KernelConfig:
    """Every behaviour switch of the Python side, in one place.  Each field is read ONCE from its environment variable when
    the package is imported and is a plain attribute afterwards: code reads ``ops.config.<field>`` at call time, tests and
    experiments change it with ``with ops.config.override(field=value): ...`` (no re-import, no environment juggling).
    Defaults are the production path; every other value exists for A/B measurements and for the parity tests of the
    alternative paths (tests/test_gpu_configs.py runs the finetune step with each result-changing default switched off).

    field              env var                  default  meaning
    -----------------  -----------------------  -------  ---------------------------------------------------------------------
    winograd           TMDIFF_WINOGRAD          True     3x3x3 convolutions on the Winograd-along-the-bands kernels (conv3d_wf,
                                                         fallback.conv3d_wino); False = direct kernels everywhere
    wf                 TMDIFF_WF                True     ... with the input transform inside the kernel (conv3d_wf) where it applies
    wf_pair            TMDIFF_WF_PAIR           True     8 bands x 8 columns: two images per 8x16 tile (pair mode); False = fallback
    wf_splitk          TMDIFF_WF_SPLITK         True     small conv3d_wf grids split their input channels (+ reduction kernel)
    wf_min_fill        TMDIFF_WF_MIN_FILL       0.7      planes that fill less of conv3d_wf's tiles go to the fallback
    wino_min_blocks    TMDIFF_WINO_MIN_BLOCKS   256      Winograd grids below this many workgroups go to the direct kernels
                                                         (1 = force the production family onto small test batches)
    ll_compose         TMDIFF_LL_COMPOSE        True     main-branch Conv_0 + halved LL band as one composed convolution
    wfll               TMDIFF_WFLL              True     ... with Winograd on top (conv3d_wf's composed-LL mode) in inference
    train_ll_wino      TMDIFF_TRAIN_LL_WINO     True     ... and in the finetune forward; False = conv3d_ll there
    conv2_after_ll     TMDIFF_CONV2_AFTER_LL    True     down blocks: the 1x1x1 Conv_2 after the LL band it commutes with
    epilogue_fuse      TMDIFF_EPILOGUE_FUSE     True     conv20 -> conv21: the producer's epilogue writes the consumer's prologue
    producer_fuse      TMDIFF_PRODUCER_FUSE     True     ... likewise for DWT / IDWT / stem producers
    fp32_staged        TMDIFF_FP32_STAGED       "auto"   direct fp32 kernels: "0" fused only, "1" staged wherever supported
    bf16_pack          TMDIFF_BF16_PACK         "auto"   bf16 mode: "0" fused kernel, "1"/"auto" packed input + DMA kernel
    wgrad_wino         TMDIFF_WGRAD_WINO        True     3x3x3 weight gradients in the Winograd domain (F(3,4)); False = direct
    wgrad_wino_bias    TMDIFF_WGRAD_WINO_BIAS   True     ... whose g pass sums the bias gradient on the side
    wgrad_bias         TMDIFF_WGRAD_BIAS        False    direct weight gradient accumulates the bias gradient in-kernel
    wino_multipack     TMDIFF_WINO_MULTIPACK    True     finetune step: all Winograd weight forms re-packed by one launch
    fuse_res_conv      TMDIFF_FUSE_RES_CONV     True     inference: a ResBlock's 1x1x1 res_conv folded into conv21's epilogue (conv3d_wf)
    emit_ll            TMDIFF_EMIT_LL           True     inference: the ResBlock in front of a down block writes LL(y) / 2 instead of y
    emit_dwt           TMDIFF_EMIT_DWT          True     inference: Conv_0 of a down block whose high bands are kept writes the Haar
                                                         transform of its output instead of the output (no DWT pass)
    side_xp            TMDIFF_SIDE_XP           True     inference: a ResBlock's res_conv launch (segmented input) also writes conv20's
                                                         prologue output (no prologue pass in front of conv20)
    train_fused_resblock TMDIFF_TRAIN_FUSED_RESBLOCK True finetune: a ResBlock with a res_conv is ONE autograd node (conv20's input gradients
                                                         are added to res_conv's by the prologue-backward kernel, not by a sum launch)
    train_graph        TMDIFF_TRAIN_GRAPH       False    (model.DDPM) capture the finetune step into a HIP graph
    train_two_streams  TMDIFF_TRAIN_STREAMS     True     forward_train runs the condition branch on a second stream beside the
                                                         main branch's down path (at a local batch of 8 most launches fill half
                                                         of the CU slots); autograd then runs their backward passes side by side too

    Library-side experiment variables (read by libtmdiff_hip.so itself, C getenv): TMDIFF_SPLITK, TMDIFF_SPLITK_LONG,
    TMDIFF_WF_STAGGER, TMDIFF_WINO_STAGGER, TMDIFF_WINO_F4, TMDIFF_EPILOGUE_VEC, TMDIFF_SMALLGRID, TMDIFF_WW_PHASES,
    TMDIFF_ATTN_SIMPLE, TMDIFF_CONV1_VEC, TMDIFF_CONV1_DWORD -- timing experiments only; TMDIFF_HIP_LIB selects a diagnostic
    build of the library (_lib.py)."""

    _FLAG = lambda default: (lambda v: (v != "0") if default else (v == "1"))
    _FIELDS = {
        "winograd": ("TMDIFF_WINOGRAD", _FLAG(True), True), "wf": ("TMDIFF_WF", _FLAG(True), True),
        "wf_pair": ("TMDIFF_WF_PAIR", _FLAG(True), True), "wf_splitk": ("TMDIFF_WF_SPLITK", _FLAG(True), True),
        "wf_min_fill": ("TMDIFF_WF_MIN_FILL", float, 0.7), "wino_min_blocks": ("TMDIFF_WINO_MIN_BLOCKS", int, 256),
        "ll_compose": ("TMDIFF_LL_COMPOSE", _FLAG(True), True), "wfll": ("TMDIFF_WFLL", _FLAG(True), True),
        "train_ll_wino": ("TMDIFF_TRAIN_LL_WINO", _FLAG(True), True),
        "conv2_after_ll": ("TMDIFF_CONV2_AFTER_LL", _FLAG(True), True),
        "epilogue_fuse": ("TMDIFF_EPILOGUE_FUSE", _FLAG(True), True), "producer_fuse": ("TMDIFF_PRODUCER_FUSE", _FLAG(True), True),
        "fp32_staged": ("TMDIFF_FP32_STAGED", str, "auto"), "bf16_pack": ("TMDIFF_BF16_PACK", str, "auto"),
        "wgrad_wino": ("TMDIFF_WGRAD_WINO", _FLAG(True), True), "wgrad_wino_bias": ("TMDIFF_WGRAD_WINO_BIAS", _FLAG(True), True),
        "wgrad_bias": ("TMDIFF_WGRAD_BIAS", _FLAG(False), False), "wino_multipack": ("TMDIFF_WINO_MULTIPACK", _FLAG(True), True),
        "fuse_res_conv": ("TMDIFF_FUSE_RES_CONV", _FLAG(True), True), "emit_ll": ("TMDIFF_EMIT_LL", _FLAG(True), True),
        "emit_dwt": ("TMDIFF_EMIT_DWT", _FLAG(True), True), "side_xp": ("TMDIFF_SIDE_XP", _FLAG(True), True),
        "train_fused_resblock": ("TMDIFF_TRAIN_FUSED_RESBLOCK", _FLAG(True), True),
        "train_graph": ("TMDIFF_TRAIN_GRAPH", _FLAG(False), False),
        "train_two_streams": ("TMDIFF_TRAIN_STREAMS", _FLAG(True), True),
    }

    def __init__(self, env=None):
        env = os.environ if env is None else env
        for name, (var, parse, default) in self._FIELDS.items():
            object.__setattr__(self, name, parse(env[var]) if var in env else default)

    def __setattr__(self, name, value):
        if name not in self._FIELDS:
            raise AttributeError(f"ops.config has no switch {name!r}")
        object.__setattr__(self, name, value)

    def key(self):
        """The current values as a tuple (cache keys of routing decisions)."""
        return tuple(getattr(self, n) for n in self._FIELDS)

    def as_dict(self):
        return {n: getattr(self, n) for n in self._FIELDS}

    def override(self, **values):
        """Context manager: the given switches take the given values inside the block."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old = {k: getattr(self, k) for k in values}
            try:
                for k, v in values.items():
                    setattr(self, k, v)
                yield self
            finally:
                for k, v in old.items():
                    setattr(self, k, v)
        return ctx()


config = KernelConfig()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """hipStream_t of torch's current stream on the current device (the raw getter: torch.cuda.current_stream() builds
    a Stream object per call, ~15 us -- 4 ms per training step at ~300 kernel launches)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t, name):
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: need a contiguous float32 tensor on the GPU, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")
    return t.data_ptr()


def pack_conv_weight(w, groups=1, mode=0):
    """[Cout, Cin/g, k,k,k] -> packed [g][ci][tap][co] (mode 0) or the data-gradient packing (mode 1)."""
    cout, cin_g, k = w.shape[0], w.shape[1], w.shape[2]
    out = torch.empty(w.numel(), device=w.device, dtype=torch.float32)
    check(lib.tmdiff_conv3d_pack_weights(_chk(w.detach(), "w"), out.data_ptr(), cout, cin_g * groups, k, groups, mode,
                                         stream_ptr()), "conv3d_pack_weights")
    return out


class PackedWeights:
    """Forward and data-gradient packings of a list of convolution weights, refreshed by ONE launch
    (tmdiff_conv3d_pack_weights_multi).  ``refresh()`` re-packs when any weight changed (``_version``); ``lookup(w)``
    returns (packed_fwd, packed_dgrad) for a weight whose current version is packed, else None."""

    def __init__(self, convs):
        """convs: list of (weight [Cout, Cin/g, k, k, k] on the GPU, groups)"""
        self.convs = [(w, g) for w, g in convs]
        dev = self.convs[0][0].device
        self.fwd = [torch.empty(w.numel(), device=dev, dtype=torch.float32) for w, _ in self.convs]
        self.dgrad = [torch.empty(w.numel(), device=dev, dtype=torch.float32) for w, _ in self.convs]
        self._table_key, self.versions = None, None
        self._index = {}

    def _build_tables(self):
        ent = bytearray()
        ct, ci = [], []
        import struct
        for k, ((w, g), f, d) in enumerate(zip(self.convs, self.fwd, self.dgrad)):
            _chk(w.detach(), "weight")
            ent += struct.pack("<QQQiiii", w.data_ptr(), f.data_ptr(), d.data_ptr(), w.shape[0], w.shape[1] * g, w.shape[2], g)
            if w.shape[2] not in (1, 3):
                raise ValueError("PackedWeights: kernel size 1 or 3")
            na = C.c_int32(0)
            nck = lib.tmdiff_conv3d_pack_weights_multi_chunks(w.shape[0], w.shape[1] * g, g, C.byref(na))
            ct += [k] * nck
            ci += list(range(na.value)) + [(1 << 30) | t for t in range(nck - na.value)]
        dev = self.convs[0][0].device
        self.entries = torch.frombuffer(bytearray(ent), dtype=torch.uint8).clone().to(dev)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32).to(dev)
        self.chunk_index = torch.tensor(ci, dtype=torch.int32).to(dev)
        self.n_chunks = len(ct)
        self._table_key = tuple(w.data_ptr() for w, _ in self.convs)
        self._index = {w.data_ptr(): k for k, (w, _) in enumerate(self.convs)}

    def refresh(self):
        ptrs = tuple(w.data_ptr() for w, _ in self.convs)
        if ptrs != self._table_key:
            self._build_tables()
            self.versions = None
        vers = tuple(w._version for w, _ in self.convs)
        if vers != self.versions:
            check(lib.tmdiff_conv3d_pack_weights_multi(self.entries.data_ptr(), self.chunk_tensor.data_ptr(),
                                                       self.chunk_index.data_ptr(), self.n_chunks, stream_ptr()),
                  "conv3d_pack_weights_multi")
            self.versions = vers
        return self

    def lookup(self, w):
        k = self._index.get(w.data_ptr())
        if k is None or self.versions is None or self.versions[k] != w._version or self.convs[k][0].shape != w.shape:
            return None
        return self.fwd[k], self.dgrad[k]


class WinoPackedWeights:
    """The Winograd (conv3d_wf) packings -- forward (mode 2) / data-gradient (mode 3) form, F(4,3), natural column order -- of the
    3x3x3 weights a finetune step ACTUALLY runs through that kernel, refreshed by ONE launch per step
    (tmdiff_conv3d_wino_pack_weights_multi).  Which (weight, form) pairs those are depends on the tensor sizes (the 8x8 level
    and small grids take other kernels, and the deep levels hold most of the parameters), so the set is learnt: ``get``
    packs a pair it does not hold by itself and registers it; ``refresh`` (once per step, before the forward) re-packs the
    registered pairs whose weight changed.

    ``owner`` = the weights of the network this pack belongs to (WavBEST.forward_train passes its convolution weights): a weight
    that is not among them -- another module's, a test's -- is packed on the spot by ``get`` and NOT registered, so a foreign
    call can neither keep tensors alive here nor dirty the table.  The device table holds raw addresses; it is
    rebuilt whenever the (weight address, packed address) pairs it was built from are not the live ones any more -- a
    ``copy.deepcopy`` of a network that had already trained used to launch the multi-pack on the SOURCE network's memory."""

    def __init__(self, owner=None):
        self.items = {}               # (weight address, mode) -> [weight, groups, packed tensor, packed version]
        self.owner = None if owner is None else list(owner)
        self._owner_ptrs = None
        self._table_key = None
        self.n_chunks = 0

    def _live_key(self):
        return tuple((it[0].data_ptr(), it[2].data_ptr(), mode) for (_, mode), it in self.items.items())

    def _build_tables(self):
        import struct
        chunk = lib.tmdiff_conv3d_wino_pack_weights_multi_chunk()
        ent, ct, ci = bytearray(), [], []
        for n, ((_, mode), (w, g, out, _)) in enumerate(self.items.items()):
            ent += struct.pack("<QQiiiiii", w.data_ptr(), out.data_ptr(), w.shape[0], w.shape[1] * g, g, mode, 6, 0)
            nck = (w.numel() // 27 + chunk - 1) // chunk          # (chunks of weight rows: 27 taps each)
            ct += [n] * nck
            ci += list(range(nck))
        dev = next(iter(self.items.values()))[0].device
        self.entries = torch.frombuffer(bytearray(ent), dtype=torch.uint8).clone().to(dev)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32).to(dev)
        self.chunk_index = torch.tensor(ci, dtype=torch.int32).to(dev)
        self.n_chunks = len(ct)
        self._table_key = self._live_key()

    def refresh(self):
        if self.owner is not None:
            self._owner_ptrs = {w.data_ptr() for w in self.owner}
        # a registered weight whose storage moved (module.to(), parameters re-created) is forgotten and re-learnt
        self.items = {k: it for k, it in self.items.items()
                      if it[0].data_ptr() == k[0] and it[0].device == it[2].device}
        if not self.items:
            self._table_key = None
            return self
        if self._table_key != self._live_key():      # new pairs since the table was built
            self._build_tables()
            for it in self.items.values():
                it[3] = None
        if any(it[3] != it[0]._version for it in self.items.values()):
            check(lib.tmdiff_conv3d_wino_pack_weights_multi(self.entries.data_ptr(), self.chunk_tensor.data_ptr(),
                                                            self.chunk_index.data_ptr(), self.n_chunks, stream_ptr()),
                  "conv3d_wino_pack_weights_multi")
            for it in self.items.values():
                it[3] = it[0]._version
        return self

    def __deepcopy__(self, memo):
        """A copy starts empty and is re-learnt by the copy's first step: the packed tensors and the device table belong to
        the tensors of the network they were built from (WavBEST.__getstate__ drops the pack from copies anyway)."""
        return WinoPackedWeights()

    def get(self, w, groups, mode):
        key = (w.data_ptr(), mode)
        it = self.items.get(key)
        if it is not None and it[0].shape != w.shape:
            it = None
        if it is not None and it[3] == w._version:
            return it[2]
        if self.owner is not None and self._owner_ptrs is None:
            self._owner_ptrs = {t.data_ptr() for t in self.owner}
        foreign = self._owner_ptrs is not None and w.data_ptr() not in self._owner_ptrs
        if it is not None:              # registered, but the weight changed since the last refresh
            out = it[2]
        else:
            _chk(w.detach(), "weight")
            out = torch.empty(w.numel() * 2, device=w.device, dtype=torch.float32)   # six planes for three taps
        cout, cin = w.shape[0], w.shape[1] * groups
        if mode & 1:
            cout, cin = cin, cout
        check(lib.tmdiff_conv3d_wino_pack_weights(w.detach().data_ptr(), out.data_ptr(), cout, cin, groups, mode, 6, stream_ptr()),
              "conv3d_wino_pack_weights")
        if not foreign:
            self.items[key] = [w, groups, out, w._version]
        return out


PACKED = None      # the PackedWeights of the network being trained (set by WavBEST.forward_train; autograd.py consults it)
WINO_PACKED = None # ... and its WinoPackedWeights


def bf16_conv_supported(cout, cin, ksize, groups=1, seg_channels=None):
    """Shapes tmdiff_conv3d_fwd_bf16 accepts (include/tmdiff_hip.h); other layers stay on the fp32 kernel."""
    if ksize not in (1, 3) or cin % groups or cout % groups or (cout // groups) % 32:
        return False
    if (cin // groups) % (8 if ksize == 3 else 16):
        return False
    return all(c % 8 == 0 for c in (seg_channels or ()))


def pack_conv_weight_bf16(w, groups=1):
    """[Cout, Cin/g, k,k,k] fp32 -> bf16 packing [g][Cin_g/8][tap slots][Cout_g][8] (returned as an int16 tensor);
    28 tap slots for 3x3x3, one for 1x1x1."""
    cout, cin_g, k = w.shape[0], w.shape[1], w.shape[2]
    nbytes = lib.tmdiff_conv3d_packed_bf16_bytes(cout, cin_g * groups, k, groups) if tuple(w.shape[2:]) == (k, k, k) else 0
    if nbytes == 0:
        raise ValueError(f"conv weight {tuple(w.shape)} (groups={groups}) has no bf16 packing")
    out = torch.empty(nbytes // 2, device=w.device, dtype=torch.int16)
    check(lib.tmdiff_conv3d_pack_weights_bf16(_chk(w.detach(), "w"), out.data_ptr(), cout, cin_g * groups, k, groups,
                                              stream_ptr()), "conv3d_pack_weights_bf16")
    return out


def make_conv_desc(segs, w_packed, cout, ksize, y, groups=1, bias=None, bias_scale=1.0, in_shift=None, in_scale=None,
                   shift_stride=0, scale_stride=0, in_act=False, in_mask=None, residual=None, out_scale=1.0,
                   y2=None, y2_shift=None, y2_scale=None, y2_shift_stride=0, y2_scale_stride=0, y2_act=False,
                   x_bf16_shape=None, drop=None, out_div=1, y2_s2d=False, x_s2d=False, res_conv=None, y_ll=None, y_hi=None,
                   side_xp=None):
    """Fill a tmdiff_conv3d_desc.  `segs` = list of 1..3 tensors [B, c_i, N, H, W] (concat-free input).
    in_shift / in_scale may be tensors or raw (ptr) ints into a projection bank.  y may be None when only the second
    output y2 = act2(y + y2_shift) * y2_scale (the consumer's prologue, same pointer conventions) is wanted.
    A y2 of dtype int16 is written as bf16 units [B, Cout/8, N*H*W, 8]; x_bf16_shape = (N, H, W) says that segs[0] is
    such a tensor (bf16 entry point only).  drop = (seed, p): in-kernel dropout of the prologue output (no mask tensor).
    out_div = 2: outputs / residual at half the H and W of the input (tmdiff_conv3d_ll_fwd).  y2_s2d: y2 in space-to-depth
    form [B, 4 Cout, N, H/2, W/2] (tmdiff_conv3d_wf_fwd only); x_s2d: segs[0] is such a tensor and the descriptor is that of
    the convolution on the full-resolution tensor it stands for (tmdiff_conv3d_wfll_fwd).  res_conv = (x_raw [B, Cx, N, H, W],
    the 1x1x1 weight [Cout, Cx, 1, 1, 1] itself (contiguous fp32, not packed), Cx): a ResBlock's res_conv folded into the epilogue (tmdiff_conv3d_wf_fwd
    only; desc.rc_*) -- the caller passes res_conv's bias as `bias` and no residual.  side_xp = dict(out= [B, Cin, N, H, W], shift=, shift_stride=,
    act=): a 1x1x1 convolution on the bandwidth kernel also writes act(x + shift) of its (segmented) input there -- the prologue
    output another convolution of the same input wants (desc.xp_*; routing.k1_side_xp)."""
    d = Conv3dDesc()
    if side_xp is not None:
        so, ssh = side_xp["out"], side_xp.get("shift")
        if ksize != 1 or tuple(so.shape) != (segs[0].shape[0], sum(s.shape[1] for s in segs), *segs[0].shape[2:]):
            raise ValueError("conv3d: side_xp goes with a 1x1x1 convolution; out = [B, Cin, N, H, W]")
        d.xp_out, d.xp_shift = _chk(so, "side_xp out"), (ssh if isinstance(ssh, int) else _chk(ssh, "side_xp shift"))
        d.xp_shift_stride, d.xp_act = int(side_xp.get("shift_stride", 0)), 1 if side_xp.get("act") else 0
    if res_conv is not None:
        rx, rw, rcin = res_conv
        if residual is not None or tuple(rx.shape) != (segs[0].shape[0], rcin, *segs[0].shape[2:]) or rw.numel() != rcin * cout:
            raise ValueError("conv3d: res_conv = (x [B, Cx, N, H, W], weight [Cout, Cx, 1, 1, 1], Cx) and no residual tensor")
        d.rc_x, d.rc_w, d.rc_cin = _chk(rx, "res_conv input"), _chk(rw, "res_conv weights"), int(rcin)
    if drop is not None:
        if in_mask is not None:
            raise ValueError("conv3d: give either in_mask (a mask tensor) or drop=(seed, p)")
        d.drop_seed, d.drop_p = int(drop[0]) & 0xFFFFFFFFFFFFFFFF, float(drop[1])
        if DROP_WORD is not None:         # the per-step part of the seed lives in device memory (HIP-graph replays)
            d.drop_seed_dev = DROP_WORD.data_ptr()
    if x_bf16_shape is not None:
        xp = segs[0]
        if len(segs) != 1 or xp.dtype != torch.int16 or xp.dim() != 4 or xp.shape[3] != 8 or not xp.is_contiguous():
            raise ValueError("conv3d: a bf16-packed input is one contiguous int16 tensor [B, Cin/8, N*H*W, 8]")
        b, (n, h, w) = xp.shape[0], x_bf16_shape
        if xp.shape[2] != n * h * w:
            raise ValueError("conv3d: packed input does not match x_bf16_shape")
        d.B, d.N, d.H, d.W = b, n, h, w
        d.Cin = xp.shape[1] * 8
        d.Cout, d.groups, d.ksize, d.nseg = cout, groups, ksize, 1
        d.seg_c[0], d.seg_x[0], d.x_bf16 = d.Cin, xp.data_ptr(), 1
    elif x_s2d:
        xs = segs[0]
        if len(segs) != 1 or xs.dim() != 5 or xs.shape[1] % 4:
            raise ValueError("conv3d: a space-to-depth input is one tensor [B, 4 Cin, N, H/2, W/2]")
        b, c4, n, h, w = xs.shape
        h, w = 2 * h, 2 * w
        d.B, d.N, d.H, d.W = b, n, h, w
        d.Cin, d.Cout, d.groups, d.ksize, d.nseg = c4 // 4, cout, groups, ksize, 1
        d.seg_c[0], d.seg_x[0] = c4 // 4, _chk(xs, "segment 0")
    else:
        b, _, n, h, w = segs[0].shape
        d.B, d.N, d.H, d.W = b, n, h, w
        d.Cin = sum(s.shape[1] for s in segs)
        d.Cout, d.groups, d.ksize, d.nseg = cout, groups, ksize, len(segs)
        for i, s in enumerate(segs):
            if tuple(s.shape[2:]) != (n, h, w) or s.shape[0] != b:
                raise ValueError("conv3d: input segments disagree on [B, N, H, W]")
            d.seg_c[i] = s.shape[1]
            d.seg_x[i] = _chk(s, f"segment {i}")
    if not isinstance(w_packed, int) and not (w_packed.is_cuda and w_packed.is_contiguous()):
        raise ValueError("w_packed: need a contiguous packed-weight tensor on the GPU")
    d.w_packed = w_packed if isinstance(w_packed, int) else w_packed.data_ptr()
    d.bias = _chk(bias, "bias")
    d.bias_scale = bias_scale
    d.in_shift = in_shift if isinstance(in_shift, int) else _chk(in_shift, "in_shift")
    d.in_scale = in_scale if isinstance(in_scale, int) else _chk(in_scale, "in_scale")
    d.in_shift_stride, d.in_scale_stride = shift_stride, scale_stride
    d.in_mask = _chk(in_mask, "in_mask")
    d.in_act = 1 if in_act else 0
    oshape = (b, cout, n, h // out_div, w // out_div)
    if residual is not None and tuple(residual.shape) != oshape:
        raise ValueError("conv3d: residual shape != output shape")
    d.residual = _chk(residual, "residual")
    d.out_scale = out_scale
    y2_packed = y2 is not None and y2.dtype == torch.int16
    for t, nm in ((y, "y"), (None if y2_packed else y2, "y2")):
        want = (b, 4 * cout, n, h // (2 * out_div), w // (2 * out_div)) if (y2_s2d and nm == "y2") else oshape
        if t is not None and tuple(t.shape) != want:
            raise ValueError(f"conv3d: {nm} shape {tuple(t.shape)} != {want}")
    if y2_packed and (tuple(y2.shape) != (b, cout // 8, n * h * w, 8) or not (y2.is_cuda and y2.is_contiguous())):
        raise ValueError(f"conv3d: packed y2 shape {tuple(y2.shape)} != {(b, cout // 8, n * h * w, 8)}")
    if y is None and y2 is None and not (y_ll is not None and y_hi is not None):
        raise ValueError("conv3d: no output")
    d.y = _chk(y, "y")
    d.y2 = y2.data_ptr() if y2_packed else _chk(y2, "y2")
    d.y2_shift = y2_shift if isinstance(y2_shift, int) else _chk(y2_shift, "y2_shift")
    d.y2_scale = y2_scale if isinstance(y2_scale, int) else _chk(y2_scale, "y2_scale")
    d.y2_shift_stride, d.y2_scale_stride = y2_shift_stride, y2_scale_stride
    d.y2_act = 1 if y2_act else 0
    d.y2_bf16 = 1 if y2_packed else 0
    d.y2_s2d = 1 if (y2_s2d and y2 is not None) else 0
    if y_ll is not None:       # third output: the halved LL band of y (tmdiff_conv3d_wf_fwd; y itself not written) ...
        q = (b, cout, n, h // 2, w // 2)
        if y is not None or (y2 is None) == (y_hi is None) or tuple(y_ll.shape) != q:
            raise ValueError("conv3d: y_ll [B, Cout, N, H/2, W/2] goes with y=None and EITHER a second output OR the three high bands")
        d.y_ll = _chk(y_ll, "y_ll")
        if y_hi is not None:   # ... or, with the three high bands, the whole Haar transform (LL through the y2 prologue constants)
            if len(y_hi) != 3 or any(tuple(t.shape) != q for t in y_hi):
                raise ValueError("conv3d: y_hi = (LH, HL, HH), each [B, Cout, N, H/2, W/2]")
            for i, t in enumerate(y_hi):
                d.y_hi[i] = _chk(t, "y_hi")
    return d


class ConvTimer:
    """Optional per-launch HIP-event timing of tmdiff_conv3d_fwd (used by bench.py for the roofline
    object).  Events are recorded on the launch stream; ``summary()`` resolves them after a sync."""

    def __init__(self):
        self.records = []        # (start_event, end_event, flops, ksize, entry point)

    def summary(self, by_entry=False):
        """{ksize: (launches, ms, flops)}; with by_entry the key is (ksize, C entry point that served the launch)."""
        torch.cuda.synchronize()
        out = {}
        for e0, e1, fl, k, what, tag, *_ in self.records:
            key = (k, what, tag) if by_entry == "layer" else ((k, what) if by_entry else k)
            n, ms, f = out.get(key, (0, 0.0, 0.0))
            out[key] = (n + 1, ms + e0.elapsed_time(e1), f + fl)
        return out

    def bytes_summary(self, ksize=1):
        """(launches, ms, algorithmic HBM bytes) of the launches of one kernel size that recorded their bytes (the 1x1x1
        convolutions: input + output (+ residual) tensors once -- they are bandwidth kernels, priced against the copy rate)."""
        torch.cuda.synchronize()
        n, ms, nbytes = 0, 0.0, 0.0
        for rec in self.records:
            if rec[3] == ksize and len(rec) > 6:
                n, ms, nbytes = n + 1, ms + rec[0].elapsed_time(rec[1]), nbytes + rec[6]
        return n, ms, nbytes


def _tag(d):
    """Shape label of a launch for ConvTimer.summary(by_entry="layer"): channels, plane, what the epilogue reads / writes."""
    return (f"{d.Cin}->{d.Cout} g{d.groups} {d.N}x{d.H}x{d.W} b{d.B}" + (" seg%d" % d.nseg if d.nseg > 1 else "") +
            (" pro" if (d.in_act or d.in_shift or d.in_scale) else "") + (" +res" if d.residual else "") +
            (" y" if d.y else "") + (" y2" if d.y2 else "") + (" dwt" if d.y_hi[0] else (" ll" if d.y_ll else "")))


TIMER = None      # set to a ConvTimer() to time every conv launch
COUNTS = None     # set to a collections.Counter() to count convolution launches per C entry point ("conv3d_fwd",
                  # "conv3d_fwd_staged", "conv3d_fwd_bf16", "conv3d_wino4_fwd", "conv3d_wino2_fwd", "conv3d_ll_fwd"): the
                  # tests and bench.py's parity leg assert from it WHICH kernel family produced a result


DROP_WORD = None  # an int64 [1] device tensor: its value is added to every in-kernel dropout seed when the kernel starts
                  # (tmdiff_conv3d_desc.drop_seed_dev).  The captured finetune step (tmdiff_amd.model) sets it and bumps it
                  # inside the graph, so that replays of the same recorded launches draw fresh masks.


FLOPS = None      # set to [0.0] to add up the multiply-add FLOPs the convolution launches EXECUTE on the matrix pipe (Winograd
                  # and composed kernels execute fewer than the reference's operator order): bench.py's train_step object


def _count(what, flops=0.0):
    if COUNTS is not None:
        COUNTS[what] += 1
    if FLOPS is not None:
        FLOPS[0] += flops


_WS = {}     # (device index, stream) -> grow-only scratch tensor (prologue outputs / bf16-packed conv inputs); launches
             # on one stream are ordered, so consecutive convolutions can reuse it; other streams get their own


def scratch_like(segs, tag):
    """A [B, sum of channels, N, H, W] fp32 view of the per-stream scratch `tag` (contents live until its next user on the stream)."""
    b, _, n, h, w = segs[0].shape
    shape = (b, sum(s.shape[1] for s in segs), n, h, w)
    numel = shape[0] * shape[1] * n * h * w
    return _workspace(segs[0].device, numel * 4, tag).view(torch.float32)[:numel].view(shape)


def _workspace(device, nbytes, tag="x"):
    key = (device.index, stream_ptr(), tag)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty(nbytes, device=device, dtype=torch.uint8)
    return ws


def conv3d(segs, w_packed, cout, ksize, out=None, math="fp32", pack_input=None, staged=None, emit=None, keep_y=True,
           x_bf16_shape=None, xp_out=None, **kw):
    """math="fp32": exact-fp32 MFMA kernel (w_packed from pack_conv_weight); "bf16": bf16 operands / fp32
    accumulation (w_packed from pack_conv_weight_bf16).  pack_input (bf16 only): True = pack the prologue output to
    bf16 once and run the staging-free kernel (default), False = one fused kernel.  staged (fp32 only): True = prologue
    pass + global_load_lds staged kernel (default where the shape allows), False = the fused kernel.
    emit = dict(act=, shift=, scale=, shift_stride=, scale_stride=): also produce y2 = act(y + shift) * scale, the
    consumer's prologue applied in this convolution's epilogue; returns (y, y2), or y2 alone with keep_y=False.  In the
    bf16 mode y2 is the packed bf16 tensor [B, Cout/8, N*H*W, 8] (int16 storage) that a following bf16 convolution takes
    as its input with x_bf16_shape=(N, H, W), skipping its pack pass.
    xp_out (fp32, 3x3x3): a [B, Cin, N, H, W] tensor that receives the prologue output x' (forces the staged kernel, whose
    prologue pass writes it there instead of the shared scratch) -- the training path keeps it for the weight gradient."""
    if kw.get("res_conv") is not None or kw.get("y_ll") is not None or (emit is not None and (emit.get("ll") or emit.get("dwt"))):
        raise ValueError("conv3d: only conv3d_wf folds a residual convolution into its epilogue / writes the LL band")
    if x_bf16_shape is not None:
        b, (n, h, w) = segs[0].shape[0], x_bf16_shape
        kw = dict(kw, x_bf16_shape=x_bf16_shape)
    else:
        b, _, n, h, w = segs[0].shape
    dev = segs[0].device
    y = out if out is not None else (torch.empty(b, cout, n, h, w, device=dev, dtype=torch.float32) if keep_y else None)
    y2 = None
    if emit is not None:
        y2 = (torch.empty(b, cout // 8, n * h * w, 8, device=dev, dtype=torch.int16) if math == "bf16"
              else torch.empty(b, cout, n, h, w, device=dev, dtype=torch.float32))
        kw = dict(kw, y2=y2, y2_act=emit.get("act", False), y2_shift=emit.get("shift"), y2_scale=emit.get("scale"),
                  y2_shift_stride=emit.get("shift_stride", 0), y2_scale_stride=emit.get("scale_stride", 0))
    elif y is None:
        raise ValueError("conv3d: keep_y=False needs emit=")
    d = make_conv_desc(segs, w_packed, cout, ksize, y, **kw)
    if d.xp_out and (math != "fp32" or not lib.tmdiff_conv3d_fwd_xp_supported(C.byref(d))):
        raise ValueError("conv3d: side_xp needs the 16-byte 1x1x1 bandwidth kernel (ask routing.k1_side_xp first)")
    ret = y if y2 is None else ((y, y2) if y is not None else y2)
    if math == "bf16":
        if w_packed.dtype != torch.int16:
            raise TypeError("conv3d(math='bf16') needs weights from pack_conv_weight_bf16")
        if pack_input is None:
            pack_input = {"0": False, "1": True}.get(config.bf16_pack, True)   # measured: the two-kernel variant wins on every production layer
        ws = (_workspace(dev, lib.tmdiff_conv3d_bf16_workspace_bytes(C.byref(d))).data_ptr()
              if pack_input and ksize == 3 and x_bf16_shape is None else None)
        fwd, what = (lambda dd, st: lib.tmdiff_conv3d_fwd_bf16(dd, ws, st)), "conv3d_fwd_bf16"
    elif math == "fp32":
        if not isinstance(w_packed, int) and w_packed.dtype != torch.float32:
            raise TypeError("conv3d(math='fp32') needs weights from pack_conv_weight")
        if ksize == 3:      # lend the split-K workspace (small grids only: B = 1, the 8x8 / 16x16 levels at small batches)
            nsk = lib.tmdiff_conv3d_fwd_splitk_workspace_bytes(C.byref(d))
            if nsk:
                d.splitk_ws, d.splitk_ws_bytes = _workspace(dev, nsk, "splitk").data_ptr(), nsk
        if staged is None:       # (the rule and its measurements: routing.direct_family)
            plain = len(segs) == 1 and not (kw.get("in_act") or kw.get("in_shift") is not None or
                                            kw.get("in_scale") is not None or kw.get("in_mask") is not None)
            plain = plain and kw.get("drop") is None
            staged = ksize == 3 and routing.direct_family(d.Cin, cout, d.groups, plain, kw.get("in_mask") is not None,
                                                          kw.get("drop") is not None) == "staged"
        if xp_out is not None:
            staged = True
        if staged and lib.tmdiff_conv3d_fwd_staged_supported(C.byref(d)):
            nb = lib.tmdiff_conv3d_fwd_staged_workspace_bytes(C.byref(d))
            if xp_out is not None and nb:
                if xp_out.numel() * 4 != nb or not (xp_out.is_cuda and xp_out.is_contiguous()):
                    raise ValueError("conv3d: xp_out must be a contiguous fp32 [B, Cin, N, H, W] tensor")
                ws32 = xp_out.data_ptr()
            else:
                ws32 = _workspace(dev, nb).data_ptr() if nb else None
            fwd, what = (lambda dd, st: lib.tmdiff_conv3d_fwd_staged(dd, ws32, st)), "conv3d_fwd_staged"
        else:
            fwd, what = lib.tmdiff_conv3d_fwd, "conv3d_fwd"
    else:
        raise ValueError(f"conv3d: unknown math {math!r}")
    _count(what if ksize == 3 else what + "_k1", 2.0 * b * cout * (d.Cin // d.groups) * ksize ** 3 * n * h * w)
    if TIMER is None:
        check(fwd(C.byref(d), stream_ptr()), what)
        return ret
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fwd(C.byref(d), stream_ptr()), what)
    e1.record()
    # algorithmic HBM bytes of the launch: input once (2 B per element for packed bf16 units, else 4), outputs once
    in_b = 2.0 if d.x_bf16 else 4.0
    y2_b = 0.0 if y2 is None else (2.0 if d.y2_bf16 else 4.0)
    # (+ the by-product x' of a side_xp launch: Cin more channels written)
    nbytes = b * n * h * w * (in_b * d.Cin + cout * (4.0 * ((1 if y is not None else 0) + (1 if kw.get("residual") is not None else 0)) + y2_b) +
                              (4.0 * d.Cin if d.xp_out else 0.0))
    TIMER.records.append((e0, e1, 2.0 * b * cout * (d.Cin // d.groups) * ksize ** 3 * n * h * w, ksize, what,
                          _tag(d) + (" +xp" if d.xp_out else ""), nbytes))
    return ret


def conv3d_prologue(desc, shape):
    """x' = the prologue output of the convolution `desc` describes (shape [B, Cin, N, H, W]), as a tensor of its own."""
    xp = torch.empty(shape, device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float32)
    check(lib.tmdiff_conv3d_prologue_fwd(C.byref(desc), xp.data_ptr(), stream_ptr()), "conv3d_prologue_fwd")
    return xp


def pack_conv_weight_wino(w, groups=1, mode=0, planes=6):
    """[Cout, Cin/groups, 3, 3, 3] -> the transformed, packed weights of the Winograd-along-n convolution (conv3d_wino);
    planes = wino_planes(N) of the tensors it will run on.  mode bit 0: the weights of the DATA-GRADIENT convolution
    (Cout -> Cin channels, transposed, taps mirrored) of the forward convolution whose weight w is; mode bit 1 (value 2):
    natural column order, the form conv3d_wf takes (conv3d_wino: interleaved 64-channel tiles)."""
    cout, cin = w.shape[0], w.shape[1] * groups
    if mode & 1:
        cout, cin = cin, cout
    nb = lib.tmdiff_conv3d_wino_packed_bytes(cout, cin, groups, planes)
    if tuple(w.shape[2:]) != (3, 3, 3) or nb == 0:
        raise ValueError(f"pack_conv_weight_wino: weight shape {tuple(w.shape)} (groups {groups}, mode {mode}, planes {planes}) not supported")
    out = torch.empty(nb // 4, device=w.device, dtype=torch.float32)
    check(lib.tmdiff_conv3d_wino_pack_weights(_chk(w.detach(), "w"), out.data_ptr(), cout, cin, groups, mode, planes, stream_ptr()),
          "conv3d_wino_pack_weights")
    return out




# routing lives in tmdiff_amd/routing.py; these names stay importable from ops (tests, tools)
wino_conv_supported = routing.wino_weight_ok
wf_route = routing.wf_route
wfll_route = routing.wfll_route


class ConvWeights:
    """The packed forms of one 3x3x3 convolution weight, each produced on first use: direct() (pack_conv_weight), wf()
    (pack_conv_weight_wino mode | 2, planes 6), wino(planes) (fallback kernels).  Any callable may be None: that family is
    then not offered for this weight (conv3d_auto falls through to the next rule)."""

    def __init__(self, direct, wf=None, wino=None):
        self.direct, self.wf, self.wino = direct, wf, wino


def conv3d_auto(segs, weights, cout, groups=1, math="fp32", emit=None, keep_y=True, xp_out=None, x_bf16_shape=None, **kw):
    """A 3x3x3 convolution on the kernel family routing.conv3_family picks for its extents (same keyword arguments and
    return convention as conv3d): conv3d_wf, the direct kernels (staged / fused), the bf16 kernel, or -- band counts other
    than 4 / 8 -- tmdiff_amd.fallback.conv3d_wino."""
    if math == "bf16" or x_bf16_shape is not None:
        return conv3d(segs, weights.direct(), cout, 3, groups=groups, math=math, emit=emit, keep_y=keep_y,
                      x_bf16_shape=x_bf16_shape, **kw)
    b, _, n, h, w = segs[0].shape
    cin = sum(s_.shape[1] for s_ in segs)
    masked, dropout = kw.get("in_mask") is not None, kw.get("drop") is not None
    plain = len(segs) == 1 and not (kw.get("in_act") or kw.get("in_shift") is not None or kw.get("in_scale") is not None or
                                    masked or dropout)
    fam = routing.conv3_family(b, cin, cout, n, h, w, groups, plain=plain, masked=masked, dropout=dropout,
                               keep_xp=xp_out is not None)
    if fam in ("wf", "wf_pair") and weights.wf is not None:
        return conv3d_wf(segs, weights.wf(), cout, emit=emit, keep_y=keep_y, groups=groups, xp_out=xp_out, **kw)
    if fam in ("wino4", "wino2") and weights.wino is not None:
        from . import fallback
        planes = 6 if fam == "wino4" else 4
        return fallback.conv3d_wino(segs, weights.wino(planes), cout, planes, emit=emit, keep_y=keep_y, groups=groups,
                                    xp_out=xp_out, **kw)
    if (emit is not None and (emit.get("s2d") or emit.get("ll") or emit.get("dwt"))) or kw.get("res_conv") is not None:
        raise ValueError("conv3d_auto: only conv3d_wf writes a space-to-depth second output / folds a residual convolution "
                         "(ask routing.wf_route first)")
    staged = fam == "staged" or (fam not in ("staged", "fused") and
                                 routing.direct_family(cin, cout, groups, plain, masked, dropout, xp_out is not None) == "staged")
    return conv3d(segs, weights.direct(), cout, 3, groups=groups, staged=staged, emit=emit, keep_y=keep_y, xp_out=xp_out, **kw)


def conv3d_wf(segs, w_packed, cout, emit=None, keep_y=True, groups=1, xp_out=None, **kw):
    """conv3d(segs, ...) (fp32, 3x3x3) through the Winograd F(4,3)-along-the-bands kernel that transforms its input INSIDE the
    kernel (csrc/conv3d_wf.hip; 8- or 4-band tensors, the whole band axis in one workgroup): no transformed copy of the input,
    no transform pass -- an input that is one plain tensor is read as it stands, any other (prologue, segments, dropout) goes
    through one elementwise prologue pass first (its output lands in xp_out when given: the finetune path keeps it for the
    weight gradient).  Same keyword arguments and return convention as conv3d; w_packed = the weights from
    pack_conv_weight_wino(w, groups, mode | 2, planes=6).  Runs whatever the grid size (routing.conv3_family is where small
    grids are sent elsewhere); raises for shapes the kernel does not take."""
    b, _, n, h, w = segs[0].shape
    dev = segs[0].device
    s2d = bool(emit is not None and emit.get("s2d"))
    want_ll = bool(emit is not None and emit.get("ll"))      # third output: LL(y) / 2 instead of y (returns (y2, y_ll))
    want_dwt = bool(emit is not None and emit.get("dwt"))    # the whole Haar transform of y instead of y: returns (LL', LH, HL, HH),
    if want_dwt and (s2d or want_ll):                        # LL' = the halved LL band through the `emit` prologue
        raise ValueError("conv3d_wf: emit dwt=True excludes s2d / ll")
    want_ll = want_ll or want_dwt
    if s2d or want_ll:
        cin = sum(s_.shape[1] for s_ in segs)
        if routing.wf_route(b, cin, cout, n, h, w, groups)[1] > 1 or h % 2 or w % 4:
            raise ValueError("conv3d_wf: this launch cannot write a space-to-depth / LL output (ask routing.wf_route first)")
    if want_ll and keep_y:
        raise ValueError("conv3d_wf: the LL output replaces y (keep_y=False)")
    y = torch.empty(b, cout, n, h, w, device=dev, dtype=torch.float32) if keep_y else None
    y2 = None
    if want_ll:
        kw = dict(kw, y_ll=torch.empty(b, cout, n, h // 2, w // 2, device=dev, dtype=torch.float32))
    if want_dwt:
        kw = dict(kw, y_hi=[torch.empty(b, cout, n, h // 2, w // 2, device=dev, dtype=torch.float32) for _ in range(3)])
    if emit is not None and not want_dwt:
        y2 = torch.empty((b, 4 * cout, n, h // 2, w // 2) if s2d else (b, cout, n, h, w), device=dev, dtype=torch.float32)
    if emit is not None:
        kw = dict(kw, y2_act=emit.get("act", False), y2_shift=emit.get("shift"), y2_scale=emit.get("scale"),
                  y2_shift_stride=emit.get("shift_stride", 0), y2_scale_stride=emit.get("scale_stride", 0), y2_s2d=s2d)
    elif y is None:
        raise ValueError("conv3d_wf: keep_y=False needs emit=")
    d = make_conv_desc(segs, w_packed, cout, 3, y, y2=y2, groups=groups, **kw)
    if not lib.tmdiff_conv3d_wf_supported(C.byref(d)):
        raise ValueError("conv3d_wf: shape not supported")
    # (a launch that writes an LL / Haar / space-to-depth output runs unsplit: no workspace is lent)
    nsk = lib.tmdiff_conv3d_wf_splitk_workspace_bytes(C.byref(d)) if (config.wf_splitk and not want_ll) else 0
    if nsk:
        d.splitk_ws, d.splitk_ws_bytes = _workspace(dev, nsk, "splitk").data_ptr(), nsk
    nws = lib.tmdiff_conv3d_wf_workspace_bytes(C.byref(d))
    ws = None
    if nws:
        if xp_out is not None:
            if not (xp_out.is_cuda and xp_out.is_contiguous() and xp_out.numel() * 4 == nws):
                raise ValueError("conv3d_wf: xp_out must be a contiguous fp32 [B, Cin, N, H, W] tensor")
            ws = xp_out.data_ptr()
        else:
            ws = _workspace(dev, nws).data_ptr()
    elif xp_out is not None:          # plain input: x' IS the input
        xp_out.copy_(segs[0] if len(segs) == 1 else torch.cat(segs, 1))
    ret = y if y2 is None else ((y, y2) if y is not None else y2)
    if want_dwt:
        ret = (kw["y_ll"], *kw["y_hi"])
    elif want_ll:
        ret = (y2, kw["y_ll"])
    # EXECUTED flops: 54 multiply-adds per (ci, co) and tile of four output bands (the direct kernel: 27 per band), plus the
    # folded res_conv's rc_cin multiply-adds per output
    flops = 2.0 * b * cout * ((d.Cin // groups) * 13.5 + d.rc_cin) * n * h * w
    _count("conv3d_wf_fwd", flops)
    if TIMER is None:
        check(lib.tmdiff_conv3d_wf_fwd(C.byref(d), ws, stream_ptr()), "conv3d_wf_fwd")
        return ret
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.tmdiff_conv3d_wf_fwd(C.byref(d), ws, stream_ptr()), "conv3d_wf_fwd")
    e1.record()
    TIMER.records.append((e0, e1, flops, 3, "conv3d_wf_fwd", _tag(d) + (f" +rc{d.rc_cin}" if d.rc_cin else "")))
    return ret


def ll_conv_supported(cout, cin, ksize=3, groups=1):
    """Shapes tmdiff_conv3d_ll_fwd takes (conv3d_ll_halved below)."""
    return ksize == 3 and groups == 1 and cin % 2 == 0 and cout % 64 == 0


def pack_conv_weight_ll(w, ll_scale=0.5):
    """[Cout, Cin, 3, 3, 3] -> the composed, packed weights of `3x3x3 convolution, then LL band * ll_scale` (conv3d_ll)."""
    cout, cin = w.shape[0], w.shape[1]
    nb = lib.tmdiff_conv3d_ll_packed_bytes(cout, cin)
    if tuple(w.shape[2:]) != (3, 3, 3) or nb == 0:
        raise ValueError(f"pack_conv_weight_ll: weight shape {tuple(w.shape)} not supported")
    out = torch.empty(nb // 4, device=w.device, dtype=torch.float32)
    check(lib.tmdiff_conv3d_ll_pack_weights(_chk(w.detach(), "w"), out.data_ptr(), cout, cin, float(ll_scale), stream_ptr()),
          "conv3d_ll_pack_weights")
    return out


def conv3d_ll(x, w_packed, cout, ll_scale=0.5, emit=None, keep_y=True, **kw):
    """haar_dwt2d(conv3d(x, w), want_high=False, ll_scale)[0] as ONE strided convolution (csrc/conv3d_ll.hip): x is one
    plain fp32 tensor [B, Cin, N, H, W] (H, W even), the result [B, Cout, N, H/2, W/2].  bias / residual / out_scale / emit
    as in conv3d (residual and the outputs at the halved size); returns y, (y, y2) or y2 alone (keep_y=False)."""
    b, _, n, h, w = x.shape
    dev = x.device
    oshape = (b, cout, n, h // 2, w // 2)
    y = torch.empty(oshape, device=dev, dtype=torch.float32) if keep_y else None
    y2 = None
    if emit is not None:
        y2 = torch.empty(oshape, device=dev, dtype=torch.float32)
        kw = dict(kw, y2_act=emit.get("act", False), y2_shift=emit.get("shift"), y2_scale=emit.get("scale"),
                  y2_shift_stride=emit.get("shift_stride", 0), y2_scale_stride=emit.get("scale_stride", 0))
    elif y is None:
        raise ValueError("conv3d_ll: keep_y=False needs emit=")
    d = make_conv_desc([x], w_packed, cout, 3, y, y2=y2, out_div=2, **kw)
    if not lib.tmdiff_conv3d_ll_supported(C.byref(d)):
        raise ValueError("conv3d_ll: shape not supported (ll_conv_supported)")
    nsk = lib.tmdiff_conv3d_ll_splitk_workspace_bytes(C.byref(d))      # small grids: lend the split-K workspace
    if nsk:
        d.splitk_ws, d.splitk_ws_bytes = _workspace(dev, nsk, "splitk").data_ptr(), nsk
    ret = y if y2 is None else ((y, y2) if y is not None else y2)
    _count("conv3d_ll_fwd", 2.0 * b * cout * d.Cin * 48 * n * (h // 2) * (w // 2))
    if TIMER is None:
        check(lib.tmdiff_conv3d_ll_fwd(C.byref(d), float(ll_scale), stream_ptr()), "conv3d_ll_fwd")
        return ret
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.tmdiff_conv3d_ll_fwd(C.byref(d), float(ll_scale), stream_ptr()), "conv3d_ll_fwd")
    e1.record()
    # EXECUTED flops: 48 multiply-adds per (ci, co, output position) -- the pair it replaces would execute 4 x 27
    TIMER.records.append((e0, e1, 2.0 * b * cout * d.Cin * 48 * n * (h // 2) * (w // 2), 3, "conv3d_ll_fwd", _tag(d)))
    return ret




def pack_conv_weight_wfll(w, ll_scale=0.5):
    """Weights of conv3d_wf_ll: the composed 3x4x4 kernel of conv3d_ll split by row / column parity (2 x 2 taps per virtual
    channel of the space-to-depth input) and transformed along the bands (F(4,3)): [Cin, 2, 2, 4, 6, Cout] floats."""
    cout, cin = w.shape[0], w.shape[1]
    out = torch.empty(cin * 96 * cout, device=w.device, dtype=torch.float32)
    check(lib.tmdiff_conv3d_wfll_pack_weights(_chk(w.detach(), "w"), out.data_ptr(), cout, cin, float(ll_scale), stream_ptr()),
          "conv3d_wfll_pack_weights")
    return out


def conv3d_wf_ll(x_s2d, w_packed, cout, ll_scale=0.5, emit=None, keep_y=True, **kw):
    """conv3d_ll (Conv_0 + halved LL band as one convolution) with Winograd F(4,3) along the bands on top
    (tmdiff_conv3d_wfll_fwd): x_s2d is the producer's space-to-depth second output [B, 4 Cin, N, H/2, W/2]
    (conv3d_wf(..., emit=dict(..., s2d=True))), w_packed from pack_conv_weight_wfll; result and conventions as conv3d_ll."""
    b, c4, n, h2, w2 = x_s2d.shape
    dev = x_s2d.device
    oshape = (b, cout, n, h2, w2)
    y = torch.empty(oshape, device=dev, dtype=torch.float32) if keep_y else None
    y2 = None
    if emit is not None:
        y2 = torch.empty(oshape, device=dev, dtype=torch.float32)
        kw = dict(kw, y2_act=emit.get("act", False), y2_shift=emit.get("shift"), y2_scale=emit.get("scale"),
                  y2_shift_stride=emit.get("shift_stride", 0), y2_scale_stride=emit.get("scale_stride", 0))
    elif y is None:
        raise ValueError("conv3d_wf_ll: keep_y=False needs emit=")
    d = make_conv_desc([x_s2d], w_packed, cout, 3, y, y2=y2, out_div=2, x_s2d=True, **kw)
    if not lib.tmdiff_conv3d_wfll_supported(C.byref(d)):
        raise ValueError("conv3d_wf_ll: shape not supported")
    nsk = lib.tmdiff_conv3d_wfll_splitk_workspace_bytes(C.byref(d)) if config.wf_splitk else 0
    if nsk:
        d.splitk_ws, d.splitk_ws_bytes = _workspace(dev, nsk, "splitk").data_ptr(), nsk
    ret = y if y2 is None else ((y, y2) if y is not None else y2)
    # EXECUTED flops: 16 taps x 6 planes per tile of four bands = 24 multiply-adds per (ci, co, output position)
    fl = 2.0 * b * cout * d.Cin * 24 * n * h2 * w2
    _count("conv3d_wfll_fwd", fl)
    if TIMER is None:
        check(lib.tmdiff_conv3d_wfll_fwd(C.byref(d), float(ll_scale), stream_ptr()), "conv3d_wfll_fwd")
        return ret
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(lib.tmdiff_conv3d_wfll_fwd(C.byref(d), float(ll_scale), stream_ptr()), "conv3d_wfll_fwd")
    e1.record()
    TIMER.records.append((e0, e1, fl, 3, "conv3d_wfll_fwd", _tag(d)))
    return ret


def stem(w, bias, out_channels, xin=None, pan=None, ms=None, silu=True, out=None, out_scale=None, out_scale_stride=0,
         pack_bf16=False):
    """out_scale (tensor [B, C0] or a raw pointer into a projection bank, row stride out_scale_stride): the consumer's
    modulation folded into the stem's output.  pack_bf16: the result as packed bf16 units [B, C0/8, N*H*W, 8] (int16
    storage), the input form of a bf16 convolution (x_bf16_shape=(N, H, W))."""
    ref = ms if ms is not None else xin
    b, n, h, wd = ref.shape
    osc = out_scale if isinstance(out_scale, int) else _chk(out_scale, "out_scale")
    if pack_bf16:
        units = torch.empty(b, out_channels // 8, n * h * wd, 8, device=ref.device, dtype=torch.int16)
        check(lib.tmdiff_stem_fwd_pack_bf16(_chk(xin, "xin"), _chk(pan, "pan"), _chk(ms, "ms"), _chk(w, "w"), _chk(bias, "bias"),
                                            osc, out_scale_stride, units.data_ptr(), b, out_channels, n, h, wd,
                                            1 if silu else 0, stream_ptr()), "stem_fwd_pack_bf16")
        return units
    y = out if out is not None else torch.empty(b, out_channels, n, h, wd, device=ref.device, dtype=torch.float32)
    osc = out_scale if isinstance(out_scale, int) else _chk(out_scale, "out_scale")
    check(lib.tmdiff_stem_fwd_scaled(_chk(xin, "xin"), _chk(pan, "pan"), _chk(ms, "ms"), _chk(w, "w"), _chk(bias, "bias"),
                                     osc, out_scale_stride, _chk(y, "y"), b, out_channels, n, h, wd, 1 if silu else 0,
                                     stream_ptr()), "stem_fwd")
    return y


def head(x, w, scale, scale_stride=0, out=None):
    b, c, n, h, wd = x.shape
    y = out if out is not None else torch.empty(b, n, h, wd, device=x.device, dtype=torch.float32)
    check(lib.tmdiff_head_fwd(_chk(x, "x"), _chk(w, "w"), scale if isinstance(scale, int) else _chk(scale, "scale"),
                              scale_stride, _chk(y, "y"), b, c, n * h * wd, stream_ptr()), "head_fwd")
    return y


def _plane_prologue(pro, channels, n_per_channel):
    """pro = dict(act=, shift=, scale=, shift_stride=, scale_stride=) (tensors or raw bank pointers) -> ctypes struct"""
    if pro is None:
        return None
    q = _lib.PlanePrologue()
    q.shift = pro.get("shift") if isinstance(pro.get("shift"), int) else _chk(pro.get("shift"), "prologue shift")
    q.scale = pro.get("scale") if isinstance(pro.get("scale"), int) else _chk(pro.get("scale"), "prologue scale")
    q.shift_stride, q.scale_stride = pro.get("shift_stride", 0), pro.get("scale_stride", 0)
    q.C, q.n_per_channel, q.act = channels, n_per_channel, 1 if pro.get("act") else 0
    return C.byref(q)


def haar_dwt2d(x, want_high=True, ll_scale=1.0, hi_scale=1.0, outs=None, ll_prologue=None, pack_bf16=False):
    """x [..., H, W] -> (ll, lh, hl, hh); the high bands are None when want_high is False.
    ll_prologue (x must then be [B, C, N, H, W]): the consumer convolution's prologue applied to the LL band as it is
    written -- dict(act=, shift=, scale=, shift_stride=, scale_stride=), per (b, c)."""
    hh_, ww = x.shape[-2:]
    planes = x.numel() // (hh_ * ww)
    shape = (*x.shape[:-2], hh_ // 2, ww // 2)
    if pack_bf16:     # bf16 mode: LL as packed bf16 units [B, C/8, N*h*w, 8] (int16 storage), high bands fp32
        b, c, n = x.shape[:3]
        units = torch.empty(b, c // 8, n * (hh_ // 2) * (ww // 2), 8, device=x.device, dtype=torch.int16)
        hi = [torch.empty(shape, device=x.device, dtype=torch.float32) for _ in range(3)] if want_high else [None] * 3
        pro = _plane_prologue(ll_prologue, c, n) if ll_prologue is not None else None
        check(lib.tmdiff_haar_dwt2d_pack_bf16(_chk(x, "x"), units.data_ptr(), *[_chk(t, "band") for t in hi], b, c, n, hh_, ww,
                                              ll_scale, hi_scale, pro, stream_ptr()), "haar_dwt2d_pack_bf16")
        return (units, *hi)
    if outs is None:
        outs = [torch.empty(shape, device=x.device, dtype=torch.float32) for _ in range(4 if want_high else 1)]
    ptrs = [_chk(o, "band") for o in outs] + [None] * (4 - len(outs))
    pro = _plane_prologue(ll_prologue, x.shape[1], x.shape[2]) if ll_prologue is not None else None
    check(lib.tmdiff_haar_dwt2d_pro(_chk(x, "x"), *ptrs, planes, hh_, ww, ll_scale, hi_scale, pro, stream_ptr()),
          "haar_dwt2d")
    return tuple(outs) + (None,) * (4 - len(outs))


def haar_idwt2d(lls, lh, hl, hh, in_scale=1.0, outs=None, stacked_bands=None, out0_prologue=None, pack_bf16=False):
    """lls: list of 1 or 2 low bands sharing the high bands; returns a list of reconstructions.
    ``stacked_bands`` [B, 3C, N, h, w] (the convH_0 output) supplies lh/hl/hh as channel slices
    without copying them out.  out0_prologue (lls[0] must then be [B, C, N, h, w]): the consumer convolution's prologue
    applied to the first reconstruction as it is written (see haar_dwt2d)."""
    ll0 = lls[0]
    h, w = ll0.shape[-2:]
    planes = ll0.numel() // (h * w)
    if pack_bf16:     # bf16 mode: first reconstruction as packed bf16 units, second one fp32; stacked bands only
        if len(lls) != 2 or stacked_bands is None:
            raise ValueError("haar_idwt2d(pack_bf16=True) takes two low bands and stacked high bands")
        b, c, n = ll0.shape[:3]
        units = torch.empty(b, c // 8, n * 4 * h * w, 8, device=ll0.device, dtype=torch.int16)
        out1 = torch.empty(b, c, n, 2 * h, 2 * w, device=ll0.device, dtype=torch.float32)
        pro = _plane_prologue(out0_prologue, c, n) if out0_prologue is not None else None
        check(lib.tmdiff_haar_idwt2d_pack_bf16(_chk(ll0, "ll0"), _chk(lls[1], "ll1"), _chk(stacked_bands, "bands"),
                                               units.data_ptr(), out1.data_ptr(), b, c, n, h, w, in_scale, pro, stream_ptr()),
              "haar_idwt2d_pack_bf16")
        return [units, out1]
    if outs is None:
        outs = [torch.empty((*ll0.shape[:-2], 2 * h, 2 * w), device=ll0.device, dtype=torch.float32) for _ in lls]
    llp = (C.c_void_p * 2)(*[_chk(t, "ll") for t in lls], *([None] * (2 - len(lls))))
    outp = (C.c_void_p * 2)(*[_chk(t, "out") for t in outs], *([None] * (2 - len(outs))))
    if stacked_bands is not None:
        b = stacked_bands.shape[0]
        ppb = planes // b
        if stacked_bands.numel() != 3 * planes * h * w:
            raise ValueError("haar_idwt2d: stacked bands must be [B, 3C, N, h, w]")
        base = _chk(stacked_bands, "bands")
        step = 4 * ppb * h * w
        hp = (base, base + step, base + 2 * step, ppb, 3 * ppb * h * w)
    else:
        hp = (_chk(lh, "lh"), _chk(hl, "hl"), _chk(hh, "hh"), 0, 0)
    pro = _plane_prologue(out0_prologue, ll0.shape[1], ll0.shape[2]) if out0_prologue is not None else None
    check(lib.tmdiff_haar_idwt2d_pro(llp, len(lls), *hp, outp, planes, h, w, in_scale, pro, stream_ptr()), "haar_idwt2d")
    return outs


def linear(x, w, bias=None, act=False, out=None):
    b, i = x.shape
    o = w.shape[0]
    y = out if out is not None else torch.empty(b, o, device=x.device, dtype=torch.float32)
    check(lib.tmdiff_linear_fwd(_chk(x, "x"), _chk(w, "w"), _chk(bias, "bias"), _chk(y, "y"), b, i, o,
                                1 if act else 0, stream_ptr()), "linear_fwd")
    return y


def gamma_embedding(t, freqs, dim, out=None):
    b = t.shape[0]
    y = out if out is not None else torch.empty(b, dim, device=t.device, dtype=torch.float32)
    check(lib.tmdiff_gamma_embedding(_chk(t, "t"), _chk(freqs, "freqs"), _chk(y, "emb"), b, dim, stream_ptr()),
          "gamma_embedding")
    return y


def ddpm_step(x, eps, noise, c_recip, c_recipm1, coef1, coef2, sigma, clip=True, ms=None, out=None, img_out=None):
    y = out if out is not None else torch.empty_like(x)
    check(lib.tmdiff_ddpm_step(_chk(x, "x"), _chk(eps, "eps"), _chk(noise, "noise"), _chk(ms, "ms"), _chk(y, "out"),
                               _chk(img_out, "img_out"), x.numel(), c_recip, c_recipm1, coef1, coef2, sigma,
                               1 if clip else 0, stream_ptr()), "ddpm_step")
    return y


def axpby(tensors, coefs, out=None):
    n_in = len(tensors)
    y = out if out is not None else torch.empty_like(tensors[0])
    ptrs = (C.c_void_p * 4)(*[_chk(t, "in") for t in tensors], *([None] * (4 - n_in)))
    cf = (C.c_float * 4)(*[float(c) for c in coefs], *([0.0] * (4 - n_in)))
    check(lib.tmdiff_axpby(ptrs, cf, n_in, _chk(y, "out"), y.numel(), stream_ptr()), "axpby")
    return y


class MultiAxpby:
    """out_t = ca * a_t + cb * b_t over a fixed list of (out, a, b) tensor triples in ONE launch (tmdiff_multi_axpby).
    The device tables are built once from the tensors' addresses; ``stale()`` says whether they moved."""

    def __init__(self, triples):
        triples = [(o, a, b) for o, a, b in triples if o.numel() > 0]
        for o, a, b in triples:
            for t in (o, a, b):
                _chk(t, "multi_axpby tensor")
            if not (o.numel() == a.numel() == b.numel()):
                raise ValueError("multi_axpby: out / a / b sizes differ")
        dev = triples[0][0].device
        self.key = tuple(t.data_ptr() for tr in triples for t in tr)
        chunk = lib.tmdiff_multi_axpby_chunk()
        ent, ct, ci = [], [], []
        for k, (o, a, b) in enumerate(triples):
            ent += [o.data_ptr(), a.data_ptr(), b.data_ptr(), o.numel()]
            nck = (o.numel() + chunk - 1) // chunk
            ct += [k] * nck
            ci += list(range(nck))
        self.entries = torch.tensor(ent, dtype=torch.int64).to(dev)           # tmdiff_mt_entry[]: 3 pointers + int64
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32).to(dev)
        self.chunk_index = torch.tensor(ci, dtype=torch.int32).to(dev)
        self.n_chunks = len(ct)
        self._keep = triples

    def stale(self, triples):
        return self.key != tuple(t.data_ptr() for tr in triples for t in tr if tr[0].numel() > 0)

    def run(self, ca, cb):
        check(lib.tmdiff_multi_axpby(self.entries.data_ptr(), self.chunk_tensor.data_ptr(), self.chunk_index.data_ptr(),
                                     self.n_chunks, float(ca), float(cb), stream_ptr()), "multi_axpby")


def x0_from_model(x, model_out, alpha, sigma, model_is_x_start=True, out=None):
    y = out if out is not None else torch.empty_like(x)
    check(lib.tmdiff_x0_from_model(_chk(x, "x"), _chk(model_out, "model_out"), _chk(y, "x0"), x.numel(), alpha, sigma,
                                   1 if model_is_x_start else 0, stream_ptr()), "x0_from_model")
    return y


def abs_quantile_clamp_(x0, q=0.995, max_val=1.0):
    """In place: per sample s = max(quantile(|x0|, q), max_val); x0 = clamp(x0, -s, s) / s.  Returns s [B]."""
    b = x0.shape[0]
    n = x0.numel() // b
    ws = torch.empty(max(1, lib.tmdiff_abs_quantile_workspace_bytes(b, n) // 4), device=x0.device, dtype=torch.float32)
    check(lib.tmdiff_abs_quantile_clamp(_chk(x0, "x0"), b, n, q, max_val, ws.data_ptr(), stream_ptr()),
          "abs_quantile_clamp")
    return ws[:b]


def add(a, b, sign_b=1.0, out=None):
    y = out if out is not None else torch.empty_like(a)
    check(lib.tmdiff_add(_chk(a, "a"), _chk(b, "b"), _chk(y, "out"), a.numel(), sign_b, stream_ptr()), "add")
    return y


def q_sample(x0, noise, a, out=None):
    b = x0.shape[0]
    y = out if out is not None else torch.empty_like(x0)
    check(lib.tmdiff_q_sample(_chk(x0, "x0"), _chk(noise, "noise"), _chk(a, "a"), _chk(y, "out"), b,
                              x0.numel() // b, stream_ptr()), "q_sample")
    return y


# ---- backward-side wrappers (finetune path) -----------------------------------------------------------


def wgrad_wino_takes(desc):
    """True when conv3d_wgrad runs this weight gradient in the Winograd domain (its g pass then sums the bias gradient on
    the side for free)."""
    return bool(config.wgrad_wino and desc.ksize == 3 and lib.tmdiff_conv3d_wgrad_wino_supported(C.byref(desc)))


def conv3d_wgrad(desc, g, weight_shape, want_bias=False):
    """dL/dw [Cout, Cin/g, k,k,k] for the convolution described by `desc` (a filled Conv3dDesc) given g = dL/dy; with
    want_bias also dL/dbias = desc.bias_scale * sum_{b,pos} g, accumulated inside the same kernel: returns (dw, dbias)."""
    dw = torch.empty(weight_shape, device=g.device, dtype=torch.float32)
    if config.wgrad_wino and desc.ksize == 3 and lib.tmdiff_conv3d_wgrad_wino_supported(C.byref(desc)):
        # Winograd F(3,4) along the bands (csrc/wgrad_wino.hip): 13.5 executed multiply-adds per element instead of 27
        _count("conv3d_wgrad_wino", 2.0 * desc.B * desc.Cout * (desc.Cin // desc.groups) * 13.5 * desc.N * desc.H * desc.W)
        nbytes = lib.tmdiff_conv3d_wgrad_wino_workspace_bytes(C.byref(desc))
        ws = _workspace(g.device, max(16, nbytes), "wgrad")
        db = torch.empty(weight_shape[0], device=g.device, dtype=torch.float32) if want_bias else None
        check(lib.tmdiff_conv3d_wgrad_wino_bias(C.byref(desc), _chk(g, "g"), dw.data_ptr(), _chk(db, "db"), ws.data_ptr(),
                                                stream_ptr()), "conv3d_wgrad_wino")
        return (dw, db) if want_bias else dw
    _count("conv3d_wgrad", 2.0 * desc.B * desc.Cout * (desc.Cin // desc.groups) * desc.ksize ** 3 * desc.N * desc.H * desc.W)
    nbytes = lib.tmdiff_conv3d_wgrad_workspace_bytes(C.byref(desc))
    ws = _workspace(g.device, max(4, nbytes), "wgrad")
    if not want_bias:
        check(lib.tmdiff_conv3d_wgrad(C.byref(desc), _chk(g, "g"), dw.data_ptr(), ws.data_ptr(), stream_ptr()),
              "conv3d_wgrad")
        return dw
    db = torch.empty(weight_shape[0], device=g.device, dtype=torch.float32)
    check(lib.tmdiff_conv3d_wgrad_bias(C.byref(desc), _chk(g, "g"), dw.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                       stream_ptr()), "conv3d_wgrad_bias")
    return dw, db


def channel_sum(x, scale=1.0):
    """x [B, C, ...] -> [C]: scale * sum over batch and positions."""
    b, c = x.shape[:2]
    out = torch.empty(c, device=x.device, dtype=torch.float32)
    check(lib.tmdiff_channel_sum(_chk(x, "x"), out.data_ptr(), b, c, x.numel() // (b * c), scale, stream_ptr()),
          "channel_sum")
    return out


def conv3d_prologue_bwd(desc, gp, dx_segs, accumulate, want_shift, want_scale, add_segs=None):
    """dL/dx (per segment, into dx_segs; accumulate[i]: added to what dx_segs[i] holds), dL/dshift, dL/dscale of a convolution's
    prologue from dL/dx'.  add_segs (then accumulate must be all False): dx_segs[i] = add_segs[i] + dL/dx_i -- another consumer's
    gradient of the same segment, only read (tmdiff_conv3d_prologue_bwd_add)."""
    b, cin = desc.B, desc.Cin
    d_shift = torch.empty(b, cin, device=gp.device, dtype=torch.float32) if want_shift else None
    d_scale = torch.empty(b, cin, device=gp.device, dtype=torch.float32) if want_scale else None
    dxp = (C.c_void_p * 3)(*[_chk(t, "dx") for t in dx_segs], *([None] * (3 - len(dx_segs))))
    acc = (C.c_int32 * 3)(*[1 if a else 0 for a in accumulate], *([0] * (3 - len(accumulate))))
    nws = lib.tmdiff_conv3d_prologue_bwd_workspace_bytes(C.byref(desc))
    ws = _workspace(gp.device, nws, "prologue_bwd").data_ptr() if nws else None
    if add_segs is not None:
        if any(accumulate):
            raise ValueError("conv3d_prologue_bwd: add_segs goes with accumulate all False")
        addp = (C.c_void_p * 3)(*[_chk(t, "add") for t in add_segs], *([None] * (3 - len(add_segs))))
        check(lib.tmdiff_conv3d_prologue_bwd_add(C.byref(desc), _chk(gp, "gp"), dxp, addp, _chk(d_shift, "d_shift"),
                                                 _chk(d_scale, "d_scale"), ws, stream_ptr()), "conv3d_prologue_bwd_add")
        return d_shift, d_scale
    check(lib.tmdiff_conv3d_prologue_bwd_ws(C.byref(desc), _chk(gp, "gp"), dxp, acc, _chk(d_shift, "d_shift"),
                                            _chk(d_scale, "d_scale"), ws, stream_ptr()), "conv3d_prologue_bwd")
    return d_shift, d_scale


def stem_bwd(w, bias, gy, xin=None, pan=None, ms=None):
    b, c0, n, h, wd = gy.shape
    dwb = torch.empty(b, c0, 2, device=gy.device, dtype=torch.float32)
    check(lib.tmdiff_stem_bwd(_chk(xin, "xin"), _chk(pan, "pan"), _chk(ms, "ms"), _chk(w, "w"), _chk(bias, "bias"),
                              _chk(gy, "gy"), dwb.data_ptr(), b, c0, n, h, wd, stream_ptr()), "stem_bwd")
    return dwb


def stem_bwd_input(w, bias, gy, xin=None, pan=None, ms=None, need_x=True, need_pan=False):
    """Gradients w.r.t. the stem's inputs: (d_xin, None) or (d_ms, d_pan)."""
    b, c0, n, h, wd = gy.shape
    dx = torch.empty(b, n, h, wd, device=gy.device, dtype=torch.float32) if need_x else None
    dpan = torch.empty(b, 1, h, wd, device=gy.device, dtype=torch.float32) if (need_pan and ms is not None) else None
    check(lib.tmdiff_stem_bwd_input(_chk(xin, "xin"), _chk(pan, "pan"), _chk(ms, "ms"), _chk(w, "w"), _chk(bias, "bias"),
                                    _chk(gy, "gy"), _chk(dx, "dx"), _chk(dpan, "dpan"), b, c0, n, h, wd, stream_ptr()),
          "stem_bwd_input")
    return dx, dpan


def head_bwd(x, w, scale, gy, need_dx=True):
    b, c, n, h, wd = x.shape
    dx = torch.empty_like(x) if need_dx else None
    dws = torch.empty(b, c, device=x.device, dtype=torch.float32)
    check(lib.tmdiff_head_bwd(_chk(x, "x"), _chk(w, "w"), _chk(scale, "scale"), _chk(gy, "gy"), _chk(dx, "dx"),
                              dws.data_ptr(), b, c, n * h * wd, stream_ptr()), "head_bwd")
    return dx, dws


def linear_bwd(x, w, bias, gy, act=False, need_dx=True, need_dw=True, need_db=True):
    b, i = x.shape
    o = w.shape[0]
    dx = torch.empty(b, i, device=x.device, dtype=torch.float32) if need_dx else None
    dw = torch.empty(o, i, device=x.device, dtype=torch.float32) if need_dw else None
    db = torch.empty(o, device=x.device, dtype=torch.float32) if need_db else None
    gu = torch.empty(b, o, device=x.device, dtype=torch.float32) if act else None
    check(lib.tmdiff_linear_bwd(_chk(x, "x"), _chk(w, "w"), _chk(bias, "bias"), _chk(gy, "gy"), _chk(gu, "gu"),
                                _chk(dx, "dx"), _chk(dw, "dw"), _chk(db, "db"), b, i, o, 1 if act else 0,
                                stream_ptr()), "linear_bwd")
    return dx, dw, db


# ---- standalone attention operators (core/Attention.py) ----------------------------------------------------
def attention(q, k, v, scale, heads=1, key_mask=None, layout="bnd"):
    """softmax(q k^T * scale) v.  layout 'bnd': q [B, Nq, H*D], k/v [B, Nk, H*D] (heads split along the last
    dim, as CrossAttention's 'b n (h d)'); returns [B, Nq, H*D]."""
    assert layout == "bnd"
    b, nq, hd = q.shape
    nk = k.shape[1]
    d = hd // heads
    out = torch.empty_like(q)
    st = lambda n: (C.c_int64 * 3)(n * hd, d, hd)          # batch, head, row strides in elements
    m = None
    if key_mask is not None:
        m = key_mask.to(device=q.device, dtype=torch.uint8).contiguous()
    check(lib.tmdiff_attn_fwd(_chk(q, "q"), _chk(k, "k"), _chk(v, "v"), out.data_ptr(),
                              m.data_ptr() if m is not None else None, b, heads, nq, nk, d, st(nq), st(nk), st(nk),
                              st(nq), scale, stream_ptr()), "attn_fwd")
    return out


def gemm_nt(a, w, bias=None, residual=None):
    """a [..., K] @ w[N, K]^T + bias + residual -> [..., N]"""
    k = a.shape[-1]
    m = a.numel() // k
    n = w.shape[0]
    out = torch.empty(*a.shape[:-1], n, device=a.device, dtype=torch.float32)
    check(lib.tmdiff_gemm_nt(_chk(a, "a"), _chk(w, "w"), _chk(bias, "bias"), _chk(residual, "residual"),
                             out.data_ptr(), m, n, k, stream_ptr()), "gemm_nt")
    return out


def group_norm(x, gamma, beta, groups=32, eps=1e-6):
    b, c = x.shape[:2]
    y = torch.empty_like(x)
    check(lib.tmdiff_group_norm(_chk(x, "x"), _chk(gamma, "gamma"), _chk(beta, "beta"), y.data_ptr(), b, c,
                                x.numel() // (b * c), groups, eps, stream_ptr()), "group_norm")
    return y


def layer_norm(x, gamma, beta, eps=1e-5):
    d = x.shape[-1]
    y = torch.empty_like(x)
    check(lib.tmdiff_layer_norm(_chk(x, "x"), _chk(gamma, "gamma"), _chk(beta, "beta"), y.data_ptr(),
                                x.numel() // d, d, eps, stream_ptr()), "layer_norm")
    return y


def geglu(u, gelu_only=False):
    inner = u.shape[-1] if gelu_only else u.shape[-1] // 2
    y = torch.empty(*u.shape[:-1], inner, device=u.device, dtype=torch.float32)
    check(lib.tmdiff_geglu(_chk(u, "u"), y.data_ptr(), u.numel() // u.shape[-1], inner, 1 if gelu_only else 0,
                           stream_ptr()), "geglu")
    return y
