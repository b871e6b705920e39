"""Import shims that let the *reference* (/root/reference, read-only) be imported in
the build container, where four third-party modules it touches are not installed.

Used ONLY by ``make_golden.py`` (fixture generation) and never on the GPU box --
/root/reference does not exist there.  Nothing here is reference source: every
stand-in is a few lines written for this repo and replaces a *missing third-party
dependency*, not reference code (SURVEY.md 8c lists them):

  pywt            only ``Wavelet('haar').{dec_lo,dec_hi,rec_lo,rec_hi}`` is read
                  (DWT_IDWT/DWT_IDWT_layer.py:262-264, :352-356); values are the
                  published PyWavelets Haar taps +-1/sqrt(2).
  ml_collections  ``ConfigDict`` as an attribute dict (config/sample_config.py).
  torchvision,cv2 imported by utils/util.py at module scope, untouched on this path.
  core.clip       ``FrozenCLIPEmbedder`` needs CLIP weights that are not in the image;
                  the stand-in returns the same seeded [1,768] vectors the oracle and
                  the build use (``unet_ref.synthetic_text_embeddings``), keyed by the
                  prompt text's first words.
Plus: ``torch.Tensor.to("cuda")`` -> no-op on this GPU-less box (Hyper_unet_general.py:602).
"""
import math
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"

_PROMPT_KEYS = {  # first words of the paragraphs at Hyper_unet_general.py:574-585 -> prompt name
    "The QuickBird": "QB", "The WorldView-3": "WV3", "The WorldView-4": "WV4",
}


def _prompt_name(text: str) -> str:
    for head, name in _PROMPT_KEYS.items():
        if text.startswith(head):
            return name
    # "GF2" and "WV2" both start with "The GaoFen-2" (reference quirk :579-582); resolutions differ
    return "GF2" if "1.0-meter" in text else "WV2"


def install(text_embeddings):
    """Put the stand-ins into sys.modules and the reference root on sys.path."""
    s = 1.0 / math.sqrt(2.0)

    pywt = types.ModuleType("pywt")

    class Wavelet:  # noqa: D401 - minimal stand-in
        def __init__(self, name):
            assert name == "haar", "only the Haar taps are provided"
            self.dec_lo, self.dec_hi = [s, s], [-s, s]
            self.rec_lo, self.rec_hi = [s, s], [s, -s]

    pywt.Wavelet = Wavelet
    sys.modules["pywt"] = pywt

    mlc = types.ModuleType("ml_collections")

    class ConfigDict(dict):
        def __init__(self, initial_dictionary=None):
            super().__init__(initial_dictionary or {})

        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

        def get_ref(self, key):
            return self[key]

    mlc.ConfigDict = ConfigDict
    sys.modules["ml_collections"] = mlc

    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvt.ToTensor = lambda *a, **k: (lambda x: x)
    tvt.RandomHorizontalFlip = lambda *a, **k: (lambda x: x)
    tv.transforms = tvt
    tvu = types.ModuleType("torchvision.utils")
    tvu.make_grid = lambda *a, **k: None
    tv.utils = tvu
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.utils": tvu})
    sys.modules["cv2"] = types.ModuleType("cv2")

    clip = types.ModuleType("core.clip")

    class FrozenCLIPEmbedder(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def encode(self, text):
            return text_embeddings[_prompt_name(text)].clone()

    clip.FrozenCLIPEmbedder = FrozenCLIPEmbedder

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import core  # the reference's package (namespace), so the fake submodule can hang off it
    sys.modules["core.clip"] = clip
    core.clip = clip

    if not torch.cuda.is_available():
        _to = torch.Tensor.to

        def to(self, *args, **kwargs):
            if args and isinstance(args[0], str) and args[0].startswith("cuda"):
                args = ("cpu",) + tuple(args[1:])
            return _to(self, *args, **kwargs)

        torch.Tensor.to = to


def install_metrics_stand_ins():
    """Import stand-ins for the third-party modules core/metrics.py imports at module scope and this image lacks
    (skimage.metrics, sewar; torchvision / cv2 as in install()).  Only ``SAM_numpy`` (core/metrics.py:91-112, pure NumPy) is
    ever EXECUTED through them: the stand-in functions raise if touched, so SSIM / MPSNR (skimage's arithmetic) cannot be
    produced by accident -- they stay unpinned."""
    def absent(name):
        def f(*a, **k):
            raise RuntimeError(f"{name} is a stand-in: the real package is not in this image")
        return f

    sk, skm = types.ModuleType("skimage"), types.ModuleType("skimage.metrics")
    skm.structural_similarity = absent("skimage.metrics.structural_similarity")
    skm.peak_signal_noise_ratio = absent("skimage.metrics.peak_signal_noise_ratio")
    sk.metrics = skm
    sewar = types.ModuleType("sewar")
    for fn in ("ssim", "sam", "scc", "ergas", "psnr"):
        setattr(sewar, fn, absent("sewar." + fn))
    sys.modules.update({"skimage": sk, "skimage.metrics": skm, "sewar": sewar})
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")
    if "torchvision" not in sys.modules:
        tv, tvu = types.ModuleType("torchvision"), types.ModuleType("torchvision.utils")
        tvu.make_grid = absent("torchvision.utils.make_grid")
        tv.utils = tvu
        sys.modules.update({"torchvision": tv, "torchvision.utils": tvu})
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
