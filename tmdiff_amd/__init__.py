"""tmdiff_amd -- MI355X-native implementation of TMDiff's denoising hot path.

``tmdiff_amd.networks.define_General`` / ``tmdiff_amd.diffusion_general.GeneralDiffusion`` mirror
the reference's ``GeneralModel.networks`` / ``GeneralModel.diffusion_general``; every FLOP-carrying
op runs in ``libtmdiff_hip.so`` (hand-written HIP for gfx950, C ABI in include/tmdiff_hip.h).
Importing the package loads that library and fails loudly if it has not been built.
"""
from . import _lib  # noqa: F401  (raises ImportError when libtmdiff_hip.so is missing)

__version__ = "0.1.0"
