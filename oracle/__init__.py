"""CPU oracle for the TMDiff denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (fp32, CPU) restatement of the reference algorithm:

    unet_ref.py        <- GeneralModel/Hyper_unet_general.py  (WavBEST and its blocks)
    haar_ref.py        <- DWT_IDWT/DWT_IDWT_layer.py + DWT_IDWT_Functions.py (2-D Haar only)
    diffusion_ref.py   <- GeneralModel/diffusion_general.py   (GeneralDiffusion)
    dpm_solver_ref.py  <- core/dpm_solver_pytorch.py          (NoiseScheduleVP, model_wrapper, DPM_Solver)
    attention_ref.py   <- core/Attention.py                   (standalone attention ops)

It is pinned against the real reference (imported in the build container with the
shims in ``ref_shims.py``) by the golden vectors under ``tests/golden/`` that
``make_golden.py`` generated; see DESIGN.md "Oracle".

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``tmdiff_amd/`` imports this package: the product path is the
HIP extension and fails loudly when that extension is missing.
"""
