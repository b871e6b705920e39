"""CPU-only tests of the product's host side: the C-ABI library loads and exports what the header
declares, the module tree / checkpoint contract, schedule tables, DPM-Solver host maths, factory
options and error behaviour.  No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close
from oracle import unet_ref as U
from oracle.make_golden import TINY


def test_library_exports_every_declared_symbol():
    from tmdiff_amd import _lib
    header = open(os.path.join(ROOT, "include", "tmdiff_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(tmdiff_[a-z0-9_]+)\s*\(", header))
    declared -= {"tmdiff_conv3d_desc"}
    assert len(declared) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} is declared in include/tmdiff_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.ABI_VERSION == 6
    assert ctypes.sizeof(_lib.Conv3dDesc) % 8 == 0


def test_state_dict_contract_and_clip_keys_ignored():
    from tmdiff_amd.Hyper_unet_general import WavBEST
    from tmdiff_amd.diffusion_general import GeneralDiffusion, GaussianDiffusion
    assert GaussianDiffusion is GeneralDiffusion
    ref = U.fill_weights_(U.WavBESTRef(channels=TINY))
    net = WavBEST(channels=TINY)
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    assert len(net.state_dict()) == 272
    for k, v in ref.state_dict().items():
        assert net.state_dict()[k].shape == v.shape, k
    sd = dict(ref.state_dict())
    sd["clip_text_model.transformer.text_model.embeddings.token_embedding.weight"] = torch.zeros(3)
    net.load_state_dict(sd, strict=True)              # reference checkpoints carry the CLIP weights
    diff = GeneralDiffusion(net, "l1")
    diff.set_new_noise_schedule({"schedule": "linear", "n_timestep": 10}, "cpu")
    keys = list(diff.state_dict().keys())
    assert keys[:12] == ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
                         "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
                         "sqrt_recip_alphas_cumprod_1", "sqrt_recipm1_alphas_cumprod_1", "posterior_variance",
                         "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    assert keys[12] == "denoise_fn.embed.0.weight" and len(keys) == 284


def test_schedule_tables_bitwise(golden):
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    g = golden("schedules")
    for sched in ("cosine", "linear"):
        for T in (10, 50, 1000):
            d = GeneralDiffusion(None)
            d.set_new_noise_schedule({"schedule": sched, "n_timestep": T}, "cpu")
            for k, v in d.state_dict().items():
                assert np.array_equal(v.numpy(), g[f"{sched}_{T}_{k}"]), (sched, T, k)
            assert np.array_equal(d.sqrt_alphas_cumprod_prev, g[f"{sched}_{T}_sqrt_alphas_cumprod_prev"])
            assert d.num_timesteps == T and len(d._step_coef) == T
            i = T // 2
            assert d._step_coef[i][0] == float(d.sqrt_recip_alphas_cumprod_1[i])
            assert abs(d._step_coef[i][4] - float((0.5 * d.posterior_log_variance_clipped[i]).exp())) < 1e-12


def test_error_behaviour_matches_reference():
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.Hyper_unet_general import WavBEST
    d = GeneralDiffusion(None, loss_type="huber")
    with pytest.raises(NotImplementedError):
        d.set_loss("cpu")
    with pytest.raises(NotImplementedError):
        d.set_new_noise_schedule({"schedule": "sigmoid", "n_timestep": 10}, "cpu")
    net = WavBEST(channels=TINY)
    x = torch.zeros(1, 4, 16, 16)
    with pytest.raises(RuntimeError, match="HIP kernels only"):     # no CPU fallback behind the product
        net(x, torch.tensor([[1]]), torch.zeros(1, 1, 16, 16), x, "QB")


def test_dpm_solver_host_math(golden):
    from tmdiff_amd.diffusion_general import GeneralDiffusion
    from tmdiff_amd.dpm_solver import DPM_Solver, NoiseScheduleVP
    g = golden("dpm_solver")
    d = GeneralDiffusion(None)
    d.set_new_noise_schedule({"schedule": "cosine", "n_timestep": 1000}, "cpu")
    ns = NoiseScheduleVP("discrete", betas=d.betas)
    tq = torch.tensor(g["ns_t"])
    assert_close(ns.marginal_log_mean_coeff(tq), g["ns_log_alpha"], 1e-6, 1e-6)
    assert_close(ns.marginal_std(tq), g["ns_std"], 1e-6, 1e-6)
    assert_close(ns.marginal_lambda(tq), g["ns_lambda"], 1e-6, 1e-6)
    assert_close(ns.inverse_lambda(ns.marginal_lambda(tq)), g["ns_inv_lambda"], 1e-5, 1e-5)
    solver = DPM_Solver(lambda x, t: x, ns, algorithm_type="dpmsolver++", correcting_x0_fn="dynamic_thresholding")
    for steps in (20, 30, 31, 32):
        outer, orders = solver.get_orders_and_timesteps_for_singlestep_solver(steps, 3, "logSNR", 1.0, 1e-3)
        assert list(orders) == list(g[f"orders_{steps}"])
        assert_close(outer, g[f"grid_{steps}"], 1e-5, 1e-5, "outer grid")
    with pytest.raises(ValueError):
        NoiseScheduleVP("sigmoid")
    with pytest.raises(ValueError):
        solver.get_time_steps("bogus", 1.0, 1e-3, 5)
    for sched in ("linear", "cosine"):                # continuous schedules: lambda round trip
        nc = NoiseScheduleVP(sched)
        t = torch.tensor([0.05, 0.3, 0.9])
        assert_close(nc.inverse_lambda(nc.marginal_lambda(t)), t, 1e-4, 1e-4, sched)


def test_define_general_reads_reference_options():
    from tmdiff_amd import networks
    opt = {"model": {"unet": {"channel_multiplier": TINY}, "diffusion": {"loss_type": "l2"}, "init_type": "orthogonal"},
           "phase": "train", "gpu_ids": None, "distributed": False}
    net = networks.define_General(opt)
    assert net.loss_type == "l2" and net.denoise_fn.channels == TINY
    assert float(net.denoise_fn.down1.conv20.conv20.bias.abs().sum()) == 0.0      # orthogonal init zeroes biases
    with pytest.raises(NotImplementedError):
        networks.init_weights(net, "xavier")
    opt["phase"] = "val"
    emb = {k: torch.ones(1, 768) * i for i, k in enumerate(("QB", "WV3", "GF2", "WV2", "WV4"))}
    opt["model"]["text_embeddings"] = emb
    net = networks.define_General(opt)
    assert float(net.denoise_fn.get_embeding("GF2")[0, 0]) == 2.0
    assert net.denoise_fn.get_embeding("LANDSAT") is None


def test_psnr_and_res2img():
    from tmdiff_amd.util import img2res, psnr, res2img
    a = torch.rand(2, 4, 8, 8)
    b = a + 0.01
    assert abs(psnr(a, b) - 40.0) < 1e-3          # fp32 (a+0.01)-a is 0.01 only to ~1e-6
    assert psnr(a, a) > 200
    assert torch.equal(img2res(res2img(a, b), b), (a + b) - b)


def test_metrics_against_direct_formulas():
    from tmdiff_amd import metrics as M
    g = torch.Generator().manual_seed(0)
    a = torch.rand(24, 20, 4, generator=g)
    b = (a + 0.05 * torch.randn(24, 20, 4, generator=g)).clamp(0, 1)
    want = np.mean([10 * np.log10(1.0 / float(((a[..., c] - b[..., c]) ** 2).mean())) for c in range(4)])
    assert abs(M.mpsnr(a, b) - want) < 1e-6
    ang = torch.arccos((a * b).sum(-1) / a.norm(dim=-1) / b.norm(dim=-1)).double()
    assert abs(M.sam(a, b) - float(ang.mean() * 180 / np.pi)) < 1e-5
    assert M.sam(a, a) < 0.05 and abs(M.ssim(a, a) - 1.0) < 1e-9
    # SSIM: direct sliding-window evaluation of the skimage definition for one band
    x, y = a[..., 0].double().numpy(), b[..., 0].double().numpy()
    vals = []
    for i in range(x.shape[0] - 6):
        for j in range(x.shape[1] - 6):
            p, q = x[i:i + 7, j:j + 7].ravel(), y[i:i + 7, j:j + 7].ravel()
            ux, uy = p.mean(), q.mean()
            vx, vy, vxy = p.var(ddof=1), q.var(ddof=1), np.cov(p, q, ddof=1)[0, 1]
            vals.append((2 * ux * uy + 1e-4) * (2 * vxy + 9e-4) / ((ux * ux + uy * uy + 1e-4) * (vx + vy + 9e-4)))
    assert abs(M.ssim(a[..., :1], b[..., :1]) - np.mean(vals)) < 1e-9
    assert abs(M.mpsnr(a.permute(2, 0, 1), b.permute(2, 0, 1), hwc=False) - want) < 1e-6


def test_sam_against_the_reference_fixture(golden):
    """N2, the pinnable part: tests/golden/metrics.npz holds SAM_numpy (core/metrics.py:91-112) of the REFERENCE on float32
    and float64 H x W x C arrays with zero / identical / parallel spectra among the pixels.  Same dtype, same operations:
    the float32 cases must agree to float32 rounding of the mean (they agree exactly here), float64 likewise."""
    from tmdiff_amd import metrics as M
    g = golden("metrics")
    for tag in ("wv3_f32", "gf2_f32", "wv3_f64"):
        hr, sr = g[f"{tag}_hr"], g[f"{tag}_sr"]
        assert abs(M.sam(sr, hr) - float(g[f"{tag}_sam"])) <= 1e-12 * max(1.0, float(g[f"{tag}_sam"])), tag
        assert abs(M.sam(hr, sr) - float(g[f"{tag}_sam_swapped"])) <= 1e-12, tag
        # tensors and the [C, H, W] layout go through the same arithmetic
        assert M.sam(torch.from_numpy(sr), torch.from_numpy(hr)) == M.sam(sr, hr)
        assert M.sam(np.moveaxis(sr, -1, 0), np.moveaxis(hr, -1, 0), hwc=False) == M.sam(sr, hr)
    # the dtype matters: the float32 fixture evaluated in float64 is a different number (identical spectra give exactly 0
    # there, a rounding-sized angle or NaN -> 0 in float32)
    hr, sr = g["wv3_f32_hr"], g["wv3_f32_sr"]
    assert M.sam(sr.astype(np.float64), hr.astype(np.float64)) != M.sam(sr, hr)


def test_tiling_helpers_roundtrip():
    from tmdiff_amd import tiling as T
    x = torch.arange(2 * 3 * 8 * 12, dtype=torch.float32).reshape(2, 3, 8, 12)
    t = T.split_tiles(x, 4, 6)
    assert t.shape == (8, 3, 4, 6)
    assert torch.equal(t[1], x[0, :, 0:4, 6:12]) and torch.equal(t[6], x[1, :, 4:8, 0:6])     # row-major tile order
    assert torch.equal(T.merge_tiles(t, 2, 2), x)
    img = torch.rand(1, 4, 16, 16)
    q = T.invPatch(img)
    assert q.shape == (4, 4, 8, 8) and torch.equal(q[3], img[0, :, 8:, 8:])                   # reference order
    cube = torch.rand(5, 32, 32)
    assert torch.equal(T.patch_16(T.unpatch_16(cube)), cube) and T.unpatch_16(cube).shape == (16, 5, 8, 8)


def _tiny_clip_dir(path):
    """A local 1-layer random CLIP text model + character-level tokenizer (stands in for clip-vit-large-patch14)."""
    import json
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    chars = [chr(c) for c in range(ord("a"), ord("z") + 1)] + list("0123456789.,-():")
    vocab = {}
    for suffix in ("", "</w>"):
        for ch in chars:
            vocab[ch + suffix] = len(vocab)
    vocab["<|startoftext|>"], vocab["<|endoftext|>"] = len(vocab), len(vocab) + 1
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    CLIPTokenizer(os.path.join(path, "vocab.json"), os.path.join(path, "merges.txt")).save_pretrained(path)
    eos = vocab["<|endoftext|>"]
    cfg = CLIPTextConfig(vocab_size=len(vocab), hidden_size=768, intermediate_size=64, num_hidden_layers=1,
                         num_attention_heads=4, max_position_embeddings=77, bos_token_id=vocab["<|startoftext|>"],
                         eos_token_id=eos, pad_token_id=eos)
    torch.manual_seed(0)
    CLIPTextModel(cfg).eval().save_pretrained(path)


def test_clip_embedder_and_embedding_cache(tmp_path):
    """FrozenCLIPEmbedder (ref core/clip.py:15-59) from a local directory, the five-paragraph cache file, and its
    injection into WavBEST (ref Hyper_unet_general.py:566-598)."""
    from tmdiff_amd import clip
    from tmdiff_amd.prompts import PROMPT_TEXT
    from tmdiff_amd.Hyper_unet_general import WavBEST, PROMPTS
    with pytest.raises(FileNotFoundError):
        clip.FrozenCLIPEmbedder("openai/clip-vit-large-patch14", device="cpu")      # never downloads
    d = str(tmp_path / "clip")
    os.makedirs(d)
    _tiny_clip_dir(d)
    emb = clip.FrozenCLIPEmbedder(d, device="cpu")
    assert not any(p.requires_grad for p in emb.parameters()) and not emb.transformer.training
    z = emb.encode(PROMPT_TEXT["QB"])
    assert z.shape == (1, 768) and torch.equal(z, emb.encode(PROMPT_TEXT["QB"]))
    assert clip.FrozenCLIPEmbedder(d, device="cpu", layer="last").encode("a b").shape == (1, 768)
    assert clip.FrozenCLIPEmbedder(d, device="cpu", layer="hidden", layer_idx=1).encode("a b").shape == (1, 77, 768)
    table = clip.build_text_embeddings(emb)
    assert set(table) == set(PROMPTS) == set(PROMPT_TEXT)
    assert not torch.equal(table["QB"], table["GF2"])           # (77 character-tokens only reach the sensor name here)
    out = str(tmp_path / "emb.pt")
    clip.main(["--clip", d, "--out", out, "--device", "cpu"])
    back = clip.load_text_embeddings(out)
    for k in PROMPTS:
        assert torch.equal(back[k], table[k])
    net = WavBEST([4, 8, 16, 32], text_embeddings=out)
    assert torch.equal(net.get_embeding("WV3"), table["WV3"]) and net.get_embeding("nope") is None
    assert net.get_prompt("WV2").startswith("The GaoFen-2") and net.get_prompt("nope") is None
    torch.save({"QB": table["QB"]}, out)
    with pytest.raises(KeyError):
        clip.load_text_embeddings(out)


def test_val_dataset_writes_mat_and_scores(tmp_path):
    """evaluate.val_dataset (ref driver :126-152) with a stand-in trainer: .mat key/scale/layout and SSIM/SAM."""
    import scipy.io as scio
    from tmdiff_amd import evaluate, metrics

    class Trainer:
        def feed_data(self, d):
            self.d = d

        def test(self, continous=False, prompt="QB"):
            self.prompt = prompt
            self.SR = torch.cat([torch.zeros_like(self.d["HR"]), self.d["HR"] * 1.5 - 0.2])   # stack; last = result

        def get_current_visuals(self):
            return {"SR": self.SR, "HR": self.d["HR"]}

    g = torch.Generator().manual_seed(3)
    loader = [{"HR": torch.rand(1, 4, 16, 16, generator=g)} for _ in range(2)]
    t = Trainer()
    score = evaluate.val_dataset(t, "GF2", loader, str(tmp_path), log=lambda *a: None)
    assert t.prompt == "GF2"
    m = scio.loadmat(os.path.join(str(tmp_path), "GF2", "output_mulExm_1.mat"))["sr"]
    want = evaluate.to_hwc01(loader[1]["HR"] * 1.5 - 0.2)
    assert m.shape == (16, 16, 4) and np.allclose(m, want * 1023.0, rtol=1e-6) and m.min() >= 0 and m.max() <= 1023.0
    hr1 = evaluate.to_hwc01(loader[1]["HR"])
    hr0, sr0 = evaluate.to_hwc01(loader[0]["HR"]), evaluate.to_hwc01(loader[0]["HR"] * 1.5 - 0.2)
    assert abs(score["ssim_GF2"] - (metrics.ssim(hr0, sr0, 1) + metrics.ssim(hr1, want, 1)) / 2) < 1e-12
    assert abs(score["sam_GF2"] - (metrics.sam(hr0, sr0) + metrics.sam(hr1, want)) / 2) < 1e-12
    evaluate.val_dataset(t, "WV3", loader[:1], str(tmp_path), log=lambda *a: None)
    assert np.allclose(scio.loadmat(os.path.join(str(tmp_path), "WV3", "output_mulExm_0.mat"))["sr"], sr0 * 2047.0, rtol=1e-6)


def _write_dataset(path, n, c, h, scale, with_gt=True, seed=0):
    g = np.random.default_rng(seed)
    arrays = {"ms": g.integers(0, scale, (n, c, h // 4, h // 4)).astype(np.float64),
              "lms": g.integers(0, scale, (n, c, h, h)).astype(np.float64),
              "pan": g.integers(0, scale, (n, 1, h, h)).astype(np.float64)}
    if with_gt:
        arrays["gt"] = g.integers(0, scale, (n, c, h, h)).astype(np.float64)
    np.savez(path, **arrays)
    return arrays


def test_config_parse_and_datasets(tmp_path):
    """Option file (ref core/logger.py:20-125) and PanCollection reader (ref data/LRHR_dataset.py:87-133)."""
    from tmdiff_amd import config as Config, data as Data
    cfg = tmp_path / "opt.json"
    cfg.write_text('''{
  "name": "best", // run name
  "phase": "val", "gpu_ids": [0],
  "path": {"log": "logs", "results": "results", "checkpoint": "checkpoint", "resume": "/nowhere/I100000"},
  "datasets": {"train_wv3": {"dataroot": "x.npz", "batch_size": 4, "num_workers": 0, "use_shuffle": true, "data_len": -1},
               "val_WV3": {"dataroot": "y.npz", "data_len": -1}},
  "model": {"beta_schedule": {"train": {"schedule": "cosine", "n_timestep": 1000}, "val": {"schedule": "cosine", "n_timestep": 1000}},
            "unet": {"channel_multiplier": [32, 64, 128, 256]}, "diffusion": {"loss_type": "l1"}, "init_type": "kaiming"},
  "train": {"val_freq": 2000, "save_checkpoint_freq": 2000, "print_freq": 50, "max_iter": 150000, "optimizer": {"lr": 1e-4}}
}''')
    opt = Config.parse(str(cfg), "train", root=str(tmp_path / "exp"))
    assert opt["phase"] == "train" and opt["distributed"] is False and opt["nothing"] is None
    assert opt["path"]["resume"] == "/nowhere/I100000"                       # resume paths are left alone
    assert opt["path"]["results"].startswith(str(tmp_path / "exp" / "best_")) and os.path.isdir(opt["path"]["results"])
    assert opt["model"]["unet"]["missing"] is None and "channel_multiplier" in Config.dict2str(opt)
    dbg = Config.parse(str(cfg), "train", gpu_ids="0,1", debug=True, root=str(tmp_path / "exp"), make_dirs=False)
    assert dbg["name"] == "debug_best" and dbg["distributed"] is True and dbg["gpu_ids"] == [0, 1]
    assert dbg["train"]["val_freq"] == 2 and dbg["model"]["beta_schedule"]["val"]["n_timestep"] == 10
    assert dbg["datasets"]["train_wv3"]["batch_size"] == 2 and dbg["datasets"]["val_WV3"]["data_len"] == 3

    raw = _write_dataset(str(tmp_path / "train_wv3.npz"), 5, 8, 16, 2047)
    ds = Data.LRHRDataset(str(tmp_path / "train_wv3.npz"), data_len=3)
    assert len(ds) == 3 and ds.img_scale == 2047.0
    item = ds[2]
    assert set(item) == {"LR", "PAN", "MS", "HR", "Res"} and item["LR"].shape == (8, 4, 4) and item["PAN"].shape == (1, 16, 16)
    np.testing.assert_allclose(item["HR"].numpy(), raw["gt"][2] / 2047.0, rtol=1e-6)
    np.testing.assert_allclose(item["Res"].numpy(), (raw["gt"][2] - raw["lms"][2]) / 2047.0, rtol=1e-5, atol=1e-7)
    raw2 = _write_dataset(str(tmp_path / "test_gf2_full.npz"), 2, 4, 16, 1023, with_gt=False)
    ds2 = Data.LRHRDataset(str(tmp_path / "test_gf2_full.npz"))
    assert ds2.img_scale == 1023.0 and len(ds2) == 2
    np.testing.assert_allclose(ds2[0]["HR"].numpy(), raw2["lms"][0] / 1023.0, rtol=1e-6)   # no gt: lms stands in
    assert float(ds2[0]["Res"].abs().max()) == 0.0
    loader = Data.create_dataloader(ds, {"batch_size": 2, "use_shuffle": False, "num_workers": 0}, "train_wv3")
    batch = next(Data.get_data_generator(loader))
    assert batch["MS"].shape == (2, 8, 16, 16)
    assert len(Data.create_dataloader(ds, {}, "val_WV3")) == 3


def test_h5_dataset_branch(tmp_path):
    """The PanCollection .h5 reader (reference data/LRHR_dataset.py:87-133) -- runs where h5py exists; this image does
    not ship it, and then the reader must say so instead of failing obscurely."""
    from tmdiff_amd import data as Data
    try:
        import h5py
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            Data.LRHRDataset(str(tmp_path / "train_wv3.h5"))
        pytest.skip("h5py is not installed in this image: .h5 branch exercised only up to its error message")
    g = np.random.default_rng(0)
    arrs = {"gt": g.integers(0, 2047, (3, 8, 16, 16)), "lms": g.integers(0, 2047, (3, 8, 16, 16)),
            "ms": g.integers(0, 2047, (3, 8, 4, 4)), "pan": g.integers(0, 2047, (3, 1, 16, 16))}
    with h5py.File(str(tmp_path / "train_wv3.h5"), "w") as f:
        for k, v in arrs.items():
            f.create_dataset(k, data=v.astype(np.float64))
    ds = Data.LRHRDataset(str(tmp_path / "train_wv3.h5"))
    assert len(ds) == 3 and ds.img_scale == 2047.0
    np.testing.assert_allclose(ds[1]["HR"].numpy(), arrs["gt"][1] / 2047.0, rtol=1e-6)
    np.testing.assert_allclose(ds[1]["Res"].numpy(), (arrs["gt"][1] - arrs["lms"][1]) / 2047.0, rtol=1e-5, atol=1e-7)


def test_driver_dataset_sampling():
    """Per-iteration choice of the training set (ref driver :45-53, :158-160): weights 4 / 4 / 8 per batch."""
    from tmdiff_amd import train
    p = train.dataset_probabilities({"train_qb": 100, "train_gf2": 50, "train_wv3": 25})
    assert abs(p["train_qb"] - 400 / 800) < 1e-12 and abs(p["train_gf2"] - 200 / 800) < 1e-12
    assert train.sample_dataset(p, 0.49) == "train_qb" and train.sample_dataset(p, 0.5) == "train_gf2"
    assert train.sample_dataset(p, 0.7499) == "train_gf2" and train.sample_dataset(p, 0.75) == "train_wv3"
    assert train.sample_dataset(train.dataset_probabilities({"train_wv3": 3}), 0.1) == "train_wv3"


def test_winograd_transform_choice_and_sizes():
    """Host-side rules of the Winograd-along-the-bands convolution (no GPU work): F(4,3) for band counts whose tiles fill
    the kernel's pairs of tiles (8, 12, 16), F(2,3) for the other even counts (a single F(4,3) tile would leave half of
    every workgroup idle), nothing for odd counts; packed weights hold 9 x planes values per (ci, co)."""
    from tmdiff_amd import _lib
    lib = _lib.lib
    assert [lib.tmdiff_conv3d_wino_planes(n) for n in (2, 3, 4, 6, 8, 12, 16)] == [4, 0, 4, 4, 6, 6, 6]
    assert lib.tmdiff_conv3d_wino_packed_bytes(64, 32, 1, 6) == 32 * 54 * 64 * 4
    assert lib.tmdiff_conv3d_wino_packed_bytes(64, 32, 1, 4) == 32 * 36 * 64 * 4
    assert lib.tmdiff_conv3d_wino_packed_bytes(96, 192, 3, 6) == 64 * 54 * 96 * 4        # groups = 3: Cin / groups rows
    assert lib.tmdiff_conv3d_wino_packed_bytes(48, 32, 1, 6) == 0                          # Cout / groups not a multiple of 32
    assert lib.tmdiff_conv3d_wino_packed_bytes(64, 32, 1, 5) == 0                          # planes is 4 or 6
    assert lib.tmdiff_conv3d_ll_packed_bytes(64, 32) == 32 * 48 * 64 * 4


def test_routing_table_of_the_baseline_configs():
    """every 3x3x3 convolution of BASELINE configs[0..4] (B in {1, 8, 32}, 4 / 8 bands, 64^2 / 256^2 planes, both
    widths, fp32 and bf16) walked through tmdiff_amd.routing -- the table committed as profiles/r04_routing_table.txt
    (tools/routing_table.py).  Every family of the product path is reached by a BASELINE case, except the general-shape direct
    kernel ("fused"), which the reference's default widths reach; the transform-pass Winograd families (tmdiff_amd.fallback) are
    reached by none."""
    import collections
    from tmdiff_amd import ops, routing
    assert ops.config.as_dict() == ops.KernelConfig(env={}).as_dict(), "this table is that of the default switches"
    reached = collections.Counter()
    for label, ch, b, n, size, math in routing.BASELINE_CASES:
        rows = routing.unet_table(ch, b, n, size, size, math)
        assert len(rows) == 51, label                              # the 51 3x3x3 convolutions of one forward
        assert all(f in routing.PRODUCT_FAMILIES for _, f in rows), (label, [f for _, f in rows if f not in routing.PRODUCT_FAMILIES])
        reached.update(f for _, f in rows)
    assert set(reached) == set(routing.PRODUCT_FAMILIES) - {"fused"}, dict(reached)
    other = collections.Counter(f for _, ch, b, n, size, math in routing.OTHER_CASES for _, f in routing.unet_table(ch, b, n, size, size, math))
    assert "fused" in other and not (set(other) & set(routing.FALLBACK_FAMILIES)), dict(other)
    # the benchmark workload: nothing but conv3d_wf (pair mode at the 8x8 level) and its composed-LL mode
    bench = collections.Counter(f for _, f in routing.unet_table(routing.FULL, 32, 8, 64, 64))
    assert bench == {"wf": 40, "wf_pair": 8, "wfll": 3}, dict(bench)
    # what DOES reach the fallback module: even band counts other than 4 / 8 on grids that fill the chip
    assert routing.conv3_family(32, 64, 64, 12, 64, 64) == "wino4" and routing.conv3_family(32, 64, 64, 6, 64, 64) == "wino2"
    assert routing.conv3_family(1, 64, 64, 12, 16, 16) in ("staged", "fused")          # ... small grids stay direct
    # the committed table is this code's table
    import subprocess, sys
    path = os.path.join(ROOT, "profiles", "r04_routing_table.txt")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "routing_table.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1000:]
    assert out.stdout == open(path).read(), "profiles/r04_routing_table.txt is stale: python tools/routing_table.py > " + path


def test_kernel_routing_rules_and_support_queries():
    """Host-side routing of the round-3 kernels (no GPU): which launches conv3d_wf takes itself and how it splits its input
    channels (the benchmark's shapes), when the composed Conv_0 + LL convolution runs with Winograd on top, the packed sizes of
    its weights, and the support queries of the Winograd-domain weight gradient."""
    import ctypes as C
    from tmdiff_amd import _lib, ops
    L = _lib.lib
    # benchmark batch (B = 32): every level is taken unsplit or with a split that still fills the chip
    assert ops.wf_route(32, 64, 64, 8, 64, 64) == (True, 1)
    assert ops.wf_route(32, 256, 256, 8, 16, 16) == (True, 1)                 # 512 tiles
    taken, split = ops.wf_route(32, 256, 256, 8, 8, 8)                        # pair mode: 16 pairs x 8 channel tiles = 128
    assert taken and split == 4                                               # (to 256, then once more: ranges of 32 chunks are long)
    assert ops.wf_route(32, 64, 64, 8, 16, 16)[1] == 2                        # 128 tiles of a 64-channel layer: short ranges
    assert ops.wf_route(32, 128, 128, 8, 16, 16)[1] == 2                      # 256 tiles, 32-chunk ranges: split once
    assert ops.wf_route(8, 128, 128, 8, 32, 32)[1] == 2 and ops.wf_route(8, 64, 64, 8, 32, 32)[1] == 2
    # a single image at the 8x8 level is half a pair: the fallback kernels; odd band counts and masks too
    assert not ops.wf_route(1, 256, 256, 8, 8, 8)[0]
    assert not ops.wf_route(32, 64, 64, 6, 64, 64)[0] and not ops.wf_route(32, 64, 64, 8, 64, 64, masked=True)[0]
    # composed Conv_0 + LL with Winograd on top: 8 bands, W % 8 == 0, Cout % 32 == 0
    assert ops.wfll_route(32, 64, 64, 8, 64, 64) and ops.wfll_route(32, 256, 256, 8, 16, 16)
    assert ops.wfll_route(32, 64, 64, 4, 64, 64) and not ops.wfll_route(32, 64, 64, 6, 64, 64) and not ops.wfll_route(32, 64, 48, 8, 64, 64)
    assert L.tmdiff_conv3d_wfll_packed_bytes(64, 32) == 32 * 96 * 64 * 4 and L.tmdiff_conv3d_wfll_packed_bytes(48, 32) == 0
    # Winograd-domain weight gradient: N % 4 == 0, W % 4 == 0, 3x3x3
    d = _lib.Conv3dDesc()
    d.B, d.N, d.H, d.W, d.Cin, d.Cout, d.groups, d.ksize, d.nseg = 8, 8, 64, 64, 64, 64, 1, 3, 1
    d.seg_c[0] = 64
    assert L.tmdiff_conv3d_wgrad_wino_supported(C.byref(d)) == 1
    ws = L.tmdiff_conv3d_wgrad_wino_workspace_bytes(C.byref(d))
    xh = 6 * 16 * 66 * 66 * 64 * 4                    # x^: six planes x (b, tile) x padded plane x channels
    gh = 6 * 16 * 64 * 64 * 64 * 4
    assert ws > xh + gh and ws < 2 * (xh + gh)
    for field, bad in (("N", 6), ("W", 62), ("ksize", 1), ("groups", 2)):
        e = _lib.Conv3dDesc.from_buffer_copy(d)
        setattr(e, field, bad)
        assert L.tmdiff_conv3d_wgrad_wino_supported(C.byref(e)) == 0, field


def test_side_prologue_rule_on_the_benchmark_workload():
    """routing.k1_side_xp (which 1x1x1 launches can also write the prologue output of their input: conv1.hip's 16-byte form on
    large grids, the channels-over-the-waves form on small ones): the four three-segment res_conv launches of the benchmark
    workload's up path all qualify -- which is why no prologue pass is left in its step -- and the shapes neither form takes
    do not."""
    from tmdiff_amd import routing
    up_path = [((32, 32, 32), 32, 64, 64), ((64, 64, 64), 32, 32, 32), ((128, 128, 128), 64, 16, 16), ((256, 256, 256), 128, 8, 8)]
    for seg_c, cout, h, w in up_path:
        assert routing.k1_side_xp(32, seg_c, cout, 8, h, w), (seg_c, cout, h, w)
    assert routing.k1_side_xp(1, (256, 256, 256), 128, 8, 8, 8)             # a single tile: still the small-grid form
    assert not routing.k1_side_xp(1, (32, 32, 32), 32, 8, 8, 8)             # a small grid of fewer than 128 input channels
    assert not routing.k1_side_xp(32, (32, 24, 40), 32, 8, 64, 64)          # a segment that is not a multiple of 16 channels
    assert not routing.k1_side_xp(32, (32, 32, 32), 48, 8, 64, 64)          # Cout not a multiple of 32
    assert not routing.k1_side_xp(32, (32, 32, 32), 32, 3, 7, 9)            # a plane that is not a multiple of 4 positions

