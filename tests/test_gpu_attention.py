"""GPU parity of the standalone core/Attention.py operators against the reference fixture and the oracle."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close
from oracle import attention_ref as R
from oracle import unet_ref as U
from oracle.make_golden import randn

pytestmark = pytest.mark.gpu


def cu(t):
    return t.cuda().contiguous()


def pair(ref_cls, hip_cls, *args, seed=3, **kw):
    ref = U.fill_weights_(ref_cls(*args, **kw), seed=seed).eval()
    hip = hip_cls(*args, **kw)
    hip.load_state_dict(ref.state_dict())
    return ref, hip.cuda().eval()


def test_primitives():
    from tmdiff_amd import ops
    for m, n, k in ((130, 70, 33), (512, 256, 128), (77, 128, 768), (5, 3, 2)):
        a, w, b, r = randn(1, m, k), randn(2, n, k) / k ** 0.5, randn(3, n), randn(4, m, n)
        assert_close(ops.gemm_nt(cu(a), cu(w), cu(b), cu(r)).cpu(), F.linear(a, w, b) + r, 1e-5, 1e-5, "gemm_nt")
    x = randn(5, 2, 64, 16, 16)
    g, be = 1 + 0.1 * randn(6, 64), 0.1 * randn(7, 64)
    assert_close(ops.group_norm(cu(x), cu(g), cu(be), 32, 1e-6).cpu(), F.group_norm(x, 32, g, be, 1e-6), 2e-5, 2e-5)
    x = randn(8, 3, 50, 128)
    g, be = 1 + 0.1 * randn(9, 128), 0.1 * randn(10, 128)
    assert_close(ops.layer_norm(cu(x), cu(g), cu(be)).cpu(), F.layer_norm(x, (128,), g, be), 2e-5, 2e-5, "layer_norm")
    u = randn(11, 4, 9, 64)
    a, gate = u.chunk(2, -1)
    assert_close(ops.geglu(cu(u)).cpu(), a * F.gelu(gate), 1e-6, 1e-6, "geglu")
    assert_close(ops.geglu(cu(u), gelu_only=True).cpu(), F.gelu(u), 1e-6, 1e-6, "gelu")
    # (head dims 64 / 128 take the pipelined LDS-DMA kernel, the others the simple one: ragged Nq / Nk for both)
    # (... and d_head 64 with at most 96 keys -- the CLIP context -- the small-context kernel: K / V resident in LDS, single-pass
    #  softmax; 77 / 80 keys (80 key rows in LDS), 96 and 81 (96 rows), one key tile, several query blocks per wave, ragged Nq)
    for b, h, nq, nk, d in ((2, 8, 256, 77, 16), (1, 4, 64, 64, 32), (2, 1, 200, 200, 64), (1, 2, 33, 5, 128), (2, 4, 300, 77, 64),
                            (1, 1, 1024, 1024, 128), (3, 2, 129, 97, 64), (1, 2, 1200, 80, 64), (2, 2, 130, 96, 64),
                            (1, 1, 64, 13, 64), (1, 8, 4096, 77, 64), (1, 3, 1000, 81, 64), (2, 1, 31, 33, 64)):
        q, k, v = randn(12, b, nq, h * d), randn(13, b, nk, h * d), randn(14, b, nk, h * d)
        split = lambda t: t.reshape(t.shape[0], t.shape[1], h, d).permute(0, 2, 1, 3)
        want = F.scaled_dot_product_attention(split(q), split(k), split(v), scale=d ** -0.5)
        want = want.permute(0, 2, 1, 3).reshape(b, nq, h * d)
        assert_close(ops.attention(cu(q), cu(k), cu(v), d ** -0.5, heads=h).cpu(), want, 1e-5, 1e-5, "attention")
        mask = torch.rand(b, nk, generator=torch.Generator().manual_seed(1)) > 0.3
        mask[:, 0] = True
        want = F.scaled_dot_product_attention(split(q), split(k), split(v), attn_mask=mask[:, None, None, :], scale=d ** -0.5)
        want = want.permute(0, 2, 1, 3).reshape(b, nq, h * d)
        assert_close(ops.attention(cu(q), cu(k), cu(v), d ** -0.5, heads=h, key_mask=mask).cpu(), want, 1e-5, 1e-5)


def test_modules_vs_reference_fixture(golden):
    from tmdiff_amd import Attention as A
    g = golden("attention")
    for c, hw in ((64, 8), (128, 16)):
        ref, hip = pair(R.SpatialSelfAttention, A.SpatialSelfAttention, c)
        x = randn(162, 2, c, hw, hw)
        y = hip(cu(x)).cpu()
        assert_close(y, g[f"ssa_c{c}"], 2e-5, 2e-5, "spatial self attention vs reference")
        with torch.no_grad():
            assert_close(y, ref(x), 2e-5, 2e-5, "spatial self attention vs oracle")
    ref, hip = pair(R.CrossAttention, A.CrossAttention, 128, context_dim=768, heads=8, dim_head=16)
    x, ctx = randn(163, 2, 256, 128), randn(164, 2, 77, 768)
    assert_close(hip(cu(x), context=cu(ctx)).cpu(), g["cross"], 2e-5, 2e-5, "cross attention")
    mask = torch.ones(2, 77, dtype=torch.bool); mask[0, 40:] = False; mask[1, 5:9] = False
    assert_close(hip(cu(x), context=cu(ctx), mask=mask).cpu(), g["cross_masked"], 2e-5, 2e-5, "masked cross attention")
    ref, hip = pair(R.CrossAttention, A.CrossAttention, 128, heads=4, dim_head=32)
    assert_close(hip(cu(randn(165, 2, 64, 128))).cpu(), g["self"], 2e-5, 2e-5, "self attention")
    # the reference's default cross-attention shape: 77 context tokens, d_head 64 (the small-context kernel)
    ref, hip = pair(R.CrossAttention, A.CrossAttention, 128, context_dim=768, heads=2, dim_head=64)
    x64 = randn(171, 2, 200, 128)
    assert_close(hip(cu(x64), context=cu(ctx)).cpu(), g["cross_d64"], 2e-5, 2e-5, "cross attention, 77 keys x d_head 64")
    assert_close(hip(cu(x64), context=cu(ctx), mask=mask).cpu(), g["cross_d64_masked"], 2e-5, 2e-5, "... masked")
    ref, hip = pair(R.BasicTransformerBlock, A.BasicTransformerBlock, 128, 8, 16, context_dim=768)
    assert_close(hip(cu(randn(166, 2, 64, 128)), context=cu(randn(167, 2, 77, 768))).cpu(), g["block"], 3e-5, 3e-5)
    ref, hip = pair(R.SpatialTransformer, A.SpatialTransformer, 128, 8, 16, depth=1, context_dim=768)
    y = hip(cu(randn(168, 2, 128, 16, 16)), context=cu(randn(169, 2, 77, 768))).cpu()
    assert_close(y, g["spatial_transformer"], 3e-5, 3e-5, "spatial transformer")
    ref, hip = pair(R.FeedForward, A.FeedForward, 64, glu=True)
    assert_close(hip(cu(randn(170, 3, 10, 64))).cpu(), g["geglu_ff"], 2e-5, 2e-5, "geglu feed-forward")
    ref, hip = pair(R.SpatialTransformer, A.SpatialTransformer, 64, 4, 16, depth=2, context_dim=[32, 32], use_linear=False)
    x, c1 = randn(171, 1, 64, 8, 8), randn(172, 1, 10, 32)
    with torch.no_grad():
        assert_close(hip(cu(x), context=[cu(c1), cu(c1)]).cpu(), ref(x, context=[c1, c1]), 3e-5, 3e-5, "depth-2 transformer")


def test_attnblockpp_vs_reference_fixture(golden):
    """AttnBlockpp / NIN (ref Hyper_unet_general.py:471-515) on the HIP kernels vs the reference fixture and the oracle."""
    from tmdiff_amd import Attention as A
    g = golden("attnpp")
    for tag, (b, c, n, hw, rescale) in {"a": (2, 16, 4, 8, True), "b": (1, 8, 8, 16, False)}.items():
        ref, hip = pair(R.AttnBlockpp, A.AttnBlockpp, c * n, skip_rescale=rescale, seed=5)
        assert set(hip.state_dict()) == set(ref.state_dict())
        x = randn(180, b, c, n, hw, hw)
        y = hip(cu(x)).cpu()
        assert_close(y, g[f"{tag}_y"], 2e-5, 2e-5, f"AttnBlockpp {tag} vs reference")
        with torch.no_grad():
            assert_close(y, ref(x), 2e-5, 2e-5, f"AttnBlockpp {tag} vs oracle")
