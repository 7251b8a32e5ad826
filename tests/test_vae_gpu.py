"""GPU parity of the VAE conv stack (fp32 HIP path through the C ABI) against the CPU restatement of the
diffusers==0.29.0 AutoencoderKL architecture (oracle/vae_ref.py — parity unpinned by the reference, which
ships neither diffusers nor a VAE fixture).

Tolerance: both sides are fp32; the MFMA accumulates k in a different order than the CPU convolution, so
rel-L2 <= 2e-5 per conv and <= 5e-4 through the ~30-layer encoder/decoder; uint8 images may differ by one
level on at most 0.5 % of the pixels (a value landing within float noise of an integer boundary)."""
import importlib

import pytest
import torch
import torch.nn.functional as F

from oracle import vae_ref as VR

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def g(seed):
    return torch.Generator("cpu").manual_seed(seed)


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 8, 64, 8, 32), (1, 3, 128, 16, 16), (3, 20, 70, 7, 37), (1, 128, 128, 32, 32)])
@pytest.mark.parametrize("mode", ["3x3", "3x3s2", "3x3up", "1x1"])
def test_conv_variants(ops, N, Cin, Cout, H, W, mode):
    if mode == "3x3s2" and (H % 2 or W % 2):
        pytest.skip("stride-2 downsample is only applied to even sizes")
    x = torch.randn(N, Cin, H, W, generator=g(1))
    k = 1 if mode == "1x1" else 3
    w = torch.randn(Cout, Cin, k, k, generator=g(2)) / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, generator=g(3))
    if mode == "3x3":
        ref = F.conv2d(x, w, b, padding=1)
        out = ops.conv2d(x.to(DEV), w.to(DEV), b.to(DEV))
    elif mode == "3x3s2":
        ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
        out = ops.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), stride=2)
    elif mode == "3x3up":
        ref = F.conv2d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
        out = ops.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), upsample=True)
    else:
        ref = F.conv2d(x, w, b)
        out = ops.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), ksize=1)
    assert out.shape == ref.shape
    assert rel_l2(out, ref) < 2e-5


def test_conv_groupnorm_silu_prologue_and_residual(ops):
    N, C, H, W, G = 2, 64, 12, 20, 8
    x = torch.randn(N, C, H, W, generator=g(4)) * 2 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(C, generator=g(5)), 0.1 * torch.randn(C, generator=g(6))
    w = torch.randn(96, C, 3, 3, generator=g(7)) / (C * 9) ** 0.5
    b = torch.randn(96, generator=g(8))
    res = torch.randn(N, 96, H, W, generator=g(9))
    st = ops.groupnorm_stats(x.to(DEV), G, 1e-6)
    xr = x.view(N, G, -1)
    assert torch.allclose(st[..., 0].cpu(), xr.mean(-1), atol=1e-5)
    assert torch.allclose(st[..., 1].cpu(), torch.rsqrt(xr.var(-1, unbiased=False) + 1e-6), rtol=1e-4)
    for silu in (1, 0):
        h = F.group_norm(x, G, gamma, beta, 1e-6)
        ref = F.conv2d(F.silu(h) if silu else h, w, b, padding=1) + res
        out = ops.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), resid=res.to(DEV),
                         gn=(st, gamma.to(DEV), beta.to(DEV), G, silu))
        assert rel_l2(out, ref) < 2e-5


def test_attention_products_through_conv(ops):
    """S^T = K^T Q (weights stored transposed, per-image), column softmax, O = V P^T."""
    N, C, H, W = 2, 64, 4, 10
    HW = H * W
    q, k, v = (torch.randn(N, C, H, W, generator=g(10 + i)) for i in range(3))
    st = ops.conv2d(q.to(DEV), k.to(DEV), ksize=1, cout=HW, w_transposed=True, ldw=HW, w_batch_stride=C * HW)
    qf, kf, vf = (t.view(N, C, HW) for t in (q, k, v))
    s_ref = torch.einsum("nck,ncq->nkq", kf, qf)
    assert rel_l2(st.view(N, HW, HW), s_ref) < 2e-5
    ops.col_softmax(st.view(N, HW, HW), 1 / C ** 0.5)
    p_ref = torch.softmax(s_ref / C ** 0.5, dim=1)
    assert rel_l2(st.view(N, HW, HW), p_ref) < 2e-5
    o = ops.conv2d(st, v.to(DEV), ksize=1, cout=C, ldw=HW, w_batch_stride=C * HW)
    assert rel_l2(o.view(N, C, HW), torch.einsum("nck,nkq->ncq", vf, p_ref)) < 5e-5


def _product_vae(cfg, p):
    V = importlib.import_module("video-gpt_amd.vae")
    vae = V.AutoencoderKL(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                          norm_num_groups=cfg.norm_num_groups, scaling_factor=cfg.scaling_factor,
                          shift_factor=cfg.shift_factor)
    vae.load_state_dict(p, strict=True)
    return vae.to(DEV, torch.float32).eval()


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("cfg,hw", [(VR.TINY_VAE, (16, 24)), (VR.VaeCfg(), (64, 64))])
def test_vae_encode_decode_match_oracle(cfg, hw, precision):
    """Same tolerances for both convolution arithmetics: exact fp32 MFMA and split-bf16 ("bf16x3") on the bf16 MFMA."""
    p = VR.make_vae_params(cfg, seed=1)
    vae = _product_vae(cfg, p)
    vae.conv_precision = precision
    f = 2 ** (len(cfg.block_out_channels) - 1)
    x = torch.randn(2, 3, *hw, generator=g(20)).clamp(-1, 1)
    noise = torch.randn(2, 4, hw[0] // f, hw[1] // f, generator=g(21))
    mean, logvar = VR.encode_moments(p, cfg, x)
    dist = vae.encode(x.to(DEV)).latent_dist
    assert rel_l2(dist.parameters, torch.cat([mean, logvar], 1)) < 5e-4
    z_ref = VR.vae_encode(p, cfg, x, noise)
    z = vae.encode_scaled(x.to(DEV), noise.to(DEV), dtype=torch.float32)
    assert rel_l2(z, z_ref) < 5e-4
    img_ref = VR.decode(p, cfg, z_ref / cfg.scaling_factor)
    img = vae.decode(z_ref.to(DEV) / cfg.scaling_factor).sample
    assert img.shape == img_ref.shape and rel_l2(img, img_ref) < 5e-4
    u8_ref = VR.decode_to_uint8(p, cfg, z_ref)
    u8 = vae.decode_to_uint8(z_ref.to(DEV)).cpu()
    diff = (u8.int() - u8_ref.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 5e-3


def test_vae_refuses_cpu():
    V = importlib.import_module("video-gpt_amd.vae")
    with pytest.raises(Exception):
        V.AutoencoderKL(block_out_channels=(32, 64), layers_per_block=1, norm_num_groups=8).decode(torch.zeros(1, 4, 4, 4))


@pytest.mark.parametrize("N,Cin,Cout,H,W,up,gn", [(2, 32, 64, 8, 40, False, False), (1, 128, 96, 20, 33, True, True),
                                                 (1, 4, 70, 7, 9, False, False), (1, 96, 64, 16, 32, False, True)])
def test_conv3x3_split_bf16(ops, N, Cin, Cout, H, W, up, gn):
    """bf16x3 convolution (split operands on the bf16 MFMA) against the fp64 convolution of the same fp32 values:
    error ~1e-5, far inside the 1e-3 of TF32 arithmetic (what cuDNN gives the reference by default)."""
    g_ = torch.Generator("cpu").manual_seed(41)
    x = torch.randn(N, Cin, H, W, generator=g_)
    w = torch.randn(Cout, Cin, 3, 3, generator=g_) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g_)
    xin = x
    gn_arg = None
    if gn:
        groups = 32
        gamma, beta = 1 + 0.1 * torch.randn(Cin, generator=g_), 0.1 * torch.randn(Cin, generator=g_)
        st = ops.groupnorm_stats(x.to(DEV), groups, 1e-6)
        gn_arg = (st, gamma.to(DEV), beta.to(DEV), groups, 1)
        xin = torch.nn.functional.silu(torch.nn.functional.group_norm(x.double(), groups, gamma.double(), beta.double(), 1e-6))
    if up:
        xin = torch.nn.functional.interpolate(xin.double(), scale_factor=2.0, mode="nearest")
    ref = torch.nn.functional.conv2d(xin.double(), w.double(), b.double(), padding=1)
    resid = torch.randn(ref.shape, generator=g_)
    packed = ops.conv_pack_bx3(w.to(DEV))
    out = ops.conv2d_bx3(x.to(DEV), packed, b.to(DEV), resid=resid.to(DEV), gn=gn_arg, upsample=up)
    err = rel_l2(out, ref + resid.double())
    exact = ops.conv2d(x.to(DEV), w.to(DEV), b.to(DEV), resid=resid.to(DEV), gn=gn_arg, upsample=up)
    assert err < 3e-5 and rel_l2(out, exact) < 3e-5, err
