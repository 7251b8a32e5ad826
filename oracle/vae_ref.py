"""CPU restatement of the VAE the reference uses: diffusers==0.29.0 `AutoencoderKL` with the
stabilityai/sdxl-vae configuration (LVM/pipeline.py:87-93).  TEST INFRASTRUCTURE.

diffusers is NOT in this container and its source is not under /root/reference, so this file is
written from the published architecture of that release — **parity unpinned**: the
reference holds no fixture for it.  Call sites restated: LVM/pipeline.py:110-117 (vae_encode),
:558-590 (decode, clamp, uint8), LVM/utils.py:99-137.

Parameters use diffusers' state_dict keys (encoder.down_blocks.0.resnets.0.norm1.weight, ...,
decoder.up_blocks.3.resnets.2.conv2.bias, quant_conv.*, post_quant_conv.*), so a real
`diffusion_pytorch_model.safetensors` loads into the same dict.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F


@dataclass
class VaeCfg:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: tuple = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.13025
    shift_factor: float = None
    eps: float = 1e-6


TINY_VAE = VaeCfg(block_out_channels=(32, 64), layers_per_block=1, norm_num_groups=8)
TINY_VAE8 = VaeCfg(block_out_channels=(16, 32, 32, 32), layers_per_block=1, norm_num_groups=8)  # /8 like sdxl-vae


def resnet(p, pre, x, groups, eps):
    """ResnetBlock2D: GN -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, 1x1 shortcut when channels change."""
    h = F.silu(F.group_norm(x, groups, p[f"{pre}.norm1.weight"], p[f"{pre}.norm1.bias"], eps))
    h = F.conv2d(h, p[f"{pre}.conv1.weight"], p[f"{pre}.conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, p[f"{pre}.norm2.weight"], p[f"{pre}.norm2.bias"], eps))
    h = F.conv2d(h, p[f"{pre}.conv2.weight"], p[f"{pre}.conv2.bias"], padding=1)
    if f"{pre}.conv_shortcut.weight" in p:
        x = F.conv2d(x, p[f"{pre}.conv_shortcut.weight"], p[f"{pre}.conv_shortcut.bias"])
    return x + h


def mid_attention(p, pre, x, groups, eps):
    """Attention block of UNetMidBlock2D: 1 head of dim C over the H*W positions, residual connection."""
    B, C, H, W = x.shape
    h = F.group_norm(x, groups, p[f"{pre}.group_norm.weight"], p[f"{pre}.group_norm.bias"], eps)
    h = h.view(B, C, H * W).transpose(1, 2)
    q = F.linear(h, p[f"{pre}.to_q.weight"], p[f"{pre}.to_q.bias"])
    k = F.linear(h, p[f"{pre}.to_k.weight"], p[f"{pre}.to_k.bias"])
    v = F.linear(h, p[f"{pre}.to_v.weight"], p[f"{pre}.to_v.bias"])
    a = torch.softmax(q @ k.transpose(1, 2) / (C ** 0.5), dim=-1)
    o = F.linear(a @ v, p[f"{pre}.to_out.0.weight"], p[f"{pre}.to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, W)


def mid_block(p, pre, x, groups, eps):
    x = resnet(p, f"{pre}.resnets.0", x, groups, eps)
    x = mid_attention(p, f"{pre}.attentions.0", x, groups, eps)
    return resnet(p, f"{pre}.resnets.1", x, groups, eps)


def encode_moments(p, cfg: VaeCfg, x):
    """Encoder + quant_conv -> (mean, logvar)."""
    g, eps = cfg.norm_num_groups, cfg.eps
    h = F.conv2d(x, p["encoder.conv_in.weight"], p["encoder.conv_in.bias"], padding=1)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            h = resnet(p, f"encoder.down_blocks.{i}.resnets.{j}", h, g, eps)
        if i != nb - 1:  # Downsample2D: pad (0,1,0,1) then conv3x3 stride 2, padding 0
            h = F.pad(h, (0, 1, 0, 1))
            h = F.conv2d(h, p[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         p[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    h = mid_block(p, "encoder.mid_block", h, g, eps)
    h = F.silu(F.group_norm(h, g, p["encoder.conv_norm_out.weight"], p["encoder.conv_norm_out.bias"], eps))
    h = F.conv2d(h, p["encoder.conv_out.weight"], p["encoder.conv_out.bias"], padding=1)
    h = F.conv2d(h, p["quant_conv.weight"], p["quant_conv.bias"])
    return h.chunk(2, dim=1)


def sample_latent(mean, logvar, noise):
    """DiagonalGaussianDistribution.sample with externally supplied noise."""
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * noise


def vae_encode(p, cfg: VaeCfg, x, noise):
    """LVM/pipeline.py:110-117."""
    z = sample_latent(*encode_moments(p, cfg, x), noise)
    if cfg.shift_factor is not None:
        return (z - cfg.shift_factor) * cfg.scaling_factor
    return z * cfg.scaling_factor


def decode(p, cfg: VaeCfg, z):
    """post_quant_conv + Decoder."""
    g, eps = cfg.norm_num_groups, cfg.eps
    h = F.conv2d(z, p["post_quant_conv.weight"], p["post_quant_conv.bias"])
    h = F.conv2d(h, p["decoder.conv_in.weight"], p["decoder.conv_in.bias"], padding=1)
    h = mid_block(p, "decoder.mid_block", h, g, eps)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(p, f"decoder.up_blocks.{i}.resnets.{j}", h, g, eps)
        if i != nb - 1:  # Upsample2D: nearest x2 then conv3x3
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, p[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         p[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    h = F.silu(F.group_norm(h, g, p["decoder.conv_norm_out.weight"], p["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(h, p["decoder.conv_out.weight"], p["decoder.conv_out.bias"], padding=1)


def decode_to_uint8(p, cfg: VaeCfg, latent):
    """LVM/pipeline.py:572-588: unscale, decode, (x*0.5+0.5).clamp(0,1)*255 -> uint8, NHWC."""
    z = latent.float()
    z = z / cfg.scaling_factor + cfg.shift_factor if cfg.shift_factor is not None else z / cfg.scaling_factor
    img = (decode(p, cfg, z) * 0.5 + 0.5).clamp(0, 1)
    return (img * 255).to(torch.uint8).permute(0, 2, 3, 1)


def make_vae_params(cfg: VaeCfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic weights: conv/linear N(0, 1/sqrt(fan_in)) (keeps activations O(1) through ~30 layers),
    biases N(0, 0.02), norm gains 1 + 0.1 N(0,1), norm biases N(0, 0.05)."""
    g = torch.Generator("cpu").manual_seed(seed)
    p = {}

    def conv(name, co, ci, k):
        p[f"{name}.weight"] = torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5
        p[f"{name}.bias"] = torch.randn(co, generator=g) * 0.02

    def lin(name, co, ci):
        p[f"{name}.weight"] = torch.randn(co, ci, generator=g) / ci ** 0.5
        p[f"{name}.bias"] = torch.randn(co, generator=g) * 0.02

    def norm(name, c):
        p[f"{name}.weight"] = 1 + 0.1 * torch.randn(c, generator=g)
        p[f"{name}.bias"] = 0.05 * torch.randn(c, generator=g)

    def res(name, ci, co):
        norm(f"{name}.norm1", ci); conv(f"{name}.conv1", co, ci, 3)
        norm(f"{name}.norm2", co); conv(f"{name}.conv2", co, co, 3)
        if ci != co:
            conv(f"{name}.conv_shortcut", co, ci, 1)

    def mid(name, c):
        res(f"{name}.resnets.0", c, c)
        norm(f"{name}.attentions.0.group_norm", c)
        for t in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(f"{name}.attentions.0.{t}", c, c)
        res(f"{name}.resnets.1", c, c)

    boc = list(cfg.block_out_channels)
    conv("encoder.conv_in", boc[0], cfg.in_channels, 3)
    ch = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            res(f"encoder.down_blocks.{i}.resnets.{j}", ch, co); ch = co
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", co, co, 3)
    mid("encoder.mid_block", boc[-1])
    norm("encoder.conv_norm_out", boc[-1]); conv("encoder.conv_out", 2 * cfg.latent_channels, boc[-1], 3)
    conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    rev = boc[::-1]
    conv("decoder.conv_in", rev[0], cfg.latent_channels, 3)
    mid("decoder.mid_block", rev[0])
    ch = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", ch, co); ch = co
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    norm("decoder.conv_norm_out", rev[-1]); conv("decoder.conv_out", cfg.out_channels, rev[-1], 3)
    return p
