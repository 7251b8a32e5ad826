"""AutoencoderKL (sdxl-vae architecture of diffusers==0.29.0) on the HIP conv kernels, fp32.

Interface mirror of what the reference touches (LVM/pipeline.py:87-117,558-590; LVM/utils.py:99-137):
`vae.encode(x).latent_dist.sample()`, `vae.decode(z).sample`, `vae.config.scaling_factor`,
`vae.config.shift_factor`, `.to()`, `.eval()`; module/parameter names equal diffusers' so a real
`diffusion_pytorch_model.safetensors` loads with `load_state_dict`.

Execution: every Conv2d/Linear is `ops.conv2d` (implicit GEMM on the fp32 MFMA); GroupNorm+SiLU is
never materialised (stats kernel + normalise-on-load prologue of the next conv); nearest upsampling is
folded into the following conv's loader; the resnet skip / attention residual is the conv epilogue;
the mid-block attention is Q K^T -> column softmax -> P V through the same conv kernel.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .ops import VgptError

F32 = torch.float32


class _Norm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


def _conv3(mod, x, prec: str, **kw):
    """3x3 stride-1 convolution of an nn.Conv2d: "fp32" = exact fp32 MFMA, "bf16x3" = split-bf16 operands on the bf16
    MFMA (include/vgpt.h, vgpt_conv2d_bx3_fwd); the split weights are cached on the module until the weight changes."""
    if prec == "bf16x3":
        key = (mod.weight.data_ptr(), mod.weight._version)
        if getattr(mod, "_bx3_key", None) != key:
            mod._bx3 = ops.conv_pack_bx3(mod.weight.detach().contiguous())
            mod._bx3_key = key
        return ops.conv2d_bx3(x, mod._bx3, mod.bias, **kw)
    if prec != "fp32":
        raise VgptError(f"unknown conv precision {prec!r} (fp32 | bf16x3)")
    return ops.conv2d(x, mod.weight, mod.bias, **kw)


def _conv1(mod, x, prec: str, **kw):
    """1x1 convolution of an nn.Conv2d / nn.Linear weight: the split-bf16 kernel when the precision asks for it and the
    input channels are whole 32-channel chunks, the exact fp32 kernel otherwise (the 4- and 8-channel quant convolutions)."""
    if prec == "bf16x3" and mod.weight.shape[1] % 32 == 0:
        key = (mod.weight.data_ptr(), mod.weight._version)
        if getattr(mod, "_bx1_key", None) != key:
            mod._bx1 = ops.conv1x1_pack_bx3(mod.weight.detach().contiguous())
            mod._bx1_key = key
        return ops.conv1x1_bx3(x, mod._bx1, mod.bias, **kw)
    return ops.conv2d(x, mod.weight, mod.bias, ksize=1, **kw)


class _Resnet(nn.Module):
    def __init__(self, ci, co):
        super().__init__()
        self.norm1, self.conv1 = _Norm(ci), nn.Conv2d(ci, co, 3, padding=1)
        self.norm2, self.conv2 = _Norm(co), nn.Conv2d(co, co, 3, padding=1)
        if ci != co:
            self.conv_shortcut = nn.Conv2d(ci, co, 1)

    def run(self, x, groups, eps, prec="fp32"):
        st = ops.groupnorm_stats(x, groups, eps)
        h = _conv3(self.conv1, x, prec, gn=(st, self.norm1.weight, self.norm1.bias, groups, 1))
        skip = x
        if hasattr(self, "conv_shortcut"):
            skip = _conv1(self.conv_shortcut, x, prec)
        st2 = ops.groupnorm_stats(h, groups, eps)
        return _conv3(self.conv2, h, prec, resid=skip, gn=(st2, self.norm2.weight, self.norm2.bias, groups, 1))


class _Attention(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.group_norm = _Norm(c)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Identity()])

    def run(self, x, groups, eps, prec="fp32"):
        N, C, H, W = x.shape
        HW = H * W
        st = ops.groupnorm_stats(x, groups, eps)
        gn = (st, self.group_norm.weight, self.group_norm.bias, groups, 0)
        q = _conv1(self.to_q, x, prec, gn=gn)   # (N, C, H, W) = [c][pos]
        k = _conv1(self.to_k, x, prec, gn=gn)
        v = _conv1(self.to_v, x, prec, gn=gn)
        # S^T[key][query] = sum_c K[c][key] Q[c][query]: "weights" = K stored [c][key] (transposed), input = Q
        # (a 1x1 conv is pointwise, so the HW query positions keep their (H, W) tiling)
        st_ = ops.conv2d(q, k, ksize=1, cout=HW, w_transposed=True, ldw=HW, w_batch_stride=C * HW)   # (N, HW, H, W)
        ops.col_softmax(st_.view(N, HW, HW), 1.0 / (C ** 0.5))
        # O[c][query] = sum_key V[c][key] P^T[key][query]
        o = ops.conv2d(st_, v, ksize=1, cout=C, ldw=HW, w_batch_stride=C * HW)                     # (N, C, H, W)
        return _conv1(self.to_out[0], o, prec, resid=x)


class _Mid(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(c, c), _Resnet(c, c)])
        self.attentions = nn.ModuleList([_Attention(c)])

    def run(self, x, groups, eps, prec="fp32"):
        x = self.resnets[0].run(x, groups, eps, prec)
        x = self.attentions[0].run(x, groups, eps, prec)
        return self.resnets[1].run(x, groups, eps, prec)


class _Sampler(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)


class _Block(nn.Module):
    def __init__(self, ci, co, n_res, kind: Optional[str]):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(ci if j == 0 else co, co) for j in range(n_res)])
        if kind == "down":
            self.downsamplers = nn.ModuleList([_Sampler(co)])
        elif kind == "up":
            self.upsamplers = nn.ModuleList([_Sampler(co)])


class Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        boc = cfg.block_out_channels
        self.conv_in = nn.Conv2d(cfg.in_channels, boc[0], 3, padding=1)
        blocks, ch = [], boc[0]
        for i, co in enumerate(boc):
            blocks.append(_Block(ch, co, cfg.layers_per_block, "down" if i != len(boc) - 1 else None))
            ch = co
        self.down_blocks = nn.ModuleList(blocks)
        self.mid_block = _Mid(boc[-1])
        self.conv_norm_out = _Norm(boc[-1])
        self.conv_out = nn.Conv2d(boc[-1], 2 * cfg.latent_channels, 3, padding=1)


class Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        rev = list(cfg.block_out_channels)[::-1]
        self.conv_in = nn.Conv2d(cfg.latent_channels, rev[0], 3, padding=1)
        self.mid_block = _Mid(rev[0])
        blocks, ch = [], rev[0]
        for i, co in enumerate(rev):
            blocks.append(_Block(ch, co, cfg.layers_per_block + 1, "up" if i != len(rev) - 1 else None))
            ch = co
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = _Norm(rev[-1])
        self.conv_out = nn.Conv2d(rev[-1], cfg.out_channels, 3, padding=1)


class DiagonalGaussianDistribution:
    """`latent_dist` of the encoder output; `sample()` draws with torch's RNG unless noise is supplied
    (identical-seed parity needs CPU-generator noise)."""

    def __init__(self, moments: torch.Tensor):
        self.parameters = moments

    def sample(self, generator=None, noise: Optional[torch.Tensor] = None):
        N, C2, h, w = self.parameters.shape
        if noise is None:
            noise = torch.randn(N, C2 // 2, h, w, device=self.parameters.device, dtype=F32, generator=generator)
        return ops.vae_sample(self.parameters, noise.to(F32).contiguous(), 0.0, 1.0)

    def sample_scaled(self, noise, shift, scaling):
        return ops.vae_sample(self.parameters, noise.to(F32).contiguous(), shift, scaling)

    def mode(self):
        return self.parameters[:, : self.parameters.shape[1] // 2]


class AutoencoderKL(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512),
                 layers_per_block=2, norm_num_groups=32, scaling_factor=0.13025, shift_factor=None, **_ignored):
        super().__init__()
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      latent_channels=latent_channels, block_out_channels=tuple(block_out_channels),
                                      layers_per_block=layers_per_block, norm_num_groups=norm_num_groups,
                                      scaling_factor=scaling_factor, shift_factor=shift_factor)
        self.encoder = Encoder(self.config)
        self.decoder = Decoder(self.config)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.eps = 1e-6
        # arithmetic of the 3x3 stride-1 convolutions (97 % of the FLOPs): "bf16x3" = split-bf16 operands on the bf16
        # MFMA, ~16 mantissa bits (the reference's torch/cuDNN default for these fp32 convolutions is TF32, 10 bits),
        # 1.8x faster; "fp32" = exact fp32 MFMA.  Both meet the same parity tolerances (tests/test_vae_gpu.py).
        self.conv_precision = "bf16x3"

    @classmethod
    def from_pretrained(cls, path):
        """Local directory with config.json + diffusion_pytorch_model.safetensors (no hub access offline)."""
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path}: hub download is unavailable offline; pass a local directory")
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        keys = ("in_channels", "out_channels", "latent_channels", "block_out_channels", "layers_per_block",
                "norm_num_groups", "scaling_factor", "shift_factor")
        vae = cls(**{k: cfg[k] for k in keys if k in cfg})
        from safetensors.torch import load_file
        vae.load_state_dict(load_file(os.path.join(path, "diffusion_pytorch_model.safetensors")))
        return vae

    def _ready(self, x):
        w = self.quant_conv.weight
        if not w.is_cuda or w.dtype != F32:
            raise VgptError("AutoencoderKL runs on the MI355X HIP path in fp32: call vae.to('cuda', torch.float32)")
        if not x.is_cuda:
            raise VgptError("AutoencoderKL: input must be on the GPU")

    @torch.no_grad()
    def encode(self, x):
        self._ready(x)
        g, eps, enc, pr = self.config.norm_num_groups, self.eps, self.encoder, self.conv_precision
        h = _conv3(enc.conv_in, x.to(F32).contiguous(), pr)
        for blk in enc.down_blocks:
            for r in blk.resnets:
                h = r.run(h, g, eps, pr)
            if hasattr(blk, "downsamplers"):
                c = blk.downsamplers[0].conv
                h = ops.conv2d(h, c.weight, c.bias, stride=2)       # 3 launches per frame: stays on the fp32 kernel
        h = enc.mid_block.run(h, g, eps, pr)
        st = ops.groupnorm_stats(h, g, eps)
        h = _conv3(enc.conv_out, h, pr, gn=(st, enc.conv_norm_out.weight, enc.conv_norm_out.bias, g, 1))
        moments = ops.conv2d(h, self.quant_conv.weight, self.quant_conv.bias, ksize=1)
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(moments))

    @torch.no_grad()
    def decode(self, z):
        self._ready(z)
        g, eps, dec, pr = self.config.norm_num_groups, self.eps, self.decoder, self.conv_precision
        h = ops.conv2d(z.to(F32).contiguous(), self.post_quant_conv.weight, self.post_quant_conv.bias, ksize=1)
        h = _conv3(dec.conv_in, h, pr)
        h = dec.mid_block.run(h, g, eps, pr)
        for blk in dec.up_blocks:
            for r in blk.resnets:
                h = r.run(h, g, eps, pr)
            if hasattr(blk, "upsamplers"):
                h = _conv3(blk.upsamplers[0].conv, h, pr, upsample=True)
        st = ops.groupnorm_stats(h, g, eps)
        h = _conv3(dec.conv_out, h, pr, gn=(st, dec.conv_norm_out.weight, dec.conv_norm_out.bias, g, 1))
        return SimpleNamespace(sample=h)

    # ---- fused helpers used by the pipeline (LVM/pipeline.py:110-117, 558-590) ----
    def encode_scaled(self, x, noise, dtype=torch.bfloat16):
        d = self.encode(x).latent_dist
        shift = self.config.shift_factor or 0.0
        return d.sample_scaled(noise, shift, self.config.scaling_factor).to(dtype)

    def decode_to_uint8(self, latents):
        """latents (N, C, h, w) bf16/fp32 in model scale -> (N, H, W, 3) uint8 on the device."""
        shift = self.config.shift_factor or 0.0
        z = ops.affine_to_f32(latents, 1.0 / self.config.scaling_factor, shift)
        return ops.vae_postprocess_u8(self.decode(z).sample)


# ---- LVM/utils.py:99-145: the VAE seam of the training loops -------------------------------------------------------
def vae_encode(vae, x, weight_dtype, seed=None, batch_encode=False):
    """list of images (1, 3, H, W) in [-1, 1] -> list of latents (1, 4, H/8, W/8): posterior sample, (z - shift) *
    scaling, cast (LVM/utils.py:99-137; call sites train_x1_stage1_noiseinput.py:354-358).  `batch_encode` runs the
    encoder once over the concatenated images; `seed` draws the posterior noise from a generator seeded with it (the
    reference reseeds and restores the global RNGs around every image, :113-135)."""
    shift, scaling = vae.config.shift_factor or 0.0, vae.config.scaling_factor
    if len(x) == 0:
        return []
    same = all(t.shape == x[0].shape for t in x)
    if batch_encode or (same and seed is None):
        if not same:
            raise VgptError("vae_encode(batch_encode=True): images must share one resolution")
        mom = vae.encode(torch.cat(list(x), dim=0)).latent_dist.parameters
        dists = [DiagonalGaussianDistribution(mom[i:i + 1]) for i in range(len(x))]
    else:
        dists = [vae.encode(img).latent_dist for img in x]
    out = []
    for d in dists:
        p = d.parameters
        gen = torch.Generator(device=p.device).manual_seed(seed) if seed is not None else None
        noise = torch.randn(p.shape[0], p.shape[1] // 2, *p.shape[2:], device=p.device, dtype=F32, generator=gen)
        out.append(d.sample_scaled(noise, shift, scaling).to(weight_dtype))
    return out


def vae_encode_list(vae, x, weight_dtype):
    """LVM/utils.py:140-145."""
    return [vae_encode(vae, img, weight_dtype) for img in x]
