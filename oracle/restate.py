"""CPU restatement (torch, fp32 by default) of the Video-GPT denoiser path.  TEST INFRASTRUCTURE.

Every function cites the reference lines it restates (paths relative to the reference checkout;
"HF" = transformers==4.47.1 models/phi3/modeling_phi3.py, which the reference imports but does
not vendor).  Parameters travel as a flat dict keyed exactly like the reference's state_dict
(SURVEY.md §8b), so the same dict loads into the product's modules.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class Phi3Cfg:
    """The subset of Phi3Config the path reads (defaults = installed Phi3Config defaults)."""
    hidden_size: int = 3072
    intermediate_size: int = 8192
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    vocab_size: int = 32064
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    hidden_act: str = "silu"
    pad_token_id: int = 2
    patch_size: int = 2
    in_channels: int = 4
    pos_embed_max_size: int = 192
    pe_interpolation: float = 1.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


TINY = Phi3Cfg(hidden_size=192, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=2, vocab_size=64, pos_embed_max_size=24)


# ------------------------------------------------------------------------------------------------
# position table, patch embed, timestep embed, final layer   (LVM/model.py:22-154, 255-289)
# ------------------------------------------------------------------------------------------------

def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """LVM/model.py:117-135 — [sin | cos] of pos * 1/10000^(i/(D/2)), float64."""
    omega = np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d(embed_dim: int, grid_size: int, interpolation_scale: float = 1.0, base_size: int = 1) -> np.ndarray:
    """LVM/model.py:86-114 — note `np.meshgrid(grid_w, grid_h)` (w first) and emb = [emb(grid[0]) | emb(grid[1])]."""
    gh = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size) / interpolation_scale
    gw = np.arange(grid_size, dtype=np.float32) / (grid_size / base_size) / interpolation_scale
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    return np.concatenate([sincos_1d(embed_dim // 2, grid[0]), sincos_1d(embed_dim // 2, grid[1])], axis=1)


def make_pos_embed(cfg: Phi3Cfg) -> torch.Tensor:
    """LVM/model.py:185-186 — (1, max*max, H) fp32 persistent buffer, base_size=64."""
    pe = sincos_2d(cfg.hidden_size, cfg.pos_embed_max_size, cfg.pe_interpolation, base_size=64)
    return torch.from_numpy(pe).float().unsqueeze(0)


def cropped_pos_embed(pos_embed: torch.Tensor, cfg: Phi3Cfg, height: int, width: int) -> torch.Tensor:
    """LVM/model.py:268-289 (height/width are latent sizes)."""
    h, w = height // cfg.patch_size, width // cfg.patch_size
    if h > cfg.pos_embed_max_size or w > cfg.pos_embed_max_size:
        raise ValueError("latent larger than pos_embed_max_size")
    top, left = (cfg.pos_embed_max_size - h) // 2, (cfg.pos_embed_max_size - w) // 2
    pe = pos_embed.reshape(1, cfg.pos_embed_max_size, cfg.pos_embed_max_size, -1)
    return pe[:, top:top + h, left:left + w, :].reshape(1, h * w, -1)


def patch_embed(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, patch: int) -> torch.Tensor:
    """LVM/model.py:149-154 — Conv2d(k=s=patch) then NCHW -> NLC."""
    return F.conv2d(x, weight, bias, stride=patch).flatten(2).transpose(1, 2)


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000) -> torch.Tensor:
    """LVM/model.py:39-58 — [cos | sin], t is NOT scaled by 1000."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def timestep_embedder(p: Dict[str, torch.Tensor], prefix: str, t: torch.Tensor, dtype) -> torch.Tensor:
    """LVM/model.py:60-63 — Linear(256,H) -> SiLU -> Linear(H,H) on the sinusoid cast to `dtype`."""
    x = timestep_embedding(t, p[f"{prefix}.mlp.0.weight"].shape[1]).to(dtype)
    x = F.linear(x, p[f"{prefix}.mlp.0.weight"], p[f"{prefix}.mlp.0.bias"])
    return F.linear(F.silu(x), p[f"{prefix}.mlp.2.weight"], p[f"{prefix}.mlp.2.bias"])


def final_layer(p: Dict[str, torch.Tensor], x: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """LVM/model.py:22-23,79-83 — adaLN(shift, scale) on LayerNorm(no affine, eps 1e-6), then Linear."""
    mod = F.linear(F.silu(c), p["final_layer.adaLN_modulation.1.weight"], p["final_layer.adaLN_modulation.1.bias"])
    shift, scale = mod.chunk(2, dim=1)
    x = F.layer_norm(x, (x.shape[-1],), eps=1e-6)
    x = x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)
    return F.linear(x, p["final_layer.linear.weight"], p["final_layer.linear.bias"])


def unpatchify(x: torch.Tensor, h: int, w: int, patch: int, c: int) -> torch.Tensor:
    """LVM/model.py:255-265."""
    x = x.reshape(x.shape[0], h // patch, w // patch, patch, patch, c)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(x.shape[0], c, h, w)


# ------------------------------------------------------------------------------------------------
# Phi3 decoder (HF 4.47.1 formulas; call sites OmniGen/transformer.py:128-214,
# LVM/transform/sdpa_transform.py:37-91)
# ------------------------------------------------------------------------------------------------

def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """HF Phi3RMSNorm.forward: fp32 variance, cast back, times weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return weight * xf.to(dt)


def rope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float, dtype, rope_scaling=None,
                 max_position_embeddings: int = 4096, original_max_position_embeddings: Optional[int] = None):
    """HF Phi3RotaryEmbedding.forward: inv_freq = 1/theta^(2i/d), fp32 freqs, emb=[f|f], cast to dtype.
    rope_scaling {"type": "su"|"longrope", short_factor, long_factor} = HF 4.47.1 Phi3LongRoPEScaledRotaryEmbedding:
    inv_freq = 1/(ext * theta^(2i/d)) with ext = long_factor when max(position_ids)+1 > original_max else short_factor,
    cos / sin scaled by sqrt(1 + ln(max/orig)/ln(orig)) when max > orig."""
    shape = torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim
    scale = 1.0
    if rope_scaling is None:
        inv_freq = 1.0 / (theta ** shape)
    else:
        orig = original_max_position_embeddings or max_position_embeddings
        long = int(position_ids.max()) + 1 > orig
        ext = torch.tensor(rope_scaling["long_factor" if long else "short_factor"], dtype=torch.float32)
        inv_freq = 1.0 / (ext * theta ** shape)
        f = max_position_embeddings / orig
        scale = 1.0 if f <= 1.0 else math.sqrt(1 + math.log(f) / math.log(orig))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return (emb.cos() * scale).to(dtype), (emb.sin() * scale).to(dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(q, k, cos, sin):
    """HF apply_rotary_pos_emb with unsqueeze_dim=1 (q,k are (B,h,L,d))."""
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin


def additive_mask(mask: torch.Tensor, dtype) -> torch.Tensor:
    """OmniGen/transformer.py:139-145 (NV branch): 0 where visible, finfo(dtype).min where masked."""
    if mask is None or mask.dim() != 3:
        raise Exception("attention_mask parameter was unavailable or invalid")
    m = -1 * (mask + -1) * torch.finfo(dtype).min
    return m.unsqueeze(1).to(dtype)


_ACT = {"silu": F.silu, "gelu": F.gelu, "gelu_new": lambda x: F.gelu(x, approximate="tanh"),
        "gelu_pytorch_tanh": lambda x: F.gelu(x, approximate="tanh")}


def attention(p, cfg: Phi3Cfg, i: int, hidden: torch.Tensor, amask: torch.Tensor, cos, sin) -> torch.Tensor:
    """LVM/transform/sdpa_transform.py:37-91 at SP=1: qkv_proj, split, RoPE, repeat_kv, SDPA(additive mask), o_proj."""
    B, L, _ = hidden.shape
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    qkv = F.linear(hidden, p[f"llm.layers.{i}.self_attn.qkv_proj.weight"])
    q = qkv[..., : nh * hd].view(B, L, nh, hd).transpose(1, 2)
    k = qkv[..., nh * hd: nh * hd + nkv * hd].view(B, L, nkv, hd).transpose(1, 2)
    v = qkv[..., nh * hd + nkv * hd:].view(B, L, nkv, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if nkv != nh:
        k = k.repeat_interleave(nh // nkv, dim=1)
        v = v.repeat_interleave(nh // nkv, dim=1)
    w = torch.matmul(q, k.transpose(2, 3)) / math.sqrt(hd) + amask
    w = torch.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(w, v).transpose(1, 2).reshape(B, L, nh * hd)
    return F.linear(o, p[f"llm.layers.{i}.self_attn.o_proj.weight"])


def mlp(p, cfg: Phi3Cfg, i: int, x: torch.Tensor) -> torch.Tensor:
    """HF Phi3MLP.forward: gate, up = gate_up_proj(x).chunk(2); down(up * act(gate))."""
    gate, up = F.linear(x, p[f"llm.layers.{i}.mlp.gate_up_proj.weight"]).chunk(2, dim=-1)
    return F.linear(up * _ACT[cfg.hidden_act](gate), p[f"llm.layers.{i}.mlp.down_proj.weight"])


def transformer(p, cfg: Phi3Cfg, inputs_embeds: torch.Tensor, attention_mask: torch.Tensor,
                position_ids: torch.Tensor, return_layers: bool = False):
    """OmniGen/transformer.py:128-214 + HF Phi3DecoderLayer (resid dropouts are p=0)."""
    amask = additive_mask(attention_mask, inputs_embeds.dtype)
    cos, sin = rope_cos_sin(position_ids, cfg.head_dim, cfg.rope_theta, inputs_embeds.dtype)
    h = inputs_embeds
    layers = []
    for i in range(cfg.num_hidden_layers):
        a = attention(p, cfg, i, rmsnorm(h, p[f"llm.layers.{i}.input_layernorm.weight"], cfg.rms_norm_eps), amask, cos, sin)
        h = h + a
        h = h + mlp(p, cfg, i, rmsnorm(h, p[f"llm.layers.{i}.post_attention_layernorm.weight"], cfg.rms_norm_eps))
        if return_layers:
            layers.append(h)
    out = rmsnorm(h, p["llm.norm.weight"], cfg.rms_norm_eps)
    return (out, layers) if return_layers else out


# ------------------------------------------------------------------------------------------------
# LVM.frame_block_forward (+ CFG)   (LVM/model.py:292-327, 399-501, 519-566)
# ------------------------------------------------------------------------------------------------

def patch_multiple_resolutions(p, cfg: Phi3Cfg, pos_embed, latents: List[torch.Tensor], is_input_images: bool):
    """LVM/model.py:292-316 (list branch, no padding latents)."""
    pre = "input_x_embedder" if is_input_images else "x_embedder"
    out, num_tokens, shapes = [], [], []
    for lat in latents:
        h, w = lat.shape[-2:]
        tok = patch_embed(lat, p[f"{pre}.proj.weight"], p[f"{pre}.proj.bias"], cfg.patch_size)
        pe = cropped_pos_embed(pos_embed, cfg, h, w)
        out.append(tok + pe.to(tok.dtype))
        num_tokens.append(pe.size(1))
        shapes.append([h, w])
    return out, num_tokens, shapes


def frame_block_forward(p, cfg: Phi3Cfg, x: List[torch.Tensor], timestep: torch.Tensor, input_ids, input_img_latents,
                        input_image_sizes, attention_mask, position_ids, denoise_image_sizes, time_emb_inx,
                        return_hidden: bool = False, input_output_return: bool = False):
    """LVM/model.py:399-501 at world_size 1 (also LVMTraining.forward :752-845, same body).  input_output_return
    (LVM/model.py:488-497): also the `input_final_layer` head -- a plain Linear on the last hidden state -- on every condition
    frame's rows, unpatchified."""
    pos_embed = p["pos_embed"]
    xs, _, shapes = patch_multiple_resolutions(p, cfg, pos_embed, x, False)
    dtype = xs[0].dtype
    time_token = timestep_embedder(p, "time_token", timestep, dtype)
    input_latents = []
    if input_img_latents is not None:
        input_latents, _, input_shapes = patch_multiple_resolutions(p, cfg, pos_embed, input_img_latents, True)
    emb = F.embedding(input_ids, p["llm.embed_tokens.weight"]).clone()
    n = 0
    for b in input_image_sizes.keys():
        for s, e in input_image_sizes[b]:
            emb[b, s:e] = input_latents[n]
            n += 1
    tn = 0
    for b in time_emb_inx.keys():
        for tok in time_emb_inx[b]:
            emb[b, tok] = time_token[tn]
            tn += 1
    dn = 0
    for b in denoise_image_sizes.keys():
        for s, e in denoise_image_sizes[b]:
            emb[b, s:e] = xs[dn]
            dn += 1
    assert n == len(input_latents) and tn == time_token.shape[0] and dn == len(xs)
    out = transformer(p, cfg, emb, attention_mask, position_ids)
    time_emb = timestep_embedder(p, "t_embedder", timestep, dtype)
    latents, k = [], 0
    for b in denoise_image_sizes.keys():
        for s, e in denoise_image_sizes[b]:
            y = final_layer(p, out[b:b + 1, s:e], time_emb[k:k + 1])
            latents.append(unpatchify(y, shapes[k][0], shapes[k][1], cfg.patch_size, cfg.in_channels))
            k += 1
    if input_output_return:
        preds, k = [], 0
        for b in input_image_sizes.keys():
            for s, e in input_image_sizes[b]:
                y = F.linear(out[b:b + 1, s:e], p["input_final_layer.weight"], p["input_final_layer.bias"])
                preds.append(unpatchify(y, input_shapes[k][0], input_shapes[k][1], cfg.patch_size, cfg.in_channels))
                k += 1
        return latents, preds
    return (latents, out) if return_hidden else latents


def lvm_forward(p, cfg: Phi3Cfg, x: torch.Tensor, timestep, input_ids, input_img_latents, input_image_sizes,
                attention_mask, position_ids):
    """LVM.forward (LVM/model.py:330-397), tensor branch at world_size 1: [condition | time_token | x] -> last N tokens."""
    pos_embed = p["pos_embed"]
    h, w = x.shape[-2:]
    tok = patch_embed(x, p["x_embedder.proj.weight"], p["x_embedder.proj.bias"], cfg.patch_size)
    tok = tok + cropped_pos_embed(pos_embed, cfg, h, w).to(tok.dtype)
    n_tok = tok.size(1)
    time_token = timestep_embedder(p, "time_token", timestep, tok.dtype).unsqueeze(1)
    if input_ids is not None:
        cond = F.embedding(input_ids, p["llm.embed_tokens.weight"]).clone()
        if input_img_latents is not None:
            lat, _, _ = patch_multiple_resolutions(p, cfg, pos_embed, input_img_latents, True)
            n = 0
            for b in input_image_sizes.keys():
                for s, e in input_image_sizes[b]:
                    cond[b, s:e] = lat[n]
                    n += 1
        emb = torch.cat([cond, time_token, tok], dim=1)
    else:
        emb = torch.cat([time_token, tok], dim=1)
    out = transformer(p, cfg, emb, attention_mask, position_ids)
    y = final_layer(p, out[:, -n_tok:], timestep_embedder(p, "t_embedder", timestep, tok.dtype))
    return unpatchify(y, h, w, cfg.patch_size, cfg.in_channels)


def frame_block_forward_with_cfg(p, cfg, x, timestep, use_img_cfg, img_cfg_scale, prediction_type="v", **kw):
    """LVM/model.py:519-566 — CFG on the model output only for prediction_type 'v'."""
    out = frame_block_forward(p, cfg, x, timestep, **kw)
    if use_img_cfg and prediction_type == "v":
        half = len(out) // 2
        cond, uncond = out[:half], out[half:]
        cond = [uncond[i] + img_cfg_scale * (cond[i] - uncond[i]) for i in range(half)]
        out = cond + cond
    return out


# ------------------------------------------------------------------------------------------------
# LVMScheduler   (LVM/scheduler.py:120-130, 161-208)
# ------------------------------------------------------------------------------------------------

def scheduler_sigma(num_steps: int = 50, time_shifting_factor: float = 1, begin_time: Optional[float] = None):
    t = torch.linspace(0 if begin_time is None else begin_time, 1, num_steps + 1)
    return t / (t + time_shifting_factor - time_shifting_factor * t)


def scheduler_call(sigma: torch.Tensor, z: List[torch.Tensor], func, use_img_cfg: bool, img_cfg_scale: float,
                   prediction_type: str = "v"):
    """LVM/scheduler.py:161-208, list branch.  `func(z, timesteps) -> list of predictions`."""
    z = [t.clone() for t in z]
    for i in range(len(sigma) - 1):
        timesteps = torch.zeros(len(z)) + sigma[i]
        pred = list(func(z, timesteps))
        s, s_next = sigma[i], sigma[i + 1]
        if prediction_type == "x1":
            pred = [(pred[j] - z[j]) / (1.0 - s) for j in range(len(z))]
            if use_img_cfg:
                half = len(pred) // 2
                cond, uncond = pred[:half], pred[half:]
                cond = [uncond[j] + img_cfg_scale * (cond[j] - uncond[j]) for j in range(half)]
                pred = cond + cond
        z = [z[j] + (s_next - s) * pred[j] for j in range(len(z))]
    return z


# ------------------------------------------------------------------------------------------------
# LVMCollator builders   (LVM/processor.py:128-274, 442-534, 575-731, 812-838, 964-1000)
# ------------------------------------------------------------------------------------------------

SPECIAL = {"img": 10, "img_end": 11, "diffusion": 12, "slot": 0}  # synthetic ids (no tokenizer offline)


def ids_inference(C: int, G: int, N: int, special=SPECIAL):
    """LVM/processor.py:128-179 with one-token chunks: clean = <img> N*slot </img>, noisy = <|diffusion|> slot(time) N*slot."""
    ids, sizes = [], []
    for _ in range(C):
        ids.append(special["img"]); s = len(ids); sizes.append([s, s + N]); ids.extend([special["slot"]] * N)
        ids.append(special["img_end"])
    for _ in range(G):
        ids.append(special["diffusion"]); ids.append(special["slot"]); s = len(ids); sizes.append([s, s + N])
        ids.extend([special["slot"]] * N)
    return ids, sizes


def ids_stage1(F_: int, N: int, special=SPECIAL):
    """LVM/processor.py:181-218: noisy_0, clean_0, noisy_1, ..., noisy_{F-1} (2F-1 blocks)."""
    ids, sizes = [], []
    for i in range(2 * F_ - 1):
        if i % 2 == 0:
            ids.append(special["diffusion"]); ids.append(special["slot"]); s = len(ids); sizes.append([s, s + N])
            ids.extend([special["slot"]] * N)
        else:
            ids.append(special["img"]); s = len(ids); sizes.append([s, s + N]); ids.extend([special["slot"]] * N)
            ids.append(special["img_end"])
    return ids, sizes


def pad_ids(id_rows, size_rows, pad_id: int, sp: int = 1):
    """LVM/processor.py:812-838 — left-pad to the longest row (rounded up to a multiple of sp)."""
    max_l = max(len(r) for r in id_rows)
    if max_l % sp:
        max_l += sp - max_l % sp
    ids = np.full((len(id_rows), max_l), pad_id, dtype=np.int64)
    valid = np.zeros((len(id_rows), max_l), dtype=np.uint8)
    sizes = {}
    for b, row in enumerate(id_rows):
        pad = max_l - len(row)
        ids[b, pad:] = row
        valid[b, pad:] = 1
        sizes[b] = [[s + pad, e + pad] for s, e in size_rows[b]]
    return ids, valid, sizes


def _clean_block(m, r0, c0, bl):
    """clean block at rows>=r0 / cols [c0,c0+bl): LVM/processor.py:700-702 (same at :649-651)."""
    m[r0:, c0] = 1
    m[r0 + 1:, c0 + 1: c0 + bl - 1] = 1
    m[r0 + bl - 1:, c0 + bl - 1] = 1


def _noisy_clip(m, r0, c0, bl, n):
    """n noisy blocks forming one clip: LVM/processor.py:708-720 (same at :635-647, :657-669)."""
    for i in range(n):
        c = c0 + i * bl
        m[r0:r0 + bl, c] = 1
        m[r0 + 1:r0 + bl, c + 1] = 1
        m[r0 + 2:r0 + bl, c + 2:c + bl] = 1
    for i in range(1, n):
        m[r0 + i * bl:r0 + (i + 1) * bl, c0:c0 + n * bl] = m[r0:r0 + bl, c0:c0 + n * bl]


def _with_pad(m, pad):
    """LVM/processor.py:722-727: pad columns masked, pad rows all ones."""
    if pad == 0:
        return m
    L = m.shape[0] + pad
    out = np.zeros((L, L), dtype=np.uint8)
    out[:pad, :] = 1
    out[pad:, pad:] = m
    return out


def mask_inference(valid_len: int, pad: int, bl: int, frame_blocks):
    """LVM/processor.py:682-731."""
    m = np.zeros((valid_len, valid_len), dtype=np.uint8)
    r = c = 0
    for k, fb in enumerate(frame_blocks):
        if k != len(frame_blocks) - 1:
            for _ in range(fb):
                _clean_block(m, r, c, bl); r += bl; c += bl
        else:
            _noisy_clip(m, r, c, bl, fb)
    return _with_pad(m, pad)


def positions_inference(sizes, frame_blocks):
    """LVM/processor.py:502-534 (note the row-0 special case `pad = first_start - 1`)."""
    pos, bls = [], []
    for b in sizes.keys():
        pad = sizes[b][0][0] - (1 if b == 0 else 2)
        token_l = sizes[b][-1][-1] - pad
        assert token_l % len(sizes[b]) == 0
        bl = token_l // len(sizes[b]); bls.append(bl)
        n_blocks = sum(frame_blocks[b])
        pos.append([0] * pad + list(range(n_blocks * bl)))
    return np.asarray(pos, dtype=np.int64), bls


def collate_inference(C: int, G: int, N: int, use_cfg: bool = True, pad_id: int = 2, sp: int = 1):
    """LVM/processor.py:366-421 + 916-941 + 964-1000 for row 0 = [C clean, G noisy], CFG row = [0, G]."""
    rows, sizes, fbs = [], [], {}
    i0, s0 = ids_inference(C, G, N); rows.append(i0); sizes.append(s0); fbs[0] = [C, G]
    if use_cfg:
        i1, s1 = ids_inference(0, G, N); rows.append(i1); sizes.append(s1); fbs[1] = [0, G]
    ids, valid, sizes = pad_ids(rows, sizes, pad_id, sp)
    pos, bls = positions_inference(sizes, fbs)
    L = ids.shape[1]
    mask = np.stack([mask_inference(int(valid[b].sum()), L - int(valid[b].sum()), bls[b], fbs[b]) for b in range(len(rows))])
    inp, den, tix = {}, {}, {}
    for b in sizes.keys():
        nc = fbs[b][0]
        inp[b] = sizes[b][:nc]
        den[b] = sizes[b][nc:]
        tix[b] = [s[0] - 1 for s in den[b]]
    return dict(input_ids=torch.from_numpy(ids), position_ids=torch.from_numpy(pos),
                attention_mask=torch.from_numpy(mask).to(torch.bool), input_image_sizes=inp,
                denoise_image_sizes=den, time_emb_inx=tix, frame_blocks=fbs)


def mask_stage1(valid_len: int, pad: int, bl: int):
    """LVM/processor.py:575-616 (create_mask_training)."""
    image_num = valid_len // bl // 2 + 1
    m = np.tril(np.ones((valid_len, valid_len), dtype=np.uint8))
    bs, be, ims, ime = 0, bl, 2, bl
    for i in range(image_num):
        m[be:, bs:be] = 0
        m[ims:ime, ims:ime] = 1
        if i != image_num - 1:
            bs += bl; be += bl
            ims, ime = bs + 1, be - 1
            m[ims:, ims:ime] = 1
            bs += bl; be += bl
            ims, ime = bs + 2, be
    return _with_pad(m, pad)


def positions_stage1(sizes):
    """LVM/processor.py:442-467: noisy_i and clean_i share positions [i*bl, (i+1)*bl)."""
    pos, bls = [], []
    for b in sizes.keys():
        pad = sizes[b][0][0] - 2
        token_l = sizes[b][-1][-1] - pad
        assert token_l % len(sizes[b]) == 0
        bl = token_l // len(sizes[b]); bls.append(bl)
        row = [0] * pad
        start = 0
        for i in range(len(sizes[b])):
            if i == 0:
                row.extend(range(start, start + bl))
            elif i % 2 == 0:
                row.extend(range(start, start + bl)); start += bl; row.extend(range(start, start + bl))
        pos.append(row)
    return np.asarray(pos, dtype=np.int64), bls


def collate_stage1(F_list: List[int], N: int, pad_id: int = 2, sp: int = 1):
    """TrainDataCollator (LVM/train_helper/data.py:422-458) over process_mllm_input_training (processor.py:869-891)."""
    rows, sizes = [], []
    for F_ in F_list:
        i, s = ids_stage1(F_, N); rows.append(i); sizes.append(s)
    ids, valid, sizes = pad_ids(rows, sizes, pad_id, sp)
    pos, bls = positions_stage1(sizes)
    L = ids.shape[1]
    mask = np.stack([mask_stage1(int(valid[b].sum()), L - int(valid[b].sum()), bls[b]) for b in range(len(rows))])
    den = {b: [s for i, s in enumerate(sizes[b]) if i % 2 == 0] for b in sizes}
    inp = {b: [s for i, s in enumerate(sizes[b]) if i % 2 == 1] for b in sizes}
    tix = {b: [s[0] - 1 for s in den[b]] for b in sizes}
    return dict(input_ids=torch.from_numpy(ids), position_ids=torch.from_numpy(pos),
                attention_mask=torch.from_numpy(mask).to(torch.bool), input_image_sizes=inp,
                denoise_image_sizes=den, time_emb_inx=tix)


# ------------------------------------------------------------------------------------------------
# stage-1 loss   (LVM/train_helper/loss.py:128-243) with externally supplied noise / times
# ------------------------------------------------------------------------------------------------

def stage1_loss(p, cfg, x1: List[torch.Tensor], x0, t, clean_latents, x0_in, t_in, batch, input_output_return: bool = False):
    """xt = t x1 + (1-t) x0; clean inputs noised the same way with t_in in [0.9,1]; per-frame MSE (order=2).
    input_output_return (loss.py:194-197,220-225): the model also predicts the CLEAN condition latents from the noised ones it
    was given (input_final_layer head); their per-frame MSE terms are appended to the loss vector."""
    xt = [t[i] * x1[i] + (1 - t[i]) * x0[i] for i in range(len(x1))]
    cl = [t_in[i] * clean_latents[i] + (1 - t_in[i]) * x0_in[i] for i in range(len(clean_latents))]
    pred = frame_block_forward(p, cfg, xt, t, batch["input_ids"], cl, batch["input_image_sizes"],
                               batch["attention_mask"], batch["position_ids"], batch["denoise_image_sizes"],
                               batch["time_emb_inx"], input_output_return=input_output_return)
    if input_output_return:
        pred, pred_in = pred
        loss = torch.stack([((x1[i] - pred[i]) ** 2).mean() for i in range(len(x1))])
        loss_in = torch.stack([((clean_latents[i] - pred_in[i]) ** 2).mean() for i in range(len(clean_latents))])
        return torch.cat([loss, loss_in]), xt
    return torch.stack([((x1[i] - pred[i]) ** 2).mean() for i in range(len(x1))]), xt


# ------------------------------------------------------------------------------------------------
# synthetic parameters: N(0,0.02) weights/biases, norm gains 1+0.1 N(0,1),
# zero-initialised heads re-randomised (LVM/model.py:241-244 would make every output exactly 0)
# ------------------------------------------------------------------------------------------------

def make_params(cfg: Phi3Cfg, seed: int = 0, dtype=torch.float32, pos_embed: bool = True) -> Dict[str, torch.Tensor]:
    g = torch.Generator("cpu").manual_seed(seed)
    H, I = cfg.hidden_size, cfg.intermediate_size
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    pp = cfg.patch_size * cfg.patch_size * cfg.in_channels

    def w(*shape):
        return (torch.randn(*shape, generator=g) * 0.02).to(dtype)

    def gain(n):
        return (1 + 0.1 * torch.randn(n, generator=g)).to(dtype)

    p = {}
    for pre in ("x_embedder", "input_x_embedder"):
        p[f"{pre}.proj.weight"] = w(H, cfg.in_channels, cfg.patch_size, cfg.patch_size)
        p[f"{pre}.proj.bias"] = w(H)
    for pre in ("time_token", "t_embedder"):
        p[f"{pre}.mlp.0.weight"] = w(H, 256); p[f"{pre}.mlp.0.bias"] = w(H)
        p[f"{pre}.mlp.2.weight"] = w(H, H); p[f"{pre}.mlp.2.bias"] = w(H)
    p["final_layer.linear.weight"] = w(pp, H); p["final_layer.linear.bias"] = w(pp)
    p["final_layer.adaLN_modulation.1.weight"] = w(2 * H, H); p["final_layer.adaLN_modulation.1.bias"] = w(2 * H)
    p["llm.embed_tokens.weight"] = w(cfg.vocab_size, H)
    for i in range(cfg.num_hidden_layers):
        p[f"llm.layers.{i}.self_attn.qkv_proj.weight"] = w((nh + 2 * nkv) * hd, H)
        p[f"llm.layers.{i}.self_attn.o_proj.weight"] = w(H, nh * hd)
        p[f"llm.layers.{i}.mlp.gate_up_proj.weight"] = w(2 * I, H)
        p[f"llm.layers.{i}.mlp.down_proj.weight"] = w(H, I)
        p[f"llm.layers.{i}.input_layernorm.weight"] = gain(H)
        p[f"llm.layers.{i}.post_attention_layernorm.weight"] = gain(H)
    p["llm.norm.weight"] = gain(H)
    if pos_embed:
        p["pos_embed"] = make_pos_embed(cfg).to(dtype)
    return p


def add_input_final_layer(p: Dict[str, torch.Tensor], cfg: Phi3Cfg, seed: int = 77) -> Dict[str, torch.Tensor]:
    """The optional `input_final_layer` head (LVM/model.py:246-253 creates it zero-initialised; seeded N(0, 0.02) here so
    that its outputs and gradients say something)."""
    g = torch.Generator("cpu").manual_seed(seed)
    pp = cfg.patch_size * cfg.patch_size * cfg.in_channels
    q = dict(p)
    q["input_final_layer.weight"] = (torch.randn(pp, cfg.hidden_size, generator=g) * 0.02).to(torch.bfloat16).float()
    q["input_final_layer.bias"] = (torch.randn(pp, generator=g) * 0.02).to(torch.bfloat16).float()
    return q
