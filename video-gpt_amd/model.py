"""LVM / LVMTraining: the next-clip diffusion transformer running on the HIP kernels.

Mirrors the reference's public interface (LVM/model.py:157-566 `LVM`, :569-845 `LVMTraining`;
OmniGen/transformer.py:71-232 `Phi3Transformer`) — constructor arguments, method names, argument
order/defaults, return structure and `state_dict` keys — so checkpoints and callers carry over.
The arithmetic does not go through torch: every step is a kernel of libvgpt_hip.so
(video-gpt_amd/ops.py).  torch supplies parameters, device buffers and streams.

Design differences from the reference (same results, different execution):
  * the token sequence is assembled by three kernels (embedding gather, patch-embed + position
    table scattered straight into the sequence, time-token MLP scattered into the sequence)
    instead of per-frame Python loops over tensor slices (LVM/model.py:419-454);
  * the (B,L,L) mask is bit-packed once and summarised per tile; the dense additive
    (B,1,L,L) mask of OmniGen/transformer.py:139-145 is never materialised;
  * q/k/v stay in the fused projection buffer: RoPE is applied in place and the attention
    kernel reads the strided heads directly (no transpose/contiguous copies of
    LVM/transform/sdpa_transform.py:45-47,75-77);
  * residual adds are GEMM epilogues, act(gate)*up is the gate_up GEMM's epilogue;
  * everything is shape-static and allocation-free after the first call, so a whole denoise
    step can be captured into a hipGraph (scheduler.py).
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .ops import BF16, VgptError


# ------------------------------------------------------------------------------------------------
# configuration
# ------------------------------------------------------------------------------------------------

class Phi3Config:
    """The fields of transformers.Phi3Config this path reads; accepts the HF object or kwargs."""

    _DEFAULTS = dict(vocab_size=32064, hidden_size=3072, intermediate_size=8192, num_hidden_layers=32,
                     num_attention_heads=32, num_key_value_heads=None, hidden_act="silu", rms_norm_eps=1e-5,
                     rope_theta=10000.0, pad_token_id=32000, use_cache=False, attention_dropout=0.0,
                     max_position_embeddings=4096, original_max_position_embeddings=None, rope_scaling=None,
                     partial_rotary_factor=1.0)

    def __init__(self, **kw):
        for k, v in self._DEFAULTS.items():
            setattr(self, k, kw.pop(k, v))
        if self.num_key_value_heads is None:
            self.num_key_value_heads = self.num_attention_heads
        for k, v in kw.items():
            setattr(self, k, v)
        self._check_rope()

    def _check_rope(self):
        """Rotary variants of the checkpoint's config.json (the reference builds its rotary embedding from it,
        LVM/model.py:202 -> HF Phi3Attention._init_rope): plain RoPE, or "su" / "longrope" (Phi-3-128k family: per-dim
        short / long factors + an attention factor on cos / sin).  Anything else is refused instead of silently
        rotating q / k differently from the reference."""
        if float(self.partial_rotary_factor or 1.0) != 1.0:
            raise VgptError(f"partial_rotary_factor={self.partial_rotary_factor} is not supported (the reference's "
                            "transformers 4.47.1 Phi3 rotates the whole head)")
        rs = self.rope_scaling
        if rs is None:
            return
        if not isinstance(rs, dict):
            raise VgptError(f"rope_scaling must be a dict, got {type(rs).__name__}")
        kind = rs.get("rope_type", rs.get("type", "default"))
        if kind in (None, "default"):
            self.rope_scaling = None
            return
        if kind not in ("su", "longrope"):
            raise VgptError(f"rope_scaling type {kind!r} is not supported (plain RoPE, 'su' and 'longrope' are)")
        half = self.head_dim // 2
        for key in ("short_factor", "long_factor"):
            if not isinstance(rs.get(key), (list, tuple)) or len(rs[key]) != half:
                raise VgptError(f"rope_scaling.{key} must list head_dim/2 = {half} numbers")
        if self.original_max_position_embeddings is None:
            self.original_max_position_embeddings = rs.get("original_max_position_embeddings", self.max_position_embeddings)

    @classmethod
    def _from_dict(cls, d):
        d = dict(d)
        rp = d.get("rope_parameters")
        if isinstance(rp, dict):      # transformers 5.x spelling of rope_theta / rope_scaling / partial_rotary_factor
            d.setdefault("rope_theta", rp.get("rope_theta", 10000.0))
            if d.get("rope_scaling") is None and rp.get("rope_type", "default") != "default":
                d["rope_scaling"] = rp
            if "partial_rotary_factor" in rp:
                d.setdefault("partial_rotary_factor", rp["partial_rotary_factor"])
            if d.get("original_max_position_embeddings") is None and "original_max_position_embeddings" in rp:
                d["original_max_position_embeddings"] = rp["original_max_position_embeddings"]
        return cls(**{k: d[k] for k in cls._DEFAULTS if k in d and d[k] is not None})

    @classmethod
    def from_hf(cls, cfg):
        if isinstance(cfg, cls):
            return cfg
        return cls._from_dict(cfg.to_dict() if hasattr(cfg, "to_dict") else dict(vars(cfg)))

    @classmethod
    def from_pretrained(cls, model_name):
        import json
        with open(os.path.join(model_name, "config.json")) as f:
            return cls._from_dict(json.load(f))

    def rope_spec(self, max_position: int):
        """-> (ext_factors | None, cos/sin scale) for a sequence whose largest position id is `max_position`
        (HF 4.47.1 Phi3LongRoPEScaledRotaryEmbedding.forward: seq_len = max(position_ids) + 1 picks the long factors
        beyond original_max_position_embeddings; scale = sqrt(1 + ln(max/orig) / ln(orig)) when max > orig)."""
        rs = self.rope_scaling
        if rs is None:
            return None, 1.0
        orig = int(self.original_max_position_embeddings)
        ext = rs["long_factor"] if max_position + 1 > orig else rs["short_factor"]
        factor = rs.get("factor")
        if factor is None or self.original_max_position_embeddings is not None:
            factor = self.max_position_embeddings / orig
        att = rs.get("attention_factor")
        if att is None:
            att = 1.0 if factor <= 1.0 else math.sqrt(1 + math.log(factor) / math.log(orig))
        return ext, float(att)

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


# ------------------------------------------------------------------------------------------------
# position table (LVM/model.py:86-135) — float64 numpy exactly as the reference builds it
# ------------------------------------------------------------------------------------------------

def _sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    omega = 1.0 / 10000 ** (np.arange(dim // 2, dtype=np.float64) / (dim / 2.0))
    out = np.outer(pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False, extra_tokens=0, interpolation_scale=1.0, base_size=1):
    if isinstance(grid_size, int):
        grid_size = (grid_size, grid_size)
    gh = np.arange(grid_size[0], dtype=np.float32) / (grid_size[0] / base_size) / interpolation_scale
    gw = np.arange(grid_size[1], dtype=np.float32) / (grid_size[1] / base_size) / interpolation_scale
    gx, gy = np.meshgrid(gw, gh)  # w first
    return np.concatenate([_sincos_1d(embed_dim // 2, gx), _sincos_1d(embed_dim // 2, gy)], axis=1)


# ------------------------------------------------------------------------------------------------
# leaf modules (parameter containers with the reference's names)
# ------------------------------------------------------------------------------------------------

class TimestepEmbedder(nn.Module):
    """LVM/model.py:26-63.  forward() = sinusoid kernel + two small-M Linear kernels."""

    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size, bias=True), nn.SiLU(),
                                 nn.Linear(hidden_size, hidden_size, bias=True))
        self.frequency_embedding_size = frequency_embedding_size
        self._freqs = None

    def freqs(self, device):
        if self._freqs is None or self._freqs.device != torch.device(device):
            self._freqs = ops.timestep_freqs(self.frequency_embedding_size, device)
        return self._freqs

    def forward(self, t, dtype=BF16, out=None, out_row=None, ldo=None):
        if dtype != BF16:
            raise VgptError("TimestepEmbedder: the HIP path computes in bf16")
        t = t.to(torch.float32).contiguous()
        emb = ops.timestep_sinusoid(t, self.freqs(t.device))
        h = ops.linear_small(emb, self.mlp[0].weight, self.mlp[0].bias, post_act=ops.ACT_SILU)
        return ops.linear_small(h, self.mlp[2].weight, self.mlp[2].bias, out=out, out_row=out_row, ldo=ldo)


class FinalLayer(nn.Module):
    """LVM/model.py:66-83 (parameters only; the fused kernel is called by LVM)."""

    def __init__(self, hidden_size, patch_size, out_channels):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))


class PatchEmbedMR(nn.Module):
    """LVM/model.py:138-154 (parameters only)."""

    def __init__(self, patch_size=2, in_chans=4, embed_dim=768, bias=True):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)


class Phi3RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x, out=None):
        return ops.rmsnorm(x, self.weight, self.variance_epsilon, out=out)


def rope_tables_for(config, position_ids):
    """cos / sin tables of the checkpoint's rotary variant (plain, or su / longrope: Phi3Config.rope_spec)."""
    ext, scale = config.rope_spec(int(position_ids.max()) if config.rope_scaling is not None else 0)
    inv = ops.rope_inv_freq(config.head_dim, config.rope_theta, position_ids.device, ext)
    return ops.rope_table(position_ids.contiguous(), inv, scale=scale)


class Phi3Attention(nn.Module):
    """Attention module with the reference's operator seam: `local_attn` / `dist_attn`
    (LVM/transform/sdpa_transform.py:162-169).  With the default `local_attn` the fused
    strided path is used; a user-installed callable gets (B,h,S,d) tensors like SDPA."""

    def __init__(self, config: Phi3Config, layer_idx: int):
        super().__init__()
        self.config, self.layer_idx = config, layer_idx
        self.hidden_size = config.hidden_size
        self.num_heads = config.num_attention_heads
        self.num_key_value_heads = config.num_key_value_heads
        self.num_key_value_groups = self.num_heads // self.num_key_value_heads
        self.head_dim = config.head_dim
        self.attention_dropout = config.attention_dropout
        op = self.num_heads * self.head_dim + 2 * self.num_key_value_heads * self.head_dim
        self.qkv_proj = nn.Linear(self.hidden_size, op, bias=False)
        self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
        self.local_attn = ops.sdpa
        self.dist_attn = None

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions=False, use_cache=False, cache_position=None, rope=None, residual=None):
        """-> (attn_output, None, past_key_value), as the reference's `new_forward` (:12-91)."""
        if output_attentions:
            raise VgptError("output_attentions is not supported by the fused attention kernel")
        B, L, _ = hidden_states.shape
        pm = ops.as_packed_mask(attention_mask, hidden_states.device)
        if rope is None:
            rope = rope_tables_for(self.config, position_ids)
        qkv = ops.linear_qkv_rope(hidden_states, self.qkv_proj.weight, rope[0], rope[1], self.num_heads,
                                  self.num_key_value_heads, self.head_dim)
        if self.local_attn is ops.sdpa and self.dist_attn is None:
            ctx = ops.attention_qkv(qkv, pm, self.num_heads, self.num_key_value_heads, self.head_dim)
        else:
            nq, nk, hd = self.num_heads, self.num_key_value_heads, self.head_dim
            q = qkv[..., : nq * hd].view(B, L, nq, hd).transpose(1, 2)
            k = qkv[..., nq * hd:(nq + nk) * hd].view(B, L, nk, hd).transpose(1, 2)
            v = qkv[..., (nq + nk) * hd:].view(B, L, nk, hd).transpose(1, 2)
            if self.dist_attn is None:
                ctx = self.local_attn(q, k, v, attn_mask=pm, dropout_p=0.0, is_causal=False).transpose(1, 2)
            else:   # Ulysses: (B, L/P, heads, d) in and out (LVM/transform/sdpa_transform.py:78-86)
                ctx = self.dist_attn(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), 0, attn_mask=pm,
                                     dropout_p=0.0, is_causal=False)
            ctx = ctx.reshape(B, L, nq * hd).contiguous()
        out = ops.linear(ctx, self.o_proj.weight, residual=residual)
        return out, None, past_key_value


class Phi3MLP(nn.Module):
    def __init__(self, config: Phi3Config):
        super().__init__()
        self.config = config
        self.gate_up_proj = nn.Linear(config.hidden_size, 2 * config.intermediate_size, bias=False)
        self.down_proj = nn.Linear(config.intermediate_size, config.hidden_size, bias=False)
        self.act = ops.act_code(config.hidden_act)

    def forward(self, x, residual=None):
        return ops.linear(ops.gated_mlp_act(x, self.gate_up_proj.weight, self.act), self.down_proj.weight,
                          residual=residual)


class Phi3DecoderLayer(nn.Module):
    def __init__(self, config: Phi3Config, layer_idx: int):
        super().__init__()
        self.self_attn = Phi3Attention(config, layer_idx)
        self.mlp = Phi3MLP(config)
        self.input_layernorm = Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                output_attentions=False, use_cache=False, cache_position=None, rope=None):
        h = self.self_attn(self.input_layernorm(hidden_states), attention_mask, position_ids, rope=rope,
                           residual=hidden_states)[0]
        h = self.mlp(self.post_attention_layernorm(h), residual=h)
        return (h,)


class Phi3Transformer(nn.Module):
    """OmniGen/transformer.py:32-232: embed_tokens / layers / norm with a 3-D mask forward."""

    def __init__(self, config):
        super().__init__()
        config = Phi3Config.from_hf(config)
        self.config = config
        self.padding_idx = config.pad_token_id
        self.vocab_size = config.vocab_size
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Phi3DecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.gradient_checkpointing = False
        std = getattr(config, "initializer_range", 0.02)
        for m in self.modules():
            if isinstance(m, (nn.Linear, nn.Embedding)):
                nn.init.normal_(m.weight, mean=0.0, std=std)
        self._inv_freq = None

    def gradient_checkpointing_enable(self, gradient_checkpointing_kwargs=None):
        """HF PreTrainedModel API the training script calls (train_x1_stage1_noiseinput.py:170-171); read by
        train.Stage1Trainer, which then keeps only layer inputs and recomputes each layer inside its backward."""
        self.gradient_checkpointing = True

    def gradient_checkpointing_disable(self):
        self.gradient_checkpointing = False

    def rope_tables(self, position_ids):
        if self.config.rope_scaling is not None:
            return rope_tables_for(self.config, position_ids)
        dev = position_ids.device
        if self._inv_freq is None or self._inv_freq.device != dev:
            self._inv_freq = ops.rope_inv_freq(self.config.head_dim, self.config.rope_theta, dev)
        return ops.rope_table(position_ids.contiguous(), self._inv_freq)

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                inputs_embeds=None, use_cache=None, output_attentions=None, output_hidden_states=None,
                return_dict=None, cache_position=None, offload_model=False, apply_final_norm=True):
        if (input_ids is None) ^ (inputs_embeds is not None):
            raise ValueError("You must specify exactly one of input_ids or inputs_embeds")
        if inputs_embeds is None:
            inputs_embeds = ops.embed_gather(input_ids.contiguous(), self.embed_tokens.weight)
        if isinstance(attention_mask, ops.PackedMask) or hasattr(attention_mask, "packed_mask"):
            pm = ops.as_packed_mask(attention_mask, inputs_embeds.device)   # packed already, or a layout.TokenLayout
        elif attention_mask is not None and attention_mask.dim() == 3:
            pm = ops.pack_mask(attention_mask)
        else:  # OmniGen/transformer.py:150-151
            raise Exception("attention_mask parameter was unavailable or invalid")
        if past_key_values is not None or use_cache:
            raise VgptError("KV caching is not used on the reference's sampler path (LVM/scheduler.py:174)")
        if offload_model:
            raise VgptError("layer CPU offload is not supported (288 GB HBM holds the model)")
        rope = self.rope_tables(position_ids)
        h = inputs_embeds
        for layer in self.layers:
            h = layer(h, pm, position_ids, rope=rope)[0]
        if apply_final_norm:
            h = self.norm(h)
        return SimpleNamespace(last_hidden_state=h, past_key_values=None, hidden_states=None, attentions=None)


def load_checkpoint_state_dict(ckpt_dir: str) -> dict:
    """model.pt | model.safetensors | pytorch_model.bin (one file) | pytorch_model.bin/ (directory of *.bin shards), in
    the reference's order of preference (LVM/inference/...inference.py:48-68).  torch files are read with
    weights_only=True: nothing from the file is executed."""
    pt, st, bn = (os.path.join(ckpt_dir, n) for n in ("model.pt", "model.safetensors", "pytorch_model.bin"))
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    if os.path.isfile(bn):
        return torch.load(bn, map_location="cpu", weights_only=True)
    if os.path.isdir(bn):
        sd = {}
        for f in sorted(os.listdir(bn)):
            if f.endswith(".bin"):
                sd.update(torch.load(os.path.join(bn, f), map_location="cpu", weights_only=True))
        if sd:
            return sd
    raise FileNotFoundError(f"{ckpt_dir}: no model.pt, model.safetensors or pytorch_model.bin")


# ------------------------------------------------------------------------------------------------
# LVM
# ------------------------------------------------------------------------------------------------

def _index_rows(sizes: Dict[int, list], L: int, which: str):
    """Flatten an index dict {b: [[s,e],...]} (or {b: [tok,...]}) to absolute row numbers b*L + s."""
    rows, lens = [], []
    for b in sizes.keys():
        for item in sizes[b]:
            if which == "span":
                rows.append(b * L + item[0]); lens.append(item[1] - item[0])
            else:
                rows.append(b * L + item); lens.append(1)
    return rows, lens


class LVM(nn.Module):
    """Diffusion model with a Transformer backbone (LVM/model.py:157-566)."""

    zero_init_x_embedder = False

    def __init__(self, transformer_config, patch_size=2, in_channels=4, pe_interpolation: float = 1.0,
                 pos_embed_max_size: int = 192):
        super().__init__()
        transformer_config = Phi3Config.from_hf(transformer_config)
        self.in_channels = in_channels
        self.out_channels = in_channels
        self.patch_size = patch_size
        self.pos_embed_max_size = pos_embed_max_size
        hidden_size = transformer_config.hidden_size
        self.hidden_size = hidden_size
        self.x_embedder = PatchEmbedMR(patch_size, in_channels, hidden_size, bias=True)
        self.input_x_embedder = PatchEmbedMR(patch_size, in_channels, hidden_size, bias=True)
        self.time_token = TimestepEmbedder(hidden_size)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.pe_interpolation = pe_interpolation
        pos_embed = get_2d_sincos_pos_embed(hidden_size, pos_embed_max_size, interpolation_scale=pe_interpolation,
                                            base_size=64)
        self.register_buffer("pos_embed", torch.from_numpy(pos_embed).float().unsqueeze(0), persistent=True)
        self.final_layer = FinalLayer(hidden_size, patch_size, self.out_channels)
        self.initialize_weights()
        self.llm = Phi3Transformer(transformer_config)
        self.llm.config.use_cache = False
        self._plan_cache = {}

    # -- construction / checkpoints --
    @classmethod
    def from_pretrained(cls, model_name, load_llm_ckpt=True):
        """Local checkpoint directories only (no network): config.json + one of the layouts the reference's
        inference script accepts (LVM/inference/...inference.py:48-68)."""
        if not os.path.exists(model_name):
            raise FileNotFoundError(f"{model_name}: hub download is unavailable offline; pass a local directory")
        model = cls(Phi3Config.from_pretrained(model_name))
        if load_llm_ckpt:
            model.load_state_dict(load_checkpoint_state_dict(model_name))
        return model

    def initialize_weights(self):
        """LVM/model.py:213-244 (LVMTraining zero-initialises x_embedder instead, :639-640)."""
        def _basic_init(module):
            if isinstance(module, nn.Linear):
                nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
        self.apply(_basic_init)
        for emb in (self.x_embedder, self.input_x_embedder):
            w = emb.proj.weight.data
            nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
            nn.init.constant_(emb.proj.bias, 0)
        if self.zero_init_x_embedder:
            nn.init.constant_(self.x_embedder.proj.weight.data, 0)
        for emb in (self.t_embedder, self.time_token):
            nn.init.normal_(emb.mlp[0].weight, std=0.02)
            nn.init.normal_(emb.mlp[2].weight, std=0.02)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    def init_input_final_layer(self):
        self.input_final_layer = nn.Linear(self.hidden_size, self.patch_size * self.patch_size * self.out_channels, bias=True)
        nn.init.constant_(self.input_final_layer.weight, 0)
        nn.init.constant_(self.input_final_layer.bias, 0)

    # -- small helpers kept for interface parity --
    def unpatchify(self, x, h, w):
        """(N, T, p*p*C) -> (N, C, H, W); pure index shuffle (LVM/model.py:255-265)."""
        c, p = self.out_channels, self.patch_size
        x = x.reshape(x.shape[0], h // p, w // p, p, p, c)
        return x.permute(0, 5, 1, 3, 2, 4).reshape(x.shape[0], c, h, w)

    def cropped_pos_embed(self, height, width):
        if self.pos_embed_max_size is None:
            raise ValueError("`pos_embed_max_size` must be set for cropping.")
        height, width = height // self.patch_size, width // self.patch_size
        if height > self.pos_embed_max_size:
            raise ValueError(f"Height ({height}) cannot be greater than `pos_embed_max_size`: {self.pos_embed_max_size}.")
        if width > self.pos_embed_max_size:
            raise ValueError(f"Width ({width}) cannot be greater than `pos_embed_max_size`: {self.pos_embed_max_size}.")
        top, left = (self.pos_embed_max_size - height) // 2, (self.pos_embed_max_size - width) // 2
        pe = self.pos_embed.reshape(1, self.pos_embed_max_size, self.pos_embed_max_size, -1)
        return pe[:, top:top + height, left:left + width, :].reshape(1, -1, pe.shape[-1])

    def release_engines(self):
        """Drop the sampler engines LVMScheduler keeps on this model between clips of the same sequence (a few GB of
        per-clip buffers each at 256^2 / 12 frames; scheduler.LVMScheduler.cache_engines)."""
        self.__dict__.pop("_vgpt_engine_cache", None)

    def __getstate__(self):
        # copy.deepcopy(model) (how the reference makes its EMA: train_x1_stage1_noiseinput.py:229), pickling and torch.save(model)
        # go through here: the cached sampler engines -- GBs of per-clip buffers, a hipGraph, a reference back to this model --
        # are not part of the module's state
        state = self.__dict__.copy()
        state.pop("_vgpt_engine_cache", None)
        return state

    def _check_ready(self):
        w = self.llm.norm.weight
        if not w.is_cuda or w.dtype != BF16:
            raise VgptError("LVM runs on the MI355X HIP path only: call model.to('cuda', torch.bfloat16) first")

    def _stack(self, latents):
        """list of (1,C,h,w) -> groups of equal shape: [(indices, (n,C,h,w) tensor)]."""
        if torch.is_tensor(latents):
            return [(list(range(latents.shape[0])), latents.to(BF16).contiguous())]
        groups: Dict[tuple, list] = {}
        for i, t in enumerate(latents):
            groups.setdefault(tuple(t.shape[-2:]), []).append(i)
        return [(idx, torch.cat([latents[i].to(BF16) for i in idx], dim=0).contiguous()) for idx in groups.values()]

    def _i32(self, values, device):
        return torch.tensor(values, dtype=torch.int32, device=device)

    def _plan(self, key, builder):
        if key not in self._plan_cache:
            if len(self._plan_cache) > 64:
                self._plan_cache.clear()
            self._plan_cache[key] = builder()
        return self._plan_cache[key]

    # -- the hot path --
    def assemble_sequence(self, x, timestep, input_ids, input_img_latents, input_image_sizes, denoise_image_sizes,
                          time_emb_inx, seq=None):
        """Token-embedding gather + patch embeds + time tokens scattered into (B, L, H) (LVM/model.py:419-454)."""
        B, L = input_ids.shape
        dev = input_ids.device
        H = self.hidden_size
        seq = ops.embed_gather(input_ids.contiguous(), self.llm.embed_tokens.weight, out=seq)
        seq2d = seq.view(B * L, H)
        pos = self.pos_embed[0]
        if input_img_latents is not None and len(input_img_latents) > 0:
            rows, lens = _index_rows(input_image_sizes, L, "span")
            if len(rows) != len(input_img_latents):
                raise AssertionError("input_image_sizes and input_img_latents disagree")
            for idx, stacked in self._stack(input_img_latents):
                dst = self._plan(("in", tuple(rows[i] for i in idx)), lambda: self._i32([rows[i] for i in idx], dev))
                ops.patch_embed(stacked, self.input_x_embedder.proj.weight, self.input_x_embedder.proj.bias, pos, dst,
                                seq2d, self.pos_embed_max_size)
        trows, _ = _index_rows(time_emb_inx, L, "tok")
        if len(trows) != timestep.shape[0]:
            raise AssertionError("time_emb_inx and timestep disagree")
        tdst = self._plan(("t", tuple(trows)), lambda: self._i32(trows, dev))
        self.time_token(timestep, out=seq2d, out_row=tdst, ldo=H)
        rows, lens = _index_rows(denoise_image_sizes, L, "span")
        n_x = len(x) if not torch.is_tensor(x) else x.shape[0]
        if len(rows) != n_x:
            raise AssertionError("denoise_image_sizes and x disagree")
        shapes = [None] * n_x
        for idx, stacked in self._stack(x):
            dst = self._plan(("x", tuple(rows[i] for i in idx)), lambda: self._i32([rows[i] for i in idx], dev))
            ops.patch_embed(stacked, self.x_embedder.proj.weight, self.x_embedder.proj.bias, pos, dst, seq2d,
                            self.pos_embed_max_size)
            for i in idx:
                shapes[i] = tuple(stacked.shape[-2:])
        return seq, rows, shapes

    def decode_frames(self, hidden, timestep, rows, shapes, out=None):
        """t_embedder -> adaLN -> fused final layer + unpatchify per denoised frame (LVM/model.py:478-486)."""
        dev = hidden.device
        H = self.hidden_size
        hidden2d = hidden.view(-1, H)
        time_emb = self.t_embedder(timestep)
        ada = self.final_layer.adaLN_modulation[1]
        mod = ops.linear_small(time_emb, ada.weight, ada.bias, pre_act=ops.ACT_SILU)
        n = len(rows)
        groups: Dict[tuple, list] = {}
        for i, s in enumerate(shapes):
            groups.setdefault(s, []).append(i)
        outs = [None] * n
        for (h, w), idx in groups.items():
            if out is not None and len(groups) == 1:
                buf = out
            else:
                buf = torch.empty(len(idx), self.out_channels, h, w, dtype=BF16, device=dev)
            src = self._plan(("f", tuple(rows[i] for i in idx)), lambda: self._i32([rows[i] for i in idx], dev))
            m = mod if len(groups) == 1 else mod[idx].contiguous()
            ops.final_layer(hidden2d, src, m, self.final_layer.linear.weight, self.final_layer.linear.bias, buf)
            for j, i in enumerate(idx):
                outs[i] = buf[j:j + 1]
        return outs

    def frame_block_forward(self, x, timestep, input_ids, input_img_latents, input_image_sizes, attention_mask,
                            position_ids, denoise_image_sizes, time_emb_inx, padding_latent=None, past_key_values=None,
                            return_past_key_values=True, offload_model: bool = False, vae=None,
                            input_output_return=False, out=None):
        """LVM/model.py:399-501 at sequence-parallel size 1."""
        self._check_ready()
        if padding_latent is not None and not (isinstance(padding_latent, (list, tuple)) and all(p_ is None for p_ in padding_latent)):
            # LVM/model.py:292-309 appends `padding` tokens behind a frame's patch tokens; in THIS entry the frame's span then
            # holds more than (h/p)(w/p) rows and the reference's own unpatchify (LVM/model.py:262) fails on the reshape.  A list
            # of None (what patch_multiple_resolutions expands None to) is the only value the reference can run with.
            raise VgptError("frame_block_forward: padding_latent entries other than None are not runnable in the reference "
                            "either (unpatchify reshape, LVM/model.py:262)")
        assert input_ids is not None, "input_ids is None"
        seq, rows, shapes = self.assemble_sequence(x, timestep, input_ids, input_img_latents, input_image_sizes,
                                                   denoise_image_sizes, time_emb_inx)
        from . import sequence_parallel as SP
        if SP.sp_world() > 1:   # LVM/model.py:457-473: every rank runs its L/P slice, the last hidden state is gathered
            seq_l, pos_l = SP.shard_sequence(seq, position_ids)
            output = self.llm(inputs_embeds=seq_l, attention_mask=attention_mask, position_ids=pos_l,
                              past_key_values=past_key_values, offload_model=offload_model)
            output.last_hidden_state = SP.gather_sequence(output.last_hidden_state)
        else:
            output = self.llm(inputs_embeds=seq, attention_mask=attention_mask, position_ids=position_ids,
                              past_key_values=past_key_values, offload_model=offload_model)
        latents = self.decode_frames(output.last_hidden_state, timestep, rows, shapes, out=out)
        if input_output_return:
            return latents, self.decode_input_frames(output.last_hidden_state, input_img_latents, input_image_sizes)
        if return_past_key_values:
            return latents, None
        return latents

    def decode_input_frames(self, hidden, input_img_latents, input_image_sizes):
        """LVM/model.py:488-497: the `input_final_layer` head (a plain Linear on the last hidden state, no modulation) on the
        rows of every CONDITION frame, unpatchified -- what `input_output_return=True` adds to the outputs."""
        head = self.input_final_layer        # AttributeError without init_input_final_layer(), as in the reference
        B, L, H = hidden.shape
        hidden2d = hidden.view(-1, H)
        rows, lens = _index_rows(input_image_sizes, L, "span")
        n = 0 if input_img_latents is None else len(input_img_latents)
        if len(rows) != n:
            raise AssertionError("input_image_sizes and input_img_latents disagree")
        preds = []
        for i in range(n):
            h, w = input_img_latents[i].shape[-2:]
            y = ops.linear_small(hidden2d[rows[i]:rows[i] + lens[i]], head.weight, head.bias)
            preds.append(self.unpatchify(y[None], h, w).contiguous())
        return preds

    @torch.no_grad()
    def frame_block_forward_with_cfg(self, x, timestep, input_ids, input_img_latents, input_image_sizes,
                                     attention_mask, position_ids, denoise_image_sizes, time_emb_inx, use_img_cfg,
                                     img_cfg_scale, past_key_values, use_kv_cache, offload_model, vae,
                                     prediction_type: str = "v"):
        """LVM/model.py:519-566: CFG is applied here only for 'v' predictions."""
        model_out, past_key_values = self.frame_block_forward(
            x, timestep, input_ids, input_img_latents, input_image_sizes, attention_mask, position_ids,
            denoise_image_sizes, time_emb_inx, past_key_values=past_key_values, return_past_key_values=True,
            offload_model=offload_model, vae=vae)
        if use_img_cfg and prediction_type == "v":
            half = len(model_out) // 2
            stacked = torch.cat(model_out, dim=0)
            v = torch.zeros(stacked.shape, dtype=torch.float32, device=stacked.device)
            # z=0, sigma=(0,1): one Euler-kernel call evaluates uncond + s*(cond-uncond)
            sig = self._plan(("sig01", stacked.device), lambda: torch.tensor([0.0, 1.0], device=stacked.device))
            st = self._plan(("step0", stacked.device), lambda: torch.zeros(1, dtype=torch.int32, device=stacked.device))
            vm = torch.empty_like(stacked)
            ops.euler_cfg_update(v, vm, stacked, sig, st, ops.PRED_V, True, img_cfg_scale)
            model_out = [vm[i:i + 1] for i in range(half)] * 2
        return model_out, past_key_values

    def forward(self, x, timestep, input_ids, input_img_latents, input_image_sizes, attention_mask, position_ids,
                padding_latent=None, past_key_values=None, return_past_key_values=True, offload_model: bool = False):
        """Single-target variant (LVM/model.py:330-397): sequence = [condition tokens | time token | x tokens],
        prediction read from the last N positions.  Same kernels as the frame-block path: the sequence is the
        frame-block layout with one noisy block per batch row at the end.  (The reference's unconditional
        `dist.all_gather` at world size 1 is the identity and is not issued.)"""
        self._check_ready()
        if padding_latent is not None or isinstance(x, (list, tuple)):
            raise VgptError("LVM.forward: mixed-resolution lists / padding latents are not supported on the HIP path")
        B, C, h, w = x.shape
        N = (h // self.patch_size) * (w // self.patch_size)
        dev = x.device
        Lc = 0 if input_ids is None else input_ids.shape[1]
        L = Lc + 1 + N
        pad = self.llm.config.pad_token_id if self.llm.config.pad_token_id is not None else 0
        pad = min(max(int(pad), 0), self.llm.vocab_size - 1)
        ids = torch.full((B, L), pad, dtype=torch.int64, device=dev)
        if input_ids is not None:
            ids[:, :Lc] = input_ids
        denoise = {b: [[Lc + 1, L]] for b in range(B)}
        time_inx = {b: [Lc] for b in range(B)}
        sizes = input_image_sizes if input_img_latents is not None else {}
        lat = input_img_latents if input_img_latents is not None and len(input_img_latents) > 0 else None
        seq, rows, shapes = self.assemble_sequence(x, timestep, ids, lat, sizes, denoise, time_inx)
        output = self.llm(inputs_embeds=seq, attention_mask=attention_mask, position_ids=position_ids,
                          past_key_values=past_key_values, offload_model=offload_model)
        frames = self.decode_frames(output.last_hidden_state, timestep, rows, shapes)
        latents = torch.cat(frames, dim=0)
        if return_past_key_values:
            return latents, None
        return latents

    @torch.no_grad()
    def forward_with_cfg(self, x, timestep, input_ids, input_img_latents, input_image_sizes, attention_mask, position_ids,
                         use_img_cfg, img_cfg_scale, past_key_values, use_kv_cache, offload_model,
                         prediction_type: str = "v"):
        """LVM/model.py:504-516: batch = [cond ; uncond]; CFG applied here only for 'v' predictions."""
        model_out, past_key_values = self.forward(x, timestep, input_ids, input_img_latents, input_image_sizes,
                                                  attention_mask, position_ids, past_key_values=past_key_values,
                                                  return_past_key_values=True, offload_model=offload_model)
        if use_img_cfg and prediction_type == "v":
            n = model_out.shape[0]
            v = torch.zeros(model_out.shape, dtype=torch.float32, device=model_out.device)
            sig = self._plan(("sig01", model_out.device), lambda: torch.tensor([0.0, 1.0], device=model_out.device))
            st = self._plan(("step0", model_out.device), lambda: torch.zeros(1, dtype=torch.int32, device=model_out.device))
            vm = torch.empty_like(model_out)
            ops.euler_cfg_update(v.view(n, -1), vm.view(n, -1), model_out.contiguous().view(n, -1), sig, st, ops.PRED_V,
                                 True, img_cfg_scale)
            model_out = torch.cat([vm[: n // 2], vm[: n // 2]], dim=0)
        return model_out, past_key_values


class LVMTraining(LVM):
    """LVM/model.py:569-845: same modules, x_embedder zero-initialised, forward = frame_block_forward body."""

    zero_init_x_embedder = True

    def forward(self, x, timestep, input_ids, input_img_latents, input_image_sizes, attention_mask, position_ids,
                denoise_image_sizes, time_emb_inx, padding_latent=None, past_key_values=None,
                return_past_key_values=True, offload_model: bool = False, vae=None, input_output_return=False):
        return self.frame_block_forward(x, timestep, input_ids, input_img_latents, input_image_sizes, attention_mask,
                                        position_ids, denoise_image_sizes, time_emb_inx, padding_latent, past_key_values,
                                        return_past_key_values, offload_model, vae, input_output_return)


LVMTraining_CP = LVMTraining
