"""Run the reference's own pure-torch leaf classes in the build container.  TEST INFRASTRUCTURE.

The reference package cannot be imported here (`import LVM` -> ModuleNotFoundError: diffusers;
SURVEY.md §8c), but the classes that pin this path use only torch / numpy / math / re.  This
module parses the reference files with `ast`, keeps the named top-level definitions and executes
them in a fresh namespace.  It READS /root/reference at run time, so it is used only by
`tests/make_golden.py` (fixture generation) and by CPU tests that skip when the reference checkout
is absent (it never travels to the GPU box).  No reference source text is stored in this repo.
"""
from __future__ import annotations

import ast
import copy
import gc
import math
import os
import re
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("VGPT_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "LVM", "processor.py"))


def _extract(relpath: str, names, extra_ns=None):
    path = os.path.join(REFERENCE_ROOT, relpath)
    with open(path, "r", encoding="utf-8") as f:
        tree = ast.parse(f.read(), filename=path)
    keep = [n for n in tree.body
            if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in names]
    missing = set(names) - {n.name for n in keep}
    if missing:
        raise RuntimeError(f"{relpath}: definitions not found: {sorted(missing)}")
    mod = ast.Module(body=keep, type_ignores=[])
    from typing import Dict, List, Optional, Tuple, Union
    ns = {"torch": torch, "nn": torch.nn, "np": np, "math": math, "re": re, "copy": copy, "gc": gc,
          "tqdm": (lambda it, *a, **k: it), "Dict": Dict, "List": List, "Optional": Optional,
          "Tuple": Tuple, "Union": Union, "__name__": f"reference:{relpath}"}
    if extra_ns:
        ns.update(extra_ns)
    exec(compile(mod, path, "exec"), ns)
    return types.SimpleNamespace(**{n: ns[n] for n in names})


def collator_classes():
    """LVMCollator (LVM/processor.py:426-1000) and the prompt-layout methods of LVMProcessor (:128-274)."""
    ns = _extract("LVM/processor.py", ["LVMCollator", "LVMProcessor"],
                  extra_ns={"PreTrainedTokenizer": object, "Image": None, "transforms": None,
                            "InterpolationMode": None, "crop_arr": None, "snapshot_download": None,
                            "AutoTokenizer": None, "os": os, "logging": None})
    return ns


def scheduler_class():
    """LVMScheduler (LVM/scheduler.py:119-208)."""
    return _extract("LVM/scheduler.py", ["LVMScheduler"], extra_ns={"DynamicCache": None}).LVMScheduler


def model_leaf_classes():
    """modulate, TimestepEmbedder, FinalLayer, sincos helpers, PatchEmbedMR (LVM/model.py:22-154)."""
    return _extract("LVM/model.py", ["modulate", "TimestepEmbedder", "FinalLayer", "get_2d_sincos_pos_embed",
                                     "get_2d_sincos_pos_embed_from_grid", "get_1d_sincos_pos_embed_from_grid",
                                     "PatchEmbedMR"])


class StubTokenizer:
    """Stands in for the (unavailable) Phi-3 tokenizer: one id per special tag, BOS=1 in front of
    every chunk exactly like a llama-family tokenizer (the reference strips it, processor.py:140-142)."""

    TABLE = {"<img>": 10, "</img>": 11, "<|diffusion|>": 12}

    def __call__(self, text):
        ids = [1]
        pos = 0
        while pos < len(text):
            for tag, tid in self.TABLE.items():
                if text.startswith(tag, pos):
                    ids.append(tid)
                    pos += len(tag)
                    break
            else:
                raise ValueError(f"StubTokenizer: unexpected text {text[pos:pos + 20]!r}")
        return types.SimpleNamespace(input_ids=ids)


def reference_inference_batch(C: int, G: int, N: int, use_cfg: bool = True, sp: int = 1):
    """Run the reference's prompt layout + collator for one next-clip request
    (the prompt strings are built as LVM/pipeline.py:426-448 does)."""
    ns = collator_classes()
    side = int(round(math.sqrt(N))) * 16
    assert (side // 16) ** 2 == N, "N must be a square number of tokens"
    proc = types.SimpleNamespace(text_tokenizer=StubTokenizer())
    layout = ns.LVMProcessor.process_multi_modal_prompt_frame_block
    prompt = "".join(f"<img><|image_{i + 1}|></img>" if i < C else f"<|diffusion|><|image_{i + 1}|>"
                     for i in range(C + G))
    images = [torch.zeros(3, side, side) for _ in range(C)]
    rows = []
    r0 = layout(proc, prompt, images, [C, G])
    r0["frame_blocks"] = [C, G]
    rows.append(r0)
    if use_cfg:
        prompt_ = "".join(f"<|diffusion|><|image_{i + 1}|>" for i in range(G))
        r1 = layout(proc, prompt_, None, [0, G], height=side, width=side)
        r1["frame_blocks"] = [0, G]
        rows.append(r1)
    coll = ns.LVMCollator(pad_token_id=2, hidden_size=8, sequence_parallel_size=sp)
    return coll.process_mllm_input_frame_block_call(rows)


def reference_stage1_batch(F_list, N: int, sp: int = 1):
    """Stage-1 layout: prompt as LVM/train_helper/data.py:203-215, collated as TrainDataCollator (:422-458)."""
    ns = collator_classes()
    side = int(round(math.sqrt(N))) * 16
    proc = types.SimpleNamespace(text_tokenizer=StubTokenizer())
    layout = ns.LVMProcessor.process_multi_modal_prompt_training
    rows = []
    for F_ in F_list:
        prompt = "".join(
            f"<|diffusion|><|image_{i + 1}|><img><|image_{i + 1}|></img>" if i < F_ - 1 else f"<|diffusion|><|image_{i + 1}|>"
            for i in range(F_))
        rows.append(layout(proc, prompt, [torch.zeros(3, side, side) for _ in range(F_)]))
    coll = ns.LVMCollator(pad_token_id=2, hidden_size=8, sequence_parallel_size=sp)
    ids, pos, mask, pixel_values, sizes = coll.process_mllm_input_training(rows, block_aware=False)
    den = {b: [s for i, s in enumerate(sizes[b]) if i % 2 == 0] for b in sizes}
    inp = {b: [s for i, s in enumerate(sizes[b]) if i % 2 == 1] for b in sizes}
    tix = {b: [s[0] - 1 for s in den[b]] for b in sizes}
    return dict(input_ids=ids, position_ids=pos, attention_mask=mask, input_image_sizes=inp,
                denoise_image_sizes=den, time_emb_inx=tix)


def reference_frame_block_training_batch(frame_blocks_list, N: int, sp: int = 1):
    """Stage-2+ layout: prompt as LVM/train_helper/data.py:358-380, collated by
    process_mllm_input_frame_block_training (LVM/processor.py:893-914)."""
    ns = collator_classes()
    side = int(round(math.sqrt(N))) * 16
    proc = types.SimpleNamespace(text_tokenizer=StubTokenizer())
    layout = ns.LVMProcessor.process_multi_modal_prompt_frame_block_training
    rows = []
    for fbs in frame_blocks_list:
        prompt, i, j, n_img = "", 0, 0, 0
        for k, fb in enumerate(fbs):
            for _ in range(fb):
                prompt += f"<|diffusion|><|image_{i + 1}|>"; i += 1; n_img += 1
            if k != len(fbs) - 1:
                for _ in range(fb):
                    prompt += f"<img><|image_{j + 1}|></img>"; j += 1
        row = layout(proc, prompt, [torch.zeros(3, side, side) for _ in range(n_img)], fbs)
        row["frame_blocks"] = fbs
        rows.append(row)
    coll = ns.LVMCollator(pad_token_id=2, hidden_size=8, sequence_parallel_size=sp)
    ids, pos, mask, pixel_values, sizes, fb = coll.process_mllm_input_frame_block_training(rows)
    return dict(input_ids=ids, position_ids=pos, attention_mask=mask, image_sizes=sizes, frame_blocks=fb)


# ------------------------------------------------------------------------------------------------
# LVM-owned glue executed from the reference's own source (SURVEY §8a rows 3,4,5,9,10,11,15,16):
# LVM / LVMTraining (LVM/model.py:157-566, 569-857), Phi3Transformer.forward (OmniGen/transformer.py:35-232),
# new_forward / new_attn_forward / replace_attention (LVM/transform/sdpa_transform.py:12-169),
# training_losses_x1_noise_input + samplers (LVM/train_helper/loss.py:73-250), broadcast_data (LVM/utils.py:177-311).
#
# What is NOT the reference here (third-party pins absent from the container):
#   * transformers==4.47.1 Phi3RMSNorm / Phi3MLP / Phi3RotaryEmbedding / apply_rotary_pos_emb / repeat_kv come from the
#     INSTALLED transformers (5.x: same formulas; adapters below only restore the 4.47.1 call signatures the reference
#     uses), and the 4.47.1 decoder-layer glue `x + attn(norm(x))`, `h + mlp(norm(h))` is `_DecoderLayer447`;
#   * deepspeed==0.15.2 DistributedAttention / _SeqAllToAll are world-size-1 identities (`_DistributedAttention`,
#     `_SeqAllToAllIdentity`); torch.distributed is the real one on a 1-rank gloo group.
# ------------------------------------------------------------------------------------------------

def _single_rank_group():
    """1-rank gloo group: LVM.forward all-gathers unconditionally (LVM/model.py:371-377) and the loss
    broadcasts its noise (LVM/train_helper/loss.py:150,168-172), so the reference needs a process group."""
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", store=dist.HashStore(), rank=0, world_size=1)
    return dist


def _hf():
    from transformers import Phi3Config
    from transformers.models.phi3 import modeling_phi3 as M
    return Phi3Config, M


class _SeqAllToAllIdentity:
    """deepspeed.sequence.layer._SeqAllToAll at sequence-parallel size 1: the all-to-all is the identity."""

    @staticmethod
    def apply(group, x, scatter_idx, gather_idx, batch_dim_idx, stream, handles, kind):
        return x


class _DistributedAttention(torch.nn.Module):
    """Attribute surface of deepspeed 0.15.2 DistributedAttention that new_attn_forward reads."""

    def __init__(self, local_attention, sequence_process_group, scatter_idx=2, gather_idx=1, sp_stream=None):
        super().__init__()
        self.local_attn = local_attention
        self.spg = sequence_process_group
        self.scatter_idx, self.gather_idx = scatter_idx, gather_idx
        self.sp_overlap_comm = False
        self.overlap_handles = None
        self.sp_stream = sp_stream

    def layer_sync(self, layer):
        return None


def hf_config(cfg):
    """Installed-transformers Phi3Config for an oracle.restate.Phi3Cfg."""
    Phi3Config, _ = _hf()
    return Phi3Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_key_value_heads, rms_norm_eps=cfg.rms_norm_eps,
                      hidden_act=cfg.hidden_act, pad_token_id=cfg.pad_token_id, eos_token_id=None, bos_token_id=None,
                      rope_parameters={"rope_theta": cfg.rope_theta, "rope_type": "default"})


def lvm_reference_classes():
    """Returns a namespace with the reference's LVM, LVMTraining, Phi3Transformer, replace_attention,
    training_losses_x1_noise_input (and hccl_info) — reference code, executable on CPU at world size 1."""
    Phi3Config, M = _hf()
    from transformers.modeling_outputs import BaseModelOutputWithPast
    from transformers.cache_utils import Cache, DynamicCache
    from transformers.utils import logging as hf_logging
    from types import MethodType
    from typing import Any
    import random
    dist = _single_rank_group()
    hccl_info = types.SimpleNamespace(world_size=1, rank=0, group=None)
    nn = torch.nn

    class _Rotary447(nn.Module):
        """4.47.1 call form `rotary_emb(x, position_ids, seq_len=None)` over the installed Phi3RotaryEmbedding."""

        def __init__(self, config):
            super().__init__()
            self.inner = M.Phi3RotaryEmbedding(config)

        def forward(self, x, position_ids, seq_len=None):
            return self.inner(x, position_ids)

    def apply_rotary_pos_emb(q, k, cos, sin, position_ids=None, unsqueeze_dim=1):
        """4.47.1 signature (position_ids is deprecated and unused there as well)."""
        return M.apply_rotary_pos_emb(q, k, cos, sin, unsqueeze_dim=unsqueeze_dim)

    class Phi3SdpaAttention(nn.Module):
        """Module attributes of transformers 4.47.1 Phi3Attention that new_forward reads; no forward of its own."""

        def __init__(self, config, layer_idx):
            super().__init__()
            self.config, self.layer_idx = config, layer_idx
            self.attention_dropout = config.attention_dropout
            self.hidden_size = config.hidden_size
            self.num_heads = config.num_attention_heads
            self.head_dim = self.hidden_size // self.num_heads
            self.num_key_value_heads = config.num_key_value_heads
            self.num_key_value_groups = self.num_heads // self.num_key_value_heads
            op_size = self.num_heads * self.head_dim + 2 * (self.num_key_value_heads * self.head_dim)
            self.o_proj = nn.Linear(self.num_heads * self.head_dim, self.hidden_size, bias=False)
            self.qkv_proj = nn.Linear(self.hidden_size, op_size, bias=False)
            self.rotary_emb = _Rotary447(config)

    class _DecoderLayer447(nn.Module):
        """transformers 4.47.1 Phi3DecoderLayer.forward (third-party glue, restated): returns a tuple."""

        def __init__(self, config, layer_idx):
            super().__init__()
            self.self_attn = Phi3SdpaAttention(config, layer_idx)
            self.mlp = M.Phi3MLP(config)
            self.input_layernorm = M.Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
            self.post_attention_layernorm = M.Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

        def forward(self, hidden_states, attention_mask=None, position_ids=None, past_key_value=None,
                    output_attentions=False, use_cache=False, cache_position=None):
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
            attn, _, present = self.self_attn(hidden_states=hidden_states, attention_mask=attention_mask,
                                              position_ids=position_ids, past_key_value=past_key_value,
                                              output_attentions=output_attentions, use_cache=use_cache,
                                              cache_position=cache_position)
            hidden_states = residual + attn
            residual = hidden_states
            hidden_states = self.mlp(self.post_attention_layernorm(hidden_states))
            return (residual + hidden_states,)

    class Phi3Model(nn.Module):
        """Module tree of transformers Phi3Model (embed_tokens, layers, norm) without its forward."""

        def __init__(self, config):
            super().__init__()
            self.config = config
            self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size, config.pad_token_id)
            self.layers = nn.ModuleList([_DecoderLayer447(config, i) for i in range(config.num_hidden_layers)])
            self.norm = M.Phi3RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
            self.gradient_checkpointing = False
            self._gradient_checkpointing_func = (
                lambda fn, *a: torch.utils.checkpoint.checkpoint(fn, *a, use_reentrant=False))

    tr = _extract("OmniGen/transformer.py", ["Phi3Transformer"],
                  extra_ns={"Phi3Model": Phi3Model, "torch_npu": None, "logger": hf_logging.get_logger("reference"),
                            "Cache": Cache, "DynamicCache": DynamicCache,
                            "BaseModelOutputWithPast": BaseModelOutputWithPast})
    sd = _extract("LVM/transform/sdpa_transform.py", ["new_forward", "new_attn_forward", "replace_attention"],
                  extra_ns={"Phi3SdpaAttention": Phi3SdpaAttention, "DistributedAttention": _DistributedAttention,
                            "_SeqAllToAll": _SeqAllToAllIdentity, "hccl_info": hccl_info, "MethodType": MethodType,
                            "apply_rotary_pos_emb": apply_rotary_pos_emb, "repeat_kv": M.repeat_kv, "Tensor": torch.Tensor,
                            "Any": Any, "Cache": Cache, "logger": hf_logging.get_logger("reference")})

    class PeftAdapterMixin:  # diffusers.loaders.PeftAdapterMixin: LoRA plumbing only, no arithmetic
        pass

    from safetensors.torch import load_file
    md = _extract("LVM/model.py", ["modulate", "TimestepEmbedder", "FinalLayer", "get_2d_sincos_pos_embed",
                                   "get_2d_sincos_pos_embed_from_grid", "get_1d_sincos_pos_embed_from_grid",
                                   "PatchEmbedMR", "LVM", "LVMTraining"],
                  extra_ns={"PeftAdapterMixin": PeftAdapterMixin, "Phi3Config": Phi3Config,
                            "Phi3Transformer": tr.Phi3Transformer, "hccl_info": hccl_info, "dist": dist, "os": os,
                            "snapshot_download": None, "load_file": load_file})
    ut = _extract("LVM/utils.py", ["_broadcast_tensor", "is_numeric_sequence", "broadcast_data"],
                  extra_ns={"dist": dist})
    ls = _extract("LVM/train_helper/loss.py",
                  ["sample_x0", "sample_timestep", "sample_exp_timestep", "sample_frame_block_timestep",
                   "sample_timestep_max_noise", "training_losses_x1_noise_input", "mean_flat"],
                  extra_ns={"dist": dist, "hccl_info": hccl_info, "broadcast_data": ut.broadcast_data, "random": random})
    return types.SimpleNamespace(LVM=md.LVM, LVMTraining=md.LVMTraining, Phi3Transformer=tr.Phi3Transformer,
                                 replace_attention=sd.replace_attention, new_forward=sd.new_forward,
                                 training_losses_x1_noise_input=ls.training_losses_x1_noise_input,
                                 sample_frame_block_timestep=ls.sample_frame_block_timestep, hccl_info=hccl_info)


def build_reference_model(cfg, params, cls_name="LVM"):
    """Reference LVM / LVMTraining with the oracle's seeded parameters (same state_dict keys, strict) and the
    reference's own attention seam installed (replace_attention, LVM/inference/...inference.py:70)."""
    ns = lvm_reference_classes()
    model = getattr(ns, cls_name)(hf_config(cfg), patch_size=cfg.patch_size, in_channels=cfg.in_channels,
                                  pe_interpolation=cfg.pe_interpolation, pos_embed_max_size=cfg.pos_embed_max_size)
    sd = {k: v for k, v in params.items()}
    if "input_final_layer.weight" in sd:
        model.init_input_final_layer()            # LVM/model.py:246-253: the optional head of input_output_return
    own = model.state_dict()
    extra = [k for k in own if k not in sd]
    # the only keys the oracle dict lacks are the rotary inv_freq buffers of the installed transformers module
    assert all("rotary_emb" in k for k in extra), extra
    model.load_state_dict(sd, strict=False)
    assert not [k for k in sd if k not in own], [k for k in sd if k not in own]
    ns.replace_attention(model.llm)
    return model.eval(), ns


def reference_single_target_batch(n_images: int, side: int, out_hw, use_cfg: bool = True, sp: int = 1, via="call"):
    """LVMProcessor.__call__ / prompt_condition_inference (LVM/processor.py:282-364) + LVMCollator.__call__ (:943-962) for
    one instruction `<img><|image_i|></img>...` with `n_images` condition images of side x side pixels (what
    LVMPipeline.__call__ builds, LVM/pipeline.py:220-238)."""
    ns = collator_classes()
    proc = types.SimpleNamespace(text_tokenizer=StubTokenizer(), process_image=lambda x: x)
    P = ns.LVMProcessor
    proc.add_prefix_instruction = lambda prompt: P.add_prefix_instruction(proc, prompt)
    proc.process_multi_modal_prompt = lambda text, imgs: P.process_multi_modal_prompt(proc, text, imgs)
    proc.collator = ns.LVMCollator(pad_token_id=2, hidden_size=8, sequence_parallel_size=sp)
    prompt = "".join(f"<img><|image_{i + 1}|></img>" for i in range(n_images))
    images = [torch.zeros(3, side, side) for _ in range(n_images)] if n_images else None
    if via == "call":
        return P.__call__(proc, [prompt], [images] if images is not None else None, height=out_hw[0], width=out_hw[1],
                          use_img_cfg=use_cfg)
    return P.prompt_condition_inference(proc, [prompt, ""], [images, None] if images is not None else None,
                                        height=out_hw[0], width=out_hw[1], use_img_cfg=use_cfg)


# ------------------------------------------------------------------------------------------------
# LVMPipeline.prompt_condition_frame_block_autoregressive_inference (LVM/pipeline.py:347-595) -- the reference's own
# orchestration (windowing, prompts, noise draws, re-noising, CFG duplication, halving, decode / uint8), executed on CPU
# with the reference's LVMProcessor, LVMCollator, LVMScheduler and LVM.  Stand-ins for what the container lacks:
#   * torchvision.transforms -> the three transforms the processor composes (Lambda, ToTensor, Normalize), restated;
#   * the tokenizer -> StubTokenizer above;  * diffusers' AutoencoderKL -> the oracle's VAE restatement (oracle/vae_ref.py:
#     third-party arithmetic, unpinned as DESIGN.md §2 says) behind the three members the pipeline touches
#     (.config, .encode(x).latent_dist.sample(), .decode(z).sample), logging the noise every sample() draws.
# ------------------------------------------------------------------------------------------------
class _Transforms:
    """torchvision.transforms.{Compose, Lambda, ToTensor, Normalize} as LVM/processor.py:31-35 uses them."""

    class Compose:
        def __init__(self, ts):
            self.ts = ts

        def __call__(self, x):
            for t in self.ts:
                x = t(x)
            return x

    class Lambda:
        def __init__(self, fn):
            self.fn = fn

        def __call__(self, x):
            return self.fn(x)

    class ToTensor:
        def __call__(self, pil):   # HWC uint8 -> CHW float in [0, 1]
            return torch.from_numpy(np.ascontiguousarray(np.array(pil))).permute(2, 0, 1).float().div(255.0)

    class Normalize:
        def __init__(self, mean, std, inplace=False):
            self.mean, self.std = torch.tensor(mean).view(-1, 1, 1), torch.tensor(std).view(-1, 1, 1)

        def __call__(self, t):
            return (t - self.mean) / self.std


class OracleVaeStandIn(torch.nn.Module):
    """The members of diffusers' AutoencoderKL that LVMPipeline touches, over oracle/vae_ref.py."""

    def __init__(self, params, cfg):
        super().__init__()
        self.p, self.cfg = params, cfg
        self.config = types.SimpleNamespace(scaling_factor=cfg.scaling_factor, shift_factor=cfg.shift_factor)
        self.noise_log = []

    def encode(self, x):
        from . import vae_ref as VR
        mean, logvar = VR.encode_moments(self.p, self.cfg, x.float())
        log = self.noise_log

        class _Dist:
            def sample(self_inner):
                n = torch.randn(mean.shape)      # diffusers draws from the global generator (randn_tensor, generator=None)
                log.append(n.clone())
                return VR.sample_latent(mean, logvar, n)
        return types.SimpleNamespace(latent_dist=_Dist())

    def decode(self, z):
        from . import vae_ref as VR
        return types.SimpleNamespace(sample=VR.decode(self.p, self.cfg, z.float()))


def reference_pipeline(cfg, params, vae_params, vae_cfg):
    """(pipe, log): the reference's LVMPipeline over its own processor / collator / scheduler / LVM (tiny, CPU fp32);
    log collects, per round, what the scheduler was called with and returned."""
    from PIL import Image
    model, ns = build_reference_model(cfg, params, "LVM")
    proc_ns = _extract("LVM/processor.py", ["LVMCollator", "LVMProcessor"],
                       extra_ns={"PreTrainedTokenizer": object, "Image": Image, "transforms": _Transforms,
                                 "InterpolationMode": None, "snapshot_download": None, "AutoTokenizer": None, "os": os,
                                 "logging": None})
    sched_cls = scheduler_class()
    log = {"rounds": []}

    class RecordingScheduler(sched_cls):
        def __call__(self, z, func, model_kwargs, **kw):
            z_in = [t.clone() for t in z]      # the reference sampler updates the list in place
            out = super().__call__(z, func, model_kwargs, **kw)
            log["rounds"].append({"latents": z_in,
                                  "input_img_latents": [t.clone() for t in model_kwargs["input_img_latents"]],
                                  "samples": [t.clone() for t in out]})
            return out

    logger = types.SimpleNamespace(info=lambda *a, **k: None, warning=lambda *a, **k: None)
    import inspect
    pns = _extract("LVM/pipeline.py", ["LVMPipeline"],
                   extra_ns={"Image": Image, "os": os, "inspect": inspect, "Any": object, "Callable": object,
                             "snapshot_download": None, "LoraConfig": None, "PeftModel": None, "AutoencoderKL": object,
                             "load_file": None, "LVMProcessor": proc_ns.LVMProcessor, "LVM": ns.LVM,
                             "LVMScheduler": RecordingScheduler, "hccl_info": ns.hccl_info, "logger": logger,
                             "EXAMPLE_DOC_STRING": "", "replace_example_docstring": lambda s: (lambda f: f)})
    vae = OracleVaeStandIn(vae_params, vae_cfg)
    processor = proc_ns.LVMProcessor(StubTokenizer())
    pipe = pns.LVMPipeline(vae, model, processor, device=torch.device("cpu"))
    return pipe, log
