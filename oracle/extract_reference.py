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
