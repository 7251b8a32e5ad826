"""Input assembly for the clip-token sequence: prompt layout, left padding, RoPE positions, block
attention masks and the index dictionaries the model consumes.

Mirrors the interface of the reference's `LVMProcessor` / `LVMCollator` for the video path
(LVM/processor.py:128-274 prompt layouts, :442-534 positions, :575-731 masks, :812-838 padding,
:869-1000 batch assembly) — same method names, argument meaning and return structure — but the
masks are produced from a closed-form visibility rule over per-token attributes instead of
painting slices block by block:

  every frame is a block of `bl = N + 2` tokens; a CLEAN block is `<img>`, N image slots, `</img>`,
  a NOISY block is `<|diffusion|>`, one time slot, N image slots; consecutive noisy blocks that are
  denoised together form a CLIP.

  * a clean key k is visible to every row q >= thr(k) (sequence order), with thr = k for `<img>` and
    `</img>` and thr = block_start + 1 for the image slots (bidirectional inside the frame);
  * a noisy key is visible only to rows of the same clip: its `<|diffusion|>` column to all of them,
    its time column to rows at in-block offset >= 1, its image columns to image rows (offset >= 2);
  * pad columns are never visible, pad rows see everything.

The three layouts of the reference (next-clip inference, stage-1 interleaved, stage-2+ frame-block
groups) differ only in block order and clip membership.  Bit-exactness against the reference's own
collator is pinned by tests/golden/collator_*.npz (tests/test_collator.py).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .layout import TokenLayout

PAD, CLEAN, NOISY = 0, 1, 2


# ------------------------------------------------------------------------------------------------
# block plans: list of (kind, clip_id) in sequence order, plus RoPE block indices
# ------------------------------------------------------------------------------------------------

def plan_inference(frame_blocks: Sequence[int]):
    """[C, G] -> C clean blocks then one clip of G noisy blocks; positions run on (LVM/processor.py:502-534)."""
    *clean, g = frame_blocks
    kinds = [(CLEAN, -1)] * int(sum(clean)) + [(NOISY, 0)] * int(g)
    return kinds, list(range(len(kinds)))


def plan_stage1(n_blocks: int):
    """noisy_0, clean_0, noisy_1, ... (2F-1 blocks); noisy_i and clean_i share position block i (:442-467)."""
    kinds, pos = [], []
    for i in range(n_blocks):
        if i % 2 == 0:
            kinds.append((NOISY, i // 2))
        else:
            kinds.append((CLEAN, -1))
        pos.append(i // 2)
    return kinds, pos


def plan_frame_block_training(frame_blocks: Sequence[int]):
    """[noisy x fb, clean x fb] per group, last group noisy only; separate position counters (:469-500)."""
    kinds, pos = [], []
    noisy_ctr = clean_ctr = 0
    for k, fb in enumerate(frame_blocks):
        for _ in range(fb):
            kinds.append((NOISY, k)); pos.append(noisy_ctr); noisy_ctr += 1
        if k != len(frame_blocks) - 1:
            for _ in range(fb):
                kinds.append((CLEAN, -1)); pos.append(clean_ctr); clean_ctr += 1
    return kinds, pos


def block_mask(kinds, bl: int, pad: int, out: Optional[np.ndarray] = None) -> np.ndarray:
    """(L, L) uint8 visibility of one row of the batch from its block plan, written block by block (a few slice
    assignments per pair of blocks instead of element-wise rules over L x L):
      clean key block at s:  `<img>` (s) is seen by rows >= s, its slots by rows >= s+1, `</img>` by rows >= s+bl-1;
      noisy key block at s:  only by the noisy blocks of the same clip -- `<|diffusion|>` (s) by all of their rows, the
                             time slot (s+1) by rows with offset >= 1, the image slots by rows with offset >= 2;
      pad rows see everything."""
    n = len(kinds) * bl
    L = n + pad
    m = np.zeros((L, L), dtype=np.uint8) if out is None else out    # `out`: a zero-filled (L, L) uint8 view
    starts = [pad + i * bl for i in range(len(kinds))]
    for (kd, cl), s in zip(kinds, starts):
        if kd == CLEAN:
            m[s:, s] = 1
            m[s + 1:, s + 1:s + bl - 1] = 1
            m[s + bl - 1:, s + bl - 1] = 1
        elif kd == NOISY:
            for (qd, qc), sq in zip(kinds, starts):
                if qd == NOISY and qc == cl:
                    m[sq:sq + bl, s] = 1
                    m[sq + 1:sq + bl, s + 1] = 1
                    m[sq + 2:sq + bl, s + 2:s + bl] = 1
    m[:pad, :] = 1
    return m


# ------------------------------------------------------------------------------------------------

class LVMCollator:
    """mask_format "bool" (default): `attention_mask` is the reference's dense (B,L,L) bool tensor; "layout": it is a
    layout.TokenLayout (8 bytes per token) from which the device generates the packed mask — the model, the engine
    and the trainer take either."""

    def __init__(self, pad_token_id: int = 2, hidden_size: int = 3072, sequence_parallel_size: int = 1,
                 mask_format: str = "bool"):
        if mask_format not in ("bool", "layout"):
            raise ValueError(f"unknown mask_format {mask_format!r}")
        self.mask_format = mask_format
        self.pad_token_id = pad_token_id
        self.hidden_size = hidden_size
        self.sequence_parallel_size = sequence_parallel_size

    # -- padding (LVM/processor.py:812-838) --
    def pad_input_ids_training(self, input_ids: List[List[int]], image_sizes: Dict[int, list]):
        max_l = max(len(r) for r in input_ids)
        sp = self.sequence_parallel_size
        if max_l % sp != 0:
            max_l += sp - max_l % sp
        ids = np.full((len(input_ids), max_l), self.pad_token_id, dtype=np.int64)
        valid = np.zeros((len(input_ids), max_l), dtype=np.uint8)
        for i, row in enumerate(input_ids):
            pad = max_l - len(row)
            ids[i, pad:] = row
            valid[i, pad:] = 1
            if i in image_sizes:
                image_sizes[i] = [[s + pad, e + pad] for s, e in image_sizes[i]]
        return torch.from_numpy(ids), torch.from_numpy(valid), image_sizes

    # -- positions --
    @staticmethod
    def _block_len(sizes, lead: int):
        pad = sizes[0][0] - lead
        token_l = sizes[-1][-1] - pad
        if token_l % len(sizes) != 0:
            raise AssertionError("sequence is not a whole number of equal frame blocks")
        return pad, token_l // len(sizes)

    @staticmethod
    def _positions(pad: int, bl: int, pos_blocks):
        row = np.zeros(pad + len(pos_blocks) * bl, dtype=np.int64)
        for i, pb in enumerate(pos_blocks):
            row[pad + i * bl: pad + (i + 1) * bl] = np.arange(pb * bl, (pb + 1) * bl)
        return row

    def create_position_frame_block_inference(self, image_sizes, frame_blocks):
        rows, block_ls = [], []
        for b in image_sizes.keys():
            # row 0 starts with a clean block (<img> precedes the first image), later rows (the CFG
            # branch) with a noisy one (<|diffusion|>, time) — LVM/processor.py:508-511
            pad, bl = self._block_len(image_sizes[b], 1 if b == 0 else 2)
            block_ls.append(bl)
            _, pos_blocks = plan_inference(frame_blocks[b])
            rows.append(self._positions(pad, bl, pos_blocks))
        return torch.from_numpy(np.stack(rows)), block_ls

    def create_position_training(self, image_sizes):
        rows, block_ls = [], []
        for b in image_sizes.keys():
            pad, bl = self._block_len(image_sizes[b], 2)
            block_ls.append(bl)
            _, pos_blocks = plan_stage1(len(image_sizes[b]))
            rows.append(self._positions(pad, bl, pos_blocks))
        return torch.from_numpy(np.stack(rows)), block_ls

    def create_position_frame_block_training(self, image_sizes, frame_blocks):
        rows, block_ls = [], []
        for b in image_sizes.keys():
            pad, bl = self._block_len(image_sizes[b], 2)
            block_ls.append(bl)
            _, pos_blocks = plan_frame_block_training(frame_blocks[b])
            rows.append(self._positions(pad, bl, pos_blocks))
        return torch.from_numpy(np.stack(rows)), block_ls

    # -- masks --
    def _masks(self, attention_mask, block_ls, plans):
        seq_len = attention_mask.size(-1)
        if self.mask_format == "layout":
            rows = []
            for i in range(attention_mask.size(0)):
                valid = int(attention_mask[i].sum())
                rows.append((plans(i, valid // block_ls[i]), block_ls[i], seq_len - valid))
            return TokenLayout.from_plans(rows, seq_len)
        out = np.zeros((attention_mask.size(0), seq_len, seq_len), dtype=np.uint8)   # written in place, 0/1 bytes
        for i in range(attention_mask.size(0)):
            valid = int(attention_mask[i].sum())
            kinds = plans(i, valid // block_ls[i])
            if len(kinds) * block_ls[i] != valid:
                raise AssertionError("block plan does not cover the valid tokens")
            block_mask(kinds, block_ls[i], seq_len - valid, out=out[i])
        return torch.from_numpy(out.view(np.bool_))

    def create_mask_frame_block_inference(self, attention_mask, block_ls, frame_blocks):
        return self._masks(attention_mask, block_ls, lambda i, n: plan_inference(frame_blocks[i])[0])

    def create_mask_training(self, attention_mask, block_ls):
        return self._masks(attention_mask, block_ls, lambda i, n: plan_stage1(n)[0])

    def create_mask_frame_block_training(self, attention_mask, block_ls, frame_blocks):
        return self._masks(attention_mask, block_ls, lambda i, n: plan_frame_block_training(frame_blocks[i])[0])

    # -- batch assembly --
    @staticmethod
    def _gather(mllm_inputs, with_frame_blocks: bool):
        pixel_values, image_sizes, frame_blocks = [], {}, {}
        for b, x in enumerate(mllm_inputs):
            if x["pixel_values"] is not None:
                pixel_values.extend(x["pixel_values"])
                for size in x["image_sizes"]:
                    image_sizes.setdefault(b, []).append(size)
                    if with_frame_blocks:
                        frame_blocks[b] = x["frame_blocks"]
        pixel_values = [p.unsqueeze(0) for p in pixel_values]
        return pixel_values, image_sizes, frame_blocks

    def process_mllm_input_frame_block_inference(self, mllm_inputs, block_aware=False):
        pixel_values, image_sizes, frame_blocks = self._gather(mllm_inputs, True)
        ids, valid, image_sizes = self.pad_input_ids_training([x["input_ids"] for x in mllm_inputs], image_sizes)
        position_ids, block_ls = self.create_position_frame_block_inference(image_sizes, frame_blocks)
        mask = self.create_mask_frame_block_inference(valid, block_ls, frame_blocks)
        return ids, position_ids, mask, pixel_values, image_sizes, frame_blocks

    def process_mllm_input_training(self, mllm_inputs, block_aware=False):
        if block_aware:
            raise NotImplementedError("block_aware masks are not used by the reference's scripts")
        pixel_values, image_sizes, _ = self._gather(mllm_inputs, False)
        ids, valid, image_sizes = self.pad_input_ids_training([x["input_ids"] for x in mllm_inputs], image_sizes)
        position_ids, block_ls = self.create_position_training(image_sizes)
        mask = self.create_mask_training(valid, block_ls)
        return ids, position_ids, mask, pixel_values, image_sizes

    def process_mllm_input_frame_block_training(self, mllm_inputs, block_aware=False):
        pixel_values, image_sizes, frame_blocks = self._gather(mllm_inputs, True)
        ids, valid, image_sizes = self.pad_input_ids_training([x["input_ids"] for x in mllm_inputs], image_sizes)
        position_ids, block_ls = self.create_position_frame_block_training(image_sizes, frame_blocks)
        mask = self.create_mask_frame_block_training(valid, block_ls, frame_blocks)
        return ids, position_ids, mask, pixel_values, image_sizes, frame_blocks

    def process_mllm_input_frame_block_call(self, features):
        """LVM/processor.py:964-1000: split the per-row image slots into condition / denoise / time indices."""
        ids, position_ids, mask, pixel_values, sizes, frame_blocks = self.process_mllm_input_frame_block_inference(features)
        denoise, inputs, time_inx, input_images = {}, {}, {}, []
        pv = 0
        for b in sizes.keys():
            n_clean = int(sum(frame_blocks[b][:-1]))
            inputs[b] = sizes[b][:n_clean]
            denoise[b] = sizes[b][n_clean:]
            time_inx[b] = [s[0] - 1 for s in denoise[b]]
            input_images.extend(pixel_values[pv:pv + n_clean])
            pv += n_clean
        return {"input_ids": ids, "attention_mask": mask, "position_ids": position_ids,
                "input_pixel_values": input_images, "input_image_sizes": inputs, "denoise_image_sizes": denoise,
                "output_images": [], "time_emb_inx": time_inx, "frame_blocks": frame_blocks}

    # -- single-target path (LVMPipeline.__call__ -> LVMProcessor.__call__ -> LVMCollator.__call__): the sequence is
    #    [left pad | condition tokens | time token | target image tokens]  (LVM/processor.py:432-440, 536-573, 776-810, 841-866, 943-962)
    def pad_input_ids(self, input_ids, image_sizes, num_tokens_for_output_images):
        """LVM/processor.py:783-810: left-pad so that condition + 1 time token + image tokens end at a common length."""
        n_out = num_tokens_for_output_images
        max_l = max(len(r) + n_out[i] + 1 for i, r in enumerate(input_ids))
        sp = self.sequence_parallel_size
        if max_l % sp != 0:
            max_l += sp - max_l % sp
        rows_ids, rows_valid = [], []
        for i, row in enumerate(input_ids):
            pad = max_l - len(row) - n_out[i] - 1
            rows_ids.append([self.pad_token_id] * pad + list(row))
            rows_valid.append([0] * pad + [1] * len(row))
            if i in image_sizes:
                image_sizes[i] = [[s + pad, e + pad] for s, e in image_sizes[i]]
        if len({len(r) for r in rows_ids}) != 1:   # the reference builds a ragged LongTensor here and fails the same way
            raise ValueError("rows of one batch must have condition length + target tokens of one total")
        return torch.LongTensor(rows_ids), torch.ByteTensor(rows_valid), image_sizes

    def create_position(self, attention_mask, num_tokens_for_output_images):
        """LVM/processor.py:432-440: 0 on the pad, then 0..valid+img_length (one extra for the time token)."""
        text_l = attention_mask.size(-1)
        img_l = max(num_tokens_for_output_images)
        rows = np.zeros((attention_mask.size(0), text_l + img_l + 1), dtype=np.int64)
        for i in range(attention_mask.size(0)):
            valid = int(attention_mask[i].sum())
            rows[i, text_l - valid:] = np.arange(valid + img_l + 1)
        return torch.from_numpy(rows)

    def create_mask(self, attention_mask, num_tokens_for_output_images):
        """LVM/processor.py:536-573: condition + time rows causal, target rows see everything, pad columns hidden, pad
        rows all ones, columns of a shorter target's padding hidden everywhere; uint8 like the reference."""
        text_l = attention_mask.size(-1)
        img_l = max(num_tokens_for_output_images)
        L = text_l + img_l + 1
        out = np.zeros((attention_mask.size(0), L, L), dtype=np.uint8)
        padding_images = []
        for i in range(attention_mask.size(0)):
            valid = int(attention_mask[i].sum())
            pad, T = text_l - valid, valid + 1
            m = out[i]
            m[pad:pad + T, pad:pad + T] = np.tril(np.ones((T, T), dtype=np.uint8))
            m[pad + T:, pad:] = 1
            m[:pad, :] = 1
            short = img_l - num_tokens_for_output_images[i]
            if short > 0:
                m[:, L - short:] = 0
                padding_images.append(torch.zeros(1, short, self.hidden_size))
            else:
                padding_images.append(None)
        return torch.from_numpy(out), padding_images

    def adjust_attention_for_input_images(self, attention_mask, image_sizes):
        """LVM/processor.py:776-781: the tokens of one condition image see each other."""
        for b in image_sizes.keys():
            for s, e in image_sizes[b]:
                attention_mask[b][s:e, s:e] = 1
        return attention_mask

    def process_mllm_input(self, mllm_inputs, target_img_size):
        """LVM/processor.py:841-866."""
        n_out = [sz[0] * sz[1] // 16 // 16 for sz in target_img_size]
        pixel_values, image_sizes, _ = self._gather(mllm_inputs, False)
        ids, valid, image_sizes = self.pad_input_ids([x["input_ids"] for x in mllm_inputs], image_sizes, n_out)
        position_ids = self.create_position(valid, n_out)
        mask, padding_images = self.create_mask(valid, n_out)
        mask = self.adjust_attention_for_input_images(mask, image_sizes)
        return ids, position_ids, mask, padding_images, pixel_values, image_sizes

    def __call__(self, features):
        """LVM/processor.py:943-962: features = [(mllm_input, img_cfg_mllm_input | None, [height, width]), ...]."""
        mllm_inputs = [f[0] for f in features]
        cfg_inputs = [f[1] for f in features]
        target = [f[2] for f in features]
        if cfg_inputs[0] is not None:
            mllm_inputs = mllm_inputs + cfg_inputs
            target = target + target
        ids, position_ids, mask, padding_images, pixel_values, image_sizes = self.process_mllm_input(mllm_inputs, target)
        return {"input_ids": ids, "attention_mask": mask, "position_ids": position_ids, "input_pixel_values": pixel_values,
                "input_image_sizes": image_sizes, "padding_images": padding_images}

    def collate_stage1(self, mllm_inputs, frame_num: int):
        """TrainDataCollator.__call__ (LVM/train_helper/data.py:422-458) without the video I/O."""
        ids, position_ids, mask, pixel_values, sizes = self.process_mllm_input_training(mllm_inputs)
        denoise = {b: [s for i, s in enumerate(sizes[b]) if i % 2 == 0] for b in sizes}
        inputs = {b: [s for i, s in enumerate(sizes[b]) if i % 2 == 1] for b in sizes}
        time_inx = {b: [s[0] - 1 for s in denoise[b]] for b in sizes}
        per = 2 * frame_num - 1
        return {"input_ids": ids, "attention_mask": mask, "position_ids": position_ids,
                "input_pixel_values": [p for i, p in enumerate(pixel_values) if i % per % 2 == 1],
                "input_image_sizes": inputs, "denoise_image_sizes": denoise,
                "output_images": [p for i, p in enumerate(pixel_values) if i % per % 2 == 0],
                "time_emb_inx": time_inx}


# ------------------------------------------------------------------------------------------------

class SpecialTokenizer:
    """Minimal stand-in for the Phi-3 tokenizer on the video path, which only ever tokenises the
    special tags `<img>`, `</img>`, `<|diffusion|>` (LVM/pipeline.py:426-439).  Pass a real
    `transformers` tokenizer to LVMProcessor when a checkpoint directory is available."""

    def __init__(self, img: int = 32001, img_end: int = 32002, diffusion: int = 32003, bos: int = 1):
        self.table = {"<img>": img, "</img>": img_end, "<|diffusion|>": diffusion}
        self.bos = bos

    def __call__(self, text: str):
        ids, pos = [self.bos], 0
        while pos < len(text):
            for tag, tid in self.table.items():
                if text.startswith(tag, pos):
                    ids.append(tid)
                    pos += len(tag)
                    break
            else:
                raise ValueError(f"SpecialTokenizer cannot tokenise {text[pos:pos + 16]!r}; pass a real tokenizer")

        class _Out:
            input_ids = ids
        return _Out()


class LVMProcessor:
    """Prompt layout + collation for the next-clip path (LVM/processor.py:23-421).  Image decoding /
    resizing (PIL, torchvision) is host preprocessing outside the accelerated path: images are
    accepted as float tensors (3, H, W) in [-1, 1] with H, W multiples of 16."""

    _TAG = re.compile(r"<\|image_\d+\|>")

    def __init__(self, text_tokenizer=None, max_image_size: int = 1024, sequence_parallel_size: int = 1,
                 mask_format: str = "bool"):
        self.text_tokenizer = text_tokenizer if text_tokenizer is not None else SpecialTokenizer()
        self.max_image_size = max_image_size
        self.sequence_parallel_size = sequence_parallel_size
        self.collator = LVMCollator(sequence_parallel_size=sequence_parallel_size, mask_format=mask_format)

    @classmethod
    def from_pretrained(cls, model_name, sequence_parallel_size: int = 1):
        import os
        from transformers import AutoTokenizer
        if not os.path.isdir(model_name):
            raise FileNotFoundError(f"{model_name}: local checkpoint directories only (no hub access)")
        # a missing / unreadable tokenizer is an error exactly as in the reference (LVM/processor.py:70-78): falling back
        # to SpecialTokenizer's made-up ids would select the wrong embed_tokens rows without any failure
        return cls(AutoTokenizer.from_pretrained(model_name), sequence_parallel_size=sequence_parallel_size)

    def crop_arr(self, pil_image):
        """LVM/processor.py:39-64: shrink to max_image_size, enlarge to >= 16, centre-crop to multiples of 16."""
        from PIL import Image
        while min(*pil_image.size) >= 2 * self.max_image_size:
            pil_image = pil_image.resize(tuple(x // 2 for x in pil_image.size), resample=Image.BOX)
        if max(*pil_image.size) > self.max_image_size:
            scale = self.max_image_size / max(*pil_image.size)
            pil_image = pil_image.resize(tuple(round(x * scale) for x in pil_image.size), resample=Image.BICUBIC)
        if min(*pil_image.size) < 16:
            scale = 16 / min(*pil_image.size)
            pil_image = pil_image.resize(tuple(round(x * scale) for x in pil_image.size), resample=Image.BICUBIC)
        arr = np.array(pil_image)
        y1, x1 = (arr.shape[0] % 16) // 2, (arr.shape[1] % 16) // 2
        y2, x2 = arr.shape[0] % 16 - y1, arr.shape[1] % 16 - x1
        return arr[y1:arr.shape[0] - y2, x1:arr.shape[1] - x2]

    def process_image(self, image):
        """PIL image / path -> (3, H, W) float tensor in [-1, 1] (ToTensor + Normalize(0.5, 0.5),
        LVM/processor.py:31-35,78-86); tensors already in that form pass through."""
        if torch.is_tensor(image):
            if image.dtype == torch.uint8 and image.dim() == 3 and image.shape[-1] == 3:  # (H, W, 3) uint8 frame
                image = (image.permute(2, 0, 1).float().div(255.0) - 0.5) / 0.5
            if image.dim() != 3 or image.shape[-1] % 16 or image.shape[-2] % 16:
                raise ValueError("tensor images must be (3, H, W) with sides that are multiples of 16")
            return image
        try:
            from PIL import Image
        except ImportError as e:  # pragma: no cover
            raise ValueError("Input must be a PIL.Image object") from e
        if isinstance(image, str):
            image = Image.open(image).convert("RGB")
        elif not isinstance(image, Image.Image):
            raise ValueError("Input must be a PIL.Image object")
        arr = self.crop_arr(image)
        t = torch.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1).float().div(255.0)
        return (t - 0.5) / 0.5

    def _chunks(self, text: str):
        chunks = [list(self.text_tokenizer(c).input_ids) for c in self._TAG.split(text)]
        chunks = [c[1:] if c and c[0] == 1 else c for c in chunks]
        tags = self._TAG.findall(text)
        ids = [int(t.split("|")[1].split("_")[-1]) for t in tags]
        uniq = sorted(set(ids))
        assert uniq == list(range(1, len(uniq) + 1)), f"image_ids must be continuous from 1, got {uniq}"
        return chunks, ids, uniq

    @staticmethod
    def _ntok(img) -> int:
        return img.size(-2) * img.size(-1) // 16 // 16

    def process_multi_modal_prompt_frame_block(self, text, input_images, frame_blocks, height=None, width=None):
        """LVM/processor.py:128-179."""
        input_images = input_images or []
        chunks, image_ids, uniq = self._chunks(text)
        assert len(uniq) == len(input_images) + frame_blocks[-1], "image tags and images disagree"
        input_images = [input_images[x - 1] for x in image_ids[:frame_blocks[0]]]
        ids, sizes, idx = [], [], 0
        for k, fb in enumerate(frame_blocks):
            last = k == len(frame_blocks) - 1
            for _ in range(fb):
                ids.extend(chunks[idx])
                if last:
                    ids.append(0)  # time slot
                    n = height * width // 256 if height is not None and width is not None else self._ntok(input_images[0])
                else:
                    n = self._ntok(input_images[idx])
                sizes.append([len(ids), len(ids) + n])
                ids.extend([0] * n)
                idx += 1
        return {"input_ids": ids, "pixel_values": input_images, "image_sizes": sizes}

    def process_multi_modal_prompt_training(self, text, input_images):
        """LVM/processor.py:181-218 (stage 1: every even chunk is followed by a time slot)."""
        chunks, image_ids, uniq = self._chunks(text)
        assert len(uniq) == len(input_images)
        input_images = [input_images[x - 1] for x in image_ids]
        ids, sizes = [], []
        for i, c in enumerate(chunks):
            ids.extend(c)
            if i != len(chunks) - 1:
                if i % 2 == 0:
                    ids.append(0)
                n = self._ntok(input_images[i])
                sizes.append([len(ids), len(ids) + n])
                ids.extend([0] * n)
        return {"input_ids": ids, "pixel_values": input_images, "image_sizes": sizes}

    def process_multi_modal_prompt_frame_block_training(self, text, input_images, frame_blocks):
        """LVM/processor.py:220-274."""
        chunks, image_ids, uniq = self._chunks(text)
        assert len(uniq) == len(input_images)
        input_images = [input_images[x - 1] for x in image_ids]
        ids, sizes, idx = [], [], 0

        def emit(noisy: bool):
            nonlocal idx
            ids.extend(chunks[idx])
            if noisy:
                ids.append(0)
            n = self._ntok(input_images[idx])
            sizes.append([len(ids), len(ids) + n])
            ids.extend([0] * n)
            idx += 1

        for k, fb in enumerate(frame_blocks):
            for _ in range(fb):
                emit(True)
            if k != len(frame_blocks) - 1:
                for _ in range(fb):
                    emit(False)
        return {"input_ids": ids, "pixel_values": input_images, "image_sizes": sizes}

    # -- single-target path (LVM/processor.py:90-126, 276-317, 319-364) --
    def add_prefix_instruction(self, prompt):
        return f"{prompt}<|diffusion|>"

    def process_multi_modal_prompt(self, text, input_images):
        """LVM/processor.py:90-126: `<|image_i|>` tags become N zero slots; the prompt ends with `<|diffusion|>`."""
        text = self.add_prefix_instruction(text)
        if input_images is None or len(input_images) == 0:
            ids = list(self.text_tokenizer(text).input_ids)
            if ids[0] == 1:
                ids = ids[1:]
            return {"input_ids": ids, "pixel_values": None, "image_sizes": None}
        chunks, image_ids, uniq = self._chunks(text)
        assert len(uniq) == len(input_images), (
            f"total images must be the same as the number of image tags, got {len(uniq)} image tags and {len(input_images)} images")
        input_images = [input_images[x - 1] for x in image_ids]
        ids, sizes = [], []
        for i, c in enumerate(chunks):
            ids.extend(c)
            if i != len(chunks) - 1:
                n = self._ntok(input_images[i])
                sizes.append([len(ids), len(ids) + n])
                ids.extend([0] * n)
        return {"input_ids": ids, "pixel_values": input_images, "image_sizes": sizes}

    def _single_target_rows(self, instruction, images):
        images = [self.process_image(x) for x in images] if images is not None and len(images) > 0 else None
        if images is None:
            assert "<img><|image_1|></img>" not in instruction
        return self.process_multi_modal_prompt(instruction, images)

    def __call__(self, instructions, input_images=None, height: int = 1024, width: int = 1024, use_img_cfg: bool = True,
                 use_input_image_size_as_output: bool = False) -> Dict:
        """LVM/processor.py:282-317: one row per instruction; the CFG row is the empty prompt (`<|diffusion|>` only)."""
        if input_images is None:
            use_img_cfg = False
        if isinstance(instructions, str):
            instructions, input_images = [instructions], [input_images]
        data = []
        for i, ins in enumerate(instructions):
            row = self._single_target_rows(ins, None if input_images is None else input_images[i])
            cfg_row = self.process_multi_modal_prompt("", None) if use_img_cfg else None
            size = ([row["pixel_values"][0].size(-2), row["pixel_values"][0].size(-1)] if use_input_image_size_as_output
                    else [height, width])
            data.append((row, cfg_row, size))
        return self.collator(data)

    def prompt_condition_inference(self, instructions, input_images=None, height: int = 1024, width: int = 1024,
                                   use_img_cfg: bool = True, use_input_image_size_as_output: bool = False) -> Dict:
        """LVM/processor.py:319-364: instructions[0] is the conditional prompt, instructions[1] the CFG prompt."""
        if input_images is None:
            use_img_cfg = False
        if isinstance(instructions, str):
            instructions, input_images = [instructions], [input_images]
        row = self._single_target_rows(instructions[0], None if input_images is None else input_images[0])
        cfg_row = None
        if use_img_cfg:
            cfg_row = self._single_target_rows(instructions[1], None if input_images is None else input_images[1])
        size = ([row["pixel_values"][0].size(-2), row["pixel_values"][0].size(-1)] if use_input_image_size_as_output
                else [height, width])
        return self.collator([(row, cfg_row, size)])

    def prompt_condition_frame_block_inference(self, instructions, input_images=None, height: int = 1024,
                                               width: int = 1024, use_img_cfg: bool = True,
                                               use_input_image_size_as_output: bool = False,
                                               frame_blocks: Optional[List[int]] = None) -> Dict:
        """LVM/processor.py:366-421."""
        if input_images is None:
            use_img_cfg = False
        if isinstance(instructions, str):
            instructions, input_images = [instructions], [input_images]
        imgs = input_images[0]
        imgs = [self.process_image(x) for x in imgs] if imgs else None
        row = self.process_multi_modal_prompt_frame_block(instructions[0], imgs, frame_blocks)
        row["frame_blocks"] = frame_blocks
        rows = [row]
        if use_img_cfg:
            imgs1 = input_images[1]
            imgs1 = [self.process_image(x) for x in imgs1] if imgs1 else None
            cfg_blocks = [0, frame_blocks[-1]]
            cfg_row = self.process_multi_modal_prompt_frame_block(
                instructions[1], imgs1, cfg_blocks, height=row["pixel_values"][0].size(-2),
                width=row["pixel_values"][0].size(-1))
            cfg_row["frame_blocks"] = cfg_blocks
            rows.append(cfg_row)
        return self.collator.process_mllm_input_frame_block_call(rows)
