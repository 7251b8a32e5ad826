"""Batch collators of the training loops: `TrainDataCollator` (stage 1) and `TrainDataCollator_FrameBlock` (stage 2+),
mirrors of LVM/train_helper/data.py:404-458 and :461-537.  The datasets that feed them (`DatasetFromVideo*`, cv2 /
decord frame sampling) are video I/O outside the accelerated path; a feature here is what their `__getitem__` returns:
the dict of `LVMProcessor.process_multi_modal_prompt_training` / `..._frame_block_training` (+ "frame_blocks")."""
from __future__ import annotations

import copy

from .processor import LVMCollator


class _TrainCollatorBase(LVMCollator):
    def __init__(self, pad_token_id: int, hidden_size: int, keep_raw_resolution: bool, frame_num: int,
                 sequence_parallel_size=1, batch_size=1, block_aware=False, mask_format: str = "bool"):
        super().__init__(pad_token_id=pad_token_id, hidden_size=hidden_size, sequence_parallel_size=sequence_parallel_size,
                         mask_format=mask_format)
        if not keep_raw_resolution:
            raise ValueError("keep_raw_resolution=False references undefined names in the reference too (data.py:432-437)")
        self.keep_raw_resolution = keep_raw_resolution
        self.batch_size = batch_size
        self.frame_num = frame_num
        self.block_aware = block_aware

    def _fill(self, features):
        """A lone sample is repeated up to the batch size (data.py:426-430)."""
        rows = copy.deepcopy(features)
        if len(features) == 1 and len(features) < self.batch_size:
            rows.extend(copy.deepcopy(features) * (self.batch_size - 1))
        return rows


class TrainDataCollator(_TrainCollatorBase):
    """Stage-1 layout noisy_0, clean_0, noisy_1, ...: even image slots are denoise targets, odd ones clean inputs."""

    def __call__(self, features):
        ids, position_ids, mask, pixel_values, sizes = self.process_mllm_input_training(self._fill(features),
                                                                                       block_aware=self.block_aware)
        denoise = {b: [s for i, s in enumerate(sizes[b]) if i % 2 == 0] for b in sizes}
        inputs = {b: [s for i, s in enumerate(sizes[b]) if i % 2 == 1] for b in sizes}
        time_inx = {b: [s[0] - 1 for s in denoise[b]] for b in sizes}
        per = 2 * self.frame_num - 1
        return {"input_ids": ids, "attention_mask": mask, "position_ids": position_ids,
                "input_pixel_values": [p for i, p in enumerate(pixel_values) if i % per % 2 == 1],
                "input_image_sizes": inputs, "denoise_image_sizes": denoise,
                "output_images": [p for i, p in enumerate(pixel_values) if i % per % 2 == 0], "time_emb_inx": time_inx}


class TrainDataCollator_FrameBlock(_TrainCollatorBase):
    """Stage-2+ layout [noisy x fb, clean x fb] per group, the last group noisy only (data.py:503-523)."""

    def __call__(self, features):
        ids, position_ids, mask, pixel_values, sizes, frame_blocks = self.process_mllm_input_frame_block_training(
            self._fill(features), block_aware=self.block_aware)
        denoise, inputs, time_inx, output_images, input_images = {}, {}, {}, [], []
        for b in sizes.keys():
            denoise[b], inputs[b], time_inx[b], idx = [], [], [], 0
            base = b * self.frame_num
            for k, fb in enumerate(frame_blocks[b]):
                last = k == len(frame_blocks[b]) - 1
                for _ in range(fb):
                    denoise[b].append(sizes[b][idx])
                    time_inx[b].append(sizes[b][idx][0] - 1)
                    output_images.append(pixel_values[idx + base])
                    if not last:
                        inputs[b].append(sizes[b][idx + fb])
                        input_images.append(pixel_values[idx + fb + base])
                    idx += 1
                if not last:
                    idx += fb
        return {"input_ids": ids, "attention_mask": mask, "position_ids": position_ids, "input_pixel_values": input_images,
                "input_image_sizes": inputs, "denoise_image_sizes": denoise, "output_images": output_images,
                "time_emb_inx": time_inx, "frame_blocks": frame_blocks}
