"""`training_losses_x1_noise_input` and its samplers (mirror of LVM/train_helper/loss.py:73-250) on the HIP path.

The reference's function mixes noise into the targets and the clean condition latents, calls the model and returns
per-frame MSE terms whose `.mean()` the training script back-propagates through torch.autograd
(LVM/train/train_x1_stage1_noiseinput.py:378-380).  The product has no autograd graph: gradients come from the explicit
backward kernels of `train.Stage1Trainer`.  So
  * `draw_training_noise(...)` performs the reference's draws, in the reference's order, from torch's global RNG
    (x0 per target frame, t, x0 per clean frame, t_input) -- the thing a trainer needs to stay on the reference's
    random stream;
  * `training_losses_x1_noise_input(model, x1, model_kwargs, ...)` keeps the reference's signature and returns
    {"loss": (frames,)} computed by the HIP forward (evaluation, logging); pass a `Stage1Trainer` as `model` and the
    same call also runs the backward, clipping and AdamW step (`update=True`), which is how a loop written against the
    reference's API trains here.
At sequence-parallel size 1 the reference's `broadcast_data` calls (:150,168-172) are identities and are not issued.
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional

import torch


def sample_x0(x1):
    """loss.py:73-89."""
    if isinstance(x1, (list, tuple)):
        return [torch.randn_like(a) for a in x1]
    return torch.randn_like(x1)


def sample_timestep(x1):
    """loss.py:92-95."""
    return torch.rand(len(x1)).to(x1[0])


def sample_exp_timestep(x1):
    """loss.py:98-102."""
    u = torch.normal(mean=0.0, std=1.0, size=(len(x1),))
    return (1 / (1 + torch.exp(-u))).to(x1[0])


def sample_frame_block_timestep(x1, frame_blocks):
    """loss.py:105-113: one python-RNG draw per frame block, shared by its frames."""
    t = []
    for b_inx in frame_blocks.keys():
        for frame_block in frame_blocks[b_inx]:
            t.extend([random.random()] * frame_block)
    t = torch.tensor(t)
    assert len(t) == len(x1)
    return t.to(x1[0])


def sample_timestep_max_noise(x1, max_noise_level=0):
    """loss.py:116-119."""
    return (max_noise_level + (1 - max_noise_level) * torch.rand(len(x1))).to(x1[0])


def mean_flat(x):
    """loss.py:246-250."""
    return torch.mean(x, dim=list(range(1, len(x.size()))))


def is_all_equal(data):
    """loss.py:14-54 (imported by the training script, never called there)."""
    if isinstance(data, list):
        if not data:
            return True
        if all(isinstance(x, torch.Tensor) for x in data):
            return all(torch.equal(x, data[0]) for x in data)
        return all(x == data[0] for x in data)
    if isinstance(data, torch.Tensor):
        return True if data.shape[0] < 2 else bool(torch.all(data == data[0]))
    raise TypeError("list or tensor expected")


def draw_training_noise(x1: List[torch.Tensor], input_img_latents: List[torch.Tensor], input_noise: float = 0.9,
                        frame_blocks: Optional[Dict[int, list]] = None, exp_time: bool = False):
    """The draws of loss.py:155-166 in their order: x0 ~ N per target frame; t ~ U (stage 1), logit-normal (`exp_time`)
    or one python-RNG value per frame block (stage 2+); x0_input per clean frame; t_input = input_noise + (1 -
    input_noise) U when there are clean frames."""
    x0 = sample_x0(x1)
    if frame_blocks is None:
        t = sample_exp_timestep(x1) if exp_time else sample_timestep(x1)
    else:
        t = sample_frame_block_timestep(x1, frame_blocks)
    x0_input = sample_x0(input_img_latents)
    t_input = sample_timestep_max_noise(input_img_latents, max_noise_level=input_noise) if len(input_img_latents) > 0 else None
    return x0, t, x0_input, t_input


def training_losses_x1_noise_input(model, x1, model_kwargs=None, snr_type="uniform", patch_weight=None, input_noise=0.9,
                                   cls_weight=None, order=None, frame_blocks=None, exp_time=False, device=None,
                                   update: bool = False):
    """loss.py:128-243 for list inputs, `order` None or 2, no patch / class weights (what the scripts use).
    `model`: an LVMTraining (forward only) or a train.Stage1Trainer (forward + backward [+ optimizer step])."""
    from .train import Stage1Trainer
    from .ops import VgptError
    if patch_weight is not None or cls_weight is not None or order not in (None, 2):
        raise VgptError("training_losses_x1_noise_input: patch_weight / cls_weight / order != 2 are not used by the "
                        "reference's scripts and not built")
    if model_kwargs is None:
        model_kwargs = {}
    if not isinstance(x1, (list, tuple)):
        x1 = list(x1.split(1))
    clean = list(model_kwargs.get("input_img_latents") or [])
    x0, t, x0_in, t_in = draw_training_noise(x1, clean, input_noise, frame_blocks, exp_time)
    trainer = model if isinstance(model, Stage1Trainer) else Stage1Trainer.for_evaluation(model)
    batch = {k: model_kwargs[k] for k in ("input_ids", "attention_mask", "position_ids", "input_image_sizes",
                                          "denoise_image_sizes", "time_emb_inx")}
    cat = lambda xs: torch.cat(list(xs), dim=0) if len(xs) > 0 else None
    ioret = bool(model_kwargs.get("input_output_return", False))      # loss.py:194-197,220-225: the input head's terms appended
    loss = trainer.step(batch, cat(x1), cat(x0), t, cat(clean), cat(x0_in) if clean else None, t_in,
                        update=update and isinstance(model, Stage1Trainer), backward=isinstance(model, Stage1Trainer),
                        input_output_return=ioret)
    if ioret:
        return {"loss": loss, "input_loss": loss[len(x1):]}
    return {"loss": loss}
