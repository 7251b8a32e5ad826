"""Seeded inputs shared by tests/make_golden.py (which runs the REFERENCE's own LVM / LVMTraining / loss on them
through oracle/extract_reference.py), tests/test_oracle_pins.py (oracle restatement == those vectors) and the GPU
parity tests (HIP path == those vectors).  Inputs are regenerated from CPU generators; only outputs are stored."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle import restate as R

BF = torch.bfloat16
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BATCH_KEYS = ("input_ids", "input_image_sizes", "attention_mask", "position_ids", "denoise_image_sizes", "time_emb_inx")


def single_target_case(cfg=R.TINY):
    """LVM.forward inputs (LVM/model.py:330-397): [condition ids | time token | x] with an OmniGen-style mask."""
    g = torch.Generator("cpu").manual_seed(11)
    B, Lc, hw, N = 2, 21, (8, 8), 16
    L = Lc + 1 + N
    ids = torch.randint(3, cfg.vocab_size, (B, Lc), generator=g)
    x = torch.randn(B, 4, *hw, generator=g).to(BF).float()
    t = torch.tensor([0.2, 0.9])
    lat = [torch.randn(1, 4, *hw, generator=g).to(BF).float()]
    sizes = {0: [[2, 2 + N]]}
    mask = torch.tril(torch.ones(L, L)).bool()[None].repeat(B, 1, 1)
    mask[:, -N:, -N:] = True
    pos = torch.arange(L)[None].repeat(B, 1)
    return dict(ids=ids, x=x, t=t, lat=lat, sizes=sizes, mask=mask, pos=pos, Lc=Lc, N=N)


def stage1_case(cfg=R.TINY, seed=3):
    """Stage-1 batch (F = 3 and 2 frames, 8x8 latents) with externally drawn noise / times."""
    p = {k: v.to(BF).float() for k, v in R.make_params(cfg, seed).items()}
    batch = R.collate_stage1([3, 2], 16)
    gen = torch.Generator("cpu").manual_seed(0)
    nd = sum(len(v) for v in batch["denoise_image_sizes"].values())
    nc = sum(len(v) for v in batch["input_image_sizes"].values())
    mk = lambda n: torch.randn(n, 4, 8, 8, generator=gen)
    x1, x0, clean, x0i = mk(nd), mk(nd), mk(nc), mk(nc)
    t = torch.rand(nd, generator=gen)
    ti = 0.9 + 0.1 * torch.rand(nc, generator=gen)
    return p, batch, x1, x0, t, clean, x0i, ti


def frame_block_training_batch(fbs_list=((1, 2, 2), (3, 1)), N=16):
    """Stage-2+ batch from the reference-generated collator fixture (ref_collator_fbtrain_3x2N16.npz) with the index
    dicts of TrainDataCollator_FrameBlock (LVM/train_helper/data.py:503-523)."""
    d = np.load(os.path.join(GOLD, "ref_collator_fbtrain_3x2N16.npz"))
    shape = tuple(int(v) for v in d["mask_shape"])
    mask = torch.from_numpy(np.unpackbits(d["mask_bits"], axis=-1)[:, : shape[1] * shape[2]].reshape(shape)).bool()
    sizes = {}
    for b, s, e in d["image_sizes"]:
        sizes.setdefault(int(b), []).append([int(s), int(e)])
    den, inp, tix = {}, {}, {}
    for b, fbs in enumerate(fbs_list):
        den[b], inp[b], tix[b], idx = [], [], [], 0
        for k, f in enumerate(fbs):
            if k != len(fbs) - 1:
                for _ in range(f):
                    den[b].append(sizes[b][idx]); inp[b].append(sizes[b][idx + f]); tix[b].append(sizes[b][idx][0] - 1)
                    idx += 1
                idx += f
            else:
                for _ in range(f):
                    den[b].append(sizes[b][idx]); tix[b].append(sizes[b][idx][0] - 1); idx += 1
    return dict(input_ids=torch.from_numpy(d["input_ids"]), position_ids=torch.from_numpy(d["position_ids"]),
                attention_mask=mask, input_image_sizes=inp, denoise_image_sizes=den, time_emb_inx=tix,
                frame_blocks={b: list(f) for b, f in enumerate(fbs_list)})


def sampled_grad(gr: torch.Tensor, cap: int = 6000) -> np.ndarray:
    """Every stride-th element of a flattened gradient (whole tensor when it has <= cap elements)."""
    flat = gr.detach().reshape(-1)
    stride = max(1, (flat.numel() + cap - 1) // cap)
    return flat[::stride].float().numpy()


def load(name):
    return np.load(os.path.join(GOLD, name))
