"""Ulysses sequence parallelism: two ranks on the one GPU of the test box over gloo (RCCL refuses
two ranks on one device; the transport falls back to an all-gather through host memory, the layout code is the one
RCCL's all_to_all_single serves).  Every rank runs its L/2 slice of the sequence with heads/2 of the attention and the
gathered result must equal the sequence-parallel-size-1 forward BIT FOR BIT: rows of a GEMM do not depend on M, a head's
attention is computed by one rank over the full sequence with the same kernel.  Also checked against the CPU oracle."""
import importlib
import socket

import numpy as np
import pytest
import torch

from oracle import restate as R
from tests import smoke_case as SC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _worker(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    importlib.import_module("video-gpt_amd")
    SP = importlib.import_module("video-gpt_amd.sequence_parallel")
    TF = importlib.import_module("video-gpt_amd.transform")
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg, C=2, G=2)
    model = SC.build_product_model(cfg, p, DEV)
    kw = SC.model_kwargs(batch, cond, DEV)
    kw = {k: v for k, v in kw.items() if k not in ("img_cfg_scale", "use_img_cfg", "use_kv_cache")}
    t = torch.full((len(z),), 0.3, device=DEV)
    x = [v.to(DEV, BF) for v in z]
    TF.replace_attention(model.llm)                       # no group yet: dist_attn stays None
    assert model.llm.layers[0].self_attn.dist_attn is None
    base, _ = model.frame_block_forward(x, t, **kw)
    SP.initialize_sequence_parallel_state(world)
    TF.replace_attention(model.llm)                       # installs DistributedAttention
    assert model.llm.layers[0].self_attn.dist_attn is not None
    out, _ = model.frame_block_forward(x, t, **kw)
    torch.cuda.synchronize()
    same = all(torch.equal(a, b) for a, b in zip(out, base))
    q.put((rank, bool(same), torch.cat(out).float().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_ulysses_two_ranks_match_single_rank_and_oracle():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda x: x[0])
    for p_ in procs:
        p_.join(timeout=120)
        assert p_.exitcode == 0
    (_, same0, o0), (_, same1, o1) = res
    assert same0 and same1                      # SP=2 == SP=1, bit for bit, on both ranks
    assert np.array_equal(o0, o1)
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg, C=2, G=2)
    ref = R.frame_block_forward(p, cfg, z, torch.full((len(z),), 0.3), input_ids=batch["input_ids"], input_img_latents=cond,
                                input_image_sizes=batch["input_image_sizes"], attention_mask=batch["attention_mask"],
                                position_ids=batch["position_ids"], denoise_image_sizes=batch["denoise_image_sizes"],
                                time_emb_inx=batch["time_emb_inx"])
    assert SC.rel_l2(torch.from_numpy(o0), torch.cat(ref)) < SC.tol("forward_latents")
