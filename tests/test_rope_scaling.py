"""rope_scaling ("su" / "longrope", Phi-3-128k family) — the reference builds its rotary embedding from the checkpoint's
config.json (LVM/model.py:202 -> transformers 4.47.1 Phi3Attention._init_rope), which the reference tree does not ship:
parity unpinned by the reference; the oracle is cross-checked against the installed transformers' longrope
implementation, the product's config handling and table kernel against the oracle."""
import importlib
import math

import pytest
import torch

from oracle import restate as R

M = importlib.import_module("video-gpt_amd.model")
VgptError = importlib.import_module("video-gpt_amd.ops").VgptError

HD = 96
SHORT = [1.0 + 0.01 * i for i in range(HD // 2)]
LONG = [1.0 + 0.5 * i for i in range(HD // 2)]
SCALING = {"type": "su", "short_factor": SHORT, "long_factor": LONG}
CFG = dict(hidden_size=192, num_attention_heads=2, max_position_embeddings=131072, original_max_position_embeddings=4096,
           rope_scaling=SCALING)


def _positions(long: bool):
    pos = torch.arange(0, 300)[None] * (20 if long else 3)
    return torch.cat([pos, pos.flip(1)], 0)


@pytest.mark.parametrize("long", [False, True])
def test_oracle_longrope_matches_installed_transformers(long):
    try:
        from transformers import Phi3Config
        from transformers.models.phi3.modeling_phi3 import Phi3RotaryEmbedding
    except Exception as e:  # pragma: no cover
        pytest.skip(str(e))
    hc = Phi3Config(hidden_size=192, num_attention_heads=2, num_key_value_heads=2, num_hidden_layers=1, intermediate_size=64,
                    vocab_size=64, pad_token_id=2, eos_token_id=None, bos_token_id=None, max_position_embeddings=131072,
                    original_max_position_embeddings=4096,
                    rope_parameters={"rope_type": "longrope", "rope_theta": 10000.0, "short_factor": SHORT,
                                     "long_factor": LONG, "original_max_position_embeddings": 4096})
    pos = _positions(long)
    assert (int(pos.max()) + 1 > 4096) == long
    cos, sin = Phi3RotaryEmbedding(hc)(torch.zeros(1, dtype=torch.float32), pos)
    c2, s2 = R.rope_cos_sin(pos, HD, 10000.0, torch.float32, SCALING, 131072, 4096)
    assert float((cos - c2).abs().max()) <= 1e-6 and float((sin - s2).abs().max()) <= 1e-6
    # the product's host-side spec (which factors, which scale) agrees
    ext, scale = M.Phi3Config(**CFG).rope_spec(int(pos.max()))
    assert ext == (LONG if long else SHORT)
    assert abs(scale - math.sqrt(1 + math.log(32) / math.log(4096))) < 1e-12
    # transformers-5 style config objects are read too
    ext2, scale2 = M.Phi3Config.from_hf(hc).rope_spec(int(pos.max()))
    assert ext2 == ext and scale2 == scale


def test_config_reads_or_refuses_rope_variants(tmp_path):
    import json
    base = dict(hidden_size=192, num_attention_heads=2, vocab_size=64, intermediate_size=64, num_hidden_layers=1)
    (tmp_path / "config.json").write_text(json.dumps(dict(base, rope_scaling=SCALING, max_position_embeddings=131072,
                                                          original_max_position_embeddings=4096)))
    c = M.Phi3Config.from_pretrained(str(tmp_path))
    assert c.rope_scaling["type"] == "su" and c.original_max_position_embeddings == 4096
    assert M.Phi3Config(**base).rope_spec(10 ** 6) == (None, 1.0)
    assert M.Phi3Config(**base, rope_scaling={"type": "default"}).rope_scaling is None
    with pytest.raises(VgptError, match="not supported"):
        M.Phi3Config(**base, rope_scaling={"type": "yarn", "factor": 4.0})
    with pytest.raises(VgptError, match="partial_rotary_factor"):
        M.Phi3Config(**base, partial_rotary_factor=0.5)
    with pytest.raises(VgptError, match="head_dim/2"):
        M.Phi3Config(**base, rope_scaling={"type": "longrope", "short_factor": [1.0] * 3, "long_factor": [1.0] * 3})
    (tmp_path / "config.json").write_text(json.dumps(dict(base, rope_scaling={"type": "linear", "factor": 2.0})))
    with pytest.raises(VgptError):
        M.Phi3Config.from_pretrained(str(tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("long", [False, True])
def test_device_rope_table_with_longrope(long):
    """vgpt_rope_table with rescaled inv_freq + attention factor == the oracle's bf16-rounded tables; and q/k rotated
    with them == the oracle's apply_rope."""
    ops = importlib.import_module("video-gpt_amd.ops")
    cfg = M.Phi3Config(**CFG)
    pos = _positions(long)
    cos, sin = M.rope_tables_for(cfg, pos.to("cuda:0"))
    c2, s2 = R.rope_cos_sin(pos, HD, 10000.0, torch.bfloat16, SCALING, 131072, 4096)
    c2, s2 = c2.float()[..., : HD // 2].reshape(-1, HD // 2), s2.float()[..., : HD // 2].reshape(-1, HD // 2)
    # fp32 cosf/sinf on the device vs torch CPU at angles up to ~6000 rad: one bf16 ulp on a few entries
    assert float((cos.cpu() - c2).abs().max()) <= 2 ** -7 and float((sin.cpu() - s2).abs().max()) <= 2 ** -7
    assert float(((cos.cpu() - c2).abs() > 1e-6).float().mean()) < 0.02
    g = torch.Generator("cpu").manual_seed(0)
    B, L = pos.shape
    qkv = torch.randn(B, L, 3 * 2 * HD, generator=g).to(torch.bfloat16)
    out = ops.rope_qk_inplace(qkv.clone().to("cuda:0"), cos, sin, 2, 2, HD).cpu().float()
    q = qkv.float()[..., : 2 * HD].view(B, L, 2, HD).transpose(1, 2)
    k = qkv.float()[..., 2 * HD: 4 * HD].view(B, L, 2, HD).transpose(1, 2)
    cf, sf = R.rope_cos_sin(pos, HD, 10000.0, torch.bfloat16, SCALING, 131072, 4096)
    qr, kr = R.apply_rope(q, k, cf.float(), sf.float())
    ref = torch.cat([qr.transpose(1, 2).reshape(B, L, -1), kr.transpose(1, 2).reshape(B, L, -1), qkv.float()[..., 4 * HD:]], -1)
    assert float((out - ref).norm() / ref.norm()) < 1e-2


@pytest.mark.gpu
def test_engine_hoisted_rows_use_the_whole_sequence_rope_factors():
    """su / longrope picks short or long factors from the largest position of the WHOLE sequence (HF 4.47.1
    Phi3LongRoPEScaledRotaryEmbedding.forward).  Layout where only the image rows cross original_max_position_embeddings
    while the hoisted time rows (computed in a per-clip pass of their own) stay below it: the hoisted engine must rotate
    them with the same (long) factors as the full recompute does."""
    from tests import smoke_case as SC
    S = importlib.import_module("video-gpt_amd.scheduler")
    P = importlib.import_module("video-gpt_amd.processor")
    LY = importlib.import_module("video-gpt_amd.layout")
    cfg, C, G, hw, steps = R.TINY, 2, 2, (16, 16), 3
    bl = (hw[0] // 2) * (hw[1] // 2) + 2
    p, batch, z, cond = SC.build_case(cfg, C=C, G=G, hw=hw, use_cfg=True)
    pos = batch["position_ids"]
    t_rows = [t for b in batch["time_emb_inx"] for t in batch["time_emb_inx"][b]]
    t_max = max(int(pos[b, t]) for b in batch["time_emb_inx"] for t in batch["time_emb_inx"][b])
    orig = t_max + 2                      # time rows alone: short factors; the whole sequence: long factors
    assert int(pos.max()) + 1 > orig > t_max + 1 and t_rows
    half = cfg.head_dim // 2
    scaling = {"type": "longrope", "short_factor": [1.0 + 0.01 * i for i in range(half)],
               "long_factor": [1.5 + 0.25 * i for i in range(half)]}
    pc = M.Phi3Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_key_value_heads, hidden_act=cfg.hidden_act, rms_norm_eps=cfg.rms_norm_eps,
                      rope_theta=cfg.rope_theta, pad_token_id=cfg.pad_token_id, max_position_embeddings=4 * orig,
                      original_max_position_embeddings=orig, rope_scaling=scaling)
    model = M.LVM(pc, pos_embed_max_size=cfg.pos_embed_max_size)
    model.load_state_dict(p, strict=True)
    model = model.to("cuda:0", torch.bfloat16).eval()
    lay = LY.TokenLayout.from_plans([(P.plan_inference([C, G])[0], bl, 0), (P.plan_inference([0, G])[0], bl, C * bl)],
                                    (C + G) * bl)
    outs = {}
    for mode in ("hoist", "none"):
        sched = S.LVMScheduler(num_steps=steps)
        sched.reuse_condition_prefix = sched.hoist_special_rows = mode == "hoist"
        kw = SC.model_kwargs(batch, cond, "cuda:0", use_cfg=True)
        kw["attention_mask"] = lay
        outs[mode] = torch.cat(sched([x.to("cuda:0", torch.bfloat16) for x in z], model.frame_block_forward_with_cfg, kw,
                                     prediction_type="x1"))
        assert bool(sched.last_engine.hoist) == (mode == "hoist")
    assert SC.rel_l2(outs["hoist"], outs["none"]) < 5e-3
