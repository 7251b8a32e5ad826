"""GPU parity of the assembled model and sampler (product HIP path vs CPU fp32 oracle on identical
bf16-representable weights, identical noise seeds).

Tolerances are measured, not guessed: 2 x the rel-L2 error torch's stock bf16 ops make on the same inputs against the
fp32 oracle (scripts/calibrate_tolerances.py -> tests/golden/tolerance_calibration.json): 2.3e-2 on
one-forward latents, 2.9e-2 on sampled latents, 1.1e-2 on hidden states, 1.6e-2 on the single-target forward.
"""
import importlib

import pytest
import torch

from oracle import restate as R
from tests import smoke_case as SC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16
# tolerances: 2 x the measured error of stock bf16 ops on the same inputs (tests/golden/tolerance_calibration.json,
# scripts/calibrate_tolerances.py), per quantity
TOL_FWD, TOL_SMP, TOL_HID, TOL_ONE = (SC.tol(q) for q in ("forward_latents", "sampler_latents", "llm_hidden", "single_forward_latents"))


@pytest.fixture(scope="module")
def case():
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg)
    model = SC.build_product_model(cfg, p, DEV)
    return cfg, p, batch, z, cond, model


def test_transformer_hidden_states(case):
    """Phi3Transformer.forward on a 3-D bool mask vs the oracle (OmniGen/transformer.py:128-214)."""
    cfg, p, batch, z, cond, model = case
    g = torch.Generator("cpu").manual_seed(5)
    B, L = batch["input_ids"].shape
    emb = (torch.randn(B, L, cfg.hidden_size, generator=g) * 0.5).to(BF).float()
    out = model.llm(inputs_embeds=emb.to(DEV, BF), attention_mask=batch["attention_mask"].to(DEV),
                    position_ids=batch["position_ids"].to(DEV)).last_hidden_state
    ref = R.transformer(p, cfg, emb, batch["attention_mask"], batch["position_ids"])
    valid = batch["input_ids"] != cfg.pad_token_id  # pad rows are don't-care for the model outputs, but still match
    assert SC.rel_l2(out, ref) < TOL_HID
    assert SC.rel_l2(out.cpu()[valid], ref[valid]) < TOL_HID


def test_transformer_rejects_2d_mask(case):
    cfg, p, batch, z, cond, model = case
    B, L = batch["input_ids"].shape
    with pytest.raises(Exception, match="attention_mask parameter was unavailable or invalid"):
        model.llm(inputs_embeds=torch.zeros(B, L, cfg.hidden_size, device=DEV, dtype=BF),
                  attention_mask=torch.ones(B, L, device=DEV), position_ids=batch["position_ids"].to(DEV))


@pytest.mark.parametrize("prediction_type", ["x1", "v"])
def test_frame_block_forward_with_cfg(case, prediction_type):
    cfg, p, batch, z, cond, model = case
    kw = SC.model_kwargs(batch, cond, DEV)
    t = torch.full((len(z),), 0.3)
    out, cache = model.frame_block_forward_with_cfg([x.to(DEV, BF) for x in z], t.to(DEV), past_key_values=None,
                                                    prediction_type=prediction_type, **kw)
    assert cache is None and len(out) == len(z) and out[0].shape == z[0].shape
    okw = {k: batch[k] for k in ("input_ids", "input_image_sizes", "attention_mask", "position_ids",
                                 "denoise_image_sizes", "time_emb_inx")}
    ref = R.frame_block_forward_with_cfg(p, cfg, z, t, True, 1.6, prediction_type, input_img_latents=cond, **okw)
    assert SC.rel_l2(torch.cat(out), torch.cat(ref)) < TOL_FWD


def test_seam_replace_attention_matches_fused(case):
    """replace_attention installs the SDPA-signature HIP op; a user-supplied local_attn gets (B,h,S,d)."""
    cfg, p, batch, z, cond, model = case
    T = importlib.import_module("video-gpt_amd.transform")
    ops = importlib.import_module("video-gpt_amd.ops")
    kw = SC.model_kwargs(batch, cond, DEV)
    t = torch.full((len(z),), 0.7).to(DEV)
    xs = [x.to(DEV, BF) for x in z]
    base = torch.cat(model.frame_block_forward(xs, t, kw["input_ids"], kw["input_img_latents"], kw["input_image_sizes"],
                                               kw["attention_mask"], kw["position_ids"], kw["denoise_image_sizes"],
                                               kw["time_emb_inx"], return_past_key_values=False))
    seen = []

    def spy(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
        seen.append(tuple(q.shape))
        return ops.sdpa(q, k, v, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal)

    for mod in model.modules():
        if hasattr(mod, "local_attn"):
            mod.local_attn = spy
    via = torch.cat(model.frame_block_forward(xs, t, kw["input_ids"], kw["input_img_latents"], kw["input_image_sizes"],
                                              kw["attention_mask"], kw["position_ids"], kw["denoise_image_sizes"],
                                              kw["time_emb_inx"], return_past_key_values=False))
    T.replace_attention(model)
    B, L = kw["input_ids"].shape
    assert seen and seen[0] == (B, cfg.num_attention_heads, L, cfg.head_dim)
    assert torch.equal(base, via)


@pytest.mark.parametrize("prediction_type", ["x1", "v"])
@pytest.mark.parametrize("use_graph,pack", [(True, True), (False, True), (True, False)])
def test_sampler_fast_path(case, prediction_type, use_graph, pack):
    cfg, p, batch, z, cond, model = case
    S = importlib.import_module("video-gpt_amd.scheduler")
    steps = 3
    sched = S.LVMScheduler(num_steps=steps, time_shifting_factor=1)
    sched.use_graph = use_graph
    sched.pack_padding = pack
    out = sched([x.to(DEV, BF) for x in z], model.frame_block_forward_with_cfg, SC.model_kwargs(batch, cond, DEV),
                prediction_type=prediction_type)
    assert sched.last_engine is not None and sched.last_engine.packed == pack
    ref = SC.oracle_sample(cfg, p, batch, z, cond, steps, prediction_type)
    assert SC.rel_l2(torch.cat(out), torch.cat(ref)) < TOL_SMP


def test_sampler_generic_path_matches_fast_path(case):
    cfg, p, batch, z, cond, model = case
    S = importlib.import_module("video-gpt_amd.scheduler")
    kw = SC.model_kwargs(batch, cond, DEV)
    sched = S.LVMScheduler(num_steps=2)
    fast = torch.cat(sched([x.to(DEV, BF) for x in z], model.frame_block_forward_with_cfg, kw, prediction_type="x1"))

    def func(zl, ts, past_key_values=None, prediction_type="x1", **mk):  # not a bound LVM method -> generic path
        return model.frame_block_forward_with_cfg(zl, ts, past_key_values=past_key_values,
                                                  prediction_type=prediction_type, **mk)
    sched2 = S.LVMScheduler(num_steps=2)
    gen = torch.cat(sched2([x.to(DEV, BF) for x in z], func, kw, prediction_type="x1"))
    assert sched2.last_engine is None
    assert SC.rel_l2(gen, fast) < 1e-2


def test_shift_and_sigma_table():
    S = importlib.import_module("video-gpt_amd.scheduler")
    for steps, shift, begin in [(50, 1, None), (7, 3.0, None), (5, 2, 0.25)]:
        assert torch.equal(S.LVMScheduler(steps, shift, begin).sigma, R.scheduler_sigma(steps, shift, begin))


def test_smoke_entry():
    assert SC.run_smoke(verbose=False) < TOL_SMP


def test_single_target_forward(case):
    """LVM.forward / forward_with_cfg (LVM/model.py:330-397,504-516): [condition | time token | x], any 3-D mask."""
    cfg, p, batch, z, cond, model = case
    g = torch.Generator("cpu").manual_seed(11)
    B, Lc, hw = 2, 21, (8, 8)
    N = 16
    L = Lc + 1 + N
    ids = torch.randint(3, cfg.vocab_size, (B, Lc), generator=g)
    x = torch.randn(B, 4, *hw, generator=g).to(BF).float()
    t = torch.tensor([0.2, 0.9])
    lat = [torch.randn(1, 4, *hw, generator=g).to(BF).float()]
    sizes = {0: [[2, 2 + N]]}
    mask = torch.tril(torch.ones(L, L)).bool()[None].repeat(B, 1, 1)
    mask[:, -N:, -N:] = True                      # image tokens see each other (OmniGen-style)
    pos = torch.arange(L)[None].repeat(B, 1)
    ref = R.lvm_forward(p, cfg, x, t, ids, lat, sizes, mask, pos)
    out, cache = model.forward(x.to(DEV, BF), t.to(DEV), ids.to(DEV), [lat[0].to(DEV, BF)], sizes, mask.to(DEV), pos.to(DEV))
    assert cache is None and out.shape == x.shape
    assert SC.rel_l2(out, ref) < TOL_ONE
    # 'v' CFG on the batch halves
    o2, _ = model.forward_with_cfg(x.to(DEV, BF), t.to(DEV), ids.to(DEV), [lat[0].to(DEV, BF)], sizes, mask.to(DEV),
                                   pos.to(DEV), True, 1.6, None, False, False, prediction_type="v")
    c = ref[1:2] + 1.6 * (ref[0:1] - ref[1:2])
    assert SC.rel_l2(o2, torch.cat([c, c])) < TOL_ONE
    # no condition tokens
    out3 = model.forward(x.to(DEV, BF), t.to(DEV), None, None, None, mask[:, Lc:, Lc:].contiguous().to(DEV),
                         pos[:, : N + 1].contiguous().to(DEV), return_past_key_values=False)
    ref3 = R.lvm_forward(p, cfg, x, t, None, None, None, mask[:, Lc:, Lc:], pos[:, : N + 1])
    assert SC.rel_l2(out3, ref3) < TOL_ONE


def test_condition_prefix_reuse_matches_full_recompute():
    """The clean condition rows are step-invariant: computing them once (prefill) and running only the remaining
    rows per step must give the reference's result (which recomputes everything every step, LVM/scheduler.py:174)."""
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg, C=2, G=2, hw=(16, 16))       # block = 66 tokens -> 132-token prefix
    model = SC.build_product_model(cfg, p, DEV)
    S = importlib.import_module("video-gpt_amd.scheduler")
    outs = {}
    for reuse in (True, False):
        sched = S.LVMScheduler(num_steps=3)
        sched.reuse_condition_prefix = reuse
        outs[reuse] = torch.cat(sched([x.to(DEV, BF) for x in z], model.frame_block_forward_with_cfg,
                                      SC.model_kwargs(batch, cond, DEV), prediction_type="x1"))
        assert (sched.last_engine.S > 0) == reuse
        if reuse:
            eng = sched.last_engine
            assert eng.S == 256 and eng.Ma == eng.L - 256      # 132-row prefix padded to 2 x 128
    ref = torch.cat(SC.oracle_sample(cfg, p, batch, z, cond, 3, "x1"))
    assert SC.rel_l2(outs[True], ref) < TOL_SMP
    assert SC.rel_l2(outs[True], outs[False]) < 5e-3


@pytest.mark.parametrize("use_cfg,C,G,hw", [(True, 2, 2, (16, 16)), (False, 2, 2, (16, 16)), (True, 5, 3, (12, 8))])
def test_special_row_hoisting_matches_full_recompute(use_cfg, C, G, hw):
    """`<|diffusion|>` rows are step-invariant and time rows depend on the step only (neither sees an image column), so
    the engine computes them for all steps in one pass and runs only the image rows per step.  Same result as the
    reference's full recompute, and as prefix reuse alone.  (12, 8): a non-square latent whose 24-token frames are
    not a multiple of anything the kernels tile by.)"""
    cfg = R.TINY
    steps = 3
    N = (hw[0] // 2) * (hw[1] // 2)
    bl = N + 2
    p, batch, z, cond = SC.build_case(cfg, C=C, G=G, hw=hw, use_cfg=use_cfg)
    P = importlib.import_module("video-gpt_amd.processor")
    LY = importlib.import_module("video-gpt_amd.layout")
    plans = [(P.plan_inference([C, G])[0], bl, 0)]
    if use_cfg:
        plans.append((P.plan_inference([0, G])[0], bl, C * bl))
    lay = LY.TokenLayout.from_plans(plans, (C + G) * bl)
    assert torch.equal(lay.to_bool_tensor(), batch["attention_mask"].to(torch.bool))
    model = SC.build_product_model(cfg, p, DEV)
    S = importlib.import_module("video-gpt_amd.scheduler")
    outs = {}
    for mode in ("hoist", "prefix", "none"):
        sched = S.LVMScheduler(num_steps=steps)
        sched.reuse_condition_prefix = mode != "none"
        sched.hoist_special_rows = mode == "hoist"
        kw = SC.model_kwargs(batch, cond, DEV, use_cfg=use_cfg)
        kw["attention_mask"] = lay
        outs[mode] = torch.cat(sched([x.to(DEV, BF) for x in z], model.frame_block_forward_with_cfg, kw,
                                     prediction_type="x1"))
        eng = sched.last_engine
        assert bool(eng.hoist) == (mode == "hoist")
        if mode == "hoist":
            nf = len(z)
            assert eng.S0 == C * bl and eng.S == (C * bl + 2 * nf + 127) // 128 * 128 and eng.Ma == nf * N   # image rows only
            assert eng.time_qkv.shape[:3] == (steps, cfg.num_hidden_layers, nf)
    ref = torch.cat(SC.oracle_sample(cfg, p, batch, z, cond, steps, "x1", use_cfg=use_cfg))
    assert SC.rel_l2(outs["hoist"], ref) < TOL_SMP
    assert SC.rel_l2(outs["hoist"], outs["none"]) < 5e-3 and SC.rel_l2(outs["hoist"], outs["prefix"]) < 5e-3


def test_scheduler_noise_level_premix():
    """LVM/scheduler.py:162-163: z <- noise_level * z + (1 - noise_level) * randn before the first step (HIP lerp kernel,
    torch's generator for the noise) == sampling from latents mixed the same way beforehand."""
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg)
    model = SC.build_product_model(cfg, p, DEV)
    S = importlib.import_module("video-gpt_amd.scheduler")
    zs = [x.to(DEV, BF) for x in z]
    torch.manual_seed(5)
    a = torch.cat(S.LVMScheduler(num_steps=2)(zs, model.frame_block_forward_with_cfg, SC.model_kwargs(batch, cond, DEV),
                                              prediction_type="x1", noise_level=0.3))
    torch.manual_seed(5)
    mixed = [(f.float() * 0.3 + torch.randn_like(f).float() * 0.7).to(BF) for f in zs]
    b = torch.cat(S.LVMScheduler(num_steps=2)(mixed, model.frame_block_forward_with_cfg, SC.model_kwargs(batch, cond, DEV),
                                              prediction_type="x1"))
    assert SC.rel_l2(a, b) < 1e-2


def test_engine_is_reused_for_the_next_clip_of_the_same_sequence():
    """Rounds of a rollout whose window is full present the same sequence with new condition latents: the scheduler re-binds
    the engine cached on the model (buffers, attention plan, captured graph kept; per-clip pass redone) instead of building
    a new one.  Same bits as a fresh engine on the same inputs; another layout, other parameter storage or
    cache_engines = False build a new engine."""
    cfg = R.TINY
    P = importlib.import_module("video-gpt_amd.processor")
    LY = importlib.import_module("video-gpt_amd.layout")
    S = importlib.import_module("video-gpt_amd.scheduler")
    C, G, hw, steps = 2, 2, (16, 16), 3
    bl = (hw[0] // 2) * (hw[1] // 2) + 2
    p, batch, z, cond = SC.build_case(cfg, C=C, G=G, hw=hw)
    lay = lambda: LY.TokenLayout.from_plans([(P.plan_inference([C, G])[0], bl, 0), (P.plan_inference([0, G])[0], bl, C * bl)],
                                            (C + G) * bl)
    model = SC.build_product_model(cfg, p, DEV)
    cond2 = [torch.randn(1, 4, *hw, generator=torch.Generator("cpu").manual_seed(2000 + i)).to(BF).float() for i in range(C)]
    z2 = [torch.randn(1, 4, *hw, generator=torch.Generator("cpu").manual_seed(3000 + i)).to(BF).float() for i in range(G)] * 2

    def sample(zz, cc, cache=True, prec="bf16"):
        sched = S.LVMScheduler(num_steps=steps)
        sched.cache_engines = cache
        sched.attention_precision = prec
        kw = SC.model_kwargs(batch, cc, DEV)
        kw["attention_mask"] = lay()                    # a NEW layout object with the same attributes, as the collator makes
        out = torch.cat(sched([x.to(DEV, BF) for x in zz], model.frame_block_forward_with_cfg, kw, prediction_type="x1"))
        return out, sched
    for prec in ("bf16", "fp8"):
        model.__dict__.pop("_vgpt_engine_cache", None)
        a1, s1 = sample(z, cond, prec=prec)
        a2, s2 = sample(z2, cond2, prec=prec)           # same sequence, new latents: the cached engine
        assert not s1.last_engine_reused and s2.last_engine_reused and s2.last_engine is s1.last_engine
        assert s2.last_engine.hoist and s2.last_engine.graph is not None
        b2, f2 = sample(z2, cond2, cache=False, prec=prec)
        assert not f2.last_engine_reused and f2.last_engine is not s1.last_engine
        assert torch.equal(a2, b2)
        a3, s3 = sample(z, cond, prec=prec)             # and back: nothing of clip 2 survives in the cached engine
        assert s3.last_engine_reused and torch.equal(a3, a1)
    ref = torch.cat(SC.oracle_sample(cfg, p, batch, z2, cond2, steps, "x1"))
    assert SC.rel_l2(b2, ref) < 6e-2
    # another sequence (one more generated frame) -> another engine; two layouts are kept
    p3, batch3, z3, cond3 = SC.build_case(cfg, C=C, G=3, hw=hw)
    sched = S.LVMScheduler(num_steps=steps)
    sched([x.to(DEV, BF) for x in z3], model.frame_block_forward_with_cfg, SC.model_kwargs(batch3, cond3, DEV), prediction_type="x1")
    assert not sched.last_engine_reused
