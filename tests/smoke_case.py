"""Shared tiny end-to-end case: next-clip denoising of a 2-condition / 2-generated-frame clip with
CFG on a 2-layer, H=192 model (SURVEY.md §7 tiny config), product (HIP) vs oracle (CPU fp32).

Used by `__graft_entry__.smoke()` and tests/test_model_gpu.py.  The oracle is the checker only.
"""
from __future__ import annotations

import importlib

import torch

from oracle import restate as R

BF = torch.bfloat16


_TOL = None


def tol(quantity: str) -> float:
    """Tolerance of a model-level parity check = 2 x the rel-L2 error torch's stock bf16 ops make on the same inputs
    against the fp32 oracle, as measured by scripts/calibrate_tolerances.py and committed in
    tests/golden/tolerance_calibration.json -- not a hand-set constant."""
    global _TOL
    if _TOL is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tolerance_calibration.json")
        _TOL = json.load(open(path))["quantities"]
    return float(_TOL[quantity]["tolerance"])


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def build_case(cfg=R.TINY, C=2, G=2, hw=(8, 8), seed=0, use_cfg=True):
    """Returns (oracle params with bf16-representable values, batch dict, latents, cond latents)."""
    p = {k: v.to(BF).float() for k, v in R.make_params(cfg, seed).items()}
    N = (hw[0] // 2) * (hw[1] // 2)
    batch = R.collate_inference(C, G, N, use_cfg=use_cfg, pad_id=cfg.pad_token_id)
    g = torch.Generator("cpu").manual_seed(42)  # LVM/inference/...inference.py:98 uses seed 42
    noise = [torch.randn(1, 4, *hw, generator=g).to(BF).float() for _ in range(G)]
    z = noise * (2 if use_cfg else 1)
    cond = [torch.randn(1, 4, *hw, generator=torch.Generator("cpu").manual_seed(1000 + i)).to(BF).float()
            for i in range(C)]
    return p, batch, z, cond


def build_product_model(cfg, params, device="cuda:0", cls_name="LVM"):
    M = importlib.import_module("video-gpt_amd.model")
    pc = M.Phi3Config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                      num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                      num_key_value_heads=cfg.num_key_value_heads, hidden_act=cfg.hidden_act,
                      rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta, pad_token_id=cfg.pad_token_id)
    model = getattr(M, cls_name)(pc, pos_embed_max_size=cfg.pos_embed_max_size)
    if "input_final_layer.weight" in params:
        model.init_input_final_layer()        # the optional head of input_output_return (LVM/model.py:246-253)
    missing, unexpected = model.load_state_dict(params, strict=True), None
    return model.to(device, BF).eval()


def model_kwargs(batch, cond, device, use_cfg=True, scale=1.6):
    return dict(input_ids=batch["input_ids"].to(device), input_img_latents=[c.to(device, BF) for c in cond],
                input_image_sizes=batch["input_image_sizes"], attention_mask=batch["attention_mask"].to(device),
                position_ids=batch["position_ids"].to(device), denoise_image_sizes=batch["denoise_image_sizes"],
                time_emb_inx=batch["time_emb_inx"], img_cfg_scale=scale, use_img_cfg=use_cfg, use_kv_cache=False,
                offload_model=False, vae=None)


def oracle_sample(cfg, p, batch, z, cond, steps, prediction_type, use_cfg=True, scale=1.6):
    sigma = R.scheduler_sigma(steps, 1.0)

    def func(zl, t):
        return R.frame_block_forward_with_cfg(
            p, cfg, zl, t, use_cfg, scale, prediction_type, input_ids=batch["input_ids"], input_img_latents=cond,
            input_image_sizes=batch["input_image_sizes"], attention_mask=batch["attention_mask"],
            position_ids=batch["position_ids"], denoise_image_sizes=batch["denoise_image_sizes"],
            time_emb_inx=batch["time_emb_inx"])
    return R.scheduler_call(sigma, z, func, use_cfg, scale, prediction_type)


def run_smoke(device="cuda:0", steps=2, tol=None, verbose=True):
    """One tiny next-clip denoise (x1 prediction, CFG 1.6) through the hipGraph sampler vs the oracle."""
    S = importlib.import_module("video-gpt_amd.scheduler")
    if tol is None:
        tol = globals()["tol"]("sampler_latents")
    cfg = R.TINY
    p, batch, z, cond = build_case(cfg)
    model = build_product_model(cfg, p, device)
    kw = model_kwargs(batch, cond, device)
    sched = S.LVMScheduler(num_steps=steps, time_shifting_factor=1)
    out = sched([t.to(device, BF) for t in z], model.frame_block_forward_with_cfg, kw, prediction_type="x1")
    torch.cuda.synchronize()
    ref = oracle_sample(cfg, p, batch, z, cond, steps, "x1")
    err = rel_l2(torch.cat(out), torch.cat(ref))
    if verbose:
        print(f"smoke: tiny next-clip denoise, {steps} steps, rel-L2 vs CPU oracle = {err:.3e} (tol {tol})")
    if not err < tol:
        raise AssertionError(f"smoke parity failed: rel-L2 {err} >= {tol}")
    # the same through the default product path of LVMPipeline: per-token mask layout (mask expanded on the device),
    # cached condition prefix, special rows hoisted out of the per-step row set
    P = importlib.import_module("video-gpt_amd.processor")
    LY = importlib.import_module("video-gpt_amd.layout")
    C, G, hw = 2, 2, (16, 16)
    bl = (hw[0] // 2) * (hw[1] // 2) + 2
    p2, batch2, z2, cond2 = build_case(cfg, C=C, G=G, hw=hw)
    lay = LY.TokenLayout.from_plans([(P.plan_inference([C, G])[0], bl, 0), (P.plan_inference([0, G])[0], bl, C * bl)],
                                    (C + G) * bl)
    model2 = build_product_model(cfg, p2, device)
    kw2 = model_kwargs(batch2, cond2, device)
    kw2["attention_mask"] = lay
    sched2 = S.LVMScheduler(num_steps=steps, time_shifting_factor=1)
    out2 = sched2([t.to(device, BF) for t in z2], model2.frame_block_forward_with_cfg, kw2, prediction_type="x1")
    torch.cuda.synchronize()
    err2 = rel_l2(torch.cat(out2), torch.cat(oracle_sample(cfg, p2, batch2, z2, cond2, steps, "x1")))
    if verbose:
        print(f"smoke: same with the token-layout mask, cached prefix and hoisted special rows "
              f"(hoisted={bool(sched2.last_engine.hoist)}): rel-L2 = {err2:.3e}")
    if not (err2 < tol and sched2.last_engine.hoist):
        raise AssertionError(f"smoke parity failed on the hoisted path: rel-L2 {err2} (tol {tol})")
    return err
