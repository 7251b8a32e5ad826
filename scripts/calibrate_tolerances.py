"""Measure the tolerances of the model-level GPU parity tests instead of guessing them ("2x the error of
stock bf16 ops on the same inputs").

For every quantity the `-m gpu` tests compare against the fp32 oracle, this script runs oracle/restate.py TWICE on the
same bf16-representable weights and the same inputs: once in fp32 (the checker the tests use) and once with torch's stock
CPU bf16 ops (bf16 parameters and activations: F.linear / matmul accumulate in fp32 and round their OUTPUT to bf16, the
norm output, RoPE, softmax probabilities, residual stream and the sampler state z are bf16 -- the points at which the
reference's bf16 model rounds on an accelerator).  rel-L2(bf16 run, fp32 run) is what "stock bf16 ops" cost; the test
tolerance is 2x the largest value over the seeds, written to tests/golden/tolerance_calibration.json and read by
tests/smoke_case.py::tol().  CPU only, about a minute; re-run after changing a test case's shape.

  python scripts/calibrate_tolerances.py            # tiny cases (what the tests use)
  python scripts/calibrate_tolerances.py --fullwidth   # + one full-width layer (H 3072) over a 516-token clip, ~2 min
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import restate as R          # noqa: E402
from tests import glue_cases as GC       # noqa: E402
from tests import smoke_case as SC       # noqa: E402

BF = torch.bfloat16


def to_bf(p):
    return {k: v.to(BF) for k, v in p.items()}


def rel(a, b):
    return SC.rel_l2(a.float(), b.float())


def fwd_kwargs(batch, cond):
    return dict(input_ids=batch["input_ids"], input_img_latents=cond, input_image_sizes=batch["input_image_sizes"],
                attention_mask=batch["attention_mask"], position_ids=batch["position_ids"],
                denoise_image_sizes=batch["denoise_image_sizes"], time_emb_inx=batch["time_emb_inx"])


def forward_and_sampler(cfg, seeds, hw_list):
    out = {"forward_latents": [], "sampler_latents": [], "llm_hidden": []}
    for seed in seeds:
        for hw in hw_list:
            for use_cfg in (True, False):
                p, batch, z, cond = SC.build_case(cfg, hw=hw, seed=seed, use_cfg=use_cfg)
                pb, zb, cb = to_bf(p), [t.to(BF) for t in z], [t.to(BF) for t in cond]
                for pt in ("x1", "v"):
                    t = torch.full((len(z),), 0.3)
                    f32 = R.frame_block_forward_with_cfg(p, cfg, z, t, use_cfg, 1.6, pt, **fwd_kwargs(batch, cond))
                    b16 = R.frame_block_forward_with_cfg(pb, cfg, zb, t, use_cfg, 1.6, pt, **fwd_kwargs(batch, cb))
                    out["forward_latents"].append(rel(torch.cat(b16), torch.cat(f32)))
                for steps in (2, 3):
                    f32 = SC.oracle_sample(cfg, p, batch, z, cond, steps, "x1", use_cfg=use_cfg)
                    b16 = SC.oracle_sample(cfg, pb, batch, zb, cb, steps, "x1", use_cfg=use_cfg)
                    out["sampler_latents"].append(rel(torch.cat(b16), torch.cat(f32)))
                _, h32 = R.frame_block_forward(p, cfg, z, torch.full((len(z),), 0.3), return_hidden=True, **fwd_kwargs(batch, cond))
                _, h16 = R.frame_block_forward(pb, cfg, zb, torch.full((len(z),), 0.3), return_hidden=True, **fwd_kwargs(batch, cb))
                out["llm_hidden"].append(rel(h16, h32))
    return out


def single_forward(cfg, seeds):
    vals = []
    for seed in seeds:
        c = GC.single_target_case(cfg)
        p = {k: v.to(BF).float() for k, v in R.make_params(cfg, seed).items()}
        f32 = R.lvm_forward(p, cfg, c["x"], c["t"], c["ids"], c["lat"], c["sizes"], c["mask"], c["pos"])
        b16 = R.lvm_forward(to_bf(p), cfg, c["x"].to(BF), c["t"], c["ids"], [t_.to(BF) for t_ in c["lat"]], c["sizes"],
                            c["mask"], c["pos"])
        vals.append(rel(b16, f32))
    return {"single_forward_latents": vals}


def loss_and_grads(cfg, seeds):
    res = {"loss": [], "param_grads": [], "xt": []}
    for seed in seeds:
        p, batch, x1, x0, t, clean, x0i, ti = GC.stage1_case(cfg, seed=seed)
        runs = {}
        for name, dt in (("f32", torch.float32), ("bf16", BF)):
            pr = {k: v.to(dt).clone().requires_grad_(True) for k, v in p.items() if k != "pos_embed"}
            pr["pos_embed"] = p["pos_embed"].to(dt)
            c = lambda a: list(a.to(dt).split(1))
            loss, xt = R.stage1_loss(pr, cfg, c(x1), c(x0), t, c(clean), c(x0i), ti, batch)
            loss.float().mean().backward()
            runs[name] = (loss.detach().float(), torch.cat(xt).detach().float(),
                          {k: v.grad.detach().float() for k, v in pr.items() if v.requires_grad and v.grad is not None})
        res["loss"].append(rel(runs["bf16"][0], runs["f32"][0]))
        res["xt"].append(rel(runs["bf16"][1], runs["f32"][1]))
        g32, g16 = runs["f32"][2], runs["bf16"][2]
        res["param_grads"].append(max(rel(g16[k], g32[k]) for k in g32 if float(g32[k].norm()) > 0))
    return res


def fullwidth_layer(seeds):
    """One decoder layer at the product's width (H 3072, 32 x 96 heads, I 8192) on a short next-clip sequence: the error
    of stock bf16 ops per layer at full reduction lengths (K = 3072 / 8192), for tests/test_fullwidth_parity_gpu.py."""
    cfg = R.Phi3Cfg(hidden_size=3072, intermediate_size=8192, num_hidden_layers=1, num_attention_heads=32,
                    num_key_value_heads=32, vocab_size=64, pad_token_id=2)
    res = {"fullwidth_hidden": [], "fullwidth_step_latents": []}
    for seed in seeds:
        p, batch, z, cond = SC.build_case(cfg, C=2, G=2, hw=(16, 16), seed=seed)
        pb, zb, cb = to_bf(p), [t.to(BF) for t in z], [t.to(BF) for t in cond]
        t = torch.full((len(z),), 0.3)
        l32, h32 = R.frame_block_forward(p, cfg, z, t, return_hidden=True, **fwd_kwargs(batch, cond))
        l16, h16 = R.frame_block_forward(pb, cfg, zb, t, return_hidden=True, **fwd_kwargs(batch, cb))
        res["fullwidth_hidden"].append(rel(h16, h32))
        f32 = SC.oracle_sample(cfg, p, batch, z, cond, 1, "x1")
        b16 = SC.oracle_sample(cfg, pb, batch, zb, cb, 1, "x1")
        res["fullwidth_step_latents"].append(rel(torch.cat(b16), torch.cat(f32)))
    return res


def fullwidth_stage1():
    """The cfg-3 stage-1 batch of tests/test_fullwidth_parity_gpu.py::test_cfg3_stage1_step_full_width (2 x 3870 tokens, one
    full-width decoder layer): per-frame loss and the gradients of the parameters that test compares, stock bf16 ops against
    fp32 (several minutes on 8 host cores: bf16 matmuls of K = 3072 / 8192 over 7740 rows, forward and backward)."""
    cfg = R.Phi3Cfg(hidden_size=3072, intermediate_size=8192, num_hidden_layers=1, num_attention_heads=32,
                    num_key_value_heads=32, vocab_size=64, pos_embed_max_size=24)
    p = {k: v.to(BF).float() for k, v in R.make_params(cfg, seed=11).items()}
    batch = R.collate_stage1([8, 8], 256)
    gen = torch.Generator("cpu").manual_seed(3)
    nd, nc = 16, 14
    mk = lambda n: torch.randn(n, 4, 32, 32, generator=gen)
    x1, x0, clean, x0i = mk(nd), mk(nd), mk(nc), mk(nc)
    t = torch.rand(nd, generator=gen)
    ti = 0.9 + 0.1 * torch.rand(nc, generator=gen)
    names = ["llm.layers.0.self_attn.qkv_proj.weight", "llm.layers.0.self_attn.o_proj.weight",
             "llm.layers.0.mlp.gate_up_proj.weight", "llm.layers.0.mlp.down_proj.weight",
             "llm.layers.0.input_layernorm.weight", "llm.layers.0.post_attention_layernorm.weight", "llm.norm.weight",
             "final_layer.linear.weight", "x_embedder.proj.weight", "input_x_embedder.proj.weight",
             "time_token.mlp.2.weight", "final_layer.adaLN_modulation.1.weight"]
    runs = {}
    for name, dt in (("f32", torch.float32), ("bf16", BF)):
        pr = {k: v.to(dt).clone().requires_grad_(k in names) for k, v in p.items()}
        c = lambda a: list(a.to(dt).split(1))
        loss, _ = R.stage1_loss(pr, cfg, c(x1), c(x0), t, c(clean), c(x0i), ti, batch)
        loss.float().mean().backward()
        runs[name] = (loss.detach().float(), {k: pr[k].grad.detach().float() for k in names})
        print(f"fullwidth stage-1: {name} run done", flush=True)
    g32, g16 = runs["f32"][1], runs["bf16"][1]
    per = {k: rel(g16[k], g32[k]) for k in names}
    print("fullwidth stage-1 per-parameter:", {k: round(v, 5) for k, v in per.items()}, flush=True)
    return {"fullwidth_stage1_loss": [rel(runs["bf16"][0], runs["f32"][0])], "fullwidth_stage1_param_grads": [max(per.values())]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fullwidth", action="store_true")
    ap.add_argument("--fullwidth-stage1", action="store_true", help="only the cfg-3 full-width stage-1 quantities (slow)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "tolerance_calibration.json"))
    args = ap.parse_args()
    torch.manual_seed(0)
    cfg = R.TINY
    seeds = (0, 1, 2)
    meas = {}
    if args.fullwidth_stage1:
        meas.update(fullwidth_stage1())
    else:
        with torch.no_grad():
            meas.update(forward_and_sampler(cfg, seeds, ((8, 8), (16, 16))))
            meas.update(single_forward(cfg, seeds))
        meas.update(loss_and_grads(cfg, (3, 4, 5)))
    path = args.out
    prev = json.load(open(path)) if os.path.exists(path) else {"quantities": {}}
    if args.fullwidth:
        with torch.no_grad():
            meas.update(fullwidth_layer((0, 1)))
    q = dict(prev.get("quantities", {}))
    for k, v in meas.items():
        q[k] = {"stock_bf16_rel_l2_max": round(max(v), 6), "stock_bf16_rel_l2_mean": round(sum(v) / len(v), 6),
                "samples": len(v), "tolerance": round(2 * max(v), 6)}
    doc = {"what": "rel-L2 of oracle/restate.py run with torch's stock CPU bf16 ops against its fp32 self on the same "
                   "bf16-representable weights and inputs; tolerance = 2 x the largest value (SURVEY.md section 8d)",
           "generated_by": "scripts/calibrate_tolerances.py", "torch": torch.__version__, "quantities": q}
    json.dump(doc, open(path, "w"), indent=1)
    for k, v in q.items():
        print(f"{k:26s} stock bf16 max {v['stock_bf16_rel_l2_max']:.3e}  mean {v['stock_bf16_rel_l2_mean']:.3e}  -> tolerance {v['tolerance']:.3e}")


if __name__ == "__main__":
    main()
