#!/usr/bin/env python3
"""bench.py — next-clip denoise throughput of the HIP path (BASELINE.json metric, config[1]).

Workload (N=1): 256^2 px -> latent (4,32,32) -> 256 tokens/frame, 258-token blocks; C=4 condition
frames + G=8 generated frames, image CFG on (B=2, L=3096, 5160 real tokens), Phi-3-mini-class
denoiser (H 3072, 32 layers, 32x96 heads, I 8192, SiLU), x1 prediction, bf16, random-init weights,
synthetic latents, noise from torch.Generator("cpu").manual_seed(42).  One "step" = one Euler step
of the sampler = one full forward over condition+denoise tokens (the reference recomputes
everything every step, LVM/scheduler.py:174) + x1->v + CFG + update, replayed from a hipGraph.
metric = denoised clip-tokens/sec = G*N*steps / time (whole job: summed over ranks; replicas only,
each rank samples its own clip — weak scaling, no data-path collective).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

BF = torch.bfloat16
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def full_config(M, layers=32):
    return M.Phi3Config(vocab_size=32064, hidden_size=3072, intermediate_size=8192, num_hidden_layers=layers,
                        num_attention_heads=32, num_key_value_heads=32, hidden_act="silu", rms_norm_eps=1e-5,
                        rope_theta=10000.0, pad_token_id=2)


def build_model(M, cfg, device, seed=0):
    """Random-init weights on the device: N(0, 0.02) Linear/Conv/Embedding weights and biases, norm gains
    1 + 0.1 N(0,1), zero-initialised heads re-randomised."""
    with torch.device("meta"):
        model = M.LVM(cfg)
    model = model.to_empty(device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("layernorm.weight") or name == "llm.norm.weight":
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g, device=device))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device=device))
        pe = M.get_2d_sincos_pos_embed(cfg.hidden_size, model.pos_embed_max_size, interpolation_scale=1.0, base_size=64)
        model.pos_embed.copy_(torch.from_numpy(pe).float().unsqueeze(0))
    return model.to(BF).eval()


def visible_pairs(mask, valid):
    """number of visible (q,k) pairs over real (non-pad) query rows — attention FLOPs = 4*H*pairs per layer."""
    return int(mask[valid].sum().item())


def calibrate(device, stream, ops, tag):
    """What THIS box sustains right now, measured in this process: the pure-MFMA loop on random
    register operands (vgpt_calib_mfma), a 1-GiB streaming copy (vgpt_calib_copy) and an 8192^3 bf16 GEMM through the
    product kernel (ops.linear), each warmed up once and timed with HIP events on `stream`.  Reported beside the
    timed region, never inside it; a step time is read against these to tell a slow box from slow code."""
    L = importlib.import_module("video-gpt_amd._lib")
    lib = L.load()
    res = {"when": tag}
    with torch.cuda.stream(stream):
        sp = stream.cuda_stream

        def timed(fn, reps):
            fn()
            s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s_.record(stream)
            for _ in range(reps):
                fn()
            e_.record(stream)
            stream.synchronize()
            return s_.elapsed_time(e_) * 1e-3 / reps
        out = torch.zeros(4, dtype=torch.float32, device=device)
        iters = 40000
        t = timed(lambda: L.call("vgpt_calib_mfma", out.data_ptr(), iters, sp), 3)
        res["mfma_loop_tflops"] = round(lib.vgpt_calib_mfma_flops(iters) / t / 1e12, 1)
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device=device).fill_(1)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
        t = timed(lambda: L.call("vgpt_calib_copy", src.data_ptr(), dst.data_ptr(), nbytes, sp), 5)
        res["hbm_copy_gbs"] = round(2 * nbytes / t / 1e9, 1)
        del src, dst
        g = torch.Generator(device=device).manual_seed(1234)
        a = (torch.randn(8192, 8192, generator=g, device=device) * 0.5).to(BF)
        w = (torch.randn(8192, 8192, generator=g, device=device) * 0.02).to(BF)
        c = torch.empty(8192, 8192, dtype=BF, device=device)
        t = timed(lambda: ops.linear(a, w, out=c), 10)
        res["gemm_8192_tflops"] = round(2 * 8192.0 ** 3 / t / 1e12, 1)
        del a, w, c
    torch.cuda.empty_cache()
    return res


def pmc_records(kind):
    """Counter evidence of one kernel kind from the committed rocprofv3 --pmc passes of THIS command (separate runs, as
    MI355X_MICROARCH.md prescribes; scripts/pmc_traffic.py / pmc_mfma.py): beyond-L2 bytes per launch and matrix-pipe busy
    fraction, each with the file it came from.  Not measured in this run -- `source` says so."""
    pat = {"gate_up": ("gemm_w4_kernel<1", ""), "qkv_rope": ("gemm_w4_kernel<2", ""),
           "o_proj": ("gemm_w4_kernel<0", ""), "down_proj": ("gemm_w4_kernel<0", ""),
           "attn_fwd": ("attn_fwd_kernel<96", "")}[kind]
    out = {}
    for key, field, stems in (("traffic_bytes_per_launch", "traffic_bytes_per_launch", ("r04_pmc_traffic",)),
                              ("mfma_busy", "mfma_busy_frac", ("r04_pmc_mfma",))):
        for stem in stems:
            path = os.path.join(ROOT, "profiles", stem + ".json")
            if not os.path.exists(path):
                continue
            try:
                recs = [(rec["launches"], rec[field]) for name, rec in json.load(open(path))["kernels"].items()
                        if pat[0] in name and pat[1] in name]
            except Exception:
                recs = []
            if recs:
                n = sum(a for a, _ in recs)
                out[key] = round(sum(a * v for a, v in recs) / n, 4) if key == "mfma_busy" else int(sum(a * v for a, v in recs) / n)
                out[key + "_source"] = f"profiles/{stem}.json (separate --pmc pass, not this run)"
                break
    if kind in ("o_proj", "down_proj") and out:
        out["note"] = "o_proj and down_proj share one kernel: the counter figures are their launch-weighted mix"
    return out


def cpu_baseline(cfg_full, batch, layers_sample=2, with_vae=False):
    """CPU restatement of the reference path (oracle/restate.py) on a bounded sample: `layers_sample`
    of the 32 layers at full width over the full cfg-2 sequence; throughput scaled by layers."""
    from oracle import restate as R
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    oc = R.Phi3Cfg(hidden_size=cfg_full.hidden_size, intermediate_size=cfg_full.intermediate_size,
                   num_hidden_layers=layers_sample, num_attention_heads=cfg_full.num_attention_heads,
                   num_key_value_heads=cfg_full.num_key_value_heads, vocab_size=64, pad_token_id=2)
    p = R.make_params(oc, seed=0, pos_embed=False)
    B, L = batch["input_ids"].shape
    x = torch.randn(B, L, oc.hidden_size, generator=torch.Generator("cpu").manual_seed(1)) * 0.5
    with torch.no_grad():
        t0 = time.perf_counter()
        R.transformer(p, oc, x, batch["attention_mask"], batch["position_ids"])
        dt = time.perf_counter() - t0
    step_s = dt * cfg_full.num_hidden_layers / layers_sample
    # cfg-1 of BASELINE.json (4 condition frames, ONE denoise step, CPU fp32, VAE included): composed from the step above
    # and, with --cpu-baseline-vae, the VAE restatement (oracle/vae_ref.py, sdxl-vae configuration) timed on one 256^2
    # frame each way.  Off by default: on the GPU box's 256 host threads torch's CPU convolutions take 23 s (encode) and
    # 29 s (decode) per frame whatever the resolution (first-call primitive set-up dominates), which alone would
    # overrun the default run's CPU budget.
    cfg1 = {"denoise_step_s": round(step_s, 1), "vae": "not timed in this run (--cpu-baseline-vae)",
            "what": "cfg-1 shape (4 condition frames encoded, 1 denoise step of the 8-frame clip with CFG, 8 frames decoded), "
                    "fp32 on the host cores: the step scaled from the layer sample"}
    if with_vae:
        from oracle import vae_ref as VR
        vc = VR.VaeCfg()
        vp = VR.make_vae_params(vc, seed=0)
        gen = torch.Generator("cpu").manual_seed(2)
        with torch.no_grad():
            t0 = time.perf_counter()
            VR.vae_encode(vp, vc, torch.rand(1, 3, 256, 256, generator=gen) * 2 - 1, torch.randn(1, 4, 32, 32, generator=gen))
            enc_s = time.perf_counter() - t0
            t0 = time.perf_counter()
            VR.decode_to_uint8(vp, vc, torch.randn(1, 4, 32, 32, generator=gen))
            dec_s = time.perf_counter() - t0
        cfg1.update({"vae": "one 256^2 frame each way, first call", "vae_encode_s_per_frame": round(enc_s, 3),
                     "vae_decode_s_per_frame": round(dec_s, 3), "round_s": round(4 * enc_s + step_s + 8 * dec_s, 1)})
    return {"value": 8 * 256 / step_s, "unit": "clip-tokens/s", "cores": threads, "kind": "port", "cfg1_cpu_round": cfg1,
            "sample": f"{layers_sample} of {cfg_full.num_hidden_layers} decoder layers (fp32, torch CPU) over the full "
                      f"B=2 x L={L} cfg-2 sequence took {dt:.2f}s; scaled x{cfg_full.num_hidden_layers // layers_sample} "
                      "to one denoise step (embedders/final layer/Euler update are <0.1% and omitted)"}


def synthetic_vae(device):
    V = importlib.import_module("video-gpt_amd.vae")
    vae = V.AutoencoderKL()
    g = torch.Generator("cpu").manual_seed(0)
    with torch.no_grad():
        for n_, p_ in vae.named_parameters():
            if p_.dim() > 1:
                p_.copy_(torch.randn(p_.shape, generator=g) / p_[0].numel() ** 0.5)
            elif "norm" in n_ and n_.endswith("weight"):
                p_.copy_(1 + 0.1 * torch.randn(p_.shape, generator=g))
            else:
                p_.copy_(0.02 * torch.randn(p_.shape, generator=g))
    return vae.to(device, torch.float32).eval()


def bench_pipeline(args, rank, world, device, M, P, D):
    """End-to-end LVMPipeline.prompt_condition_frame_block_autoregressive_inference (LVM/pipeline.py:347-595):
    VAE-encode 4 condition frames (256^2), sample 8-frame clips with CFG (`--steps` Euler steps each, x1 prediction,
    hipGraph sampler with condition-prefix reuse), VAE-decode, uint8 frames; `--rounds` chained clips with a
    16-frame window (cfg-5: --rounds 8).  One "step" of this workload = one whole round."""
    PL = importlib.import_module("video-gpt_amd.pipeline")
    cfg = full_config(M, args.layers)
    model = build_model(M, cfg, device, seed=0)
    pipe = PL.LVMPipeline(synthetic_vae(device), model, P.LVMProcessor(P.SpecialTokenizer(10, 11, 12)), device=device)
    pipe.vae.conv_precision = args.vae_precision
    pipe.attention_precision = args.attn_precision
    g = torch.Generator("cpu").manual_seed(7 + rank)
    frames = [torch.rand(3, 256, 256, generator=g) * 2 - 1 for _ in range(4)]
    kw = dict(input_images=frames, height=256, width=256, num_inference_steps=args.steps, use_img_guidance=True,
              img_guidance_scale=1.6, seed=42, output_type="pt", prediction_type="x1", clean_image_noise_level=0.05,
              max_frame_window=16)
    pipe.prompt_condition_frame_block_autoregressive_inference(gen_nums=[8], **dict(kw, num_inference_steps=2))  # warm-up
    out = []

    def run():
        out.append(pipe.prompt_condition_frame_block_autoregressive_inference(gen_nums=[8] * args.rounds, **kw))
    elapsed = D.timed_region(run, torch.cuda.synchronize, device)
    n_gen = 8 * args.rounds
    if rank == 0:
        print(json.dumps(tag_rehearsal({"metric": "end-to-end denoised clip-tokens/sec incl. VAE encode/decode (256^2, 8-frame next-clip, CFG, x1)",
                          "value": round(world * n_gen * 256 * args.steps / elapsed, 1), "unit": "clip-tokens/s", "n_gpus": world,
                          "steps": args.rounds, "warmup": 1, "ms_per_step": round(elapsed / args.rounds * 1e3, 1),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "bf16" if args.attn_precision == "bf16" else "bf16 (attention operands MX-fp8 e4m3)", "data": "synthetic",
                          "config": {"workload": f"LVMPipeline next-clip rollout: {args.rounds} round(s) x 8 frames, {args.steps} Euler steps, "
                                                 f"C=4 condition frames on round 0 then a 16-frame window, fp32 VAE ({args.vae_precision} convolutions), bf16 denoiser"
                                                 + (" with MX-fp8 attention in the sampler steps" if args.attn_precision == "fp8" else ""),
                                     "frames_returned": len(out[0]), "generated_frames_per_s": round(world * n_gen / elapsed, 2)},
                          "roofline": None}, args)), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


def measure_vae(steps, warmup, device, D, precision="bf16x3", nfr=8, world=1):
    """VAE decode (+uint8) and encode of `nfr` 256^2 frames per step (sdxl-vae architecture, fp32 tensors like the
    reference, LVM/pipeline.py:558-590 / 60-68).  Algorithmic work per 256^2 frame: decode 0.622 TFLOP /
    1.68 GB fp32 ideal-fusion traffic, encode 0.273 TFLOP / 0.97 GB.  The 3x3 convolutions dominate; with
    conv_precision "bf16x3" (default) each fp32 product is three bf16 MFMA products (hi*hi + hi*lo + lo*hi), so the
    matrix pipe ISSUES 3x the algorithmic FLOPs: priced against the dense bf16 peak; "fp32" uses the fp32-input MFMA
    (157.3 TFLOP/s peak)."""
    vae = synthetic_vae(device)
    vae.conv_precision = precision
    g = torch.Generator("cpu").manual_seed(0)
    z = torch.randn(nfr, 4, 32, 32, generator=g).to(device)
    x = (torch.rand(nfr, 3, 256, 256, generator=g) * 2 - 1).to(device)
    issued, peak, kern = (3.0, 2500.0, "conv_bx3_kernel (3x3 conv as implicit GEMM, 3 bf16 MFMA products per fp32 product)") \
        if precision == "bf16x3" else (1.0, 157.3, "conv_kernel (3x3 conv as implicit GEMM on the fp32-input MFMA)")
    res = {"conv_precision": precision, "frames_per_step": nfr, "kernel": kern}
    for name, fn, tf, gb in (("decode", lambda: vae.decode_to_uint8(z), 0.622, 1.68), ("encode", lambda: vae.encode(x), 0.273, 0.97)):
        for _ in range(warmup):
            fn()

        def run():
            for _ in range(steps):
                fn()
        el = D.timed_region(run, torch.cuda.synchronize, device)
        per_frame = el / steps / nfr
        res[name] = {"ms_per_frame": round(per_frame * 1e3, 3), "frames_per_s": round(world / per_frame, 1),
                     "mfma": {"alg_tflop_per_frame": tf, "issued_tflop_per_frame": round(issued * tf, 3),
                              "achieved_issued_tflops": round(issued * tf / per_frame, 1), "peak": peak,
                              "frac": round(issued * tf / per_frame / peak, 4),
                              "alg_frac": round(tf / per_frame / peak, 4)},
                     "hbm_ideal_fusion": {"gb_per_frame": gb, "achieved_gbs": round(gb / per_frame, 1), "peak": 8000.0,
                                          "frac": round(gb / per_frame / 8000.0, 4)}}
    return res


def bench_vae(args, rank, world, device, D):
    res = measure_vae(args.steps, args.warmup, device, D, args.vae_precision, world=world)
    if rank == 0:
        d = res["decode"]
        nfr = res["frames_per_step"]
        print(json.dumps(tag_rehearsal({"metric": f"VAE decode frames/sec (256^2, sdxl-vae, fp32 tensors, {args.vae_precision} convolutions)", "value": d["frames_per_s"], "unit": "frames/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(d["ms_per_frame"] * nfr, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "f32 (bf16x3 products)" if args.vae_precision == "bf16x3" else "f32", "data": "synthetic",
                          "config": {"workload": f"VAE decode+uint8 of {nfr} 256^2 frames per step (also encode)", "detail": res},
                          "roofline": {"bound": "mfma", "kernel": res["kernel"], "achieved": d["mfma"]["achieved_issued_tflops"],
                                       "peak": d["mfma"]["peak"], "unit": "TFLOP/s", "frac": d["mfma"]["frac"], "traffic": None,
                                       "note": "issued MFMA FLOPs (3x algorithmic for bf16x3) against the dense peak of the MFMA used; "
                                               "alg_frac in config.detail prices the algorithmic FLOPs",
                                       "hbm_view": d["hbm_ideal_fusion"]}}, args)), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


REHEARSAL_TAG = " [REHEARSAL: ranks share one GPU over gloo: INVALID]"


def tag_rehearsal(line, args):
    """--rehearse-on-one-gpu exercises the N > 1 code path with every rank on cuda:0: no such line is a measurement."""
    if getattr(args, "rehearse_on_one_gpu", False):
        line["config"]["workload"] += REHEARSAL_TAG
        line["valid"] = False
    return line


def bench_stage1(args, rank, world, device, M, P, D, ops):
    cal = None
    if rank == 0 and not args.no_calibration:
        cal = [calibrate(device, torch.cuda.Stream(device=device), ops, "before the timed region")]
    line = measure_stage1(args.steps, args.warmup, args.layers, rank, world, device, M, P, D, grad_ckpt=args.grad_ckpt)
    if cal is not None:
        cal.append(calibrate(device, torch.cuda.Stream(device=device), ops, "right after the timed region"))
    line["calibration"] = cal
    line = tag_rehearsal(line, args)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


def measure_stage1(steps, warmup, layers, rank, world, device, M, P, D, model=None, F=8, hw=(32, 32), bs=2, grad_ckpt=False):
    """cfg-3: stage-1 pre-training, bs 2 clips/GPU of F=8 frames at 256^2 (2 x 3870 tokens), bf16 params with fp32
    master AdamW, gradient all-reduce over RCCL (one bucket per decoder layer, overlapped with backward).
    One step = forward + backward + all-reduce + clip + AdamW.  samples/sec = 2*world*steps/time."""
    TR = importlib.import_module("video-gpt_amd.train")
    N = (hw[0] // 2) * (hw[1] // 2)
    cfg = full_config(M, layers)
    if model is None:
        model = build_model(M, cfg, device, seed=0)
    proc = P.LVMProcessor(P.SpecialTokenizer(10, 11, 12))
    rows = []
    for _ in range(bs):
        prompt = "".join(f"<|diffusion|><|image_{i + 1}|><img><|image_{i + 1}|></img>" if i < F - 1
                         else f"<|diffusion|><|image_{i + 1}|>" for i in range(F))
        rows.append(proc.process_multi_modal_prompt_training(prompt, [torch.zeros(3, hw[0] * 8, hw[1] * 8) for _ in range(F)]))
    batch = proc.collator.collate_stage1(rows, F)
    batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items() if k not in ("input_pixel_values", "output_images")}
    g = torch.Generator("cpu").manual_seed(100 + rank)
    nd, nc = bs * F, bs * (F - 1)
    mk = lambda n: torch.randn(n, 4, *hw, generator=g).to(device)
    x1, x0, clean, x0i = mk(nd), mk(nd), mk(nc), mk(nc)
    t = torch.rand(nd, generator=g).to(device)
    ti = (0.9 + 0.1 * torch.rand(nc, generator=g)).to(device)
    # the update of step k runs on its own stream under the forward of step k + 1 (Stage1Trainer.overlap_optimizer); the timed
    # region's closing device-wide synchronize includes the last update
    trainer = TR.Stage1Trainer(model, lr=1e-4, weight_decay=0.1, max_grad_norm=1.0, gradient_checkpointing=grad_ckpt,
                               overlap_optimizer=os.environ.get("VGPT_OPT_OVERLAP", "1") == "1")
    for _ in range(warmup):
        loss = trainer.step(batch, x1, x0, t, clean, x0i, ti)
    losses = []

    def run():
        for _ in range(steps):
            losses.append(trainer.step(batch, x1, x0, t, clean, x0i, ti))
    elapsed = D.timed_region(run, torch.cuda.synchronize, device)
    ms = elapsed / max(steps, 1) * 1e3
    H, I, nl = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    real = bs * (2 * F - 1) * (N + 2)
    mask = batch["attention_mask"]
    pairs = int(mask.sum().item())  # no pad rows at equal lengths
    fwd = (2 * (4 * H * H + 3 * H * I) * real + 4 * H * pairs) * nl
    loss_v = [float(l.mean()) for l in (losses[0], losses[-1])]
    if True:
        line = {"metric": f"train samples/sec ({hw[0] * 8}^2, {F}-frame clips, bs {bs}/GPU, DP)", "value": round(world * bs * steps / elapsed, 3),
                "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
                "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "bf16", "data": "synthetic",
                "config": {"workload": f"stage-1-layout pretrain step: bs {bs}/GPU x F={F} frames {hw[0] * 8}^2 ({real} tokens/GPU), "
                                       f"Phi-3-mini-class denoiser {nl} layers, bf16 + fp32-master AdamW, clip 1.0"
                                       + (", gradient checkpointing per decoder layer" if grad_ckpt else "")
                                       + ("" if nl == 32 else " [DEBUG layer count: INVALID]"),
                           "global_batch": world * bs, "parallelism": f"dp{world}", "loss_first_last": loss_v},
                "roofline": {"bound": "mfma", "kernel": "whole step (fwd + bwd ~ 3 x fwd FLOPs)", "achieved": round(3 * fwd / (ms * 1e-3) / 1e12, 1),
                             "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(3 * fwd / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                             "traffic": None, "alg_tflop_per_step": round(3 * fwd / 1e12, 1)}}
    # ---- data-parallel runs: how much of the gradient all-reduce is exposed.  Two more short legs AFTER the measurement
    #      above: the same steps with the all-reduce skipped (the ranks' weights drift apart: timing only), and the
    #      buckets reduced back to back with nothing to overlap with ----
    comm = None
    if world > 1:
        import torch.distributed as dist
        trainer.skip_allreduce = True
        trainer.step(batch, x1, x0, t, clean, x0i, ti)

        def run_nocomm():
            for _ in range(steps):
                trainer.step(batch, x1, x0, t, clean, x0i, ti)
        ms_nocomm = D.timed_region(run_nocomm, torch.cuda.synchronize, device) / max(steps, 1) * 1e3
        trainer.skip_allreduce = False
        # the same steps with every bucket reduced BEHIND the backward instead of under it (VGPT_DP_OVERLAP=0): tells whether
        # the overlap pays on this node or RCCL's kernels cost the backward's GEMMs more CUs than they hide
        was = trainer.overlap_allreduce
        trainer.overlap_allreduce = False
        trainer.step(batch, x1, x0, t, clean, x0i, ti)

        def run_serial():
            for _ in range(steps):
                trainer.step(batch, x1, x0, t, clean, x0i, ti)
        ms_serial = D.timed_region(run_serial, torch.cuda.synchronize, device) / max(steps, 1) * 1e3
        trainer.overlap_allreduce = was
        buckets = list(trainer.layer_buckets) + [trainer.small_bucket]

        def run_comm():
            for _ in range(steps):
                hs = [dist.all_reduce(b, async_op=True) for b in buckets]
                for h in hs:
                    h.wait()
        run_comm()
        ms_comm = D.timed_region(run_comm, torch.cuda.synchronize, device) / max(steps, 1) * 1e3
        nbytes = sum(b.numel() * b.element_size() for b in buckets)
        comm = {"ms_per_step_without_allreduce": round(ms_nocomm, 2), "exposed_ms_per_step": round(ms - ms_nocomm, 2),
                "allreduce_alone_ms": round(ms_comm, 2), "ms_per_step_allreduce_after_backward": round(ms_serial, 2),
                "allreduce_bytes_per_step": nbytes, "buckets": len(buckets),
                "allreduce_alone_busbw_gbs": round(2 * (world - 1) / world * nbytes / (ms_comm * 1e-3) / 1e9, 1),
                "hidden_frac": round(max(0.0, 1.0 - (ms - ms_nocomm) / ms_comm), 3) if ms_comm > 0 else None}
    line["comm"] = comm
    del trainer
    return line


def self_launch_argv(argv, n_gpus, port=None):
    """The command a bare `python bench.py --gpus N ...` (N > 1, no WORLD_SIZE in the environment) re-launches itself with:
    N ranks of this same file with the same flags under torch.distributed.run on 127.0.0.1 (the driver's own launch line)."""
    if port is None:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer layers => INVALID as a benchmark")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-vae", action="store_true", help="also time the CPU VAE restatement (about a minute of CPU)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-stage1", action="store_true",
                    help="skip the short stage-1 data-parallel training measurement appended to the default line")
    ap.add_argument("--stage1-steps", type=int, default=10, help="timed steps of the stage-1 leg of the default line (2 warm-up)")
    ap.add_argument("--gemm-family", type=int, choices=[0, 1], default=0,
                    help="0 = default kernels (four-wave hand-scheduled GEMM where it applies), 1 = eight-wave GEMM kernels only "
                         "(same-box A/B; include/vgpt.h vgpt_gemm_set_family)")
    ap.add_argument("--no-calibration", action="store_true", help="skip the box calibration launches around the timed region")
    ap.add_argument("--attn-precision", choices=["bf16", "fp8"], default="bf16",
                    help="infer / pipeline workloads: operands of the sampler steps' attention (fp8 = the cfg-5 option; "
                         "the headline metric is quoted on bf16)")
    ap.add_argument("--grad-ckpt", action="store_true",
                    help="stage1 / stage4 workloads: gradient checkpointing per decoder layer (OmniGen/transformer.py:182-192)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debug only: all ranks share cuda:0 and talk over gloo (numbers are INVALID as a benchmark)")
    ap.add_argument("--no-vae", action="store_true", help="infer workload: skip the VAE decode / encode leg of the JSON line")
    ap.add_argument("--no-prefix-reuse", action="store_true",
                    help="recompute the condition frames at every step exactly as the reference does")
    ap.add_argument("--no-hoist", action="store_true",
                    help="keep the <|diffusion|> / time rows of the noisy frames in the per-step row set (engine.py)")
    ap.add_argument("--vae-precision", choices=["fp32", "bf16x3"], default="bf16x3",
                    help="vae / pipeline workloads: arithmetic of the 3x3 convolutions (video-gpt_amd/vae.py)")
    ap.add_argument("--breakdown", action="store_true", help="add per-operator HIP-event times of one eager denoise step")
    ap.add_argument("--rounds", type=int, default=1, help="pipeline workload: chained next-clip rounds (cfg-5 uses 8)")
    ap.add_argument("--workload", choices=["infer", "stage1", "stage4", "vae", "pipeline"], default="infer",
                    help="infer = cfg-2 next-clip denoise (default, BASELINE metric part 1); "
                         "stage1 = cfg-3 stage-1 pre-training step, bs 2/GPU, data-parallel (metric part 2)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a bare `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under torch.distributed.run, the
        # launch of LVM/script/train/pretrain_stage1_nv.sh:15-49) as a CHILD process, before anything here touches the GPU,
        # and leave with its exit code
        import subprocess
        raise SystemExit(subprocess.call(self_launch_argv(sys.argv[1:], args.gpus)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if args.rehearse_on_one_gpu:   # every rank on cuda:0, gloo collectives: exercises the N > 1 code path on a 1-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    importlib.import_module("video-gpt_amd")
    D = importlib.import_module("video-gpt_amd.dist_utils")
    D.init_from_env("gloo" if args.rehearse_on_one_gpu else "nccl", device)
    M = importlib.import_module("video-gpt_amd.model")
    P = importlib.import_module("video-gpt_amd.processor")
    E = importlib.import_module("video-gpt_amd.engine")
    S = importlib.import_module("video-gpt_amd.scheduler")
    ops = importlib.import_module("video-gpt_amd.ops")
    importlib.import_module("video-gpt_amd._lib").load().vgpt_gemm_set_family(args.gemm_family)

    if args.workload == "stage1":
        return bench_stage1(args, rank, world, device, M, P, D, ops)
    if args.workload == "vae":
        return bench_vae(args, rank, world, device, D)
    if args.workload == "pipeline":
        return bench_pipeline(args, rank, world, device, M, P, D)
    if args.workload == "stage4":   # cfg-4 shapes: 512^2, 16-frame clips (L = 31 806), bs 1/GPU, stage-1 interleaved layout
        line = measure_stage1(args.steps, args.warmup, args.layers, rank, world, device, M, P, D, F=16, hw=(64, 64), bs=1,
                              grad_ckpt=args.grad_ckpt)
        line["peak_memory_gb"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
        if rank == 0:
            print(json.dumps(tag_rehearsal(line, args)), flush=True)
        return

    # ---- workload: cfg-2 ----
    C, G, hw = 4, 8, (32, 32)
    N = (hw[0] // 2) * (hw[1] // 2)
    cfg = full_config(M, args.layers)
    model = build_model(M, cfg, device, seed=0)
    tok = P.SpecialTokenizer(10, 11, 12)
    proc = P.LVMProcessor(tok, mask_format="layout")   # per-token mask attributes, as LVMPipeline asks for
    prompt = "".join(f"<img><|image_{i + 1}|></img>" if i < C else f"<|diffusion|><|image_{i + 1}|>" for i in range(C + G))
    prompt_ = "".join(f"<|diffusion|><|image_{i + 1}|>" for i in range(G))
    imgs = [torch.zeros(3, hw[0] * 8, hw[1] * 8) for _ in range(C)]
    batch = proc.prompt_condition_frame_block_inference([prompt, prompt_], [imgs, []], height=hw[0] * 8, width=hw[1] * 8,
                                                        use_img_cfg=True, frame_blocks=[C, G])
    dense_mask = batch["attention_mask"].to_bool_tensor()   # only to count the visible pairs (algorithmic FLOPs)
    g = torch.Generator("cpu").manual_seed(42 + rank)
    noise = [torch.randn(1, 4, *hw, generator=g) for _ in range(G)]
    z = [n.to(device, BF) for n in noise] * 2
    cond = [torch.randn(1, 4, *hw, generator=torch.Generator("cpu").manual_seed(1000 + i)).to(device, BF) for i in range(C)]
    total_steps = args.warmup + args.steps
    sched = S.LVMScheduler(num_steps=max(total_steps, 1), time_shifting_factor=1)
    eng = E.StaticDenoiser(model, batch["input_ids"].to(device), batch["position_ids"].to(device),
                           batch["attention_mask"], cond, batch["input_image_sizes"],
                           batch["denoise_image_sizes"], batch["time_emb_inx"], len(z), hw, True, 1.6, "x1",
                           sigma=sched.sigma, reuse_condition_prefix=not args.no_prefix_reuse,
                           hoist_special_rows=not args.no_hoist, attention_precision=args.attn_precision)

    B, L = batch["input_ids"].shape
    valid = batch["input_ids"] != 2
    real_tokens = int(valid.sum())
    H, I, nl = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
    pairs = visible_pairs(dense_mask, valid)
    reuse = eng.S > 0
    if reuse:
        # condition prefix computed once per clip (timed: one prefill per timed region): per-step algorithmic work
        # is the reduced count of SURVEY.md §8d (34.11 TF at cfg-2), the prefill (7.6 TF) is added to the total
        static_tok = C * (N + 2)
        pairs_static = int(dense_mask[0, :static_tok].sum().item())
        real_tokens_step, pairs_step = real_tokens - static_tok, pairs - pairs_static
    else:
        real_tokens_step, pairs_step = real_tokens, pairs
    flops_linear = 2 * (4 * H * H + 3 * H * I) * real_tokens_step * nl
    flops_attn = 4 * H * pairs_step * nl
    flops_step = flops_linear + flops_attn
    # the prefill runs the condition rows only (they never see a later row)
    flops_prefill = (2 * (4 * H * H + 3 * H * I) * static_tok + 4 * H * pairs_static) * nl if reuse else 0

    stream = torch.cuda.Stream(device=device)
    use_graph = not args.no_graph
    calibration = None
    if rank == 0 and not args.no_calibration:
        calibration = [calibrate(device, stream, ops, "before the timed region")]
    with torch.cuda.stream(stream):
        eng.set_latents(torch.cat(z, dim=0))
        if use_graph:
            eng.capture()
        eng.run(args.warmup, use_graph=use_graph)
        stream.synchronize()
        # barrier + synchronize on both sides, MAX over ranks (dist_utils.timed_region)
        def timed():
            eng.per_clip_setup()       # prefill + special rows + adaLN table: once per clip, inside the timed region
            eng.steps_taken = args.warmup
            eng.run(args.steps, use_graph=use_graph)
        elapsed = D.timed_region(timed, torch.cuda.synchronize, device)
        # the per-clip passes alone, once more (reported beside the timed number, never subtracted from it): with
        # --steps other than the scheduler's 50 their share of a step differs from a real clip's
        setup_s = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.per_clip_setup()
        torch.cuda.synchronize()
        setup_s = time.perf_counter() - t0
    if calibration is not None:
        calibration.append(calibrate(device, stream, ops, "right after the timed region"))
    ms_per_step = elapsed / max(args.steps, 1) * 1e3
    value = world * G * N * args.steps / elapsed
    finite = bool(torch.isfinite(eng.z).all().item())

    # ---- roofline: every hot kernel of the step timed live, back to back over the 32 layers on the step's own buffers and
    #      weights (so weights stream from HBM as in the step), one HIP-event pair per kernel KIND on the stream the
    #      launches go to; agrees with the rocprofv3 --stats averages of the graph run (profiles/).  The dominant hand-written
    #      kernel is gemm_bf16_kernel: `achieved` = the algorithmic FLOPs of its launches of a layer over their summed launch
    #      time (gate_up and qkv + RoPE; o_proj / down_proj too when the vendor library is off); `kernels` lists every GEMM of
    #      the layer with the implementation that served it, and the attention forward. ----
    roof = None
    if rank == 0:
        nq_, nk_, hd_ = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        rows = eng.Ma if reuse else real_tokens_step
        with torch.cuda.stream(stream):
            rope_ = eng.rope_a if eng.S else eng.rope
            scratch = torch.empty_like(eng.hid)
            qkv_out = lambda li: (eng.qkv_full[li][eng.S:] if eng.S else eng.qkv)

            def attn_call(l, li):
                if eng.S:
                    return ops.attention_qkv_range(eng.qkv_full[li].view(1, eng.L, -1), eng.pm, nq_, nk_, hd_, eng.S, eng.ctx,
                                                   segments=eng.seg_live, item_rows=eng.attn_item_rows)
                if eng.seg_all is not None:
                    return ops.attention_qkv_range(eng.qkv, eng.pm, nq_, nk_, hd_, 0, eng.ctx, segments=eng.seg_all,
                                                   item_rows=eng.attn_item_rows)
                return ops.attention_qkv(eng.qkv, eng.pm, nq_, nk_, hd_, out=eng.ctx)
            fz = eng.fuse      # RMSNorms folded into the GEMMs (engine.py): the step runs the *_prenorm / *_resid_ssq entries
            if fz is not None:
                ops.rms_rstd(eng.hid, 1e-5, out=fz["rstd_in"])
                ops.rms_rstd(eng.hid, 1e-5, out=fz["rstd_post"])
                rs2 = torch.empty_like(fz["rstd_in"])
                f_gate = lambda l, li: ops.gated_mlp_act_prenorm(eng.hid, fz["wgu"][li], fz["rstd_post"], l.mlp.act, out=eng.act)
                f_qkv = lambda l, li: ops.linear_qkv_rope_prenorm(eng.hid, fz["wq"][li], rope_[0], rope_[1], fz["rstd_in"], nq_, nk_, hd_,
                                                                  out=qkv_out(li))
                f_down = lambda l, li: ops.linear_resid_rstd(eng.act, l.mlp.down_proj.weight, eng.hid, rs2, fz["ws"], 1e-5, out=scratch)
                f_o = lambda l, li: ops.linear_resid_rstd(eng.ctx, l.self_attn.o_proj.weight, eng.hid, rs2, fz["ws"], 1e-5, out=scratch)
            else:
                f_gate = lambda l, li: ops.gated_mlp_act(eng.nrm, l.mlp.gate_up_proj.weight, l.mlp.act, out=eng.act)
                f_qkv = lambda l, li: ops.linear_qkv_rope(eng.nrm, l.self_attn.qkv_proj.weight, rope_[0], rope_[1], nq_, nk_, hd_,
                                                          out=qkv_out(li))
                f_down = lambda l, li: ops.linear(eng.act, l.mlp.down_proj.weight, residual=eng.hid, out=scratch)
                f_o = lambda l, li: ops.linear(eng.ctx, l.self_attn.o_proj.weight, residual=eng.hid, out=scratch)
            tag = " + folded RMSNorm" if fz is not None else ""
            kinds = (("gate_up", "gemm_w4_kernel<MODE_GATED, 8> (four-wave hand-scheduled loop, 256 x 256 tiles; gate_up_proj + act(gate) * up epilogue" + tag + ")",
                      2 * rows * H * 2 * I, f_gate),
                     ("qkv_rope", "gemm_w4_kernel<MODE_ROPE, 9> (four-wave hand-scheduled loop, 256 x 288 tiles; qkv_proj + RoPE epilogue" + tag + ")",
                      2 * rows * H * 3 * H, f_qkv),
                     ("down_proj", "gemm_w4_kernel<MODE_PLAIN, 6> (four-wave hand-scheduled loop, 256 x 192 tiles; down_proj + residual"
                      + (" + 1 / rms of the output rows for the next norm" if fz is not None else "") + ")", 2 * rows * I * H, f_down),
                     ("o_proj", "gemm_w4_kernel<MODE_PLAIN, 6> (four-wave hand-scheduled loop, 256 x 192 tiles; o_proj + residual"
                      + (" + 1 / rms of the output rows for the next norm" if fz is not None else "") + ")", 2 * rows * H * H, f_o),
                     ("attn_fwd", "attn_fwd_kernel<96, true, 4, P2> (block-masked flash attention, planned launch, hand-scheduled tile bodies)", flops_attn // nl, attn_call))
            # (1) every kind on its own, back to back over the 32 layers: one event pair per kind
            iso = {}
            for name, kname, alg, fn in kinds:
                for li in (0, 1):
                    fn(model.llm.layers[li], li)
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(stream)
                for li, l in enumerate(model.llm.layers):
                    fn(l, li)
                e_.record(stream)
                stream.synchronize()
                iso[name] = s_.elapsed_time(e_) / nl * 1e3
            # (2) inside one real eager denoise forward (every kernel of the step in its own order, norms and glue included:
            #     the clock the part holds and the cache state are the step's; 32 back-to-back launches of the most
            #     power-hungry kernel, gate_up, run 15-20 % slower than the same launches spread through the step).
            #     Pass A: a HIP-event pair around every launch of the five kinds + one pair around the whole forward;
            #     pass B: the outer pair alone.  Each inner pair costs c = (A - B) / 160 of marker / dispatch time, and a
            #     kind's in-step average = its pairs' average - c.
            pairs = {k[0]: [] for k in kinds}
            saved = {n: getattr(ops, n) for n in ("linear", "linear_qkv_rope", "gated_mlp_act", "attention_qkv_range", "attention_qkv",
                                                  "linear_resid_rstd", "linear_qkv_rope_prenorm", "gated_mlp_act_prenorm")}

            def wrap(fn, kind_of):
                def f(*a_, **k_):
                    kind = kind_of(*a_, **k_)
                    if kind is None:
                        return fn(*a_, **k_)
                    s1, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s1.record(stream)
                    out = fn(*a_, **k_)
                    e1.record(stream)
                    pairs[kind].append((s1, e1))
                    return out
                return f
            lin_kind = lambda x, w, *a_, **k_: ("o_proj" if w.shape[1] == nq_ * hd_ and w.shape[0] == H else
                                                "down_proj" if w.shape[1] == I else None)

            L_ = importlib.import_module("video-gpt_amd._lib")
            sink = torch.zeros(4, dtype=torch.float32, device=device)

            def one_forward():
                eng.step.zero_()
                ops.sampler_set_timesteps(eng.sigma, eng.step, eng.ts)
                # ~20 ms of matrix-pipe work in FRONT of the forward: the host enqueues all of the forward's launches (and
                # their event records) while it runs, so no timed interval contains a wait for the host
                L_.call("vgpt_calib_mfma", sink.data_ptr(), 80000, stream.cuda_stream)
                s1, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s1.record(stream)
                eng.forward_step(from_tables=True)
                e1.record(stream)
                stream.synchronize()
                return s1.elapsed_time(e1) * 1e3
            one_forward()
            t_b = min(one_forward() for _ in range(3))
            try:
                ops.linear = wrap(saved["linear"], lin_kind)
                ops.linear_qkv_rope = wrap(saved["linear_qkv_rope"], lambda *a_, **k_: "qkv_rope")
                ops.gated_mlp_act = wrap(saved["gated_mlp_act"], lambda *a_, **k_: "gate_up")
                ops.linear_resid_rstd = wrap(saved["linear_resid_rstd"], lin_kind)
                ops.linear_qkv_rope_prenorm = wrap(saved["linear_qkv_rope_prenorm"], lambda *a_, **k_: "qkv_rope")
                ops.gated_mlp_act_prenorm = wrap(saved["gated_mlp_act_prenorm"], lambda *a_, **k_: "gate_up")
                ops.attention_qkv_range = wrap(saved["attention_qkv_range"], lambda *a_, **k_: "attn_fwd")
                ops.attention_qkv = wrap(saved["attention_qkv"], lambda *a_, **k_: "attn_fwd")
                one_forward()
                for v_ in pairs.values():
                    v_.clear()
                t_a = one_forward()
            finally:
                for n_, f_ in saved.items():
                    setattr(ops, n_, f_)
            n_pairs = sum(len(v_) for v_ in pairs.values())
            m_us = {n: sum(a_.elapsed_time(b_) for a_, b_ in v_) / max(len(v_), 1) * 1e3 for n, v_ in pairs.items()}
            c_us = (t_a - t_b) / max(n_pairs, 1)
            layer_us = t_b / nl
            klist = []
            for name, kname, alg, fn in kinds:
                us = m_us[name] - c_us
                rec = {"name": name, "kernel": kname, "implementation": "hand-written HIP",
                       "launches_per_step": nl, "avg_us": round(us, 1),
                       "avg_us_isolated_back_to_back": round(iso[name], 1),
                       "alg_gflop_per_launch": round(alg / 1e9, 1), "achieved_tflops": round(alg / us / 1e6, 1),
                       "frac": round(alg / us / 1e6 / PEAK_BF16_TFLOPS, 4), "share_of_step": round(us * nl / (ms_per_step * 1e3), 3)}
                rec.update(pmc_records(name))
                klist.append(rec)
            timing_note = {"method": "avg_us = inside one real eager denoise forward (a HIP-event pair around every launch of the "
                                     "five kinds, minus the fixed per-pair cost c = (forward with inner pairs - forward without) "
                                     "/ pairs); avg_us_isolated_back_to_back = 32 launches of one kind in a row (lower clock for "
                                     "the MFMA-dense kinds)",
                           "event_pair_cost_us": round(c_us, 2), "pairs": n_pairs,
                           "eager_forward_us_per_layer": round(layer_us, 1)}
        # a yardstick from OUTSIDE the product, timed beside it and never part of `value`: torch.addmm / torch.matmul (hipBLASLt
        # through PyTorch) on the plain products of a layer, 32 launches back to back over the layers' own weights, to be read
        # against `avg_us_isolated_back_to_back` of the hand-written kernels (same launch pattern)
        vendor_yard = None
        try:
            with torch.cuda.stream(stream):
                yard = {}
                two_d = lambda t_: t_.reshape(-1, t_.shape[-1])[:rows]
                res_ = two_d(eng.hid)
                prods = (("o_proj", two_d(eng.ctx), lambda l: l.self_attn.o_proj.weight, True),
                         ("down_proj", two_d(eng.act), lambda l: l.mlp.down_proj.weight, True),
                         ("qkv_proj_plain_gemm_only", two_d(eng.nrm), lambda l: l.self_attn.qkv_proj.weight, False),
                         ("gate_up_plain_gemm_only", two_d(eng.nrm), lambda l: l.mlp.gate_up_proj.weight, False))
                for name, x_, wf_, with_res in prods:
                    f_ = (lambda w_: torch.addmm(res_, x_, w_.t())) if with_res else (lambda w_: torch.matmul(x_, w_.t()))
                    for l in model.llm.layers[:2]:
                        f_(wf_(l))
                    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s_.record(stream)
                    for l in model.llm.layers:
                        f_(wf_(l))
                    e_.record(stream)
                    stream.synchronize()
                    yard[name] = round(s_.elapsed_time(e_) / nl * 1e3, 1)
            vendor_yard = {"what": "torch.addmm / torch.matmul (hipBLASLt inside PyTorch) on the same operands, 32 launches back to "
                                   "back; compare with kernels[].avg_us_isolated_back_to_back; not part of the product or of `value`",
                           "avg_us_isolated_back_to_back": yard}
        except Exception as ex:   # the yardstick must never break the benchmark line
            vendor_yard = {"error": repr(ex)[:200]}
        # the dominant HAND-WRITTEN kernel: gemm_bf16_kernel, all four GEMM instantiations of a decoder layer
        gem = [k for k in klist if k["name"] != "attn_fwd" and k["implementation"] == "hand-written HIP"]
        t_gemm = sum(k["avg_us"] for k in gem) * 1e-6 * nl
        alg = sum(k["alg_gflop_per_launch"] for k in gem) * 1e9 * nl
        n_launch = len(gem) * nl
        achieved = alg / t_gemm / 1e12
        traf = [k.get("traffic_bytes_per_launch") for k in gem]
        roof = {"bound": "mfma", "kernel": "gemm_w4_kernel / gemm_bf16_kernel (the hand-written GEMMs of a decoder layer: "
                                           + ", ".join(k["name"] for k in gem) + f"; {n_launch} launches per step)",
                "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                "traffic": int(sum(traf) / len(traf)) if all(t is not None for t in traf) else None,
                "traffic_source": gem[0].get("traffic_bytes_per_launch_source"),
                "launches": n_launch, "avg_launch_us": round(t_gemm / n_launch * 1e6, 1),
                "alg_flops_per_launch": alg / n_launch,
                "gemm_vendor": {"products_per_step": 0,
                                "what": "no vendor library is linked, loaded or called by libvgpt_hip.so: every product of the step "
                                        "runs on the hand-written kernels (round 3's hipBLASLt bridge is gone from the product; "
                                        "csrc/experiments/gemm_lt.hip keeps its source)"},
                "vendor_yardstick": vendor_yard,
                "peak_note": "2500 = nominal dense bf16 peak (MI355X_MICROARCH.md); calibration.mfma_loop_tflops is what "
                             "nothing-but-MFMA loops on random operands sustain on THIS box in this run",
                "kernels": klist, "kernel_timing": timing_note,
                "whole_step": {"alg_tflop": round(flops_step / 1e12, 2), "prefill_tflop_once": round(flops_prefill / 1e12, 2),
                               "achieved": round((flops_step * args.steps + flops_prefill) / elapsed / 1e12, 1),
                               "frac": round((flops_step * args.steps + flops_prefill) / elapsed / 1e12 / PEAK_BF16_TFLOPS, 4)}}

    breakdown = None
    if rank == 0 and args.breakdown:
        names = ["linear", "linear_qkv_rope", "gated_mlp_act", "attention_qkv", "attention_qkv_range", "rmsnorm", "rope_qk_inplace"]
        evs = {n: [] for n in names}
        saved = {n: getattr(ops, n) for n in names}

        def wrap(n):
            def f(*a, **k):
                s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s_.record(stream)
                out = saved[n](*a, **k)
                e_.record(stream)
                evs[n].append((s_, e_))
                return out
            return f
        with torch.cuda.stream(stream):
            for n in names:
                setattr(ops, n, wrap(n))
            try:
                for _ in range(3):
                    for n in names:
                        evs[n].clear()
                    eng.step.zero_()
                    ops.sampler_set_timesteps(eng.sigma, eng.step, eng.ts)
                    eng.forward_step(from_tables=True)
            finally:
                for n in names:
                    setattr(ops, n, saved[n])
            stream.synchronize()
        breakdown = {n: {"calls": len(v), "ms": round(sum(a.elapsed_time(b) for a, b in v), 3)} for n, v in evs.items() if v}

    # ---- second half of the BASELINE metric: stage-1 train samples/sec at this GPU count (data parallel over RCCL),
    #      2 warm-up + --stage1-steps (default 10) timed steps on the same model; reported inside the same JSON line ----
    hoisted, rows_per_step = bool(eng.hoist), int(eng.Ma) if reuse else real_tokens_step
    stage1 = None
    if not args.no_stage1 and args.layers == 32:
        del eng
        torch.cuda.empty_cache()
        try:
            s1 = measure_stage1(args.stage1_steps, 2, args.layers, rank, world, device, M, P, D, model=model)
            stage1 = {k: s1[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup")}
            stage1["config"] = s1["config"]
            stage1["roofline"] = s1["roofline"]
            stage1["comm"] = s1.get("comm")
        except Exception as e:  # keep the headline line even if the training leg fails
            stage1 = {"error": repr(e)[:300]}
    # ---- the round's other stage (LVM/pipeline.py:558-590): VAE decode / encode per 256^2 frame, 5 steps of 8 frames ----
    vae_obj = None
    if not args.no_vae:
        try:
            vae_obj = measure_vae(5, 2, device, D, args.vae_precision, world=world)
        except Exception as e:
            vae_obj = {"error": repr(e)[:300]}
    if rank == 0:
        line = {"metric": "denoised clip-tokens/sec (256^2, 8-frame next-clip, CFG, x1)", "value": round(value, 1),
                "unit": "clip-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16" if args.attn_precision == "bf16" else "bf16 (attention operands MX-fp8 e4m3)",
                "data": "synthetic",
                "config": {"workload": "cfg-2 single-GPU inference: 256^2, C=4 cond + G=8 gen frames, CFG (B=2, L=3096, "
                                       f"{real_tokens} real tokens), Phi-3-mini-class denoiser {nl} layers, x1 prediction, "
                                       "hipGraph sampler step" + ("" if nl == 32 else " [DEBUG layer count: INVALID]"),
                           "global_batch": world, "parallelism": f"replicas x{world}", "graph": use_graph,
                           "condition_prefix_reuse": reuse, "special_row_hoisting": hoisted, "attention_precision": args.attn_precision,
                           "counting_note": "value counts the clip's 8 x 256 image tokens per step; the roofline's algorithmic FLOPs are "
                                            "those of tokens_counted_per_step rows although a hoisted step computes "
                                            "tokens_computed_per_step and the per-clip passes (timed) cover the special rows of "
                                            "warmup + steps steps: an over-count of about 0.25 %",
                           "tokens_computed_per_step": rows_per_step,
                           "tokens_counted_per_step": real_tokens_step,
                           "per_clip_setup_ms": round(setup_s * 1e3, 2),
                           "ms_per_step_of_a_50_step_clip": round(((elapsed - setup_s) / max(args.steps, 1) * 50 + setup_s) / 50 * 1e3, 3),
                           "finite": finite},
                "roofline": roof, "calibration": calibration, "stage1_train": stage1, "vae": vae_obj}
        if breakdown:
            line["breakdown_ms_per_step"] = breakdown
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, dict(batch, attention_mask=dense_mask), with_vae=args.cpu_baseline_vae)
        print(json.dumps(tag_rehearsal(line, args)), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
