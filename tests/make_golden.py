#!/usr/bin/env python3
"""Generate tests/golden/*.npz.  Run in the build container (needs /root/reference):

    python tests/make_golden.py

Vectors whose name starts with `ref_` come from the REFERENCE'S OWN classes executed through
oracle/extract_reference.py: collator, prompt layout, LVMScheduler, TimestepEmbedder, FinalLayer, PatchEmbedMR, sincos
tables (`ref_collator_*`, `ref_scheduler`, `ref_leaf_modules`) and — since round 2 — the reference's LVM /
LVMTraining / Phi3Transformer.forward / new_forward / training loss (`lvm_glue_vectors()`, `ref_lvm_glue_tiny`,
`ref_loss_*_tiny`) and its LVMPipeline (`ref_pipeline_*`), run on CPU fp32 with the third-party imports stubbed as
oracle/extract_reference.py describes (installed transformers 5.x Phi3 blocks adapted to the 4.47.1 call signatures).
They pin the oracle and the product's host logic.  `oracle_tiny_e2e.npz` is the one file produced by the CPU restatement
itself (oracle/restate.py): it freezes the oracle's end-to-end numbers so a later edit of the restatement is noticed; the
restatement in turn equals the `ref_` vectors to <= 1e-5 (tests/test_oracle_pins.py).
Only data is stored (inputs / expected outputs), never reference source text.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import extract_reference as X  # noqa: E402
from oracle import restate as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def pack_mask(m: torch.Tensor):
    a = m.numpy().astype(np.uint8)
    return np.packbits(a.reshape(a.shape[0], -1), axis=-1), np.asarray(a.shape)


def flat_sizes(d):
    """{b: [[s,e],...]} -> (n,3) int array [b,s,e]; {b: [t,...]} -> (n,2)."""
    rows = []
    for b, items in d.items():
        for it in items:
            rows.append([b, *it] if isinstance(it, (list, tuple)) else [b, it])
    return np.asarray(rows, dtype=np.int64).reshape(len(rows), -1) if rows else np.zeros((0, 3), dtype=np.int64)


def save_batch(name, batch, extra=None):
    bits, shape = pack_mask(batch["attention_mask"])
    d = dict(input_ids=batch["input_ids"].numpy(), position_ids=batch["position_ids"].numpy(), mask_bits=bits,
             mask_shape=shape)
    for k in ("input_image_sizes", "denoise_image_sizes", "time_emb_inx", "image_sizes"):
        if k in batch:
            d[k] = flat_sizes(batch[k])
    if extra:
        d.update(extra)
    np.savez_compressed(os.path.join(OUT, name), **d)


def collator_vectors():
    for C, G, N, sp in [(2, 2, 16, 1), (1, 3, 4, 4), (4, 8, 256, 1)]:
        save_batch(f"ref_collator_infer_C{C}G{G}N{N}sp{sp}.npz", X.reference_inference_batch(C, G, N, sp=sp),
                   dict(C=C, G=G, N=N, sp=sp))
    for Fl, N in [([3, 2], 16), ([8, 8], 256)]:
        save_batch(f"ref_collator_stage1_F{'_'.join(map(str, Fl))}N{N}.npz", X.reference_stage1_batch(Fl, N),
                   dict(F=np.asarray(Fl), N=N))
    for fbl, N in [([[1, 2, 2], [3, 1]], 16), ([[4, 4, 8]], 64)]:
        b = X.reference_frame_block_training_batch(fbl, N)
        flat = np.asarray([x for fb in fbl for x in fb] + [-1] + [len(fb) for fb in fbl])
        save_batch(f"ref_collator_fbtrain_{'x'.join(str(len(f)) for f in fbl)}N{N}.npz", b,
                   dict(frame_blocks_flat=flat, N=N))

    # single-target path (LVMPipeline.__call__): LVMProcessor.__call__ / prompt_condition_inference + LVMCollator.__call__
    for n_img, side, hw, cfg_on, sp, via in [(2, 64, (64, 64), True, 1, "call"), (0, 64, (32, 64), True, 1, "call"),
                                             (3, 32, (64, 64), True, 4, "cond"), (1, 64, (64, 32), False, 1, "call")]:
        b = X.reference_single_target_batch(n_img, side, hw, cfg_on, sp, via)
        bits, shape = pack_mask(b["attention_mask"])
        d = dict(input_ids=b["input_ids"].numpy(), position_ids=b["position_ids"].numpy(), mask_bits=bits, mask_shape=shape,
                 input_image_sizes=flat_sizes(b["input_image_sizes"]),
                 padding_lens=np.asarray([0 if p is None else p.shape[1] for p in b["padding_images"]]),
                 n_pixel_values=len(b["input_pixel_values"]))
        np.savez_compressed(os.path.join(OUT, f"ref_collator_call_I{n_img}s{side}o{hw[0]}x{hw[1]}cfg{int(cfg_on)}sp{sp}{via}.npz"), **d)


def scheduler_vectors():
    Sched = X.scheduler_class()
    out = {}
    g = torch.Generator("cpu").manual_seed(7)
    n, shp = 4, (1, 4, 6, 6)
    z0 = [torch.randn(*shp, generator=g) for _ in range(n)]
    a = [torch.randn(*shp, generator=g) * 0.3 for _ in range(n)]
    c = [torch.randn(*shp, generator=g) for _ in range(n)]
    out["z0"] = torch.stack(z0).numpy(); out["a"] = torch.stack(a).numpy(); out["c"] = torch.stack(c).numpy()

    def func(z, timesteps, past_key_values=None, prediction_type="v", **kw):
        pred = [a[j] * z[j] + c[j] * (1 + timesteps[j]) for j in range(len(z))]
        if kw["use_img_cfg"] and prediction_type == "v":  # what LVM.frame_block_forward_with_cfg does for 'v'
            h = len(pred) // 2
            cond = [pred[h + j] + kw["img_cfg_scale"] * (pred[j] - pred[h + j]) for j in range(h)]
            pred = cond + cond
        return pred, None

    for steps in (1, 3):
        for pt in ("x1", "v"):
            for cfg_on in (True, False):
                for shift, begin in ((1, None), (3.0, 0.2)):
                    s = Sched(num_steps=steps, time_shifting_factor=shift, begin_time=begin)
                    z = s([t.clone() for t in z0], func, dict(use_img_cfg=cfg_on, img_cfg_scale=1.6), prediction_type=pt)
                    key = f"steps{steps}_{pt}_cfg{int(cfg_on)}_shift{shift}_begin{begin}"
                    out[key] = torch.stack(z).numpy()
                    out["sigma_" + key] = s.sigma.numpy()
    np.savez_compressed(os.path.join(OUT, "ref_scheduler.npz"), **out)


def leaf_vectors():
    L = X.model_leaf_classes()
    out = {}
    H = 64
    torch.manual_seed(3)
    te = L.TimestepEmbedder(H)
    fl = L.FinalLayer(H, 2, 4)
    pe = L.PatchEmbedMR(2, 4, H, bias=True)
    with torch.no_grad():
        for m in (te, fl, pe):
            for p_ in m.parameters():
                p_.normal_(0, 0.05)
        t = torch.tensor([0.0, 0.013, 0.5, 0.77, 1.0])
        x = torch.randn(3, 9, H)
        cvec = torch.randn(3, H)
        lat = torch.randn(2, 4, 6, 10)
        out["t"] = t.numpy(); out["te_out"] = te(t).numpy(); out["te_sin"] = L.TimestepEmbedder.timestep_embedding(t, 256).numpy()
        out["fl_x"] = x.numpy(); out["fl_c"] = cvec.numpy(); out["fl_out"] = fl(x, cvec).numpy()
        out["pe_x"] = lat.numpy(); out["pe_out"] = pe(lat).numpy()
        for n_, p_ in list(te.named_parameters()):
            out["te." + n_] = p_.numpy()
        for n_, p_ in list(fl.named_parameters()):
            out["fl." + n_] = p_.numpy()
        for n_, p_ in list(pe.named_parameters()):
            out["pe." + n_] = p_.numpy()
    out["sincos_64_12_b64"] = L.get_2d_sincos_pos_embed(64, 12, interpolation_scale=1.0, base_size=64)
    out["sincos_32_7_b1_i2"] = L.get_2d_sincos_pos_embed(32, 7, interpolation_scale=2.0, base_size=1)
    np.savez_compressed(os.path.join(OUT, "ref_leaf_modules.npz"), **out)


def lvm_glue_vectors():
    """The REFERENCE'S OWN LVM (LVM/model.py:157-566) over its own Phi3Transformer.forward (OmniGen/transformer.py:71-232)
    and attention seam (LVM/transform/sdpa_transform.py:12-169), executed by oracle/extract_reference.py on the tiny
    next-clip case of tests/smoke_case.py; sampled with the reference's own LVMScheduler (LVM/scheduler.py:119-208)."""
    from tests import smoke_case as SC
    from tests import glue_cases as GC
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg)
    model, ns = X.build_reference_model(cfg, p, "LVM")
    okw = {k: batch[k] for k in GC.BATCH_KEYS}
    out = {}
    t = torch.full((len(z),), 0.3)
    Sched = X.scheduler_class()
    with torch.no_grad():
        for pt in ("x1", "v"):
            o, cache = model.frame_block_forward_with_cfg([x.clone() for x in z], t, input_img_latents=cond, use_img_cfg=True,
                                                          img_cfg_scale=1.6, past_key_values=None, use_kv_cache=False,
                                                          offload_model=False, vae=None, prediction_type=pt, **okw)
            out[f"fwd_{pt}"] = torch.cat(o).numpy()
            kw = dict(okw, input_img_latents=cond, use_img_cfg=True, img_cfg_scale=1.6, use_kv_cache=False,
                      offload_model=False, vae=None)
            s = Sched(num_steps=3, time_shifting_factor=1)
            zs = s([x.clone() for x in z], model.frame_block_forward_with_cfg, kw, use_kv_cache=False, prediction_type=pt)
            out[f"sample3_{pt}"] = torch.cat(zs).numpy()
        # CFG off (single row), 'v'
        p1, batch1, z1, cond1 = SC.build_case(cfg, use_cfg=False)
        okw1 = {k: batch1[k] for k in GC.BATCH_KEYS}
        o, _ = model.frame_block_forward_with_cfg([x.clone() for x in z1], torch.full((len(z1),), 0.6), input_img_latents=cond1,
                                                  use_img_cfg=False, img_cfg_scale=1.6, past_key_values=None,
                                                  use_kv_cache=False, offload_model=False, vae=None, prediction_type="v", **okw1)
        out["fwd_nocfg_v"] = torch.cat(o).numpy()
        # Phi3Transformer.forward alone: 3-D bool mask -> additive mask, layer loop, final norm
        g = torch.Generator("cpu").manual_seed(5)
        B, L = batch["input_ids"].shape
        emb = (torch.randn(B, L, cfg.hidden_size, generator=g) * 0.5).to(torch.bfloat16).float()
        out["llm_hidden"] = model.llm(inputs_embeds=emb, attention_mask=batch["attention_mask"],
                                      position_ids=batch["position_ids"]).last_hidden_state.numpy()
        try:
            model.llm(inputs_embeds=emb, attention_mask=torch.ones(B, L), position_ids=batch["position_ids"])
            out["mask2d_error"] = np.asarray("")
        except Exception as e:  # the reference's own exception text
            out["mask2d_error"] = np.asarray(str(e))
        # LVM.forward / forward_with_cfg (single target)
        c = GC.single_target_case(cfg)
        o, _ = model.forward(c["x"], c["t"], c["ids"], c["lat"], c["sizes"], c["mask"], c["pos"])
        out["single_fwd"] = o.numpy()
        o, _ = model.forward_with_cfg(c["x"], c["t"], c["ids"], c["lat"], c["sizes"], c["mask"], c["pos"], True, 1.6, None,
                                      False, False, prediction_type="v")
        out["single_cfg_v"] = o.numpy()
        Lc, N = c["Lc"], c["N"]
        o = model.forward(c["x"], c["t"], None, None, None, c["mask"][:, Lc:, Lc:].contiguous(),
                          c["pos"][:, : N + 1].contiguous(), return_past_key_values=False)
        out["single_nocond"] = o.numpy()
    np.savez_compressed(os.path.join(OUT, "ref_lvm_glue_tiny.npz"), **out)


def _run_reference_loss(model, ns, x1, clean, batch, seed, frame_blocks=None, input_output_return=False):
    """training_losses_x1_noise_input (LVM/train_helper/loss.py:128-243) on the reference LVMTraining; the noise / times
    it draws from torch's (and python's) global RNG are recovered by replaying the same draws after the same seed."""
    import random
    from tests import glue_cases as GC
    x1l, cl = list(x1.split(1)), list(clean.split(1))
    torch.manual_seed(seed); random.seed(seed)
    x0 = [torch.randn_like(a) for a in x1l]                                   # sample_x0(x1)            loss.py:155
    if frame_blocks is None:
        t = torch.rand(len(x1l))                                             # sample_timestep          :160
    else:
        t = torch.tensor([v for b in frame_blocks for fb in frame_blocks[b] for v in [random.random()] * fb])  # :162
    x0i = [torch.randn_like(a) for a in cl]                                   # sample_x0(inputs)        :164
    ti = 0.9 + (1 - 0.9) * torch.rand(len(cl))                                # sample_timestep_max_noise :166
    seen = {}
    inner = model.forward

    def spy(xt, tt, **kw):
        seen["xt"], seen["t"], seen["inp"] = [a.clone() for a in xt], tt.clone(), [a.clone() for a in kw["input_img_latents"]]
        res = inner(xt, tt, **kw)
        if kw.get("input_output_return", False):
            seen["pred"] = [a.detach().clone() for a in res[0]]
            seen["pred_in"] = [a.detach().clone() for a in res[1]]
        else:
            seen["pred"] = [a.detach().clone() for a in res]
        return res
    model.forward = spy
    kw = {k: batch[k] for k in GC.BATCH_KEYS}
    kw.update(input_img_latents=[a.clone() for a in cl], return_past_key_values=False)
    if input_output_return:
        kw["input_output_return"] = True
    torch.manual_seed(seed); random.seed(seed)
    model.zero_grad()
    terms = ns.training_losses_x1_noise_input(model, [a.clone() for a in x1l], kw, frame_blocks=frame_blocks, device="cpu")
    model.forward = inner
    terms["loss"].mean().backward()                                           # train_x1_stage1_noiseinput.py:378-380
    assert torch.equal(seen["t"], t)
    for i in range(len(x1l)):
        assert torch.equal(seen["xt"][i], t[i] * x1l[i] + (1 - t[i]) * x0[i])
    for i in range(len(cl)):
        assert torch.equal(seen["inp"][i], ti[i] * cl[i] + (1 - ti[i]) * x0i[i])
    out = dict(x0=torch.cat(x0).numpy(), t=t.numpy(), x0_in=torch.cat(x0i).numpy(), t_in=ti.numpy(),
               xt=torch.cat(seen["xt"]).numpy(), pred=torch.cat(seen["pred"]).numpy(), loss=terms["loss"].detach().numpy())
    if input_output_return:
        out["pred_in"] = torch.cat(seen["pred_in"]).numpy()
    for name, prm in model.named_parameters():
        if prm.grad is not None:
            out["gnorm." + name] = np.asarray(float(prm.grad.double().norm()))
            out["grad." + name] = GC.sampled_grad(prm.grad)
        else:
            out["gnorm." + name] = np.asarray(-1.0)
    return out


def loss_vectors():
    """The reference's LVMTraining.forward (LVM/model.py:752-845) + stage-1 loss and every parameter gradient
    (torch autograd through reference code), stage-1 layout and stage-2+ frame-block layout."""
    from tests import glue_cases as GC
    cfg = R.TINY
    p, batch, x1, x0, t, clean, x0i, ti = GC.stage1_case(cfg)
    model, ns = X.build_reference_model(cfg, p, "LVMTraining")
    model.train()
    np.savez_compressed(os.path.join(OUT, "ref_loss_stage1_tiny.npz"), **_run_reference_loss(model, ns, x1, clean, batch, 123))
    fb = GC.frame_block_training_batch()
    nd = sum(len(v) for v in fb["denoise_image_sizes"].values())
    nc = sum(len(v) for v in fb["input_image_sizes"].values())
    gen = torch.Generator("cpu").manual_seed(21)
    x1b, cleanb = torch.randn(nd, 4, 8, 8, generator=gen), torch.randn(nc, 4, 8, 8, generator=gen)
    out = _run_reference_loss(model, ns, x1b, cleanb, fb, 321, frame_blocks=fb["frame_blocks"])
    np.savez_compressed(os.path.join(OUT, "ref_loss_fbtrain_tiny.npz"), **out)


def input_output_return_vectors():
    """`input_output_return=True` (LVM/model.py:488-497, 832-841; loss branch LVM/train_helper/loss.py:194-197,220-225): the
    reference's LVM.frame_block_forward on the tiny next-clip case and its LVMTraining + training_losses_x1_noise_input on the
    stage-1 case, both with a seeded (non-zero) `input_final_layer` head -- outputs, the appended input-loss terms and every
    parameter gradient of loss.mean()."""
    from tests import smoke_case as SC
    from tests import glue_cases as GC
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg)
    p = R.add_input_final_layer(p, cfg)
    model, ns = X.build_reference_model(cfg, p, "LVM")
    okw = {k: batch[k] for k in GC.BATCH_KEYS}
    out = {}
    with torch.no_grad():
        lat, pin = model.frame_block_forward([x.clone() for x in z], torch.full((len(z),), 0.3), input_img_latents=cond,
                                             input_output_return=True, **okw)
    out["fwd"] = torch.cat(lat).numpy(); out["fwd_in"] = torch.cat(pin).numpy()
    p2, batch2, x1, x0, t, clean, x0i, ti = GC.stage1_case(cfg)
    p2 = R.add_input_final_layer(p2, cfg)
    model2, ns2 = X.build_reference_model(cfg, p2, "LVMTraining")
    model2.train()
    for k, v in _run_reference_loss(model2, ns2, x1, clean, batch2, 456, input_output_return=True).items():
        out["loss." + k] = v
    np.savez_compressed(os.path.join(OUT, "ref_input_output_return_tiny.npz"), **out)


def pipeline_vectors():
    """The reference's LVMPipeline.prompt_condition_frame_block_autoregressive_inference (LVM/pipeline.py:347-595) executed
    on CPU fp32 (oracle/extract_reference.py::reference_pipeline): two chained rounds (gen_nums [2, 1], window 4, CFG 1.6,
    x1 prediction, 2 Euler steps, condition re-noising 0.1) on the tiny denoiser + tiny /8 VAE stand-in.  use_kv_cache is
    off: the reference sampler passes past_key_values=None at every step (LVM/scheduler.py:174), so its cache only ever
    holds one call's keys and changes no result, and the installed transformers' DynamicCache lacks the 4.47.1 methods
    new_forward calls on it.  Recorded: input frames, every global-generator noise draw in order (VAE posterior samples,
    re-noising), per round what the scheduler received and returned, the returned images."""
    from PIL import Image
    from oracle import vae_ref as VR
    cfg, vcfg = R.TINY, VR.TINY_VAE8
    p = {k: v.to(torch.bfloat16).float() for k, v in R.make_params(cfg, 0).items()}
    vp = VR.make_vae_params(vcfg, seed=2)
    pipe, log = X.reference_pipeline(cfg, p, vp, vcfg)
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    renoise = []
    orig = torch.randn_like

    def recording_randn_like(t, *a, **k):
        n = orig(t, *a, **k)
        renoise.append(n.clone())
        return n
    torch.randn_like = recording_randn_like
    torch.manual_seed(1234)
    try:
        out = pipe.prompt_condition_frame_block_autoregressive_inference(
            input_images=[Image.fromarray(f) for f in frames], height=64, width=64, gen_nums=[2, 1], num_inference_steps=2,
            use_img_guidance=True, img_guidance_scale=1.6, dtype=torch.float32, seed=42, output_type="pil",
            prediction_type="x1", clean_image_noise_level=0.1, max_frame_window=4, use_kv_cache=False)
    finally:
        torch.randn_like = orig
    d = {"frames_in": frames, "vae_noise": torch.stack(pipe.vae.noise_log).numpy(), "renoise": torch.stack(renoise).numpy(),
         "images_out": np.stack([np.array(im) for im in out]), "n_rounds": np.array(len(log["rounds"]))}
    for k, r in enumerate(log["rounds"]):
        d[f"r{k}_latents"] = torch.cat(r["latents"]).numpy()
        d[f"r{k}_input_img_latents"] = torch.cat(r["input_img_latents"]).numpy()
        d[f"r{k}_samples"] = torch.cat(r["samples"]).numpy()
    np.savez_compressed(os.path.join(OUT, "ref_pipeline_tiny.npz"), **d)

    # LVMPipeline.__call__ (LVM/pipeline.py:138-343): single-target rounds, every generated image joining the conditions
    pipe, log = X.reference_pipeline(cfg, p, vp, vcfg)
    frames = np.random.default_rng(6).integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
    renoise.clear()
    torch.randn_like = recording_randn_like
    torch.manual_seed(99)
    try:
        out = pipe(input_images=[Image.fromarray(f) for f in frames], height=64, width=64, gen_num=2, num_inference_steps=2,
                   use_img_guidance=True, img_guidance_scale=1.6, dtype=torch.float32, seed=42, output_type="pil",
                   prediction_type="x1", clean_image_noise_level=0.1, use_kv_cache=False)
    finally:
        torch.randn_like = orig
    d = {"frames_in": frames, "vae_noise": torch.stack(pipe.vae.noise_log).numpy(), "renoise": torch.stack(renoise).numpy(),
         "images_out": np.stack([np.array(im) for im in out]), "n_rounds": np.array(len(log["rounds"]))}
    for k, r in enumerate(log["rounds"]):
        d[f"r{k}_latents"] = torch.stack(r["latents"]).numpy()
        d[f"r{k}_input_img_latents"] = torch.cat(r["input_img_latents"]).numpy()
        d[f"r{k}_samples"] = torch.stack(r["samples"]).numpy()
    np.savez_compressed(os.path.join(OUT, "ref_pipeline_call_tiny.npz"), **d)


def oracle_e2e_vectors():
    """Tiny next-clip case (tests/smoke_case.py) frozen from the restatement."""
    from tests import smoke_case as SC
    cfg = R.TINY
    p, batch, z, cond = SC.build_case(cfg)
    out = {}
    t = torch.full((len(z),), 0.3)
    okw = {k: batch[k] for k in ("input_ids", "input_image_sizes", "attention_mask", "position_ids",
                                 "denoise_image_sizes", "time_emb_inx")}
    for pt in ("x1", "v"):
        out[f"fwd_{pt}"] = torch.cat(R.frame_block_forward_with_cfg(p, cfg, z, t, True, 1.6, pt, input_img_latents=cond, **okw)).numpy()
        out[f"sample3_{pt}"] = torch.cat(SC.oracle_sample(cfg, p, batch, z, cond, 3, pt)).numpy()
    g = torch.Generator("cpu").manual_seed(5)
    B, L = batch["input_ids"].shape
    emb = (torch.randn(B, L, cfg.hidden_size, generator=g) * 0.5).to(torch.bfloat16).float()
    out["llm_hidden"] = R.transformer(p, cfg, emb, batch["attention_mask"], batch["position_ids"]).numpy()
    np.savez_compressed(os.path.join(OUT, "oracle_tiny_e2e.npz"), **out)


if __name__ == "__main__":
    if not X.available():
        raise SystemExit("reference checkout not found: golden vectors can only be regenerated in the build container")
    os.makedirs(OUT, exist_ok=True)
    collator_vectors()
    scheduler_vectors()
    leaf_vectors()
    lvm_glue_vectors()
    loss_vectors()
    input_output_return_vectors()
    pipeline_vectors()
    oracle_e2e_vectors()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
