"""LVMPipeline.prompt_condition_frame_block_autoregressive_inference on the HIP path vs the oracle's
restatement of one round (LVM/pipeline.py:404-590): VAE-encode the condition frames, sample the next clip
with CFG, VAE-decode, uint8.  Tiny denoiser (2 layers, H=192) + tiny /8 VAE, noise from CPU generators.

Tolerance: condition latents fp32 VAE -> bf16 cast (<= 5e-3 rel-L2); sampled latents <= 2 x the error of stock bf16 ops
(2.9e-2, tests/golden/tolerance_calibration.json; bf16 denoiser vs fp32 oracle); decoded frames mean |diff| <= 2 grey levels."""
import importlib

import pytest
import torch

from oracle import restate as R
from oracle import vae_ref as VR
from tests import smoke_case as SC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def pipe_case():
    cfg, vcfg = R.TINY, VR.TINY_VAE8
    p = {k: v.to(BF).float() for k, v in R.make_params(cfg, 0).items()}
    vp = VR.make_vae_params(vcfg, seed=2)
    model = SC.build_product_model(cfg, p, DEV)
    V = importlib.import_module("video-gpt_amd.vae")
    vae = V.AutoencoderKL(block_out_channels=vcfg.block_out_channels, layers_per_block=vcfg.layers_per_block,
                          norm_num_groups=vcfg.norm_num_groups)
    vae.load_state_dict(vp)
    vae = vae.to(DEV, torch.float32).eval()
    P = importlib.import_module("video-gpt_amd.processor")
    PL = importlib.import_module("video-gpt_amd.pipeline")
    pipe = PL.LVMPipeline(vae, model, P.LVMProcessor(P.SpecialTokenizer(10, 11, 12)), device=DEV)
    frames = [torch.rand(3, 64, 64, generator=torch.Generator("cpu").manual_seed(50 + i)) * 2 - 1 for i in range(2)]
    return cfg, vcfg, p, vp, pipe, frames


def test_one_round_matches_oracle(pipe_case):
    cfg, vcfg, p, vp, pipe, frames = pipe_case
    C, G, steps, seed = 2, 2, 2, 42
    vnoise = [torch.randn(1, 4, 8, 8, generator=torch.Generator("cpu").manual_seed(70 + i)) for i in range(C)]
    out = pipe.prompt_condition_frame_block_autoregressive_inference(
        input_images=frames, height=64, width=64, gen_nums=[G], num_inference_steps=steps, use_img_guidance=True,
        img_guidance_scale=1.6, seed=seed, output_type="pt", prediction_type="x1", generator_device="cpu",
        vae_noise=vnoise)
    assert len(out) == C + G and out[0].shape == (64, 64, 3) and out[0].dtype == torch.uint8
    # ---- oracle round ----
    cond = [VR.vae_encode(vp, vcfg, frames[i][None], vnoise[i]).to(BF).float() for i in range(C)]
    for i in range(C):
        assert SC.rel_l2(pipe.last_latents[i], cond[i]) < 5e-3
    g = torch.Generator("cpu").manual_seed(seed)
    noise = [torch.randn(1, 4, 8, 8, generator=g).to(BF).float() for _ in range(G)]
    batch = R.collate_inference(C, G, 16, use_cfg=True, pad_id=2)
    ref = SC.oracle_sample(cfg, p, batch, noise * 2, cond, steps, "x1")[:G]
    got = torch.cat(pipe.last_samples[0])
    assert SC.rel_l2(got, torch.cat(ref)) < SC.tol("sampler_latents")
    ref_imgs = [VR.decode_to_uint8(vp, vcfg, x)[0] for x in cond + ref]
    for a, b in zip(out, ref_imgs):
        assert float((a.cpu().int() - b.int()).abs().float().mean()) <= 2.0


def test_two_rounds_chain_and_window(pipe_case):
    cfg, vcfg, p, vp, pipe, frames = pipe_case
    out = pipe.prompt_condition_frame_block_autoregressive_inference(
        input_images=frames, height=64, width=64, gen_nums=[2, 1], num_inference_steps=1, seed=1, output_type="pil",
        prediction_type="x1", clean_image_noise_level=0.1, max_frame_window=4)
    # round 0: 2 decoded condition frames + 2 generated; round 1: window keeps the last 3, generates 1
    assert len(out) == 5 and out[0].size == (64, 64)


def test_cfg_off_mirrors_reference_halving(pipe_case):
    cfg, vcfg, p, vp, pipe, frames = pipe_case
    kw = dict(input_images=frames, height=64, width=64, gen_nums=[2], num_inference_steps=1, seed=1, output_type="pt",
              prediction_type="x1", use_img_guidance=False)
    assert len(pipe.prompt_condition_frame_block_autoregressive_inference(**kw)) == 2 + 1   # reference's [:len//2]
    assert len(pipe.prompt_condition_frame_block_autoregressive_inference(halve_without_cfg=False, **kw)) == 2 + 2


def test_layout_and_dense_masks_sample_the_same_clip(pipe_case):
    """The collator's two mask forms (dense bool tensor as in the reference; per-token layout expanded on the device)
    describe the same mask bit for bit, so the sampled latents are identical."""
    cfg, vcfg, p, vp, pipe, frames = pipe_case
    vnoise = [torch.randn(1, 4, 8, 8, generator=torch.Generator("cpu").manual_seed(70 + i)) for i in range(2)]
    got = {}
    for fmt in ("layout", "bool"):
        pipe.mask_format = fmt
        pipe.prompt_condition_frame_block_autoregressive_inference(
            input_images=frames, height=64, width=64, gen_nums=[2], num_inference_steps=2, use_img_guidance=True,
            img_guidance_scale=1.6, seed=7, output_type="pt", prediction_type="x1", generator_device="cpu",
            vae_noise=vnoise)
        got[fmt] = torch.cat(pipe.last_samples[0]).clone()
    pipe.mask_format = "layout"
    assert torch.equal(got["layout"], got["bool"])


@pytest.mark.parametrize("pt", ["x1", "v"])
def test_single_target_call_matches_oracle(pipe_case, pt):
    """LVMPipeline.__call__ (LVM/pipeline.py:138-343): [<img> image </img> x2, <|diffusion|> | time | target] through
    LVM.forward_with_cfg + the Euler sampler, CFG row = the empty prompt; one round vs the oracle's LVM.forward."""
    cfg, vcfg, p, vp, pipe, frames = pipe_case
    steps, seed = 2, 5
    vnoise = [torch.randn(1, 4, 8, 8, generator=torch.Generator("cpu").manual_seed(90 + i)) for i in range(2)]
    out = pipe(input_images=frames, height=64, width=64, gen_num=1, num_inference_steps=steps, use_img_guidance=True,
               img_guidance_scale=1.6, seed=seed, output_type="pt", prediction_type=pt, generator_device="cpu",
               vae_noise=vnoise)
    assert len(out) == 3 and out[-1].shape == (64, 64, 3) and out[-1].dtype == torch.uint8
    cond = [VR.vae_encode(vp, vcfg, frames[i][None], vnoise[i]).to(BF).float() for i in range(2)]
    P = importlib.import_module("video-gpt_amd.processor")
    proc = P.LVMProcessor(P.SpecialTokenizer(10, 11, 12))
    data = proc(["<img><|image_1|></img><img><|image_2|></img>"], [frames], height=64, width=64, use_img_cfg=True)
    z0 = torch.randn(1, 4, 8, 8, generator=torch.Generator("cpu").manual_seed(seed)).to(BF).float()

    def func(zl, t):
        o = R.lvm_forward(p, cfg, torch.cat(zl), t, data["input_ids"], cond, data["input_image_sizes"],
                          data["attention_mask"].bool(), data["position_ids"])
        if pt == "v":   # LVM.forward_with_cfg applies CFG itself for 'v' (LVM/model.py:508-512)
            c = o[1:2] + 1.6 * (o[0:1] - o[1:2])
            o = torch.cat([c, c])
        return [o[0:1], o[1:2]]
    ref = R.scheduler_call(R.scheduler_sigma(steps, 1.0), [z0, z0.clone()], func, True, 1.6, pt)[0]
    assert SC.rel_l2(pipe.last_samples[0], ref) < SC.tol("sampler_latents")
    ref_img = VR.decode_to_uint8(vp, vcfg, ref)[0]
    assert float((out[-1].cpu().int() - ref_img.int()).abs().float().mean()) <= 2.0


def test_single_target_call_rounds_feed_back(pipe_case):
    """gen_num rounds: every generated image becomes a (re-noised) condition image of the next round; no input images:
    CFG is switched off for the first round (LVM/pipeline.py:210-218)."""
    cfg, vcfg, p, vp, pipe, frames = pipe_case
    out = pipe(input_images=None, height=64, width=64, gen_num=3, num_inference_steps=1, seed=3, output_type="pil",
               prediction_type="x1", clean_image_noise_level=0.1)
    assert len(out) == 3 and out[0].size == (64, 64)
    out = pipe(input_images=frames[:1], height=64, width=64, gen_num=2, num_inference_steps=1, seed=3, output_type="pt",
               prediction_type="v", clean_image_noise_level=0.1)
    assert len(out) == 1 + 2


def _poison_free_device_memory(nbytes=256 << 20):
    """Fill a block the caching allocator will hand out again with 0xFF bytes (NaN as bf16, fp32 and e4m3): a buffer that
    is read before it is written then shows up in the result."""
    t = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
    t.fill_(0xFF)
    torch.cuda.synchronize()
    del t


def test_cfg5_chained_rounds_with_fp8_attention(pipe_case, monkeypatch):
    """cfg-5 of BASELINE.json in its defining combination: chained next-clip rounds (LVM/pipeline.py:418-422 window,
    491-500 re-encode + re-noise of the fed-back frames, 549 sampler call) WITH the MX-fp8 attention option in the sampler
    steps.  gen_nums [2, 2, 2], max_frame_window 6 from 2 input frames: the condition window GROWS 2 -> 4 and then SLIDES
    (round 2 drops the two oldest frames); every noise draw is replayed from fixed CPU tensors.
      * every round's sampled latents against the oracle composition run on that round's own recorded inputs:
        bf16 AND fp8 engine within the calibrated latent tolerance (the fp8 kernel's own bound against bf16
        attention on the same operands is in tests/test_attn_fp8_gpu.py);
      * fp8 rollout against the bf16 rollout on the returned uint8 frames;
      * round 2 presents the same sequence as round 1 (full window), so the scheduler RE-BINDS round 1's engine (buffers,
        per-layer fp8 workspaces, attention plan, captured graph kept; per-clip pass redone): the last round re-run ALONE on
        its recorded inputs by a fresh engine, after poisoning the allocator's free blocks, equals the chained run bit for
        bit -- nothing of an earlier round (quantised prefix, live rows, sampler state) survives in the re-used engine."""
    cfg, vcfg, p, vp, pipe, frames = pipe_case
    PL = importlib.import_module("video-gpt_amd.pipeline")
    S = importlib.import_module("video-gpt_amd.scheduler")
    steps, G, N, px = 2, 2, 64, 128
    gen_nums, window = [2, 2, 2], 6
    gv = torch.Generator("cpu").manual_seed(300)
    frames = [torch.rand(3, px, px, generator=gv) * 2 - 1 for _ in range(2)]
    vnoise = [torch.randn(1, 4, px // 8, px // 8, generator=gv) for _ in range(2 + 4 + 4)]
    rnoise = [torch.randn(1, 4, px // 8, px // 8, generator=gv) for _ in range(4 + 4)]
    rec = {}

    class Recording(S.LVMScheduler):
        def __call__(self, z, func, model_kwargs, **kw):
            out = super().__call__(z, func, model_kwargs, **kw)
            rec[self.attention_precision].append((
                [t.clone() for t in z], dict(model_kwargs, input_img_latents=[t.clone() for t in model_kwargs["input_img_latents"]]),
                [t.clone() for t in out], self.last_engine))
            return out
    monkeypatch.setattr(PL, "LVMScheduler", Recording)
    outs = {}
    try:
        for prec in ("bf16", "fp8"):
            rec[prec] = []
            pipe.attention_precision = prec
            outs[prec] = pipe.prompt_condition_frame_block_autoregressive_inference(
                input_images=frames, height=px, width=px, gen_nums=gen_nums, num_inference_steps=steps, use_img_guidance=True,
                img_guidance_scale=1.6, seed=11, output_type="pt", prediction_type="x1", clean_image_noise_level=0.1,
                max_frame_window=window, generator_device="cpu", vae_noise=vnoise, renoise_noise=rnoise)
    finally:
        pipe.attention_precision = "bf16"
    for prec in ("bf16", "fp8"):
        assert len(outs[prec]) == 2 + sum(gen_nums)
        assert [len(r[1]["input_img_latents"]) for r in rec[prec]] == [2, 4, 4]          # grows, then slides
        assert rec[prec][2][3] is rec[prec][1][3] and rec[prec][1][3] is not rec[prec][0][3]   # the slid window re-binds the engine
        for k, (z, kw, out, eng) in enumerate(rec[prec]):
            assert eng is not None and eng.attn_fp8 == (prec == "fp8") and eng.hoist        # the fast path, hoisted layout
            C = len(kw["input_img_latents"])
            batch = R.collate_inference(C, G, N, use_cfg=True, pad_id=cfg.pad_token_id)
            cond = [t.float().cpu() for t in kw["input_img_latents"]]
            ref = SC.oracle_sample(cfg, p, batch, [t.float().cpu() for t in z], cond, steps, "x1")
            err = SC.rel_l2(torch.cat(out), torch.cat(ref))
            print(f"cfg-5 rollout, {prec} attention, round {k} (C = {C}): sampled latents rel-L2 vs oracle = {err:.3e}")
            assert err < SC.tol("sampler_latents")      # fp8 attention is held to the bf16 path's calibrated tolerance (measured 9.0e-3 vs 8.9e-3)
    diffs = [float((a.int() - b.int()).abs().float().mean()) for a, b in zip(outs["fp8"], outs["bf16"])]
    print("cfg-5 rollout: fp8 vs bf16 attention, mean grey-level difference per returned frame", [round(x, 2) for x in diffs])
    assert max(diffs) <= 4.0
    # ---- the last round on its own: fresh scheduler + engine, recycled device memory full of NaN patterns ----
    z, kw, out, _ = rec["fp8"][-1]
    del rec, outs
    torch.cuda.empty_cache()
    _poison_free_device_memory()
    alone = S.LVMScheduler(num_steps=steps)
    alone.attention_precision = "fp8"
    alone.cache_engines = False            # round 2 of the chained run re-bound round 1's engine (same sequence): this one is new
    got = alone(z, pipe.model.frame_block_forward_with_cfg, kw, prediction_type="x1")
    assert alone.last_engine.attn_fp8
    assert torch.equal(torch.cat(got), torch.cat(out))
