"""LVMPipeline.prompt_condition_frame_block_autoregressive_inference on the HIP path vs the oracle's
restatement of one round (LVM/pipeline.py:404-590): VAE-encode the condition frames, sample the next clip
with CFG, VAE-decode, uint8.  Tiny denoiser (2 layers, H=192) + tiny /8 VAE, noise from CPU generators.

Tolerance: condition latents fp32 VAE -> bf16 cast (<= 5e-3 rel-L2); sampled latents <= 3e-2 (bf16 denoiser vs
fp32 oracle); decoded frames mean |diff| <= 2 grey levels."""
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
    assert SC.rel_l2(got, torch.cat(ref)) < 3e-2
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
    assert SC.rel_l2(pipe.last_samples[0], ref) < 3e-2
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
