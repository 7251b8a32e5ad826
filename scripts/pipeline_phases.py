"""Developer probe (GPU): where one LVMPipeline round spends its wall time (host + device), by wrapping its parts."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
importlib.import_module("video-gpt_amd")
M = importlib.import_module("video-gpt_amd.model"); P = importlib.import_module("video-gpt_amd.processor")
PL = importlib.import_module("video-gpt_amd.pipeline"); E = importlib.import_module("video-gpt_amd.engine")
S = importlib.import_module("video-gpt_amd.scheduler"); V = importlib.import_module("video-gpt_amd.vae")
ops = importlib.import_module("video-gpt_amd.ops")
dev = torch.device("cuda", 0)
model = bench.build_model(M, bench.full_config(M, 32), dev, seed=0)
pipe = PL.LVMPipeline(bench.synthetic_vae(dev), model, P.LVMProcessor(P.SpecialTokenizer(10, 11, 12)), device=dev)
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)
wrap(pipe, "vae_encode", "vae_encode(sample)")
wrap(pipe, "vae_posteriors", "vae_posteriors(encoder)")
wrap(E.StaticDenoiser, "rebind", "engine_rebind(incl clip pass)")
wrap(E.StaticDenoiser, "_clip_pass", "clip_pass")
wrap(E.StaticDenoiser, "set_latents", "set_latents")
wrap(pipe.vae, "decode_to_uint8", "vae_decode")
wrap(pipe.processor, "prompt_condition_frame_block_inference", "processor")
wrap(E.StaticDenoiser, "__init__", "engine_init(incl prefill)")
wrap(E.StaticDenoiser, "prefill", "prefill")
wrap(E.StaticDenoiser, "capture", "capture")
wrap(E.StaticDenoiser, "run", "run(incl capture)")
wrap(ops, "pack_mask", "pack_mask")
frames = [torch.rand(3, 256, 256) * 2 - 1 for _ in range(4)]
kw = dict(input_images=frames, height=256, width=256, num_inference_steps=50, use_img_guidance=True, img_guidance_scale=1.6,
          seed=42, output_type="pt", prediction_type="x1", clean_image_noise_level=0.05, max_frame_window=16)
pipe.prompt_condition_frame_block_autoregressive_inference(gen_nums=[8], **dict(kw, num_inference_steps=2))
for rounds in (1, 4):
    acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipe.prompt_condition_frame_block_autoregressive_inference(gen_nums=[8] * rounds, **kw)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    print(f"rounds={rounds}: total {tot*1e3:.0f} ms;", {k: round(v * 1e3, 1) for k, v in acc.items()})
