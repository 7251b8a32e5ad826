"""Developer probe (GPU): one next-clip round at 512^2 (N = 1024 tokens per frame, C = 4 + G = 8 frames, CFG: 20 520 packed
rows) through LVMPipeline with the full-size denoiser -- a shape the bench does not use."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
importlib.import_module("video-gpt_amd")
M = importlib.import_module("video-gpt_amd.model"); P = importlib.import_module("video-gpt_amd.processor")
PL = importlib.import_module("video-gpt_amd.pipeline")
dev = torch.device("cuda", 0)
model = bench.build_model(M, bench.full_config(M, 32), dev, seed=0)
pipe = PL.LVMPipeline(bench.synthetic_vae(dev), model, P.LVMProcessor(P.SpecialTokenizer(10, 11, 12)), device=dev)
frames = [torch.rand(3, 512, 512) * 2 - 1 for _ in range(4)]
kw = dict(input_images=frames, height=512, width=512, use_img_guidance=True, img_guidance_scale=1.6, seed=42, output_type="pt",
          prediction_type="x1", clean_image_noise_level=0.05, max_frame_window=16)
pipe.prompt_condition_frame_block_autoregressive_inference(gen_nums=[8], num_inference_steps=2, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
out = pipe.prompt_condition_frame_block_autoregressive_inference(gen_nums=[8], num_inference_steps=10, **kw)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"512^2 round, 10 steps: {dt * 1e3:.0f} ms total, frames {len(out)} of {tuple(out[0].shape)}, finite latents",
      bool(torch.isfinite(torch.cat(pipe.last_samples[0]).float()).all()), f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GB")
