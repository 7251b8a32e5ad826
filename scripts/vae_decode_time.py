"""Developer probe (GPU): VAE decode of 8 frames at 256^2, ms per frame (VGPT_LIB selects a diagnostic library)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
importlib.import_module("video-gpt_amd")
vae = bench.synthetic_vae("cuda:0")
vae.conv_precision = "bf16x3"
z = torch.randn(8, 4, 32, 32, generator=torch.Generator("cpu").manual_seed(0)).to("cuda:0")
for _ in range(3):
    vae.decode_to_uint8(z)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(6):
    vae.decode_to_uint8(z)
torch.cuda.synchronize()
print(os.environ.get("VGPT_LIB", "product")[-12:], f"decode {1e3 * (time.perf_counter() - t0) / 48:.3f} ms/frame")
