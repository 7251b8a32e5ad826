"""Developer probe (GPU, under rocprofv3 --kernel-trace): VAE decode only, 8 frames x 5 iterations at 256^2."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
importlib.import_module("video-gpt_amd")
vae = bench.synthetic_vae("cuda:0")
vae.conv_precision = "bf16x3"
z = torch.randn(8, 4, 32, 32, generator=torch.Generator("cpu").manual_seed(0)).to("cuda:0")
for _ in range(7):
    vae.decode_to_uint8(z)
torch.cuda.synchronize()
