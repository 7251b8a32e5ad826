"""Developer probe (GPU): vgpt_gemm_bf16 rates of whichever library VGPT_LIB selects — the diagnostic builds of
csrc/Makefile (gemm-debug-N: 1 = no LDS-DMA staging, 2 = no LDS fragment reads, 3 = both, results are garbage;
gemm-variant-MACRO) beside the product library, for same-box A/B runs."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
dev = "cuda:0"; BF = torch.bfloat16
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
line = f"{os.path.basename(os.environ.get('VGPT_LIB', 'libvgpt_hip.so')):34s}"
for (M, N, K) in ((8192, 8192, 8192), (4096, 16384, 3072), (4096, 3072, 8192)):
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev)
    t = timeit(lambda: ops.linear(x, w, out=y))
    line += f"  {M}x{N}x{K} {t:7.1f} us {2.0*M*N*K/t/1e6:5.0f} TF"
print(line)
