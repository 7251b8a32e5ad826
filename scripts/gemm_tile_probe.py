"""Developer probe (GPU): vgpt_gemm_bf16 at the per-step shapes of cfg-2 (M = 4096 image rows) under VGPT_GEMM_TILE
(0 = launch plan, 256 / 192 = forced big tile)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
dev = "cuda:0"; BF = torch.bfloat16
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
line = f"tile={os.environ.get('VGPT_GEMM_TILE', 'plan'):>4}:"
for M in (4096, 7740):
    for (N, K) in ((9216, 3072), (3072, 3072), (3072, 8192)):
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        y = torch.empty(M, N, dtype=BF, device=dev)
        t = timeit(lambda: ops.linear(x, w, out=y))
        line += f"  {M}x{N}x{K} {t:6.1f} us {2.0*M*N*K/t/1e6:5.0f} TF"
print(line)
