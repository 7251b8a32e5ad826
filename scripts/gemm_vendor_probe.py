"""Developer probe (GPU): the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the cfg-2 / cfg-3 shapes,
beside vgpt_gemm_bf16 — a yardstick only; nothing in the product calls it."""
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
for M in (4128, 7740, 8192):
    for (N, K) in ((9216, 3072), (3072, 3072), (3072, 8192), (16384, 3072), (8192, 8192)):
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        y = torch.empty(M, N, dtype=BF, device=dev)
        t1 = timeit(lambda: ops.linear(x, w, out=y))
        t2 = timeit(lambda: torch.matmul(x, w.t(), out=y))
        fl = 2.0 * M * N * K
        print(f"M={M} N={N} K={K}: vgpt {t1:7.1f} us {fl/t1/1e6:5.0f} TF | vendor {t2:7.1f} us {fl/t2/1e6:5.0f} TF")
