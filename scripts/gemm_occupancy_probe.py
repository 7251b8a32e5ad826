"""Developer probe (GPU): time per k-tile of the 256x256 GEMM kernel when only part of the chip is busy (grid of 32 /
64 / 128 / 256 tiles, K = 8192) — separates per-CU limits from shared (L2 / fabric) ones.  Run with VGPT_GEMM_TILE=256."""
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
K = 8192
for (M, N) in ((256, 256), (1024, 2048), (2048, 2048), (2048, 4096), (4096, 4096), (8192, 8192)):
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev)
    t = timeit(lambda: ops.linear(x, w, out=y))
    tiles = (M // 256) * (N // 256); rounds = -(-tiles // 256)
    print(f"{tiles:5d} tiles ({rounds} rounds): {t:7.1f} us  -> {t / rounds / (K // 64) * 1e3:6.0f} ns per k-tile, {2.0*M*N*K/t/1e6/min(tiles,256)*256:5.0f} TF/s chip-equivalent")
