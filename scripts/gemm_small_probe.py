"""Developer probe (GPU): the small-M GEMM launches (row remainders of the 256-tile launches) and the full cfg-2 shapes."""
import importlib, sys, os
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
for M in (32, 60, 544, 4128, 7740):
    line = f"M={M:5d}:"
    for (N, K) in ((9216, 3072), (3072, 3072), (3072, 8192)):
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        y = torch.empty(M, N, dtype=BF, device=dev)
        t = timeit(lambda: ops.linear(x, w, out=y))
        line += f"  N{N} K{K} {t:7.1f} us {2.0*M*N*K/t/1e6:5.0f} TF"
    x = torch.randn(M, 3072, device=dev).to(BF); w = (torch.randn(16384, 3072, device=dev) * 0.05).to(BF)
    y = torch.empty(M, 8192, dtype=BF, device=dev)
    t = timeit(lambda: ops.gated_mlp_act(x, w, out=y))
    line += f"  gate_up {t:7.1f} us {2.0*M*16384*3072/t/1e6:5.0f} TF"
    print(line)
