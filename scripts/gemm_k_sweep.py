"""Developer probe (GPU): GEMM time against K at fixed (M, N) -> per-k-tile time (slope) and the launch's fixed part
(prologue + epilogue, intercept), for the step's shapes at M = 4096; torch.matmul (vendor library) beside it."""
import importlib, sys, os
import numpy as np, torch
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
M = 4096
for N, gated in ((3072, False), (9216, False), (16384, True), (16384, False)):
    ks, ts, tv = [1024, 2048, 3072, 4096, 6144, 8192], [], []
    for K in ks:
        x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        if gated:
            y = torch.empty(M, N // 2, dtype=BF, device=dev)
            ts.append(timeit(lambda: ops.gated_mlp_act(x, w, out=y)))
        else:
            y = torch.empty(M, N, dtype=BF, device=dev)
            ts.append(timeit(lambda: ops.linear(x, w, out=y)))
        wt = w.t()
        tv.append(timeit(lambda: torch.matmul(x, wt)))
    slope, icpt = np.polyfit(np.array(ks) / 64, np.array(ts), 1)
    sv, iv = np.polyfit(np.array(ks) / 64, np.array(tv), 1)
    print(f"N={N}{' gated' if gated else ''}: " + "  ".join(f"K={k}: {t:.0f}/{v:.0f}us" for k, t, v in zip(ks, ts, tv)))
    print(f"    ours: {slope:.3f} us per k-tile + {icpt:.1f} us fixed ({2.0 * M * N * 64 / slope / 1e6:.0f} TF/s in the loop);  vendor: {sv:.3f} + {iv:.1f} ({2.0 * M * N * 64 / sv / 1e6:.0f})")
