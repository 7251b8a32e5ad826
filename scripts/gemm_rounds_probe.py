"""Developer probe (GPU): time of the gated GEMM (gate_up, N = 2 x 8192, K = 3072) against the number of 256-workgroup ROUNDS
(M = 1024 .. 8192 rows -> 1 .. 8 rounds), tile-per-workgroup launch and persistent walk (VGPT_GEMM_PERSIST=1, separate process):
slope = one round, intercept = per-launch cost.  Weights rotate over four buffers (no L2 reuse between launches)."""
import importlib, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
dev = "cuda:0"; BF = torch.bfloat16
K, I = 3072, 8192
ws = [(torch.randn(2 * I, K, device=dev) * 0.05).to(BF) for _ in range(4)]
res = {}
for M in (1024, 2048, 3072, 4096, 6144, 8192):
    x = torch.randn(M, K, device=dev).to(BF)
    y = torch.empty(M, I, dtype=BF, device=dev)
    def f(i): ops.gated_mlp_act(x, ws[i % 4], ops.ACT_SILU, out=y)
    for i in range(8): f(i)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(32): f(i)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 32 * 1e3)
    res[M] = round(best, 1)
print(json.dumps({"persist": os.environ.get("VGPT_GEMM_PERSIST", "0"), "us_by_rows": res}))
