"""Developer probe (GPU): the step's four GEMMs at M = 4096 with a rotating set of weight matrices (so that weights come
from HBM as in the sampler step, not from the Infinity Cache of a same-weights loop).  Run once with the product library
and once with VGPT_LIB=video-gpt_amd/libvgpt_hip_g4.so (make gemm-debug-4: no epilogue) to see what the epilogue costs."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
dev = "cuda:0"; BF = torch.bfloat16
M, NW = 4096, 6
def timeit(f, n=30):
    for i in range(6): f(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = []
for name, N, K, kind in (("o_proj", 3072, 3072, "resid"), ("down", 3072, 8192, "resid"), ("qkv", 9216, 3072, "plain"), ("gate_up", 16384, 3072, "gated")):
    x = torch.randn(M, K, device=dev).to(BF)
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(BF) for _ in range(NW)]
    if kind == "gated":
        y = torch.empty(M, N // 2, dtype=BF, device=dev)
        t = timeit(lambda i: ops.gated_mlp_act(x, ws[i % NW], out=y))
    elif kind == "resid":
        h = torch.randn(M, N, device=dev).to(BF); y = torch.empty_like(h)
        t = timeit(lambda i: ops.linear(x, ws[i % NW], residual=h, out=y))
    else:
        y = torch.empty(M, N, dtype=BF, device=dev)
        t = timeit(lambda i: ops.linear(x, ws[i % NW], out=y))
    out.append(f"{name} {t:6.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF/s)")
print(os.environ.get("VGPT_LIB", "product")[-20:], " | ".join(out))
