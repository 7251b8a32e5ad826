"""Developer probe (GPU): the same GEMM launches on random, constant and zero operands (rotating weight sets as in
gemm_epilogue_probe.py).  The part lowers its clock with the switching activity of the data, so the spread between the
rows is clock, not code (MI355X_MICROARCH.md, 'DVFS give-back')."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
dev = "cuda:0"; BF = torch.bfloat16
M, NW = 4096, 6
def timeit(f, n=40):
    for i in range(10): f(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): f(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for fill in ("random", "ones", "zeros"):
    out = []
    mk = {"random": lambda *s: torch.randn(*s, device=dev), "ones": lambda *s: torch.ones(*s, device=dev),
          "zeros": lambda *s: torch.zeros(*s, device=dev)}[fill]
    for name, N, K, kind in (("o_proj", 3072, 3072, "resid"), ("down", 3072, 8192, "resid"), ("qkv", 9216, 3072, "plain"),
                             ("gate_up", 16384, 3072, "gated"), ("8192^3", 8192, 8192, "plain")):
        m = 8192 if name == "8192^3" else M
        x = mk(m, K).to(BF)
        ws = [(mk(N, K) * 0.05).to(BF) for _ in range(NW if name != "8192^3" else 2)]
        if kind == "gated":
            y = torch.empty(m, N // 2, dtype=BF, device=dev)
            t = timeit(lambda i: ops.gated_mlp_act(x, ws[i % len(ws)], out=y))
        elif kind == "resid":
            h = mk(m, N).to(BF); y = torch.empty_like(h)
            t = timeit(lambda i: ops.linear(x, ws[i % len(ws)], residual=h, out=y))
        else:
            y = torch.empty(m, N, dtype=BF, device=dev)
            t = timeit(lambda i: ops.linear(x, ws[i % len(ws)], out=y))
        out.append(f"{name} {t:6.1f} us ({2.0 * m * N * K / t / 1e6:5.0f} TF/s)")
    print(f"{fill:7s}", " | ".join(out))
