"""qkv_proj GEMM with the fused RoPE epilogue vs GEMM + rope kernel at the cfg-2 / cfg-3 row counts (HIP events)."""
import importlib, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
dev, BF = "cuda:0", torch.bfloat16
nh = nk = 32; hd = 96; H = 3072; N = (nh + 2 * nk) * hd


def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M in (4096, 4128, 7740, 1040):
    x = torch.randn(M, H, device=dev).to(BF); w = (torch.randn(N, H, device=dev) * 0.02).to(BF)
    pos = torch.arange(M, device=dev)[None]
    cos, sin = ops.rope_table(pos, ops.rope_inv_freq(hd, 10000.0, dev))
    out = torch.empty(M, N, device=dev, dtype=BF)
    a = t(lambda: ops.linear(x, w, out=out))
    b = t(lambda: ops.rope_qk_inplace(out, cos, sin, nh, nk, hd))
    c = t(lambda: ops.linear_qkv_rope(x, w, cos, sin, nh, nk, hd, out=out))
    fl = 2 * M * N * H
    print(f"M={M}: gemm {a:.1f} us ({fl / a / 1e6:.0f} TF/s) + rope {b:.1f} us = {a + b:.1f}; fused {c:.1f} us ({fl / c / 1e6:.0f} TF/s)")
