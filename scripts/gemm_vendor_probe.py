"""Developer probe (GPU): vgpt_gemm_bf16 / vgpt_gemm_bf16_tr on the hand-written kernels (vendor mode 0) and through hipBLASLt
(vendor mode 2 = every plain product, csrc/gemm_lt.hip) on the plain products of the cfg-2 sampler step, the per-clip pass and
the cfg-3 training step -- the measurements behind the table of gemm_lt.hip (`table_says_vendor`).  Also checks both against
fp64 on the first shape of each kind.  Writes one JSON line per shape."""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
L = importlib.import_module("video-gpt_amd._lib")
lib = L.load()
dev = "cuda:0"; BF = torch.bfloat16


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(kind, M, N, K, check=False):
    g = torch.Generator("cpu").manual_seed(M + N + K)
    if kind == "nt":      # C = A W^T (+ residual)
        a = torch.randn(M, K, generator=g).to(dev, BF); w = (torch.randn(N, K, generator=g) * 0.05).to(dev, BF)
        r = torch.randn(M, N, generator=g).to(dev, BF); y = torch.empty(M, N, dtype=BF, device=dev)
        f = lambda: ops.linear(a, w, residual=r, out=y)
        ref = (lambda: a.double() @ w.double().t() + r.double())
    elif kind == "nn":    # dX = dY W  (W stored (K', N'))
        a = torch.randn(M, K, generator=g).to(dev, BF); w = (torch.randn(K, N, generator=g) * 0.05).to(dev, BF)
        y = torch.empty(M, N, dtype=BF, device=dev)
        f = lambda: L.call("vgpt_gemm_bf16_tr", a.data_ptr(), w.data_ptr(), y.data_ptr(), None, M, N, K, K, N, N, 0, 0, 0, 1, ops._stream())
        ref = (lambda: a.double() @ w.double())
    else:                 # dW = dY^T X  (A = dY stored (K', M), W = X stored (K', N))
        a = torch.randn(K, M, generator=g).to(dev, BF); w = (torch.randn(K, N, generator=g) * 0.05).to(dev, BF)
        y = torch.empty(M, N, dtype=BF, device=dev)
        f = lambda: L.call("vgpt_gemm_bf16_tr", a.data_ptr(), w.data_ptr(), y.data_ptr(), None, M, N, K, M, N, N, 0, 0, 1, 1, ops._stream())
        ref = (lambda: a.double().t() @ w.double())
    out = {"kind": kind, "M": M, "N": N, "K": K}
    for mode, name in ((0, "hip"), (2, "vendor"), (0, "hip2"), (2, "vendor2")):
        lib.vgpt_gemm_vendor_set_mode(mode)
        c0 = lib.vgpt_gemm_vendor_calls()
        t = timeit(f)
        out[name + "_us"] = round(t, 1)
        if mode == 2:
            out["vendor_took_it"] = lib.vgpt_gemm_vendor_calls() > c0
        if check and name in ("hip", "vendor"):
            f(); torch.cuda.synchronize()
            rr = ref()
            out[name + "_rel_l2"] = float((y.double() - rr).norm() / rr.norm())
    out["tf_hip"] = round(2.0 * M * N * K / min(out["hip_us"], out["hip2_us"]) / 1e6)
    out["tf_vendor"] = round(2.0 * M * N * K / min(out["vendor_us"], out["vendor2_us"]) / 1e6)
    out["vendor_over_hip"] = round(out["tf_vendor"] / out["tf_hip"], 3)
    print(json.dumps(out), flush=True)


print(json.dumps({"vendor_origin": lib.vgpt_gemm_vendor_origin().decode()}))
first = True
for M in (1448, 1848, 4096, 4128, 7740, 8192):
    for (N, K) in ((3072, 3072), (3072, 8192), (9216, 3072), (16384, 3072)):
        run("nt", M, N, K, check=first); first = False
first = True
for M in (4096, 7740):       # training backward: dX = dY W (N' = K of the layer), dW = dY^T X (reduction over the tokens)
    for (N, K) in ((3072, 9216), (3072, 3072), (8192, 3072), (3072, 16384)):
        run("nn", M, N, K, check=first); first = False
first = True
for T in (4096, 7740 // 64 * 64):
    for (Mo, No) in ((9216, 3072), (3072, 3072), (3072, 8192), (16384, 3072)):
        run("tn", Mo, No, T, check=first); first = False
print(json.dumps({"vendor_origin": lib.vgpt_gemm_vendor_origin().decode()}))
