"""Developer probe (GPU): the four-wave GEMM kernel (gemm_w4_kernel, hand-scheduled loop) against the eight-wave kernels and
fp64 -- parity on every epilogue incl. ragged M / N, then an interleaved same-process A/B of the decoder's shapes with rotating
weights (six weight sets, so they stream from HBM as in the step) and of torch.matmul as an outside yardstick.
  VGPT_GEMM_W4_NI=8|6 forces the tile width of the four-wave kernel."""
import importlib, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
lib = importlib.import_module("video-gpt_amd._lib").load()
dev = "cuda:0"; BF = torch.bfloat16
g = torch.Generator("cpu").manual_seed(1)


def rel(y, ref):
    return float((y.double().cpu() - ref).norm() / ref.norm())


def both(fn):
    out = {}
    for fam in (0, 1):
        lib.vgpt_gemm_set_family(fam)
        out[fam] = fn()
    lib.vgpt_gemm_set_family(0)
    torch.cuda.synchronize()
    return out


bad = 0
if "--no-check" not in sys.argv:
    for (M, N, K, epi) in ((4096, 3072, 128, "none"), (4096, 3072, 3072, "resid"), (4000, 3000, 256, "resid"), (4100, 3080, 192, "bias"),
                           (4096, 9216, 3072, "none"), (4096, 3072, 8192, "resid"), (7740, 3072, 3072, "resid")):
        a = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
        ref = a.double() @ w.double().t()
        kw = {}
        if epi == "resid":
            r = torch.randn(M, N, generator=g).to(BF); kw["residual"] = r.to(dev); ref = ref + r.double()
        if epi == "bias":
            b = torch.randn(N, generator=g).to(BF); kw["bias"] = b.to(dev); ref = ref + b.double()
        ad, wd = a.to(dev), w.to(dev)
        ys = both(lambda: ops.linear(ad, wd, **kw))
        e0, e1 = rel(ys[0], ref), rel(ys[1], ref)
        dd = float((ys[0].float() - ys[1].float()).abs().max())
        ok = e0 < 4e-3 and abs(e0 - e1) < 2e-4
        bad += not ok
        print(f"linear {M}x{N}x{K} {epi}: rel-L2 w4 {e0:.3e}  8-wave {e1:.3e}  max|w4 - 8w| {dd:.3g}  {'ok' if ok else 'FAIL'}", flush=True)
    for (M, I, K, keep) in ((4096, 8192, 3072, False), (2100, 4096, 256, False), (2100, 4100 // 16 * 16, 256, True), (7740, 8192, 3072, True)):
        x = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(2 * I, K, generator=g) * 0.05).to(BF)
        gate, up = (x.double() @ w.double().t()).chunk(2, dim=-1)
        ref = up * torch.nn.functional.silu(gate)
        xd, wd = x.to(dev), w.to(dev)
        def run():
            gu = torch.empty(M, 2 * I, dtype=BF, device=dev) if keep else None
            y = ops.gated_mlp_act(xd, wd, ops.ACT_SILU, gate_up_out=gu)
            return y, gu
        ys = both(run)
        e0, e1 = rel(ys[0][0], ref), rel(ys[1][0], ref)
        ok = e0 < (8e-3 if keep else 5e-3) and abs(e0 - e1) < 3e-4
        if keep:
            gref = torch.cat([gate, up], dim=-1)
            eg = rel(ys[0][1], gref)
            ok = ok and eg < 4e-3
        bad += not ok
        print(f"gated {M}x{I}x{K} keep={keep}: rel-L2 w4 {e0:.3e}  8-wave {e1:.3e}  {'ok' if ok else 'FAIL'}", flush=True)
    for (M, nq, nkv, hd, K) in ((4096, 32, 32, 96, 3072), (4000, 8, 4, 96, 256), (2048, 16, 16, 64, 1024)):
        N = (nq + 2 * nkv) * hd
        x = torch.randn(M, K, generator=g).to(BF).to(dev); w = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(dev)
        pos = torch.arange(M, dtype=torch.int64, device=dev)
        cos, sin = ops.rope_table(pos, ops.rope_inv_freq(hd, 10000.0, dev))
        def run():
            return ops.linear_qkv_rope(x, w, cos, sin, nq, nkv, hd)
        ys = both(run)
        lib.vgpt_gemm_set_family(1)
        ref = ops.rope_qk_inplace(ops.linear(x, w), cos, sin, nq, nkv, hd)      # the unfused pair on the eight-wave kernel
        lib.vgpt_gemm_set_family(0)
        same8 = bool(torch.equal(ys[1], ref))
        d = (ys[0].float() - ref.float())
        e = float(d.norm() / ref.float().norm())
        ok = same8 and e < 3e-3
        bad += not ok
        print(f"qkv+rope {M}x{N}x{K} hd {hd}: 8-wave fused == unfused {same8}; w4 vs unfused rel-L2 {e:.3e}  {'ok' if ok else 'FAIL'}", flush=True)
    print("PARITY", "FAILED" if bad else "ok", flush=True)

# ---- rates: interleaved rounds, rotating weights ----
NW = 6
def bench_shape(name, M, N, K, kind):
    x = torch.randn(M, K, device=dev).to(BF)
    ws = [(torch.randn((2 * N if kind == "gated" else N), K, device=dev) * 0.05).to(BF) for _ in range(NW)]
    res = torch.randn(M, N, device=dev).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev)
    if kind == "rope":
        pos = torch.arange(M, dtype=torch.int64, device=dev)
        cos, sin = ops.rope_table(pos, ops.rope_inv_freq(96, 10000.0, dev))
    def call(w):
        if kind == "plain": ops.linear(x, w, out=y)
        elif kind == "resid": ops.linear(x, w, residual=res, out=y)
        elif kind == "gated": ops.gated_mlp_act(x, w, ops.ACT_SILU, out=y)
        elif kind == "rope": ops.linear_qkv_rope(x, w, cos, sin, 32, 32, 96, out=y)
    def vend(w):
        if kind == "resid": torch.addmm(res, x, w.t())
        else: torch.matmul(x, w.t())
    def t_of(f, n=24):
        for i in range(4): f(ws[i % NW])
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(n): f(ws[i % NW])
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    out = {"w4": [], "w8": [], "vendor": []}
    for rnd in range(3):
        lib.vgpt_gemm_set_family(0); out["w4"].append(t_of(call))
        lib.vgpt_gemm_set_family(1); out["w8"].append(t_of(call))
        lib.vgpt_gemm_set_family(0)
        out["vendor"].append(t_of(vend))
    fl = 2.0 * M * (2 * N if kind == "gated" else N) * K
    rec = {"shape": name, "M": M, "N": N, "K": K, "kind": kind}
    for k, v in out.items():
        rec[k + "_us"] = round(min(v), 1); rec[k + "_tf"] = round(fl / min(v) / 1e6)
    print(json.dumps(rec), flush=True)

for spec in (("o_proj", 4096, 3072, 3072, "resid"), ("down_proj", 4096, 3072, 8192, "resid"), ("qkv_rope", 4096, 9216, 3072, "rope"),
             ("gate_up", 4096, 8192, 3072, "gated"), ("square", 8192, 8192, 8192, "plain"), ("train_qkv", 7740, 9216, 3072, "plain"),
             ("train_down", 7740, 3072, 8192, "resid")):
    bench_shape(*spec)
sys.exit(1 if bad else 0)
