"""Developer probe (GPU): correctness (against fp64 on the same bf16 inputs) and rate of vgpt_gemm_bf16 /
vgpt_gated_mlp_act_fwd under VGPT_GEMM_TILE (0 = launch plan, 256 / 192 = forced big tile; experimental kernel variants
were A/B-tested with it under further tile codes)."""
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
print("tile =", os.environ.get("VGPT_GEMM_TILE", "plan"))
g = torch.Generator("cpu").manual_seed(1)
for (M, N, K, epi) in ((4096, 3072, 256, "none"), (4000, 3000, 128, "resid"), (777, 1028, 64, "bias"), (4096, 9216, 3072, "none")):
    a = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(N, K, generator=g) * 0.05).to(BF)
    ref = a.double() @ w.double().t()
    kw = {}
    if epi == "resid":
        r = torch.randn(M, N, generator=g).to(BF); kw["residual"] = r.to(dev); ref = ref + r.double()
    if epi == "bias":
        b = torch.randn(N, generator=g).to(BF); kw["bias"] = b.to(dev); ref = ref + b.double()
    y = ops.linear(a.to(dev), w.to(dev), **kw).cpu().double()
    print(f"linear {M}x{N}x{K} {epi}: rel-L2 {float((y - ref).norm() / ref.norm()):.2e}")
for (M, I, K) in ((2100, 8192, 256), (4096, 8192, 3072)):
    x = torch.randn(M, K, generator=g).to(BF); w = (torch.randn(2 * I, K, generator=g) * 0.05).to(BF)
    gate, up = (x.double() @ w.double().t()).chunk(2, dim=-1)
    ref = up * torch.nn.functional.silu(gate)
    y = ops.gated_mlp_act(x.to(dev), w.to(dev), ops.ACT_SILU).cpu().double()
    print(f"gated {M}x{I}x{K}: rel-L2 {float((y - ref).norm() / ref.norm()):.2e}")
line = ""
for (M, N, K) in ((8192, 8192, 8192), (4096, 9216, 3072), (4096, 3072, 3072), (4096, 3072, 8192), (7740, 9216, 3072)):
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev)
    t = timeit(lambda: ops.linear(x, w, out=y))
    line += f"  {M}x{N}x{K} {t:6.1f} us {2.0*M*N*K/t/1e6:5.0f} TF"
x = torch.randn(4096, 3072, device=dev).to(BF); w = (torch.randn(16384, 3072, device=dev) * 0.05).to(BF)
y = torch.empty(4096, 8192, dtype=BF, device=dev)
t = timeit(lambda: ops.gated_mlp_act(x, w, out=y))
line += f"  gate_up4096 {t:6.1f} us {2.0*4096*16384*3072/t/1e6:5.0f} TF"
print(line)
