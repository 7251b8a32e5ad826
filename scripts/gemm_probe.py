"""Developer probe (GPU): NT / NN / TN GEMM rates on the stage-1 shapes and on a square problem."""
import importlib, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops"); T = importlib.import_module("video-gpt_amd.ops_train")
dev = "cuda:0"; BF = torch.bfloat16
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
def run(M, N, K):
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF); dy = torch.randn(M, N, device=dev).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev); dx = torch.empty(M, K, dtype=BF, device=dev); dw = torch.empty(N, K, dtype=BF, device=dev)
    fl = 2.0 * M * N * K
    t_nt = timeit(lambda: ops.linear(x, w, out=y))
    t_nn = timeit(lambda: T.linear_dx(dy, w, out=dx))
    t_tn = timeit(lambda: T.linear_dw(dy, x, out=dw))
    print(f"M={M} N={N} K={K}: NT {t_nt:7.1f} us {fl/t_nt/1e6:6.0f} TF | NN(dX) {t_nn:7.1f} us {fl/t_nn/1e6:6.0f} TF | TN(dW) {t_tn:7.1f} us {fl/t_tn/1e6:6.0f} TF")
for shp in [(8192, 8192, 8192), (7740, 9216, 3072), (7740, 3072, 3072), (7740, 16384, 3072), (7740, 3072, 8192), (4096, 4096, 4096)]:
    run(*shp)
print("same product, weight stored [N][K] (NT) vs [K][N] (NN):")
for (M, N, K) in [(4128, 9216, 3072), (4128, 3072, 3072), (4128, 3072, 8192), (4128, 16384, 3072), (5160, 9216, 3072), (8192, 8192, 8192), (8192, 8192, 8000)]:
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(N, K, device=dev) * 0.05).to(BF); wT = w.t().contiguous()
    y = torch.empty(M, N, dtype=BF, device=dev); y2 = torch.empty(M, N, dtype=BF, device=dev)
    fl = 2.0 * M * N * K
    t_nt = timeit(lambda: ops.linear(x, w, out=y)); t_nn = timeit(lambda: T.linear_dx(x, wT, out=y2))
    print(f"M={M} N={N} K={K}: NT {t_nt:7.1f} us {fl/t_nt/1e6:6.0f} TF | NN {t_nn:7.1f} us {fl/t_nn/1e6:6.0f} TF | equal {torch.equal(y, y2)}")
