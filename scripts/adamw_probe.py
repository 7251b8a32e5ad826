import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd"); T = importlib.import_module("video-gpt_amd.ops_train")
n = 113_246_208
dev = "cuda:0"
master = torch.randn(n, device=dev); param = master.to(torch.bfloat16); grad = (torch.randn(n, device=dev) * 1e-2).to(torch.bfloat16)
m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
def f(): T.adamw_step(master, param, grad, m, v, 1e-4, 0.9, 0.95, 1e-8, 0.1, 3)
for _ in range(3): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
t = s.elapsed_time(e) / 10 * 1e-3
print(f"adamw {n/1e6:.0f}M params: {t*1e6:.1f} us, {28*n/t/1e12:.2f} TB/s")
