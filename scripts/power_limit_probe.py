"""Developer probe (GPU): is a kernel bound by its own schedule or by the package power limit?  Each kernel is timed (a) as a
single launch after 100 ms of idle (the clock has recovered), (b) as the average of launches 1..10, 91..100 and 991..1000 of a
back-to-back train.  A schedule-bound kernel takes the same time in all of them; a power-bound one slows as the train goes on.
Kernels: the attention forward at the cfg-2-like launch (hand-scheduled bodies and VGPT_ATTN_P2=0), the step's gate_up and
down_proj products."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
_lib = importlib.import_module("video-gpt_amd._lib").load()
dev = "cuda:0"; BF = torch.bfloat16
H, D = 32, 96
m = torch.zeros(1, 5160, 5160, dtype=torch.bool); m[0, :3096, :3096] = True; m[0, 3096:, 3096:] = True
qkv = torch.randn(1, 5160, 3 * H * D, device=dev).to(BF)
pm = ops.pack_mask(m.to(dev))
x = torch.randn(4096, 3072, device=dev).to(BF); wgu = (torch.randn(16384, 3072, device=dev) * 0.05).to(BF)
a8 = torch.randn(4096, 8192, device=dev).to(BF); wd = (torch.randn(3072, 8192, device=dev) * 0.05).to(BF)
res = torch.randn(4096, 3072, device=dev).to(BF)
ygu = torch.empty(4096, 8192, dtype=BF, device=dev); yd = torch.empty(4096, 3072, dtype=BF, device=dev)


def attn(p2):
    def f():
        _lib.vgpt_attn_set_hand_scheduled(int(p2))
        ops.attention_qkv(qkv, pm, H, H, D)
    return f


kernels = {"attention_hand_scheduled": attn("1"), "attention_compiler_scheduled": attn("0"),
           "gate_up_4096x16384x3072": lambda: ops.gated_mlp_act(x, wgu, ops.ACT_SILU, out=ygu),
           "down_proj_4096x3072x8192": lambda: ops.linear(a8, wd, residual=res, out=yd)}
for name, f in kernels.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    cold = []
    for _ in range(8):
        time.sleep(0.1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        cold.append(s.elapsed_time(e) * 1e3)
    time.sleep(0.2)
    n = 1000
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        f(); ev[i + 1].record()
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    avg = lambda lo, hi: round(sum(t[lo:hi]) / (hi - lo), 1)
    print(json.dumps({"kernel": name, "single_launch_after_idle_us": round(sorted(cold)[len(cold) // 2], 1),
                      "train_launch_1_10_us": avg(0, 10), "train_launch_91_100_us": avg(90, 100), "train_launch_991_1000_us": avg(990, 1000),
                      "train_seconds": round(sum(t) / 1e6, 3)}), flush=True)
