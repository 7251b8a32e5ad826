"""Developer probe (GPU): time of the attention forward at the cfg-2-like launch (two packed sequences, 3096 + 2064 keys, 32 heads
x 96, all rows) for each library given on the command line (schedule variants of the hand-scheduled bodies:
VGPT_LIB=video-gpt_amd/libvgpt_hip_p2<x>.so); runs itself once per library."""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" not in sys.argv:
    for rep in range(2):
        for n in sys.argv[1:]:
            env = dict(os.environ)
            if n not in ("main", "classic"):
                env["VGPT_LIB"] = os.path.join(ROOT, "video-gpt_amd", f"libvgpt_hip_p2{n}.so")
            if n == "classic":
                env["VGPT_ATTN_P2"] = "0"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], env=env, capture_output=True, text=True, timeout=200)
            sys.stdout.write(r.stdout if r.returncode == 0 else r.stderr[-1500:]); sys.stdout.flush()
    sys.exit(0)
import torch
sys.path.insert(0, ROOT)
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
dev = "cuda:0"; H, D = 32, 96
m = torch.zeros(1, 5160, 5160, dtype=torch.bool); m[0, :3096, :3096] = True; m[0, 3096:, 3096:] = True
qkv = torch.randn(1, 5160, 3 * H * D, device=dev).to(torch.bfloat16)
pm = ops.pack_mask(m.to(dev))
run = lambda: ops.attention_qkv(qkv, pm, H, H, D)
for _ in range(10): run()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(40): run()
    e_.record(); torch.cuda.synchronize()
    best = min(best, s_.elapsed_time(e_) / 40 * 1e3)
print(json.dumps({"lib": sys.argv[-1], "us": round(best, 1)}), flush=True)
