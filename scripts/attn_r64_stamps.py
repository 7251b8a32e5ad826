"""EXPERIMENT.  In-kernel phase times of the 64-rows-per-wave attention kernel:
    make -C video-gpt_amd/csrc experiment-r64 X=-DVGPT_R64_STAMPS && python scripts/attn_r64_stamps.py
Per iteration four s_memtime stamps of workgroup 0 / wave 0 (written where the LSE would go): entry, after the Q.K^T phase,
after the P.V phase, end; prints median ticks per phase."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _r64_lib import ops, Plan256, attention_r64
dev, BF = "cuda:0", torch.bfloat16
nh, hd, L = 32, 96, 4096
pm = ops.pack_mask(torch.ones(1, L, L, dtype=torch.bool, device=dev))
qkv = torch.randn(1, L, 3 * nh * hd, device=dev).to(BF)
out = torch.empty(1, L, nh * hd, device=dev, dtype=BF)
plan = Plan256(pm, ((0, 0, L),), nh)
dbg = torch.zeros(nh * L, dtype=torch.float32, device=dev)
for _ in range(3):
    dbg.zero_()
    attention_r64(qkv, pm, plan, nh, hd, out, lse=dbg.data_ptr())
torch.cuda.synchronize()
st = dbg.view(torch.int64).cpu().numpy()
st = st[st != 0]
n = len(st) // 4 * 4
st = st[:n].reshape(-1, 4)
d_qk = np.diff(st[:, :2], axis=1)[:, 0]; d_pv = st[:, 2] - st[:, 1]; d_fl = st[:, 3] - st[:, 2]
gap = st[1:, 0] - st[:-1, 3]
it = st[1:, 0] - st[:-1, 0]
print("iterations stamped:", len(st))
for name, d in (("QK phase", d_qk), ("PV phase", d_pv), ("flush", d_fl), ("between iterations (barrier, DMA issue)", gap), ("whole iteration", it)):
    print(f"{name:45s} median {np.median(d):8.0f}  p10 {np.percentile(d, 10):8.0f}  p90 {np.percentile(d, 90):8.0f} ticks")
