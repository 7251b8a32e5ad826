"""EXPERIMENT (needs `make -C video-gpt_amd/csrc experiment-r64`).  Attention forward on the cfg-2 engine layout (hoisted: 1152-row prefix, 4096 live rows, 32 heads x 96): the
32-rows-per-wave kernel (128-row items) against the 64-rows-per-wave kernel (256-row items); and at L = 31 806 (cfg-4)."""
import importlib, math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _r64_lib import ops, Plan256, attention_r64
P = importlib.import_module("video-gpt_amd.processor")
LY = importlib.import_module("video-gpt_amd.layout")
dev, BF = "cuda:0", torch.bfloat16
nh, hd = 32, 96


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def pairs_of(pm, rows):
    bits = pm.bits[0, rows[0]:rows[1]].contiguous().view(torch.uint8)
    import numpy as np
    return int(np.unpackbits(bits.cpu().numpy()).sum())


# cfg-2 engine layout: [prefix 1032 | 16 diffusion | 16 time | gap | 2 x 2048 image rows]
C, G, bl = 4, 8, 258
lay = LY.TokenLayout.from_plans([(P.plan_inference([C, G])[0], bl, 0), (P.plan_inference([0, G])[0], bl, C * bl)], (C + G) * bl)
packed, _ = lay.pack()
import numpy as np
S0, nf, N = C * bl, 16, 256
x_old = [S0 + f * bl + 2 for f in range(G)] + [S0 + G * bl + f * bl + 2 for f in range(G)]
d_old = [x - 2 for x in x_old]; t_old = [x - 1 for x in x_old]
S = (S0 + 2 * nf + 127) // 128 * 128
perm = list(range(S0)) + d_old + t_old + [-1] * (S - S0 - 2 * nf) + [x + j for x in x_old for j in range(N)]
lp = packed.permute(np.array(perm))
pm = lp.packed_mask(dev)
L = len(perm)
qkv = torch.randn(1, L, 3 * nh * hd, device=dev).to(BF)
out = torch.empty(1, L - S, nh * hd, device=dev, dtype=BF)
segs = ((0, S, S + 2048), (0, S + 2048, L))
fl = 4 * nh * hd * pairs_of(pm, (S, L))
def report(name, us, flops):
    print(f"{name}: {us:.1f} us  ({flops / us / 1e6:.0f} TF/s, {flops / us / 1e6 / 2500:.3f} of peak)")


a = torch.empty_like(out); b = torch.empty_like(out)
report("cfg-2 live rows, 32 rows per wave (product)", t(lambda: ops.attention_qkv_range(qkv, pm, nh, nh, hd, S, a, segments=segs)), fl)
p256 = Plan256(pm, segs, nh)
report("cfg-2 live rows, 64 rows per wave (experiment)", t(lambda: attention_r64(qkv, pm, p256, nh, hd, b, q_start=S)), fl)
print("flagged waves:", int(p256.flags.sum()), " r64 vs r32: max abs", float((a.float() - b.float()).abs().max()),
      "rel", float((a.float() - b.float()).norm() / a.float().norm()))

# cfg-4: stage-1 layout, L = 31806
F, N4 = 16, 1024
kinds, _ = P.plan_stage1(2 * F - 1)
L4 = (2 * F - 1) * (N4 + 2)
pm4 = LY.TokenLayout.from_plans([(kinds, N4 + 2, 0)], L4).packed_mask(dev)
q4 = torch.randn(1, L4, 3 * nh * hd, device=dev).to(BF)
fl4 = 4 * nh * hd * pairs_of(pm4, (0, L4))
o4 = torch.empty(1, L4, nh * hd, device=dev, dtype=BF)
report(f"cfg-4 L={L4}, 32 rows per wave (product)", t(lambda: ops.attention_qkv(q4, pm4, nh, nh, hd, out=o4), n=5), fl4)
p4 = Plan256(pm4, ((0, 0, L4),), nh)
o4x = torch.empty_like(o4)
report(f"cfg-4 L={L4}, 64 rows per wave (experiment)", t(lambda: attention_r64(q4, pm4, p4, nh, hd, o4x), n=5), fl4)
print("flagged waves:", int(p4.flags.sum()), " rel", float((o4.float() - o4x.float()).norm() / o4.float().norm()))
