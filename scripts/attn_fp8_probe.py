"""Developer probe (GPU): attention forward on the cfg-2 engine layout (hoisted: 1152-row prefix, 4096 live rows (L = 5248), 32 heads
x 96) and at L = 31 806 (cfg-4 shapes): bf16 kernel against the MX-fp8 path (quantise + kernel, and the kernel alone)."""
import importlib, os, sys, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
L_ = importlib.import_module("video-gpt_amd._lib")
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
    return int(np.unpackbits(bits.cpu().numpy()).sum())


def report(name, us, flops):
    print(f"{name}: {us:.1f} us  ({flops / us / 1e6:.0f} TF/s algorithmic)")


def fp8_kernel_only(qkv, pm, plan, out, q_start):
    B, L, w = qkv.shape
    ws = ops._FP8_WS[(qkv.device, B, L, nh, nh, hd)]
    hq = nh * hd
    return lambda: L_.call("vgpt_attn_fwd_plan_fp8", ws.data_ptr(), out.data_ptr() - q_start * hq * 2, pm.bits.data_ptr(),
                           plan.items.data_ptr(), plan.summary.data_ptr(), plan.order.data_ptr(), plan.n_items, B, L, nh, nh, hd,
                           L * hq, hd, hq, ops._stream())


C, G, bl = 4, 8, 258
lay = LY.TokenLayout.from_plans([(P.plan_inference([C, G])[0], bl, 0), (P.plan_inference([0, G])[0], bl, C * bl)], (C + G) * bl)
packed, _ = lay.pack()
S0, nf, N = C * bl, 16, 256
x_old = [S0 + f * bl + 2 for f in range(G)] + [S0 + G * bl + f * bl + 2 for f in range(G)]
d_old = [x - 2 for x in x_old]; t_old = [x - 1 for x in x_old]
S = (S0 + 2 * nf + 127) // 128 * 128
perm = list(range(S0)) + d_old + t_old + [-1] * (S - S0 - 2 * nf) + [x + j for x in x_old for j in range(N)]
pm = packed.permute(np.array(perm)).packed_mask(dev)
L = len(perm)
qkv = torch.randn(1, L, 3 * nh * hd, device=dev).to(BF)
a = torch.empty(1, L - S, nh * hd, device=dev, dtype=BF); b = torch.empty_like(a)
segs = ((0, S, S + 2048), (0, S + 2048, L))
fl = 4 * nh * hd * pairs_of(pm, (S, L))
report("cfg-2 live rows, bf16", t(lambda: ops.attention_qkv_range(qkv, pm, nh, nh, hd, S, a, segments=segs)), fl)
report("cfg-2 live rows, fp8 (quantise + attention)", t(lambda: ops.attention_qkv_fp8(qkv, pm, nh, nh, hd, out=b, q_start=S, segments=segs)), fl)
report("cfg-2 live rows, fp8 attention kernel alone", t(fp8_kernel_only(qkv, pm, pm.plan(segs), b, S)), fl)
print("fp8 vs bf16: rel-L2", float((a.float() - b.float()).norm() / a.float().norm()))

F, N4 = 16, 1024
kinds, _ = P.plan_stage1(2 * F - 1)
L4 = (2 * F - 1) * (N4 + 2)
pm4 = LY.TokenLayout.from_plans([(kinds, N4 + 2, 0)], L4).packed_mask(dev)
q4 = torch.randn(1, L4, 3 * nh * hd, device=dev).to(BF)
fl4 = 4 * nh * hd * pairs_of(pm4, (0, L4))
o4 = torch.empty(1, L4, nh * hd, device=dev, dtype=BF); o4x = torch.empty_like(o4)
report(f"cfg-4 L={L4}, bf16", t(lambda: ops.attention_qkv(q4, pm4, nh, nh, hd, out=o4), n=5), fl4)
report(f"cfg-4 L={L4}, fp8 (quantise + attention)", t(lambda: ops.attention_qkv_fp8(q4, pm4, nh, nh, hd, out=o4x), n=5), fl4)
report(f"cfg-4 L={L4}, fp8 attention kernel alone", t(fp8_kernel_only(q4, pm4, pm4.plan(None), o4x, 0), n=5), fl4)
print("fp8 vs bf16: rel-L2", float((o4.float() - o4x.float()).norm() / o4.float().norm()))
