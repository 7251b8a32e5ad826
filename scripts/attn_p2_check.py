"""Developer check (GPU): the hand-scheduled attention tile bodies (VGPT_ATTN_P2, default on for head dim 96) against the
compiler-scheduled kernel -- bit for bit on dense, block-causal, causal and packed masks -- and both against an fp32 reference;
then the time of the cfg-2-like launch (two packed sequences, 3096 + 2064 keys, 32 heads x 96) under each."""
import importlib, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
_lib = importlib.import_module("video-gpt_amd._lib").load()
dev = "cuda:0"; BF = torch.bfloat16
H, D = 32, 96


def masks():
    def blockdiag(sizes):
        L = sum(sizes); m = torch.zeros(L, L, dtype=torch.bool); o = 0
        for s in sizes:
            m[o:o + s, o:o + s] = True; o += s
        return m
    def frame_causal(L, f):
        i = torch.arange(L) // f
        return i[:, None] >= i[None, :]
    yield "dense_3096", torch.ones(1, 3096, 3096, dtype=torch.bool)
    yield "packed_3096_2064", blockdiag([3096, 2064])[None]
    yield "frame_causal_2100x2", torch.stack([frame_causal(2100, 300), frame_causal(2100, 420)])
    yield "causal_1000", torch.tril(torch.ones(1000, 1000, dtype=torch.bool))[None]
    m = frame_causal(1500, 100); m[700:740] = False; m[:, 64:128] = False; m[5, 64] = True
    yield "holes_1500", m[None]
    yield "stage1_like_7740", frame_causal(7740, 1290)[None]


def run(qkv, pm, p2):
    _lib.vgpt_attn_set_hand_scheduled(1 if p2 else 0)
    return ops.attention_qkv(qkv, pm, H, H, D)


ok = True
for name, m in masks():
    B, L = m.shape[0], m.shape[1]
    torch.manual_seed(L)
    qkv = (torch.randn(B, L, 3 * H * D, device=dev) * 1.5).to(BF)
    pm = ops.pack_mask(m.to(dev))
    o0 = run(qkv, pm, False); o1 = run(qkv, pm, True)
    torch.cuda.synchronize()
    same = torch.equal(o0, o1)
    q, k, v = [t.float().view(B, L, H, D).transpose(1, 2) for t in qkv.split(H * D, dim=-1)]
    hs = slice(0, 4)
    s = (q[:, hs] @ k[:, hs].transpose(-1, -2)) / math.sqrt(D)
    s = s.masked_fill(~m.to(dev)[:, None], float("-inf"))
    p = torch.softmax(s, dim=-1).nan_to_num(0.0)
    ref = (p @ v[:, hs]).transpose(1, 2).reshape(B, L, -1)
    got = o1.float().view(B, L, H, D)[:, :, hs].reshape(B, L, -1)
    rel = float((got - ref).norm() / ref.norm())
    print(f"{name}: P2 == compiler-scheduled {same}; max|diff| {float((o0.float() - o1.float()).abs().max()):.3e}; P2 vs fp32 rel-L2 {rel:.3e}", flush=True)
    ok = ok and same and rel < 1e-2
print("PARITY", "ok" if ok else "FAILED", flush=True)

m = next(x for n, x in masks() if n == "packed_3096_2064")
qkv = (torch.randn(1, 5160, 3 * H * D, device=dev)).to(BF)
pm = ops.pack_mask(m.to(dev))
for rep in range(2):
    for p2 in (False, True):
        for _ in range(5): run(qkv, pm, p2)
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(40): run(qkv, pm, p2)
        e_.record(); torch.cuda.synchronize()
        us = s_.elapsed_time(e_) / 40 * 1e3
        fl = 4.0 * H * D * (3096 ** 2 + 2064 ** 2)
        print(json.dumps({"launch": "packed 3096 + 2064, all rows", "p2": p2, "us": round(us, 1), "tflops": round(fl / us / 1e6)}), flush=True)
