"""Developer probe (GPU): per-workgroup timeline of the attention forward on the cfg-2 engine layout.
python scripts/attn_trace.py  ->  makespan vs. mean busy time per CU slot, active tiles per workgroup."""
import importlib, json, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
vg = importlib.import_module("video-gpt_amd")
M = importlib.import_module("video-gpt_amd.model"); P = importlib.import_module("video-gpt_amd.processor")
E = importlib.import_module("video-gpt_amd.engine"); S = importlib.import_module("video-gpt_amd.scheduler")
ops = importlib.import_module("video-gpt_amd.ops"); L_ = importlib.import_module("video-gpt_amd._lib")
dev = torch.device("cuda", 0); BF = torch.bfloat16
C, G, hw = 4, 8, (32, 32)
cfg = bench.full_config(M, 1)
model = bench.build_model(M, cfg, dev, seed=0)
proc = P.LVMProcessor(P.SpecialTokenizer(10, 11, 12))
prompt = "".join(f"<img><|image_{i + 1}|></img>" if i < C else f"<|diffusion|><|image_{i + 1}|>" for i in range(C + G))
prompt_ = "".join(f"<|diffusion|><|image_{i + 1}|>" for i in range(G))
imgs = [torch.zeros(3, 256, 256) for _ in range(C)]
batch = proc.prompt_condition_frame_block_inference([prompt, prompt_], [imgs, []], height=256, width=256, use_img_cfg=True, frame_blocks=[C, G])
z = [torch.randn(1, 4, *hw).to(dev, BF) for _ in range(G)] * 2
cond = [torch.randn(1, 4, *hw).to(dev, BF) for _ in range(C)]
sched = S.LVMScheduler(num_steps=4, time_shifting_factor=1)
reuse = "--no-reuse" not in sys.argv
eng = E.StaticDenoiser(model, batch["input_ids"].to(dev), batch["position_ids"].to(dev), batch["attention_mask"].to(dev), cond,
                       batch["input_image_sizes"], batch["denoise_image_sizes"], batch["time_emb_inx"], len(z), hw, True, 1.6, "x1",
                       sigma=sched.sigma, reuse_condition_prefix=reuse)
eng.set_latents(torch.cat(z, 0)); eng.sampler_step(); torch.cuda.synchronize()
nq, nk, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
if "--solo" in sys.argv:   # only group A has rows: intrinsic phase lengths without a partner wave
    eng.seg_live = ((0, eng.S, eng.S + 128),)
IR = 128
def run():
    if eng.S:
        ops.attention_qkv_range(eng.qkv_full[0].view(1, eng.L, -1), eng.pm, nq, nk, hd, eng.S, eng.ctx, segments=eng.seg_live)
    else:
        ops.attention_qkv_range(eng.qkv, eng.pm, nq, nk, hd, 0, eng.ctx, segments=eng.seg_all)
for _ in range(5): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print("avg attention launch us:", s.elapsed_time(e) / 20 * 1e3, "S", eng.S, "L", eng.L, "Ma", eng.Ma)
plan = eng.pm.plan(eng.seg_live if eng.S else eng.seg_all)
print("items", plan.items.cpu().tolist(), "order", plan.order.cpu().tolist())
nblk = plan.n_items * nq
NSLOT = 512
tr = torch.zeros(nblk + 8 * 32, 4, dtype=torch.int64, device=dev)
L_.call("vgpt_attn_trace", tr.data_ptr(), nblk)
run(); torch.cuda.synchronize()
L_.call("vgpt_attn_trace", None, 0)
stamps = tr[nblk:].cpu().numpy().astype(np.int64).reshape(8, 32, 4)
t = tr[:nblk].cpu().numpy().astype(np.uint64)
if stamps.any():
    base = stamps[stamps > 0].min()
    for w in (0, 1, 4, 5):
        s_ = stamps[w]
        print(f"wave {w}: M dur", (s_[2:14, 1] - s_[2:14, 0]).tolist(), " V dur", (s_[2:14, 3] - s_[2:14, 2]).tolist())
        print(f"        barrier waits: M-end->V-start", (s_[2:14, 2] - s_[2:14, 1]).tolist(), " V-end->next M-start", (s_[3:15, 0] - s_[2:14, 3]).tolist())
        print(f"        tile period", (s_[3:15, 0] - s_[2:14, 0]).tolist())
t0 = t[:, 0].min()
st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us
hw_id = t[:, 2] & 0xffffffff; xcc = (t[:, 2] >> 32) & 0xf
cu = (hw_id >> 8) & 0xf; se = (hw_id >> 13) & 0x7
tiles = t[:, 3] & 0xffffffff
slot = xcc * 64 + se * 16 + cu
dur = en - st
print("blocks", nblk, "makespan us", en.max(), "mean dur", dur.mean(), "sum dur / slots", dur.sum() / NSLOT)
print("tiles/block: min", tiles.min(), "max", tiles.max(), "mean", tiles.mean(), " us/tile", (dur / np.maximum(tiles, 1)).mean())
for x in range(8):
    m = xcc == x
    print(f"xcc {x}: blocks {m.sum()} first start {st[m].min():.1f} last end {en[m].max():.1f} busy-sum {dur[m].sum():.0f} distinct CUs {len(set(slot[m]))}")
order = np.argsort(st)
np.save("gpurun_out/attn_trace.npy", np.stack([st, en, slot.astype(np.float64), tiles.astype(np.float64)], 1))
late = en > 0.8 * en.max()
print("blocks ending in the last 20% of the makespan:", late.sum(), " their tiles:", sorted(tiles[late].tolist())[:40])
