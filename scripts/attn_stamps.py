"""Developer probe (GPU): where a wave of the attention forward spends its cycles, from the s_memtime stamps of the
diagnostics build (`make -C video-gpt_amd/csrc attn-variant-VGPT_ATTN_STAMPS`, run with
VGPT_LIB=video-gpt_amd/libvgpt_hip_aVGPT_ATTN_STAMPS.so): per tile, wave 0 of every workgroup sums four stretches of the
loop -- wait + barrier | DMA issue + K fragments + QK^T MFMAs issued | V requests + softmax | P conversion + P V issued.
cfg-2 engine layout (live rows, planned launch), one full-width layer."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
importlib.import_module("video-gpt_amd")
M = importlib.import_module("video-gpt_amd.model"); P = importlib.import_module("video-gpt_amd.processor")
E = importlib.import_module("video-gpt_amd.engine"); S = importlib.import_module("video-gpt_amd.scheduler")
ops = importlib.import_module("video-gpt_amd.ops"); L_ = importlib.import_module("video-gpt_amd._lib")
dev = torch.device("cuda", 0); BF = torch.bfloat16
C, G, hw = 4, 8, (32, 32)
cfg = bench.full_config(M, 1)
model = bench.build_model(M, cfg, dev, seed=0)
proc = P.LVMProcessor(P.SpecialTokenizer(10, 11, 12), mask_format="layout")
prompt = "".join(f"<img><|image_{i + 1}|></img>" if i < C else f"<|diffusion|><|image_{i + 1}|>" for i in range(C + G))
prompt_ = "".join(f"<|diffusion|><|image_{i + 1}|>" for i in range(G))
imgs = [torch.zeros(3, 256, 256) for _ in range(C)]
batch = proc.prompt_condition_frame_block_inference([prompt, prompt_], [imgs, []], height=256, width=256, use_img_cfg=True, frame_blocks=[C, G])
z = [torch.randn(1, 4, *hw).to(dev, BF) for _ in range(G)] * 2
cond = [torch.randn(1, 4, *hw).to(dev, BF) for _ in range(C)]
sched = S.LVMScheduler(num_steps=4, time_shifting_factor=1)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    eng = E.StaticDenoiser(model, batch["input_ids"].to(dev), batch["position_ids"].to(dev), batch["attention_mask"], cond,
                           batch["input_image_sizes"], batch["denoise_image_sizes"], batch["time_emb_inx"], len(z), hw, True, 1.6,
                           "x1", sigma=sched.sigma, reuse_condition_prefix=True)
    eng.set_latents(torch.cat(z, 0)); eng.sampler_step(); torch.cuda.synchronize()
    nq, nk, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    run = lambda: ops.attention_qkv_range(eng.qkv_full[0].view(1, eng.L, -1), eng.pm, nq, nk, hd, eng.S, eng.ctx, segments=eng.seg_live)
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(st)
    for _ in range(20): run()
    e.record(st); torch.cuda.synchronize()
    print("avg attention launch us:", round(s.elapsed_time(e) / 20 * 1e3, 1), "S", eng.S, "L", eng.L, "rows", eng.Ma)
    plan = eng.pm.plan(eng.seg_live)
    cap = plan.n_items * nq
    tr = torch.zeros(3 * cap, 4, dtype=torch.int64, device=dev)
    L_.call("vgpt_attn_trace", tr.data_ptr(), cap)
    run(); torch.cuda.synchronize()
    L_.call("vgpt_attn_trace", None, 0)
t = tr.cpu().numpy().astype(np.int64)
base, ph, cl = t[:cap], t[cap:2 * cap], t[2 * cap:]
if not ph.any():
    raise SystemExit("no stamps: run with VGPT_LIB=.../libvgpt_hip_aVGPT_ATTN_STAMPS.so")
tiles = np.maximum(cl[:, 2], 1)
us = (base[:, 1] - base[:, 0]) / 100.0
cyc = cl[:, 1] - cl[:, 0]
print(f"workgroups {cap}; tiles per workgroup {tiles.min()}..{tiles.max()}; shader clock {np.median(cyc / np.maximum(us, 1e-9)) / 1e3:.2f} GHz (median)")
per = ph / tiles[:, None]
names = ("wait vmcnt(0) + barrier", "DMA issue, K fragments, QK^T MFMAs issued", "V requests + softmax (waits for the QK^T results)",
         "P conversion + P V MFMAs issued")
tot = per.sum(1)
print(f"cycles per tile (wave 0), median over workgroups: total {np.median(tot):.0f}  (kernel: {np.median(cyc / tiles):.0f} incl. prologue / epilogue)")
for i, n in enumerate(names):
    print(f"  {n:52s} {np.median(per[:, i]):7.0f}  ({100 * np.median(per[:, i] / tot):4.1f} %)   p10 {np.percentile(per[:, i], 10):6.0f}  p90 {np.percentile(per[:, i], 90):6.0f}")
