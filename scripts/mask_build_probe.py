"""Developer probe (GPU): building the cfg-4 attention mask (stage-1 layout, 512^2, F=16, L=31 806) from token
attributes on the device vs painting the dense mask on the host and packing it."""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops"); P = importlib.import_module("video-gpt_amd.processor")
LY = importlib.import_module("video-gpt_amd.layout")
dev = "cuda:0"
for F, N in ((8, 256), (16, 1024)):
    bl = N + 2; kinds, _ = P.plan_stage1(2 * F - 1); L = (2 * F - 1) * bl
    t0 = time.perf_counter(); lay = LY.TokenLayout.from_plans([(kinds, bl, 0)], L); attr = torch.from_numpy(lay.attr()); t1 = time.perf_counter()
    a = attr.to(dev); torch.cuda.synchronize()
    for _ in range(2): pm = ops.build_mask_from_layout(a, 1, L)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    for _ in range(5): pm = ops.build_mask_from_layout(a, 1, L)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    t4 = time.perf_counter(); m = P.block_mask(kinds, bl, 0); t5 = time.perf_counter()
    md = torch.from_numpy(m)[None].to(dev); torch.cuda.synchronize(); t6 = time.perf_counter()
    ref = ops.pack_mask(md); torch.cuda.synchronize(); t7 = time.perf_counter()
    print(f"L={L}: layout host {1e3*(t1-t0):.2f} ms + device expand+summary {(t3-t2)/5*1e3:.3f} ms | dense: host paint {1e3*(t5-t4):.0f} ms, "
          f"H2D {1e3*(t6-t5):.0f} ms, pack {1e3*(t7-t6):.1f} ms | equal {torch.equal(pm.bits, ref.bits) and torch.equal(pm.summary, ref.summary)}")
