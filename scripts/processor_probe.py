import importlib, time, torch, cProfile, pstats, io, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
importlib.import_module("video-gpt_amd")
P = importlib.import_module("video-gpt_amd.processor")
proc = P.LVMProcessor(P.SpecialTokenizer(10, 11, 12))
proc.collator.mask_format = "layout"
imgs = [(torch.rand(256, 256, 3, device="cuda") * 255).to(torch.uint8) for _ in range(8)]
C, G = 8, 8
prompt = "".join(f"<img><|image_{i + 1}|></img>" if i < C else f"<|diffusion|><|image_{i + 1}|>" for i in range(C + G))
prompt_ = "".join(f"<|diffusion|><|image_{i + 1}|>" for i in range(G))
def run():
    return proc.prompt_condition_frame_block_inference([prompt, prompt_], [list(imgs), []], height=256, width=256, use_img_cfg=True, frame_blocks=[C, G])
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): run()
torch.cuda.synchronize()
print("ms per call", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile(); pr.enable(); run(); torch.cuda.synchronize(); pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(10); print(st.getvalue()[:2200])
