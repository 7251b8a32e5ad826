import importlib, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import restate as R
from tests import smoke_case as SC
importlib.import_module("video-gpt_amd")
P = importlib.import_module("video-gpt_amd.processor"); TR = importlib.import_module("video-gpt_amd.train")
DEV, BF = "cuda:0", torch.bfloat16
cfg = R.Phi3Cfg(hidden_size=3072, intermediate_size=8192, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=32, vocab_size=64, pos_embed_max_size=32)
p = {k: v.to(BF).float() for k, v in R.make_params(cfg, seed=12).items()}
F, hw = 16, (64, 64)
proc = P.LVMProcessor(P.SpecialTokenizer(10, 11, 12)); proc.collator.mask_format = "layout"
prompt = "".join(f"<|diffusion|><|image_{i + 1}|><img><|image_{i + 1}|></img>" if i < F - 1 else f"<|diffusion|><|image_{i + 1}|>" for i in range(F))
row = proc.process_multi_modal_prompt_training(prompt, [torch.zeros(3, hw[0] * 8, hw[1] * 8) for _ in range(F)])
batch = proc.collator.collate_stage1([row], F)
batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items() if k not in ("input_pixel_values", "output_images")}
gen = torch.Generator("cpu").manual_seed(9)
mk = lambda n: torch.randn(n, 4, *hw, generator=gen).to(DEV)
x1, x0, clean, x0i = mk(F), mk(F), mk(F - 1), mk(F - 1)
t = torch.rand(F, generator=gen).to(DEV); ti = (0.9 + 0.1 * torch.rand(F - 1, generator=gen)).to(DEV)
outs = []
for ck in (False, False, True):
    model = SC.build_product_model(cfg, p, DEV, cls_name="LVMTraining")
    tr = TR.Stage1Trainer(model, lr=1e-4, weight_decay=0.1, gradient_checkpointing=ck)
    loss = tr.step(batch, x1, x0, t, clean, x0i, ti, update=False)
    outs.append((loss.clone(), {k: v.clone() for k, v in tr.grads.items()}))
    del tr, model; torch.cuda.empty_cache()
for a, b, name in ((0, 1, "noCk vs noCk"), (1, 2, "noCk vs ck")):
    diff = [k for k in outs[a][1] if not torch.equal(outs[a][1][k], outs[b][1][k])]
    print(name, "loss equal", torch.equal(outs[a][0], outs[b][0]), "differing grads:", diff[:12], len(diff))
