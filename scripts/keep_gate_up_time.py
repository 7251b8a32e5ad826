import importlib, os, sys, json
sys.path.insert(0, os.getcwd())
import torch
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
dev="cuda:0"; BF=torch.bfloat16
M,I,K=7740,8192,3072
x=torch.randn(M,K,device=dev).to(BF); ws=[(torch.randn(2*I,K,device=dev)*0.05).to(BF) for _ in range(2)]
gu=torch.empty(M,2*I,dtype=BF,device=dev); act=torch.empty(M,I,dtype=BF,device=dev)
f=lambda i: ops.gated_mlp_act(x, ws[i%2], ops.ACT_SILU, out=act, gate_up_out=gu)
for i in range(5): f(i)
torch.cuda.synchronize()
best=1e9
for _ in range(3):
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(20): f(i)
    e.record(); torch.cuda.synchronize()
    best=min(best,s.elapsed_time(e)/20*1e3)
print(os.environ.get("VGPT_LIB","new")[-12:], round(best,1), "us", round(2.0*M*2*I*K/best/1e6), "TF/s")
