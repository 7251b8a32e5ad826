"""Developer probe (GPU, stamps build: make -C video-gpt_amd/csrc gemm-w4-debug-16; VGPT_LIB=video-gpt_amd/libvgpt_hip_w4d16.so):
does the shader clock of a GEMM depend on how long the kernel runs?  One round of 256 workgroups (4096 x 3072 outputs, 256 x 192
tiles), the reduction length K swept: per launch the loop's cycles (s_memtime) over its wall time (s_memrealtime)."""
import ctypes, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
lib = importlib.import_module("video-gpt_amd._lib").load()
dev = "cuda:0"; BF = torch.bfloat16
dbg = torch.zeros(1 << 18, dtype=torch.int32, device=dev)
lib.vgpt_gemm_w4_debug_buffer.argtypes = [ctypes.c_void_p]
lib.vgpt_gemm_w4_debug_buffer(dbg.data_ptr())
M, N = 4096, 3072
for K in (1536, 3072, 6144, 12288, 24576, 49152):
    x = torch.randn(M, K, device=dev).to(BF)
    ws = [(torch.randn(N, K, device=dev) * 0.05).to(BF) for _ in range(2)]
    res = torch.randn(M, N, device=dev).to(BF); y = torch.empty(M, N, dtype=BF, device=dev)
    for i in range(4): ops.linear(x, ws[i % 2], residual=res, out=y)
    torch.cuda.synchronize(); dbg.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 12
    s.record()
    for i in range(n): ops.linear(x, ws[i % 2], residual=res, out=y)
    e.record(); torch.cuda.synchronize()
    d = dbg.cpu().view(-1, 8); d = d[d[:, 3] > 0].double()
    cyc, rt, nk = d[:, 0], d[:, 1], float(d[0, 3])
    print(json.dumps({"K": K, "launch_us": round(s.elapsed_time(e) / n * 1e3, 1), "loop_us_median": round(float(rt.median()) * 0.01, 1),
                      "cycles_per_ktile": round(float(cyc.median()) / nk, 1), "clock_ghz": round(float((cyc / rt).median()) * 0.1, 3),
                      "tflops": round(2.0 * M * N * K / (s.elapsed_time(e) / n * 1e-3) / 1e12)}), flush=True)
