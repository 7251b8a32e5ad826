"""Developer probe (GPU): in-kernel stamps and cost attribution of the four-wave GEMM loop.  Runs itself once per diagnostics
library (VGPT_LIB=video-gpt_amd/libvgpt_hip_w4dN.so, `make -C video-gpt_amd/csrc gemm-w4-debug-N`): N & 16 = stamps (loop
cycles, in-kernel clock, cycles at the barriers), N & 1 / 2 / 4 / 8 = no fetches / no LDS writes / no fragment reads / no
barrier (results are garbage, timing only)."""
import importlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "--child" not in sys.argv:
    libs = sys.argv[1:] or ["16", "17", "18", "20", "24", "31"]
    for n in libs:
        env = dict(os.environ)
        if n != "main":
            env["VGPT_LIB"] = os.path.join(ROOT, "video-gpt_amd", f"libvgpt_hip_w4d{n}.so")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], env=env, capture_output=True, text=True, timeout=300)
        sys.stdout.write(r.stdout); sys.stdout.flush()
        if r.returncode:
            sys.stdout.write(r.stderr[-2000:])
    sys.exit(0)
import torch
sys.path.insert(0, ROOT)
importlib.import_module("video-gpt_amd")
ops = importlib.import_module("video-gpt_amd.ops")
lib = importlib.import_module("video-gpt_amd._lib").load()
name_ = sys.argv[sys.argv.index("--child") + 1]
tag = int("".join(c for c in name_ if c.isdigit()) or 0) if name_ != "main" else 0
tag = tag if name_ == "main" or name_[0].isdigit() else 0
import re as _re
tag = int(_re.match(r"\d+", name_).group(0)) if _re.match(r"\d+", name_) else 0
dev = "cuda:0"; BF = torch.bfloat16
stamps = bool(tag & 16)
dbg = torch.zeros(1 << 18, dtype=torch.int32, device=dev)
if stamps:
    lib.vgpt_gemm_w4_debug_buffer.argtypes = [__import__("ctypes").c_void_p]
    lib.vgpt_gemm_w4_debug_buffer(dbg.data_ptr())
for name, M, N, K, kind in (("gate_up", 4096, 8192, 3072, "gated"), ("o_proj", 4096, 3072, 3072, "resid"), ("down_proj", 4096, 3072, 8192, "resid"),
                            ("qkv_rope", 4096, 9216, 3072, "rope"), ("qkv_plain", 4096, 9216, 3072, "plain"), ("square", 8192, 8192, 8192, "plain")):
    x = torch.randn(M, K, device=dev).to(BF)
    ws = [(torch.randn((2 * N if kind == "gated" else N), K, device=dev) * 0.05).to(BF) for _ in range(4)]
    res = torch.randn(M, N, device=dev).to(BF)
    y = torch.empty(M, N, dtype=BF, device=dev)
    if kind == "rope":
        cos, sin = ops.rope_table(torch.arange(M, dtype=torch.int64, device=dev), ops.rope_inv_freq(96, 10000.0, dev))
    def call(w):
        if kind == "plain": ops.linear(x, w, out=y)
        elif kind == "resid": ops.linear(x, w, residual=res, out=y)
        elif kind == "rope": ops.linear_qkv_rope(x, w, cos, sin, 32, 32, 96, out=y)
        else: ops.gated_mlp_act(x, w, ops.ACT_SILU, out=y)
    for i in range(6): call(ws[i % 4])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 24
    s.record()
    for i in range(n): call(ws[i % 4])
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    rec = {"lib": name_, "shape": name, "us": round(us, 1)}
    if stamps:
        d = dbg.cpu().view(-1, 8)
        d = d[d[:, 3] > 0].double()
        t0, t1, t2, t3 = d[:, 4], d[:, 5], d[:, 6], d[:, 7]
        base = float(t0.min())
        rec.update({"entry_spread_us": round(float(t0.max() - base) * 0.01, 2), "prologue_cpp_us": round(float((t1 - t0).median()) * 0.01, 2),
                    "asm_us_median": round(float((t2 - t1).median()) * 0.01, 2), "epilogue_us_median": round(float((t3 - t2).median()) * 0.01, 2),
                    "epilogue_us_max": round(float((t3 - t2).max()) * 0.01, 2),
                    "kernel_span_us": round(float(t3.max() - base) * 0.01, 2)})
        nk = float(d[0, 3])
        cyc, rt, bar = d[:, 0], d[:, 1], d[:, 2]
        rec.update({"workgroup_waves": int(d.shape[0]), "cycles_per_ktile_median": round(float(cyc.median() / nk), 1),
                    "cycles_per_ktile_max": round(float(cyc.max() / nk), 1),
                    "clock_ghz": round(float((cyc / rt).median() * 0.1), 3),
                    "barrier_cycles_per_ktile": round(float(bar.median() / nk), 1),
                    "loop_us_median": round(float(rt.median() * 0.01), 1)})
        dbg.zero_()
    print(json.dumps(rec), flush=True)
