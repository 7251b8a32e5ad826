import importlib, sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for v in (("all","1"), ("auto","0"), ("all","1"), ("auto","0")):
    env = dict(os.environ, VGPT_GEMM_VENDOR=v[0], VGPT_X_GU=v[1])
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-stage1", "--no-vae", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("vendor", v, "no line", r.stderr[-3000:]); continue
    d = json.loads(line[-1])
    ks = {k["name"]: k.get("avg_us") for k in d["roofline"]["kernels"]}
    print("vendor=%s %.3f ms calib %s %s" % (v, d["ms_per_step"], [round(c["mfma_loop_tflops"]) for c in d.get("calibration", [])], ks), flush=True)
