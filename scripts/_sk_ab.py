import importlib, sys, os, json, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for v in ("auto", "0", "auto", "0"):
    env = dict(os.environ, VGPT_GEMM_VENDOR=v)
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-stage1", "--no-vae", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("vendor", v, "no line", r.stderr[-3000:]); continue
    d = json.loads(line[-1])
    ks = {k["name"]: k.get("avg_us") for k in d["roofline"]["kernels"]}
    print("vendor=%s %.3f ms setup %s calib %s %s" % (v, d["ms_per_step"], d.get("per_clip_setup_ms"), [round(c["mfma_loop_tflops"]) for c in d.get("calibration", [])], ks), flush=True)
