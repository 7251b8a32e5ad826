set -o pipefail
B="--steps 30 --warmup 5 --no-stage1 --no-vae --no-cpu-baseline"
for i in 1 2; do
for p in 1 0; do
VGPT_ATTN_P2=$p timeout -k 10 200 python bench.py $B > gpurun_out/r04_ab_attn_p2_${p}_$i.json.log 2> gpurun_out/ab.err || exit 1
python - <<PY
import json
for l in open("gpurun_out/r04_ab_attn_p2_${p}_$i.json.log"):
    if l.startswith("{"):
        d=json.loads(l); ks=d["roofline"].get("kernels",{})
        print("P2=$p run $i ms/step", d["ms_per_step"], {k:(v.get("us") if isinstance(v,dict) else v) for k,v in ks.items()} if isinstance(ks,dict) else "")
PY
done; done
