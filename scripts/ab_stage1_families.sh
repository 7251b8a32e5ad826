set -o pipefail
for i in 1 2; do for f in 0 1; do
timeout -k 10 250 python bench.py --workload stage1 --steps 10 --warmup 3 --gemm-family $f > gpurun_out/r04_ab_dw_w4_family${f}_$i.json.log 2>/dev/null || exit 1
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r04_ab_dw_w4_family${f}_$i.json.log") if l.startswith("{")][-1]); print("family=$f run $i", d["ms_per_step"], d["config"].get("loss_first_last"))
PY
done; done
