#!/bin/bash
# Round-4 evidence call (developer helper, run through gpurun): GPU suite, the driver's bench command, rocprofv3 kernel stats of
# the sampler leg, stage-1 with both GEMM families.  usage: scripts/evidence_round4.sh TAG
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/${TAG}_tests.log
timeout -k 10 280 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json.log 2> gpurun_out/${TAG}_bench.err || echo "bench FAILED"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_prof -o p -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stage1 --no-vae > $ROOT/gpurun_out/${TAG}_bench_profiled_sampler.json.log 2>&1 || echo "profile FAILED"
cd $ROOT
DB=$(find gpurun_out/${TAG}_prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/${TAG}_bench_kernel_stats_sampler_only.csv
rm -rf gpurun_out/${TAG}_prof
head -12 gpurun_out/${TAG}_bench_kernel_stats_sampler_only.csv | cut -c1-150
for f in 0 1; do
  timeout -k 10 200 python bench.py --workload stage1 --steps 10 --warmup 3 --gemm-family $f > gpurun_out/${TAG}_bench_stage1_family$f.json.log 2>/dev/null || echo "stage1 $f FAILED"
done
python - <<PY
import json,glob
for p in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json.log")):
    try:
        d=json.loads([l for l in open(p) if l.startswith("{")][-1])
        print(p.split("${TAG}_")[1], d["ms_per_step"], d["value"], [c.get("mfma_loop_tflops") for c in d.get("calibration",[])])
    except Exception as e:
        print(p, "unreadable", e)
PY
