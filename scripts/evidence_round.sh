#!/bin/bash
# Round evidence in one gpurun call (developer helper): GPU suite, the driver's bench command, its rocprofv3 kernel stats, the
# three PMC passes, the other workloads.  usage: scripts/evidence_round.sh TAG  -> gpurun_out/TAG_*
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd $ROOT
scripts/gpu_round.sh $TAG || exit 1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_cmd.json.log 2> gpurun_out/${TAG}_bench_driver_cmd.err || exit 1
echo "driver-command bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_prof -o p -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/${TAG}_bench_profiled.json.log 2>&1 || exit 1
cd $ROOT
DB=$(find gpurun_out/${TAG}_prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/${TAG}_bench_kernel_stats.csv
rm -rf gpurun_out/${TAG}_prof
echo "kernel stats done"
scripts/pmc_round.sh $TAG || exit 1
timeout -k 10 300 python bench.py > gpurun_out/${TAG}_bench_default_50steps.json.log 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload stage1 --steps 10 --warmup 2 > gpurun_out/${TAG}_bench_stage1.json.log 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload pipeline --rounds 8 --steps 50 > gpurun_out/${TAG}_bench_pipeline_cfg5_bf16.json.log 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload pipeline --rounds 8 --steps 50 --attn-precision fp8 > gpurun_out/${TAG}_bench_pipeline_cfg5_fp8.json.log 2>/dev/null || exit 1
timeout -k 10 200 python scripts/setup_breakdown_probe.py > gpurun_out/${TAG}_setup_probe.log 2>&1
tail -2 gpurun_out/${TAG}_setup_probe.log
echo "all done"
