#!/bin/bash
# Second evidence call (developer helper): rocprofv3 kernel stats of the SAMPLER leg alone (what roofline.kernels[*].avg_us is
# compared with), the reference's-full-work variants, cfg-4 shapes, the VAE workload, a 1000-step run.
set -o pipefail
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_prof -o p -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stage1 --no-vae > $ROOT/gpurun_out/${TAG}_bench_profiled_sampler.json.log 2>&1 || exit 1
cd $ROOT
DB=$(find gpurun_out/${TAG}_prof -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/${TAG}_bench_kernel_stats_sampler_only.csv
rm -rf gpurun_out/${TAG}_prof
head -8 gpurun_out/${TAG}_bench_kernel_stats_sampler_only.csv | cut -c1-160
V="--no-stage1 --no-cpu-baseline --no-vae"
: > gpurun_out/${TAG}_bench_variants.json.log
for extra in "--no-prefix-reuse" "--no-hoist" "--no-graph" "--attn-precision fp8"; do
  echo "# bench.py $V $extra" >> gpurun_out/${TAG}_bench_variants.json.log
  timeout -k 10 200 python bench.py $V $extra >> gpurun_out/${TAG}_bench_variants.json.log 2>/dev/null || exit 1
done
echo "# bench.py --steps 1000 $V" >> gpurun_out/${TAG}_bench_variants.json.log
timeout -k 10 300 python bench.py --steps 1000 $V >> gpurun_out/${TAG}_bench_variants.json.log 2>/dev/null || exit 1
echo "variants done"
timeout -k 10 300 python bench.py --workload stage4 --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_stage4_cfg4.json.log 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --workload stage4 --steps 3 --warmup 1 --grad-ckpt > gpurun_out/${TAG}_bench_stage4_cfg4_gradckpt.json.log 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload vae --steps 10 --warmup 2 > gpurun_out/${TAG}_bench_vae.json.log 2>/dev/null || exit 1
echo "all done"
