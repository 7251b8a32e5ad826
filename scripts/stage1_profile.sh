cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT}
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s1prof22 -o p -- python3 $ROOT/bench.py --workload stage1 --steps 8 --warmup 2 --no-calibration > $ROOT/gpurun_out/r04_v22_stage1_profiled.json.log 2>&1 || echo FAILED
cd $ROOT
DB=$(find gpurun_out/s1prof22 -name "*.db" | head -1)
python scripts/rocpd_stats.py $DB gpurun_out/r04_v22_stage1_kernel_stats.csv
rm -rf gpurun_out/s1prof22
head -14 gpurun_out/r04_v22_stage1_kernel_stats.csv | cut -c1-170
