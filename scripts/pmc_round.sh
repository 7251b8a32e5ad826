#!/bin/bash
# Counter evidence for bench.py's roofline objects (developer helper, run through gpurun): three SEPARATE rocprofv3 --pmc
# passes (SQ, FETCH_SIZE, WRITE_SIZE; never combined with a trace) over a short eager run of the default workload, then
# scripts/pmc_mfma.py / pmc_traffic.py.  usage: scripts/pmc_round.sh TAG [extra bench args]   -> gpurun_out/TAG_pmc_*.json
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B="--steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-stage1 --no-vae --no-calibration $*"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $ROOT/gpurun_out/${TAG}_pmc_sq -o p --output-format csv -- python3 $ROOT/bench.py $B > $ROOT/gpurun_out/${TAG}_pmc_sq.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $ROOT/gpurun_out/${TAG}_pmc_fetch -o p --output-format csv -- python3 $ROOT/bench.py $B > $ROOT/gpurun_out/${TAG}_pmc_fetch.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $ROOT/gpurun_out/${TAG}_pmc_write -o p --output-format csv -- python3 $ROOT/bench.py $B > $ROOT/gpurun_out/${TAG}_pmc_write.log 2>&1 || exit 1
cd $ROOT
python scripts/pmc_mfma.py gpurun_out/${TAG}_pmc_sq gpurun_out/${TAG}_pmc_mfma.json | head -8
python scripts/pmc_traffic.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write gpurun_out/${TAG}_pmc_traffic.json | head -8
rm -rf gpurun_out/${TAG}_pmc_sq gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write
