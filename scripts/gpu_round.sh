#!/bin/bash
# One gpurun call's worth of checks (developer helper): the GPU test suite, then same-box bench legs.
# usage: scripts/gpu_round.sh TAG [pytest-args...]   (outputs under gpurun_out/TAG_*)
set -o pipefail
TAG=$1; shift
timeout -k 10 840 python -m pytest tests -m gpu -q -s "$@" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
echo "pytest rc=$rc"; tail -3 gpurun_out/${TAG}_tests.log
[ $rc -le 1 ] || exit $rc
