#!/bin/bash
# Round 5, GPU session 6: the step-per-launch sweep above order 256 against the blocked Cholesky inverse.   tools/r05_session6.sh <out-name>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5s6}; OUT=gpurun_out/$N; mkdir -p $OUT
export TMPDIR=/tmp
ORDERS="256 300 400 512 768 1023"
for v in "SKF_NONE=1" "SKF_SWEEP_BIG=0"; do
  echo "== $v" | tee -a $OUT/summary.txt
  env $v timeout 600 python tools/bench_pinv.py $ORDERS 2>&1 | grep "full rank\|rank n/2" | tee -a $OUT/summary.txt
done
timeout 900 python -m pytest tests -m gpu -x -q -k "pinv or sweep" 2>&1 | tail -3 | tee -a $OUT/summary.txt
echo done | tee -a $OUT/summary.txt
