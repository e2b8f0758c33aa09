#!/bin/bash
# Round 5, GPU session 7: hipGraph replay of the iteration in the mid-size regime after the step-per-launch pseudo-inverse
# (tools/bench_midsize.py: a 1000-step region without profiling events, so SKF_GRAPH=1 is honoured).   tools/r05_session7.sh <out-name>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5s7}; OUT=gpurun_out/$N; mkdir -p $OUT
for rep in 1 2; do
  for v in "SKF_GRAPH=0" "SKF_GRAPH=1"; do
    env $v timeout 300 python tools/bench_midsize.py 0.05 0.1 0.2 0.3 2>&1 | grep "^scale" | sed "s/^/[$v] /" | tee -a $OUT/summary.txt
  done
done
echo done | tee -a $OUT/summary.txt
