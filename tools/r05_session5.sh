#!/bin/bash
# Round 5, GPU session 5: sweep_step_kernel with the branch-free look-ahead pivot block.   tools/r05_session5.sh <out-name>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5s5}; OUT=gpurun_out/$N; mkdir -p $OUT
export TMPDIR=/tmp
ORDERS="65 96 128 160 192 224 256"
for v in "SKF_SWEEP_STEP_MIN=0" "SKF_SWEEP_STEP_MIN=65 SKF_SWEEP_ROWS=32"; do
  echo "== $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python tools/bench_pinv.py $ORDERS 2>&1 | grep "full rank\|rank n/2" | tee -a $OUT/summary.txt
done
if [ -f tools/probe/_build/libskf_stamps.so ]; then
  for v in "SKF_SWEEP_STEP_MIN=65"; do
    echo "== stamps $v" | tee -a $OUT/summary.txt
    env $v SKF_LIB_PATH=$PWD/tools/probe/_build/libskf_stamps.so timeout 300 python tools/bench_pinv.py 256 2>&1 | grep "sweep_" | sort | uniq -c | sort -rn | head -6 | tee -a $OUT/summary.txt
  done
fi
timeout 900 python -m pytest tests -m gpu -x -q -k "pinv or sweep" 2>&1 | tail -3 | tee -a $OUT/summary.txt
for v in "SKF_SWEEP_STEP_MIN=0" "SKF_SWEEP_STEP_MIN=65" "SKF_NONE=1"; do
  env $v timeout 600 python - <<PY 2>&1 | tail -1 | tee -a $OUT/summary.txt
import bench, json
r = bench.mid_size_record()
print('[$v] c3_tenth', {k: (round(v['value'], 1), v.get('launches_per_step')) for k, v in r.items() if isinstance(v, dict) and 'value' in v})
PY
done
echo done | tee -a $OUT/summary.txt
