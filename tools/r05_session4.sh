#!/bin/bash
# Round 5, GPU session 4: the blocked sweep after the look-ahead pivot block and the up-front loads of a block step --
# operator times per order, phase stamps of a probe build, kernel durations.   tools/r05_session4.sh <out-name>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5s4}; OUT=gpurun_out/$N; mkdir -p $OUT
export TMPDIR=/tmp
ORDERS="65 96 128 160 192 224 256"
for v in "SKF_SWEEP_STEP_MIN=0" "SKF_SWEEP_STEP_MIN=65 SKF_SWEEP_ROWS=32" "SKF_SWEEP_STEP_MIN=65 SKF_SWEEP_ROWS=64"; do
  echo "== $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python tools/bench_pinv.py $ORDERS 2>&1 | grep "full rank" | tee -a $OUT/summary.txt
done
if [ -f tools/probe/_build/libskf_stamps.so ]; then
  for v in "SKF_SWEEP_STEP_MIN=0" "SKF_SWEEP_STEP_MIN=65"; do
    echo "== stamps $v" | tee -a $OUT/summary.txt
    env $v SKF_LIB_PATH=$PWD/tools/probe/_build/libskf_stamps.so timeout 300 python tools/bench_pinv.py 256 2>&1 | grep "sweep_" | sort | uniq -c | sort -rn | head -8 | tee -a $OUT/summary.txt
  done
fi
R=$PWD
for v in 0 65; do
  ( cd /tmp && SKF_SWEEP_STEP_MIN=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$v -- python $R/tools/bench_pinv.py 256 ) > $OUT/prof_$v.log 2>&1
  f=$(find $OUT/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "== kernel stats SKF_SWEEP_STEP_MIN=$v" | tee -a $OUT/summary.txt; [ -n "$f" ] && grep -E "sweep|eigh_pack|unpack" $f | cut -d, -f1-7 | tee -a $OUT/summary.txt
done
timeout 900 python -m pytest tests -m gpu -x -q -k "pinv or sweep" 2>&1 | tail -3 | tee -a $OUT/summary.txt
for v in "SKF_SWEEP_STEP_MIN=0" "SKF_NONE=1"; do
  env $v timeout 600 python - <<PY 2>&1 | tail -1 | tee -a $OUT/summary.txt
import bench, json
r = bench.mid_size_record()
print('[$v] c3_tenth', {k: (round(v['value'], 1), v.get('launches_per_step')) for k, v in r.items() if isinstance(v, dict) and 'value' in v})
PY
done
AB_ARGS="--no-workloads --no-pmc --sustained-steps 0 --steps 100" bash tools/ab_env.sh $N/ab 2 "SKF_SWEEP_STEP_MIN=0" "SKF_NONE=1" 2>&1 | tail -8 | tee -a $OUT/summary.txt
echo done | tee -a $OUT/summary.txt
