#!/bin/bash
# Round 5, GPU session 3: the blocked sweep with the pivot block swept by one wave, and one launch per block step with the
# update spread over row slabs (sweep_step_kernel) against one workgroup per matrix.   tools/r05_session3.sh <out-name>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5s3b}; OUT=gpurun_out/$N; mkdir -p $OUT
export TMPDIR=/tmp
ORDERS="65 96 128 160 192 224 256"
for v in "SKF_SWEEP_STEP_MIN=0" "SKF_SWEEP_STEP_MIN=65 SKF_SWEEP_ROWS=32" "SKF_SWEEP_STEP_MIN=65 SKF_SWEEP_ROWS=64" "SKF_PINV_SWEEP=0"; do
  echo "== $v" | tee -a $OUT/summary.txt
  env $v timeout 300 python tools/bench_pinv.py $ORDERS 2>&1 | grep "full rank" | tee -a $OUT/summary.txt
done
( cd /tmp && SKF_SWEEP_STEP_MIN=65 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_step -- python $OLDPWD/tools/bench_pinv.py 256 ) > $OUT/prof_step.log 2>&1
( cd /tmp && SKF_SWEEP_STEP_MIN=0 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_one -- python $OLDPWD/tools/bench_pinv.py 256 ) > $OUT/prof_one.log 2>&1
for d in prof_step prof_one; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1)
  echo "== $d" | tee -a $OUT/summary.txt; [ -n "$f" ] && grep -E "sweep|eigh_pack|pchol|jacobi|unpack" $f | cut -d, -f1-6 | tee -a $OUT/summary.txt
done
timeout 900 python -m pytest tests -m gpu -x -q -k "pinv or sweep or mid or tenth" 2>&1 | tail -3 | tee -a $OUT/summary.txt
for v in "SKF_SWEEP_STEP_MIN=0" "SKF_NONE=1" "SKF_SWEEP_STEP_MIN=0" "SKF_NONE=1"; do
  env $v timeout 600 python - <<PY 2>&1 | tail -1 | tee -a $OUT/summary.txt
import bench, json
r = bench.mid_size_record()
print('[$v] c3_tenth', {k: (round(v['value'], 1), v.get('launches_per_step')) for k, v in r.items() if isinstance(v, dict) and 'value' in v})
PY
done
for v in "SKF_SWEEP_STEP_MIN=0" "SKF_NONE=1"; do
  for wl in c3 c5; do
    env $v timeout 300 python bench.py --emulate-rank 3/8 --steps 30 --warmup 3 --workload $wl > $OUT/emu_$wl.log 2>&1
    grep '^{' $OUT/emu_$wl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['ranks'][0]; print('[$v] $wl rank 3/8:', round(r['compute_ms_per_step'],3), 'ms,', r['launches_per_step'], 'launches')" | tee -a $OUT/summary.txt
  done
  env $v timeout 300 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline --no-engines --no-workloads --no-pmc --sustained-steps 0 > $OUT/c5.log 2>&1
  grep '^{' $OUT/c5.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$v] c5', round(d['value'],2), 'it/s')" | tee -a $OUT/summary.txt
done
echo done | tee -a $OUT/summary.txt
