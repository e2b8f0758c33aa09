#!/bin/bash
# A/B of environment settings on bench.py (config 3, bf16), settings alternating: tools/ab_env.sh <out> <reps> "VAR=1" "" ...
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
REPS=$1; shift
mkdir -p "$OUT"
for rep in $(seq 1 $REPS); do
  i=0
  for v in "$@"; do
    i=$((i+1))
    env $v python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-engines ${AB_ARGS:-} > "$OUT/v$i.$rep.log" 2>&1
    grep '^{' "$OUT/v$i.$rep.log" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$v] rep$rep', round(d['value'],2), 'it/s', round(d['ms_per_step'],3), 'ms; contractions', round(d['roofline']['avg_launch_ms']*d['roofline']['launches_per_iter'],3), 'ms/iter')" | tee -a "$OUT/ab.txt"
  done
done
