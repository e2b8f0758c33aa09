#!/bin/bash
# Round 5, GPU session 10: the independent c x c products of a relation's chain two to a launch (gemm_mfma_pair_kernel), the two
# B-sum casts of a type in one launch -- bits on the hardware, then rates.   tools/r05_session10.sh <out-name>
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5s10}; OUT=gpurun_out/$N; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -q --durations=10 ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | tee -a $OUT/summary.txt
for rep in 1 2; do
  for v in "SKF_CHAIN_PAIRS=0" "SKF_CHAIN_PAIRS=1"; do
    env $v timeout 300 python tools/bench_midsize.py 0.05 0.1 0.2 2>&1 | grep "^scale" | sed "s/^/[$v] /" | tee -a $OUT/summary.txt
  done
done
for v in "SKF_CHAIN_PAIRS=0" "SKF_CHAIN_PAIRS=1" "SKF_CHAIN_PAIRS=0" "SKF_CHAIN_PAIRS=1"; do
  for wl in c3 c5; do
    env $v timeout 300 python bench.py --emulate-rank 3/8 --steps 30 --warmup 3 --workload $wl > $OUT/emu_$wl.log 2>&1
    grep '^{' $OUT/emu_$wl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['ranks'][0]; print('[$v] $wl rank 3/8:', round(r['compute_ms_per_step'],3), 'ms,', r['launches_per_step'], 'launches')" | tee -a $OUT/summary.txt
  done
done
AB_ARGS="--no-workloads --no-pmc --sustained-steps 0 --steps 100" bash tools/ab_env.sh $N/ab 2 "SKF_CHAIN_PAIRS=0" "SKF_CHAIN_PAIRS=1" 2>&1 | tail -4 | tee -a $OUT/summary.txt
for v in "SKF_CHAIN_PAIRS=0" "SKF_CHAIN_PAIRS=1"; do
  env $v timeout 300 python bench.py --workload c5 --steps 20 --warmup 3 --no-cpu-baseline --no-engines --no-workloads --no-pmc --sustained-steps 0 > $OUT/c5.log 2>&1
  grep '^{' $OUT/c5.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('[$v] c5', round(d['value'],2), 'it/s')" | tee -a $OUT/summary.txt
done
echo done | tee -a $OUT/summary.txt
