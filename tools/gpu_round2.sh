#!/bin/bash
# Round-2 GPU session: tools/gpu_round2.sh <out-name> <step>...   outputs -> gpurun_out/<out-name>/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
for step in "$@"; do
case "$step" in
  smoke)
    ( time python __graft_entry__.py --smoke ) > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"; tail -2 "$OUT/smoke.log" | tee -a "$OUT/summary.txt" ;;
  tests)
    ( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 ) > "$OUT/pytest.log" 2>&1
    echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -25 "$OUT/pytest.log" | tee -a "$OUT/summary.txt" ;;
  tests_all)
    ( time timeout 1500 python -m pytest tests -m gpu -q --durations=10 ) > "$OUT/pytest.log" 2>&1
    echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -40 "$OUT/pytest.log" | tee -a "$OUT/summary.txt" ;;
  tnbench)
    for a in "" "--tn"; do python tools/bench_gemm_bf16.py --shapes Q12,Q13,Q23 --tiles 256 --reps 10 $a 2>&1 | grep -v -e Warning -e amdgpu.ids | tee -a "$OUT/tnbench.txt"; done ;;
  bench_*)
    dt=${step#bench_}
    ( time timeout 900 python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline ) > "$OUT/bench_$dt.log" 2>&1
    echo "bench $dt exit $?" | tee -a "$OUT/summary.txt"; grep '^{' "$OUT/bench_$dt.log" | cut -c1-600 | tee -a "$OUT/summary.txt" ;;
  dist_smoke)
    # `bench.py --gpus 2` launches itself (torch.distributed.run, one rank per GPU); on this one-GPU box the two
    # ranks share GPU 0 over a gloo group: all three multi-GPU modes end to end
    for mode in restarts relations rows owned; do
      ( SKF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.2 --mode $mode --no-cpu-baseline --no-engines ) > "$OUT/dist_$mode.log" 2>&1
      echo "dist $mode exit $?" | tee -a "$OUT/summary.txt"; grep '^{' "$OUT/dist_$mode.log" | cut -c1-260 | tee -a "$OUT/summary.txt"
    done
    # (round 5) the default mode also carries the `strong` sub-record: one fit sharded by ownership over the same two ranks
    grep '^{' "$OUT/dist_restarts.log" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist restarts: strong sub-record', json.dumps(d.get('strong')))" | tee -a "$OUT/summary.txt"
    # (round 6) a rank lost INSIDE the strong leg: the measured restarts line must still come out, exactly once -- rank 1 killed:
    # the launcher's TERM reaches rank 0's handler; rank 0 killed: the forked watchdog prints what rank 0 held
    for dead in 1 0; do
      ( SKF_BENCH_DIE_IN_STRONG=$dead SKF_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.2 --no-cpu-baseline --no-engines ) > "$OUT/dist_dead$dead.log" 2>&1
      echo "dist restarts, rank $dead killed in the strong leg: exit $?, JSON lines $(grep -c '^{' "$OUT/dist_dead$dead.log")" | tee -a "$OUT/summary.txt"
      grep '^{' "$OUT/dist_dead$dead.log" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  value', round(d['value'],1), 'n_gpus', d['n_gpus'], 'strong', d.get('strong'), 'interrupted', d.get('interrupted'))" | tee -a "$OUT/summary.txt"
    done
    for mode in rows owned; do
      ( SKF_BENCH_BACKEND=gloo timeout 600 python bench.py --workload c5 --gpus 2 --steps 2 --warmup 1 --scale 0.1 --mode $mode --no-cpu-baseline ) > "$OUT/dist_c5_$mode.log" 2>&1
      echo "dist c5 $mode exit $?" | tee -a "$OUT/summary.txt"; grep '^{' "$OUT/dist_c5_$mode.log" | cut -c1-260 | tee -a "$OUT/summary.txt"
    done ;;
  emulate)
    # per-rank compute of the ownership-sharded fit (null communicator), config 3 at 2 / 4 / 8 ranks and config 5 at 8
    for w in 2 4 8; do
      ( timeout 300 python bench.py --emulate-rank all/$w --steps 20 --warmup 3 ) > "$OUT/emulate_all$w.log" 2>&1
      echo "emulate all/$w exit $?" | tee -a "$OUT/summary.txt"
    done
    ( timeout 300 python bench.py --emulate-rank all/8 --steps 20 --warmup 3 --workload c5 ) > "$OUT/emulate_c5.log" 2>&1
    echo "emulate c5 exit $?" | tee -a "$OUT/summary.txt" ;;
  foldin)
    ( timeout 600 python tools/bench_transform.py ) > "$OUT/foldin_scale.txt" 2>&1
    echo "foldin exit $?" | tee -a "$OUT/summary.txt"; grep fold-in "$OUT/foldin_scale.txt" | cut -c1-200 | tee -a "$OUT/summary.txt" ;;
  fullbench)
    ( time timeout 1200 python bench.py ) > "$OUT/bench_full.log" 2>&1
    echo "bench exit $?" | tee -a "$OUT/summary.txt"; grep '^{' "$OUT/bench_full.log" | tee -a "$OUT/summary.txt" ;;
  die_sub)
    # (round 6) the default run killed right after its headline is measured: the forked watchdog prints the line, once
    ( SKF_BENCH_DIE_IN_SUBRECORDS=1 timeout 600 python bench.py ) > "$OUT/die_sub.log" 2>&1
    echo "bench killed during the sub-records: exit $?, JSON lines $(grep -c '^{' "$OUT/die_sub.log")" | tee -a "$OUT/summary.txt"
    grep '^{' "$OUT/die_sub.log" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  value', round(d['value'],2), 'frac', round(d['roofline']['frac'],3), '|', d.get('interrupted'))" | tee -a "$OUT/summary.txt" ;;
  c5_*)
    dt=${step#c5_}
    ( time timeout 900 python bench.py --workload c5 --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline ) > "$OUT/c5_$dt.log" 2>&1
    echo "c5 $dt exit $?" | tee -a "$OUT/summary.txt"; grep '^{' "$OUT/c5_$dt.log" | cut -c1-900 | tee -a "$OUT/summary.txt" ;;
  prof_*)
    dt=${step#prof_}
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_$dt" -o prof -- python "$OLDPWD/bench.py" --dtype $dt --steps 5 --warmup 2 --no-cpu-baseline --no-engines --no-workloads ) > "$OUT/prof_$dt.log" 2>&1
    echo "rocprof $dt exit $?" | tee -a "$OUT/summary.txt"
    f=$(find "$OUT/prof_$dt" -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && head -30 "$f" | tee -a "$OUT/summary.txt"
    find "$OUT/prof_$dt" -name '*kernel_trace.csv' -size +20M -delete; find "$OUT/prof_$dt" -name '*.db' -size +20M -delete ;;
  c5prof_*)
    dt=${step#c5prof_}
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/c5prof_$dt" -o prof -- python "$OLDPWD/bench.py" --workload c5 --dtype $dt --steps 3 --warmup 1 --no-cpu-baseline ) > "$OUT/c5prof_$dt.log" 2>&1
    echo "rocprof c5 $dt exit $?" | tee -a "$OUT/summary.txt"
    f=$(find "$OUT/c5prof_$dt" -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && head -30 "$f" | tee -a "$OUT/summary.txt"
    find "$OUT/c5prof_$dt" -name '*kernel_trace.csv' -size +20M -delete; find "$OUT/c5prof_$dt" -name '*.db' -size +20M -delete ;;
esac
done
