#!/bin/bash
# One GPU-box session: build, smoke, parity tests, bench, rocprof.  Outputs -> gpurun_out/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== build + smoke" | tee "$OUT/summary.txt"
( time python __graft_entry__.py --smoke ) > "$OUT/smoke.log" 2>&1; echo "smoke exit $?" | tee -a "$OUT/summary.txt"
tail -3 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
for step in "$@"; do
case "$step" in
  tests)
    echo "== pytest -m gpu" | tee -a "$OUT/summary.txt"
    ( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=10 ) > "$OUT/pytest.log" 2>&1
    echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -25 "$OUT/pytest.log" | tee -a "$OUT/summary.txt" ;;
  tests_all)
    echo "== pytest -m gpu (no -x)" | tee -a "$OUT/summary.txt"
    ( time timeout 1500 python -m pytest tests -m gpu -q --durations=10 ) > "$OUT/pytest.log" 2>&1
    echo "pytest exit $?" | tee -a "$OUT/summary.txt"; tail -40 "$OUT/pytest.log" | tee -a "$OUT/summary.txt" ;;
  bench_*)
    dt=${step#bench_}
    echo "== bench $dt" | tee -a "$OUT/summary.txt"
    ( time timeout 900 python bench.py --dtype $dt --steps 10 --warmup 3 ) > "$OUT/bench_$dt.log" 2>&1
    echo "bench exit $?" | tee -a "$OUT/summary.txt"; tail -4 "$OUT/bench_$dt.log" | tee -a "$OUT/summary.txt" ;;
  c5_*)
    dt=${step#c5_}
    echo "== bench c5 (Dfmc, MovieLens-style) $dt" | tee -a "$OUT/summary.txt"
    ( time timeout 900 python bench.py --workload c5 --dtype $dt --steps 5 --warmup 2 ) > "$OUT/c5_$dt.log" 2>&1
    echo "c5 exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/c5_$dt.log" | cut -c1-1500 | tee -a "$OUT/summary.txt" ;;
  c5prof_*)
    dt=${step#c5prof_}
    echo "== rocprofv3 kernel-trace c5 $dt" | tee -a "$OUT/summary.txt"
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/c5prof_$dt" -o prof -- python "$OLDPWD/bench.py" --workload c5 --dtype $dt --steps 3 --warmup 1 ) > "$OUT/c5prof_$dt.log" 2>&1
    echo "rocprof exit $?" | tee -a "$OUT/summary.txt"
    find "$OUT/c5prof_$dt" -name '*kernel_trace.csv' -size +20M -delete ;;
  c5dist)
    echo "== torchrun x2 (gloo, shared GPU) c5 rows / relations" | tee -a "$OUT/summary.txt"
    for mode in relations rows; do
      ( SKF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --workload c5 --gpus 2 --steps 2 --warmup 1 --scale 0.1 --mode $mode ) > "$OUT/c5dist_$mode.log" 2>&1
      echo "c5 dist $mode exit $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/c5dist_$mode.log" | cut -c1-900 | tee -a "$OUT/summary.txt"
    done ;;
  dist_smoke)
    # the torchrun paths of bench.py with 2 ranks sharing the one GPU of this box (gloo group)
    echo "== torchrun x2 (gloo, shared GPU) restarts / relations / rows" | tee -a "$OUT/summary.txt"
    for mode in restarts relations rows; do
      ( SKF_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.2 --mode $mode --no-cpu-baseline ) > "$OUT/dist_$mode.log" 2>&1
      echo "dist $mode exit $?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/dist_$mode.log" | cut -c1-700 | tee -a "$OUT/summary.txt"
    done ;;
  prof_*)
    dt=${step#prof_}
    echo "== rocprofv3 kernel-trace $dt" | tee -a "$OUT/summary.txt"
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_$dt" -o prof -- python "$OLDPWD/bench.py" --dtype $dt --steps 5 --warmup 2 --no-cpu-baseline ) > "$OUT/prof_$dt.log" 2>&1
    echo "rocprof exit $?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/prof_$dt.log" | tee -a "$OUT/summary.txt"
    f=$(find "$OUT/prof_$dt" -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && head -25 "$f" | tee -a "$OUT/summary.txt"
    # keep only the small summaries (traces can be large)
    find "$OUT/prof_$dt" -name '*kernel_trace.csv' -size +20M -delete ;;
  sidetile)
    # A/B of the side-update tile shape: rocprof kernel stats with 64 x 64 tiles forced
    echo "== rocprofv3 kernel-trace bf16, SKF_SIDE_BK=32" | tee -a "$OUT/summary.txt"
    ( cd /tmp && SKF_SIDE_BK=32 timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$OUT/prof_side64" -o prof -- python "$OLDPWD/bench.py" --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline ) > "$OUT/prof_side64.log" 2>&1
    echo "rocprof exit $?" | tee -a "$OUT/summary.txt"; grep '^{' "$OUT/prof_side64.log" | cut -c1-300 | tee -a "$OUT/summary.txt"
    find "$OUT/prof_side64" -name '*kernel_trace.csv' -size +20M -delete ;;
  pmc_*)
    dt=${step#pmc_}
    echo "== rocprofv3 pmc $dt" | tee -a "$OUT/summary.txt"
    ( cd /tmp && timeout 900 rocprofv3 --pmc FETCH_SIZE -d "$OLDPWD/$OUT/pmc_fetch_$dt" -o pmc -- python "$OLDPWD/bench.py" --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline ) > "$OUT/pmc_fetch_$dt.log" 2>&1
    echo "pmc fetch exit $?" | tee -a "$OUT/summary.txt"
    ( cd /tmp && timeout 900 rocprofv3 --pmc WRITE_SIZE -d "$OLDPWD/$OUT/pmc_write_$dt" -o pmc -- python "$OLDPWD/bench.py" --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline ) > "$OUT/pmc_write_$dt.log" 2>&1
    echo "pmc write exit $?" | tee -a "$OUT/summary.txt"
    python tools/pmc_summary.py "$OUT/pmc_fetch_$dt/pmc_results.db" "$OUT/pmc_write_$dt/pmc_results.db" 2>&1 | tee -a "$OUT/summary.txt"
    find "$OUT" -name '*counter_collection.csv' -size +20M -delete ;;
esac
done
rocm-smi --showmeminfo vram 2>/dev/null | head -5 >> "$OUT/summary.txt"
nproc >> "$OUT/summary.txt"
