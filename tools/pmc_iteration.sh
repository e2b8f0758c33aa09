#!/bin/bash
# HBM traffic of whole iterations: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py (config 3, bf16),
# per-kernel sums -> bytes per iteration.  tools/pmc_iteration.sh <out-name>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
STEPS=${PMC_STEPS:-2}
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c -d "$OUT/$c" -o pmc -- python "$OLDPWD/bench.py" --steps $STEPS --warmup 1 --no-cpu-baseline --no-engines --no-workloads --no-pmc --sustained-steps 0 ${PMC_ARGS:-} ) > "$OUT/$c.log" 2>&1
  echo "pmc $c exit $?"
done
python tools/pmc_iteration.py "$OUT" $((STEPS + 1)) | tee "$OUT/traffic.txt"
