#!/bin/bash
# Round 6: the Gram products of all types in one grouped launch (default) against one launch + reduce per type
# (scikit-fusion_amd/lib/libskf_before.so = the tree before the change; same-box, alternating):   tools/r06_gram_group.sh <out>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-gramgrp}; OUT=gpurun_out/$N; mkdir -p "$OUT"
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "gram_products_of_all_types or round5_schedule or two_runs_give" 2>&1 | tail -3 | tee "$OUT/tests.txt"
for rep in 1 2; do
  for v in base before; do
    if [ "$v" = base ]; then unset SKF_LIB_PATH; else export SKF_LIB_PATH=$PWD/scikit-fusion_amd/lib/libskf_$v.so; fi
    echo "[$v] mid size:" | tee -a "$OUT/mid.txt"; python tools/bench_midsize.py 0.1 0.2 2>&1 | grep scale | tee -a "$OUT/mid.txt"
    python bench.py --emulate-rank 3/8 --steps 20 --warmup 3 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['ranks'][0]; print('[$v] rank 3 of 8, config 3: %.3f ms, %d launches' % (r['compute_ms_per_step'], r['launches_per_step']))" | tee -a "$OUT/mid.txt"
  done
done
unset SKF_LIB_PATH
bash tools/ab_libs.sh ${N}_ab 3 base before 2>&1 | tee "$OUT/ab.txt"
