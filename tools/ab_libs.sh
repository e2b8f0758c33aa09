#!/bin/bash
# A/B of experimental builds of the library on the GPU box: tools/ab_libs.sh <out-name> <reps> <variant>...
# ("base" = the in-tree build; other names = scikit-fusion_amd/lib/libskf_<name>.so); prints it/s and
# the contraction TFLOP/s of bench.py (config 3, bf16) for every run, variants alternating.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
REPS=$1; shift
mkdir -p "$OUT"
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    if [ "$v" = base ]; then unset SKF_LIB_PATH; else export SKF_LIB_PATH=$PWD/scikit-fusion_amd/lib/libskf_$v.so; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline ${AB_ARGS:-} > "$OUT/$v.$rep.log" 2>&1
    grep '^{' "$OUT/$v.$rep.log" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v rep$rep', round(d['value'],2), 'it/s', round(d['roofline']['achieved'],1), 'TF')"
  done
done
