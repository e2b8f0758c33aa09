#!/bin/bash
# A/B of library builds on the stand-alone contraction: tools/probe_ab.sh <out-name> <variant>...
# (variant = name of scikit-fusion_amd/lib/libskf_<name>.so; PROBE_SHAPES / PROBE_SPLITS / PROBE_ENV select the runs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
SH=${PROBE_SHAPES:-P12,Q12,P23,Q23}
for rep in 1 2; do
for v in "$@"; do
  env SKF_LIB_PATH=$PWD/scikit-fusion_amd/lib/libskf_$v.so python tools/bench_gemm_bf16.py --shapes $SH --tiles 256 --reps 10 --splits ${PROBE_SPLITS:-0} 2>&1 | grep -v -e Warning -e amdgpu.ids | sed "s/^/[$v] /" | tee -a "$OUT/ab.txt"
done
done
