#!/bin/bash
# Probe: how the contraction (full / ingest-only / compute-only builds) scales with the number of K slices,
# i.e. with the number of busy CUs (P12 has 196 row tiles).  tools/probe_splits.sh <out-name>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
SH=${PROBE_SHAPES:-P12}
run() { name=$1; shift; python tools/bench_gemm_bf16.py --shapes $SH --tiles 256 --reps 10 "$@" 2>&1 | grep -v -e Warning -e amdgpu.ids | sed "s/^/[$name] /" | tee -a "$OUT/splits.txt"; }
unset SKF_LIB_PATH; run base --splits 1,2,3,4,5,8,13
for v in nomfma nodma; do
  SKF_LIB_PATH=$PWD/scikit-fusion_amd/lib/libskf_$v.so run $v --splits 1,2,3,4,5,8,13
done
