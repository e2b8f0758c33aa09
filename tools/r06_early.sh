#!/bin/bash
# Round 6: DMA issue schedule of gemm_bf16_v2_kernel<256,...> (SKF_V2_SCHED=1: every LDS-DMA piece in phase 1, A a half K tile
# earlier) against the product schedule: stand-alone contractions, the ingest-only builds, then the whole iteration.
#   tools/r06_early.sh <out-name> <variant libs...>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=$1; shift
OUT=gpurun_out/$N; mkdir -p "$OUT"
SH=${PROBE_SHAPES:-P12,P23,Q23,Q12}
run() { name=$1; shift; python tools/bench_gemm_bf16.py --shapes $SH --tiles 256 --reps 10 "$@" 2>&1 | grep -v Warning | sed "s/^/[$name] /" | tee -a "$OUT/standalone.txt"; }
for rep in 1 2; do
  unset SKF_LIB_PATH; run base
  for v in "$@"; do SKF_LIB_PATH=$PWD/scikit-fusion_amd/lib/libskf_$v.so run $v; done
done
unset SKF_LIB_PATH
VARS=""; for v in "$@"; do case $v in *nomfma*) ;; *) VARS="$VARS $v";; esac; done
bash tools/ab_libs.sh ${N}_ab 3 base $VARS 2>&1 | tee "$OUT/ab.txt"
