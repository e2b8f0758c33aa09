#!/bin/bash
# Bound-finding probe of the bf16 relation contraction on the GPU box (round 2):
#   tools/probe_bounds.sh <out-name>
# runs the stand-alone contraction (tools/bench_gemm_bf16.py) with the product build, with all-zero
# operands (DVFS / power probe), and with the probe builds libskf_nomfma / nodma_a / nodma_b / nodma
# (built beforehand in the container: tools/build_probe_libs.sh), plus the ds_read_b64_tr_b16 lane-map probe.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
./tools/probe/tr_probe > "$OUT/tr_probe.txt" 2>&1; tail -3 "$OUT/tr_probe.txt"
SH=${PROBE_SHAPES:-P12,Q12,P23,Q23}
run() { name=$1; shift; python tools/bench_gemm_bf16.py --shapes $SH --tiles 256 --reps 10 "$@" 2>&1 | grep -v Warning | sed "s/^/[$name] /" | tee -a "$OUT/bounds.txt"; }
unset SKF_LIB_PATH; run base; run base --zero
for v in nomfma nodma_a nodma_b nodma; do
  if [ -f scikit-fusion_amd/lib/libskf_$v.so ]; then SKF_LIB_PATH=$PWD/scikit-fusion_amd/lib/libskf_$v.so run $v; fi
done
unset SKF_LIB_PATH; run base2
