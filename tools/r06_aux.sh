#!/bin/bash
# Round 6: cache-policy bits of the two LDS-DMA streams of gemm_bf16_v2_kernel (aux operand of global_load_lds: 1 = sc0, 2 = nt,
# 16 = sc1; product: relation stream nt, G^T stream default), stand-alone contractions, variants alternating:
#   tools/r06_aux.sh <out> <variant libs...>        (tools/build_probe_libs.sh a18:-DSKF_A_AUX=18 ...)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=$1; shift
OUT=gpurun_out/$N; mkdir -p "$OUT"
run() { name=$1; shift; python tools/bench_gemm_bf16.py --shapes P12,P23,Q23,Q12 --tiles 256 --reps 10 2>&1 | grep -v -e Warning -e amdgpu.ids | sed "s/^/[$name] /" | tee -a "$OUT/standalone.txt"; }
for rep in $(seq 1 ${AUX_REPS:-2}); do
  unset SKF_LIB_PATH; run base
  for v in "$@"; do export SKF_LIB_PATH=$PWD/scikit-fusion_amd/lib/libskf_$v.so; run $v; done
done
