#!/bin/bash
# Probe: 32x32x16 vs 16x16x32 MFMA flavour in the 3-stage ring (full and compute-only builds), and the
# ingest-only build restricted to one of the two operand streams.  tools/probe_mf32.sh <out-name>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
SH=${PROBE_SHAPES:-P12,Q12,Q23}
run() { name=$1; shift; python tools/bench_gemm_bf16.py --shapes $SH --tiles 256 --reps 10 "$@" 2>&1 | grep -v -e Warning -e amdgpu.ids | sed "s/^/[$name] /" | tee -a "$OUT/mf32.txt"; }
L=$PWD/scikit-fusion_amd/lib
SKF_LIB_PATH=$L/libskf_base.so run base16 --splits 0,5
SKF_BF16_MFMA=32 SKF_LIB_PATH=$L/libskf_base.so run base32 --splits 0,5
SKF_LIB_PATH=$L/libskf_nodma.so run nodma16 --splits 0,5
SKF_BF16_MFMA=32 SKF_LIB_PATH=$L/libskf_nodma.so run nodma32 --splits 0,5
SKF_LIB_PATH=$L/libskf_ingest_a.so run ingest_a --splits 0,5
SKF_LIB_PATH=$L/libskf_ingest_b.so run ingest_b --splits 0,5
