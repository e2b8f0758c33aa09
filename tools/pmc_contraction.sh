#!/bin/bash
# PMC passes over the stand-alone bf16 contraction (tools/bench_gemm_bf16.py): SQ busy / wait / MFMA counters, TCP stall,
# TCC hit / miss, and HBM traffic (FETCH_SIZE, WRITE_SIZE in separate passes).  tools/pmc_contraction.sh <out-name>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/tools/bench_gemm_bf16.py --shapes ${PMC_SHAPES:-P12,Q23} --tiles 256 --reps 3 ${PMC_EXTRA:-}"
pass() { name=$1; shift; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o pmc -- $CMD ) > "$OUT/$name.log" 2>&1; echo "pmc $name exit $?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass tcp TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python $PWD/tools/pmc_table.py "$OUT" > "$OUT/pmc_table.txt" 2>&1
cat "$OUT/pmc_table.txt"
