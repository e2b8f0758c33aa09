#!/bin/bash
# Round 6: PMC passes over the blocked-list probe (LDS conflicts, MFMA busy, waits).  tools/r06_blk_pmc.sh <out> <arrange>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/${1:-r6blkpmc}; ARR=${2:-1}; mkdir -p $OUT
export TMPDIR=/tmp
CMD="$PWD/tools/probe/blk_probe 40000 100000 2000 $ARR 8"
pass() { name=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o pmc -- $CMD ) > "$OUT/$name.log" 2>&1; echo "pmc $name exit $?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python $PWD/tools/pmc_table.py "$OUT" _kernel > "$OUT/pmc_table.txt" 2>&1
grep -A30 "blk_pass\|srp_bf16_v6" "$OUT/pmc_table.txt" | cut -c1-150
rm -rf $OUT/sq1 $OUT/sq2 $OUT/tcc
