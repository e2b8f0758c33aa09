#!/bin/bash
# PMC passes over the stand-alone f32 relation contraction (tools/bench_gemm_f32.py --dtypes f32 --cases P12,Q12): where the
# wave cycles of gemm_mfma_kernel<float, ...> go.   tools/pmc_f32.sh <out-name>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/$1; shift
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/tools/bench_gemm_f32.py --dtypes f32 --cases ${PMC_CASES:-P12,Q12} --reps 2"
$CMD 2>&1 | grep -v amdgpu.ids | tee "$OUT/timing.txt"
pass() { name=$1; shift; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o pmc -- $CMD ) > "$OUT/$name.log" 2>&1; echo "pmc $name exit $?"; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU
for d in sq1 sq2; do db=$(find "$OUT/$d" -name '*.db' | head -1); [ -n "$db" ] && python $PWD/tools/pmc_summary.py "$db" | grep -E "gemm_mfma|^#|^kernel" ; done | tee "$OUT/pmc_f32.txt"
rm -rf "$OUT/sq1" "$OUT/sq2"
