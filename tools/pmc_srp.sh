#!/bin/bash
# PMC passes over the stand-alone list-pass probe (tools/probe/srp_probe): instruction issue, L2 hit / miss, L1 stalls of
# srp_bf16_kernel (round 3a) and srp_bf16_v6_kernel with the lists in 1 and in 8 parts.   tools/pmc_srp.sh <out-name>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$PWD/gpurun_out/$1; mkdir -p "$OUT"
export TMPDIR=/tmp
SHAPE="40000 100000 2000 128 bf16"      # config 5's column lists: 40k movies gather 25.6 MB of user rows
run() { tag=$1; shift; env "$@" bash -c "cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/$tag -o pmc -- $PWD/tools/probe/srp_probe $SHAPE" > "$OUT/$tag.log" 2>&1; echo "pmc $tag exit $?"; }
for grp in sq tcc tcp; do
  case $grp in
    sq)  PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE";;
    tcc) PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum";;
    tcp) PMC="TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY";;
  esac
  export PMC
  run tuned_p1_$grp SRP_TUNED=1 SRP_PARTS=1
  run v6_p1_$grp SRP_V6=1 SRP_PARTS=1
  run v6_p8_$grp SRP_V6=1 SRP_PARTS=8
done
for cfg in tuned_p1 v6_p1 v6_p8; do
  mkdir -p "$OUT/$cfg"; for grp in sq tcc tcp; do mv "$OUT/${cfg}_$grp" "$OUT/$cfg/$grp" 2>/dev/null; done
  echo "== $cfg"; python tools/pmc_table.py "$OUT/$cfg" srp_ 2>&1 | cut -c1-200
done > "$OUT/pmc_table.txt"
find "$OUT" -name '*.db' -delete; find "$OUT" -name '*.csv' -size +1M -delete
cat "$OUT/pmc_table.txt"
