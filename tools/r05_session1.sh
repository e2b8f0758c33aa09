#!/bin/bash
# Round 5, GPU session 1: suite on the new build, CU-mask bit layout, A/B of the round-5 schedule switches on config 3,
# rank-of-8 emulation with / without the masked contraction stream.   tools/r05_session1.sh <out-name>
# (Kept as the record of how profiles/r05_cu_mask_ab.txt was measured: SKF_MAIN_CU_DROP named the mask bits to clear on an
# internal stream that carried the iteration; the switch and the stream were removed after this session -- the runs below
# that set it now measure the plain schedule.)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5s1}; OUT=gpurun_out/$N; mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/summary.txt; tail -30 $OUT/pytest.log | tee -a $OUT/summary.txt
cp gpurun_out/test_deviations.txt $OUT/ 2>/dev/null
# --- which CUs does a mask leave?
EVERY32="31,63,95,127,159,191,223,255"
FIRST8="0-7"; LAST8="248-255"; EVERY8TH="0,32,64,96,128,160,192,224"
timeout 120 tools/probe/cumask_probe "" "$LAST8" "$FIRST8" "$EVERY32" "$EVERY8TH" "7,15,23,31,39,47,55,63" "0-15" 2>&1 | tee $OUT/cumask_probe.txt
# --- A/B on config 3 (bf16): every switch alone, together, and with three masks
AB_ARGS="--no-workloads --no-pmc --sustained-steps 0" bash tools/ab_env.sh $N/ab 2 \
   "SKF_GRAM_SYM=0 SKF_EARLY_UPDATE=0" "SKF_EARLY_UPDATE=0" "SKF_GRAM_SYM=0" "SKF_NONE=1" \
   "SKF_MAIN_CU_DROP=$LAST8" "SKF_MAIN_CU_DROP=$EVERY32" "SKF_MAIN_CU_DROP=$FIRST8" "SKF_MAIN_CU_DROP=$EVERY8TH" "SKF_MAIN_CU_DROP=0-15" 2>&1 | tail -20 | tee -a $OUT/summary.txt
# --- rank 3 of 8 (exchanges skipped): plain and masked, config 3 and config 5
for v in "SKF_NONE=1" "SKF_MAIN_CU_DROP=$LAST8" "SKF_MAIN_CU_DROP=$EVERY32" "SKF_MAIN_CU_DROP=$EVERY8TH" "SKF_MAIN_CU_DROP=0-15"; do
  for wl in c3 c5; do
    env $v timeout 300 python bench.py --emulate-rank 3/8 --steps 20 --warmup 3 --workload $wl > $OUT/emu_$wl.log 2>&1
    grep '^{' $OUT/emu_$wl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['ranks'][0]; print('[$v] $wl rank 3/8:', round(r['compute_ms_per_step'],3), 'ms,', r['launches_per_step'], 'launches')" | tee -a $OUT/summary.txt
  done
done
# --- timeline of rank 3 of 8, plain and with the best-looking mask (decided afterwards from the numbers: both kept)
for v in "SKF_NONE=1" "SKF_MAIN_CU_DROP=$EVERY32"; do
  tag=$(echo "$v" | tr -c 'A-Za-z0-9' '_')
  ( cd /tmp && env $v timeout 600 rocprofv3 --kernel-trace -d "$OLDPWD/$OUT/tl_$tag" -o prof -- python "$OLDPWD/bench.py" --emulate-rank 3/8 --steps 4 --warmup 2 ) > $OUT/tl_$tag.log 2>&1
  db=$(find $OUT/tl_$tag -name '*.db' | head -1)
  [ -n "$db" ] && python tools/timeline_owned.py $db 4 > $OUT/timeline_$tag.txt 2>&1
  rm -rf $OUT/tl_$tag
done
echo done | tee -a $OUT/summary.txt
