#!/bin/bash
# Round 5, closing evidence session: tools/gpu_evidence.sh, then one iteration of the 1/10-scale graph as a timeline
# (after the step-per-launch pseudo-inverse).   gpurun -- 'bash tools/r05_final.sh <name>'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
N=${1:-r5ev3}; G=gpurun_out/$N
bash tools/gpu_evidence.sh $N
R=$PWD
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/$G/prof_tenth -o prof -- python $R/bench.py --scale 0.1 --steps 20 --warmup 3 --no-cpu-baseline --no-engines --no-workloads --no-pmc --sustained-steps 0 ) > $G/prof_tenth.log 2>&1
{ echo "# one iteration of config 3 at 1/10 linear scale, bf16 engine, after the step-per-launch pseudo-inverse (tools/timeline.py)"
  python tools/timeline.py $(find $G/prof_tenth -name "*.db" | head -1) 3; } > $G/c3_tenth_timeline_after.txt 2>&1
rm -rf $G/prof_tenth
tail -3 $G/c3_tenth_timeline_after.txt
