#!/bin/bash
# Round 6: why does a hipGraph replay of the mid-size iteration run at a quarter of the eager rate?  A/B at 1/10 scale:
# eager two streams, eager one stream, graph two streams, graph one stream; kernel traces of the two graph forms.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6graph}; mkdir -p $OUT
export TMPDIR=/tmp
{
for cfg in "eager-two-streams:" "eager-one-stream:SKF_NO_OVERLAP=1" "graph-two-streams:SKF_GRAPH=1" "graph-one-stream:SKF_GRAPH=1 SKF_NO_OVERLAP=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  echo "## $name ($envs)"
  env $envs python tools/bench_midsize.py 0.1 2>&1 | grep scale
done
} > $OUT/graph_ab.txt 2>&1
cat $OUT/graph_ab.txt
R=$PWD
for cfg in "graph2:SKF_GRAPH=1" "graph1:SKF_GRAPH=1 SKF_NO_OVERLAP=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  ( cd /tmp && env $envs timeout 300 rocprofv3 --kernel-trace -d $R/$OUT/prof_$name -o prof -- python $R/tools/bench_midsize.py 0.1 ) > $OUT/prof_$name.log 2>&1
  { echo "# one iteration of the replay ($name) as a timeline"; python tools/timeline.py $(find $OUT/prof_$name -name "*.db" | head -1) 3; } > $OUT/timeline_$name.txt 2>&1
  rm -rf $OUT/prof_$name
  tail -3 $OUT/timeline_$name.txt
done
