#!/bin/bash
# Round 6: probe of the blocked known-entry passes (tools/probe/blk_probe.hip) on config 5's shapes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-r6blk}; mkdir -p $OUT
P=tools/probe/blk_probe
{
echo "## small"; timeout 120 $P 1000 3000 60 1 8
for arr in ${ARRS:-1 0 2}; do
  echo "## column lists (40000 movies over 100000 users, 2000 per list), arrange $arr"; timeout 300 $P 40000 100000 2000 $arr 8
  echo "## row lists (100000 users over 40000 movies, 800 per list), arrange $arr"; timeout 300 $P 100000 40000 800 $arr 8
done
} > $OUT/blk_probe.txt 2>&1
cat $OUT/blk_probe.txt
