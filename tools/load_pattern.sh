#!/bin/bash
# tools/load_pattern.sh <round-tag>: the roof of the lane-per-trial smoothers' access pattern (tools/ubench/load_pattern.hip) -> gpurun_out/<tag>_load_pattern.txt
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_load_pattern.txt
: > $OUT
for full in 0 1; do
  for ahead in 1 3; do
    for nf in 0 150 300; do
      timeout -k 10 120 tools/ubench/load_pattern 262144 512 $nf $ahead $full >> $OUT 2>&1 || { echo "load_pattern failed" >> $OUT; exit 1; }
    done
  done
done
cat $OUT
