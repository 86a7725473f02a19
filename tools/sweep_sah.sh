#!/bin/bash
# Leaf size and intersection cost of the SAH refinement over the bench's configs: tools/sweep_sah.sh "<leaf>:<ci>"...
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
for lc in "$@"; do
  leaf=${lc%%:*}; ci=${lc##*:}
  echo "== leaf $leaf ci $ci"
  for cfg in "ball 800 800 1" "redballs 1920 1080 1" "refractionballs 1920 1080 1" "fin 1920 1080 1" "lego 3840 2160 4" "final 3840 2160 4"; do
    RT_KD_LEAF=$leaf RT_KD_CI=$ci timeout -k 10 120 python tools/prof_frame.py $cfg 6 | grep "no-stats"
  done
  RT_KD_LEAF=$leaf RT_KD_CI=$ci WORLD=8 RANK=0 timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 6 | grep "no-stats"
done
