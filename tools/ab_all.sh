#!/bin/bash
# A/B timing of library builds over the bench's configs: tools/ab_all.sh <lib.so>...
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
for lib in "$@"; do
  echo "== $lib"
  for cfg in "ball 800 800 1" "redballs 1920 1080 1" "refractionballs 1920 1080 1" "fin 1920 1080 1" "lego 3840 2160 4" "final 3840 2160 4"; do
    echo -n "$cfg: "
    RT_HIP_LIB=$lib timeout -k 10 120 python tools/prof_frame.py $cfg 6 | grep "no-stats"
  done
  echo -n "final, an eighth: "
  RT_HIP_LIB=$lib WORLD=8 RANK=0 timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 6 | grep "no-stats"
done
