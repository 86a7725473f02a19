#!/bin/bash
# One environment switch of the library over the bench's configs: tools/sweep_env.sh <lib.so> <VAR> <value>...
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
lib=$1; var=$2; shift 2
for v in "$@"; do
  echo "== $var=$v"
  for cfg in "ball 800 800 1" "redballs 1920 1080 1" "refractionballs 1920 1080 1" "fin 1920 1080 1" "lego 3840 2160 4" "final 3840 2160 4"; do
    env RT_HIP_LIB=$lib $var=$v timeout -k 10 120 python tools/prof_frame.py $cfg 6 | grep "no-stats"
  done
  env RT_HIP_LIB=$lib $var=$v WORLD=8 RANK=0 timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 6 | grep "no-stats"
done
