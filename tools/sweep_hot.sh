#!/bin/bash
# Which regions go first (RT_WF_HOT: multiples of the mean duration) against frame time: tools/sweep_hot.sh <lib.so> <factor>...
set -e
lib=$1; shift
for hot in "$@"; do
  echo "== RT_WF_HOT=$hot"
  for cfg in "ball 800 800 1" "redballs 1920 1080 1" "refractionballs 1920 1080 1" "fin 1920 1080 1" "lego 3840 2160 4" "final 3840 2160 4"; do
    RT_WF_HOT=$hot RT_HIP_LIB=$lib timeout -k 10 120 python tools/prof_frame.py $cfg 6 | grep "no-stats"
  done
  RT_WF_HOT=$hot RT_HIP_LIB=$lib WORLD=8 RANK=0 timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 6 | grep "no-stats"
done
