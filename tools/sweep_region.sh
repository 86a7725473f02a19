#!/bin/bash
# Region-size sweep of the pipeline launch: tools/sweep_region.sh <S>...  (whole 4K frame and an eighth of it)
set -e
for S in "$@"; do
  echo "== RT_WF_REGION=$S"
  RT_WF_REGION=$S timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 5 | grep "no-stats"
  RT_WF_REGION=$S WORLD=8 RANK=0 timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 5 | grep "no-stats"
done
