#!/bin/bash
# A/B timing of library builds: tools/ab.sh <lib.so>...  (whole 4K frame and an eighth of it)
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
for lib in "$@"; do
  echo "== $lib"
  RT_HIP_LIB=$lib timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 5 | grep "no-stats"
  RT_HIP_LIB=$lib WORLD=8 RANK=0 timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 5 | grep "no-stats"
  RT_HIP_LIB=$lib WORLD=8 RANK=3 timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 5 | grep "no-stats"
done
