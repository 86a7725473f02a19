#!/bin/bash
# SQ counters of two frames of the headline config for several library builds: tools/pmc_ab.sh <lib.so>...
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for lib in "$@"; do
  name=$(basename $lib .so)
  out=gpurun_out/pmc_ab_$name
  rm -rf $out && mkdir -p $out
  export RT_HIP_LIB=$lib
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -o pmc -- python3 tools/prof_frame.py final 3840 2160 4 2 > $out/pmc_sq.log 2>&1
  cp $(find $out/pmc_sq -name "*counter_collection.csv" | head -1) $out/pmc_sq.csv
  rm -rf $out/pmc_sq
  echo "== $lib"
  python3 tools/pmc_digest.py $out | grep "SQ_INSTS\|lanes_active\|SQ_ACTIVE_INST_VALU\|SQ_WAVE_CYCLES"
done
