#!/bin/bash
# Bytes between the L2s and the memory side of the fabric (FETCH_SIZE, WRITE_SIZE, raw, in KB per launch) of the frame
# launch of the headline config, for several library builds: tools/pmc_rw.sh <lib.so>...
# One counter per run; each GPU step runs under its own time limit; the script stops at the first failure.
set -e
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for lib in "$@"; do
  name=$(basename $lib .so)
  out=gpurun_out/pmc_rw_$name
  rm -rf $out && mkdir -p $out
  export RT_HIP_LIB=$lib
  echo "== $lib"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $out/p_$c -o pmc -- python3 tools/prof_frame.py final 3840 2160 4 2 > $out/$c.log 2>&1
    f=$(find $out/p_$c -name "*counter_collection.csv" | head -1)
    python3 - "$f" $c <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
v = [float(r["Counter_Value"]) for r in rows if "k_wf_pipeline<false" in r["Kernel_Name"] and r["Counter_Name"] == sys.argv[2]]
n = len(set(r["Dispatch_Id"] for r in rows if "k_wf_pipeline<false" in r["Kernel_Name"]))
print(sys.argv[2], "KB per launch:", sum(v) / max(n, 1), "launches", n)
P
    rm -rf $out/p_$c
  done
done
