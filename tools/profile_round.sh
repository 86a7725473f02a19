#!/bin/bash
# Profiles of a round, on the GPU box: tools/profile_round.sh <tag> [passes]   (writes gpurun_out/prof_<tag>/)
#   passes: any of "stats sq fetch write tcp tcc ta dram" (default: all)
#   1. rocprofv3 --kernel-trace --stats around the default bench command (no CPU baseline, no extras)
#   2. PMC passes around two frames of the headline config, one counter group per run
# Every GPU step has its own time limit and the script stops at the first failure.
set -e
tag=${1:-run}
passes=${2:-stats sq fetch write tcp tcc ta dram}
want() { [[ " $passes " == *" $1 "* ]]; }
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if want stats; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench.json 2> $out/bench.err
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats.csv
rm -rf $out/stats
fi
pmc() { # name, counters...
  local name=$1; shift
  want $name || return 0
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -o pmc -- python3 tools/prof_frame.py final 3840 2160 4 2 > $out/pmc_$name.log 2>&1
  cp $(find $out/pmc_$name -name "*counter_collection.csv" | head -1) $out/pmc_$name.csv
  rm -rf $out/pmc_$name
}
pmc sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
python3 tools/pmc_digest.py $out > $out/digest.json
# requests the L2 sends on towards DRAM, as opposed to all fabric requests (FETCH_SIZE / WRITE_SIZE count both)
pmc dram TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum
python3 tools/pmc_digest.py $out > $out/digest.json
cat $out/digest.json
