#!/bin/bash
# Issue cost of vector instructions on a SIMD of this GPU (1, 2 and 7 waves resident): builds and runs the
# micro-benchmarks next to this script.  On the GPU box: tools/mb/run.sh > gpurun_out/valu_issue.txt
set -e
here=$(dirname "$0")
mkdir -p exp
hipcc --offload-arch=gfx950 -O3 $here/valu_issue.hip -o exp/valu_issue 2> /dev/null
hipcc --offload-arch=gfx950 -O3 $here/valu_issue_packed.hip -o exp/valu_issue_packed 2> /dev/null
timeout -k 10 120 exp/valu_issue
timeout -k 10 120 exp/valu_issue_packed
hipcc --offload-arch=gfx950 -O3 $here/salu_issue.hip -o exp/salu_issue 2> /dev/null
timeout -k 10 120 exp/salu_issue
