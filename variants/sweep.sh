#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 120 python -u tools/prof_frame.py final 3840 2160 4 4 2>&1 | grep "no-stats" | tail -3; }
run WORLD=1
run WORLD=2
run WORLD=2 RT_WF_REGION=64
run WORLD=4
run WORLD=4 RT_WF_REGION=128
