#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
run() { echo "== $*"; env "$@" timeout -k 10 120 python -u tools/prof_frame.py final 3840 2160 4 4 2>&1 | grep "no-stats\|wavefront:" | tail -2; }
run RT_KD_STATS=1
run RT_KD_STATS=1 RT_WF_STACK_LDS=8
run RT_KD_STATS=1 RT_WF_STACK_LDS=16
run WORLD=8
