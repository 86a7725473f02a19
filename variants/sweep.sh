#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
run() { echo "== $*"; env "$@" timeout -k 10 120 python -u tools/prof_frame.py final 3840 2160 4 4 2>&1 | grep "no-stats\|wavefront:" | tail -2; }
for b in 128 512; do
run RT_KD_STATS=1 WORLD=1 RT_HIP_LIB=/root/repo/variants/librt_b$b.so
run WORLD=8 RT_HIP_LIB=/root/repo/variants/librt_b$b.so
run WORLD=1 RT_WF_REGION=64 RT_HIP_LIB=/root/repo/variants/librt_b$b.so
done
