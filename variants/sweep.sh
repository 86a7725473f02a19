#!/bin/bash
# scratch: parameter sweep on the GPU box
cd /root/repo
run() { echo "== $*"; env "$@" timeout -k 10 120 python -u tools/prof_frame.py final 3840 2160 4 4 2>/dev/null | tail -1; }
run X=base
for w in 5 6 8; do run RT_HIP_LIB=/root/repo/variants/librt_w$w.so; done
for leaf in 2 3 6 8; do run RT_KD_LEAF=$leaf; done
for ci in 1.5 2 4 6; do run RT_KD_CI=$ci; done
run RT_KD_LEAF=6 RT_KD_CI=2
run RT_KD_LEAF=3 RT_KD_CI=4
