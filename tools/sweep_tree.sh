#!/bin/bash
# Leaf size / intersection cost of the SAH refinement against frame time: tools/sweep_tree.sh [lib.so]
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
lib=${1:-raytracing-project_amd/csrc/librt_hip.so}
for leaf in 2 4 6 8; do for ci in 1.5 3 5; do
  echo "== leaf $leaf ci $ci"
  RT_KD_LEAF=$leaf RT_KD_CI=$ci RT_HIP_LIB=$lib timeout -k 10 120 python tools/prof_frame.py final 3840 2160 4 4 | grep "no-stats\|tests/ray"
done; done
