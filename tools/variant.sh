#!/bin/bash
# Builds a variant of the HIP library into exp/ (git-ignored): tools/variant.sh <name> [extra hipcc flags]
set -e
name=$1; shift
mkdir -p exp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize "$@" -shared raytracing-project_amd/csrc/rt_api.hip -o exp/$name.so -ldl
echo exp/$name.so
