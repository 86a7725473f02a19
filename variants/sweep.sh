#!/bin/bash
cd /root/repo && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; tail -2 gpurun_out/gpu_tests.log
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU -d gpurun_out/pmc6a -o p --output-format csv -- python3 tools/prof_frame.py final 3840 2160 4 2 > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES -d gpurun_out/pmc6b -o p --output-format csv -- python3 tools/prof_frame.py final 3840 2160 4 2 > /dev/null 2>&1 || exit 1
echo pmc done
