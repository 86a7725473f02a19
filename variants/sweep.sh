#!/bin/bash
cd /root/repo && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 120 python -u tools/prof_frame.py final 3840 2160 4 4 2>&1 | grep -v loaded | tail -2
export RT_WF_MULTILAUNCH=1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d gpurun_out/pmc7 -o p --output-format csv -- python3 tools/prof_frame.py final 3840 2160 4 1 > /dev/null 2>&1 || exit 1
echo pmc done
