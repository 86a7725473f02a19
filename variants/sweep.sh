#!/bin/bash
cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -5 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
timeout -k 10 300 python -u bench.py > gpurun_out/bench_v6.json 2> gpurun_out/bench_v6.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_v6 -o v6 --output-format csv -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_v6_prof.json 2>gpurun_out/prof_v6.err || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_v6_fetch -o p --output-format csv -- python3 tools/prof_frame.py final 3840 2160 4 1 > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_v6_write -o p --output-format csv -- python3 tools/prof_frame.py final 3840 2160 4 1 > /dev/null 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU -d gpurun_out/pmc_v6_sq -o p --output-format csv -- python3 tools/prof_frame.py final 3840 2160 4 1 > /dev/null 2>&1 || exit 1
timeout -k 10 200 python -u tools/gpu_quick.py 2>/dev/null
echo all done
