#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1; tail -3 gpurun_out/gpu_tests.log
timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 10 > gpurun_out/b1.json 2> gpurun_out/b1.err || { tail gpurun_out/b1.err; exit 1; }
cut -c1-300 gpurun_out/b1.json
export RT_BENCH_BACKEND=gloo RT_BENCH_CHECK=1
for n in 2 4; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 3 --warmup 1 > gpurun_out/bench_gloo$n.json 2> gpurun_out/bench_gloo$n.err || { tail -20 gpurun_out/bench_gloo$n.err; exit 1; }
grep "partition check" gpurun_out/bench_gloo$n.err; grep "^{" gpurun_out/bench_gloo$n.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['partition'], d['config'].get('load_balance_calibration'), d['roofline']['frac'])"
done
