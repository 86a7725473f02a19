#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
export RT_BENCH_BACKEND=gloo RT_BENCH_CHECK=1
for n in 2 4; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $n --steps 5 --warmup 2 > gpurun_out/bench_gloo$n.json 2> gpurun_out/bench_gloo$n.err || { tail -20 gpurun_out/bench_gloo$n.err; exit 1; }
grep "partition check" gpurun_out/bench_gloo$n.err; cut -c1-400 gpurun_out/bench_gloo$n.json
done
