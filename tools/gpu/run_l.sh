#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/l
timeout 900 python -m pytest tests/test_gpu_multiprocess.py -x -q -m gpu > gpurun_out/l/tests_mp.log 2>&1
timeout 600 python bench.py > gpurun_out/l/bench.json 2> gpurun_out/l/bench.err
GS_DIST_BACKEND=gloo GS_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 3 > gpurun_out/l/bench_2rank_gloo.json 2> gpurun_out/l/bench_2rank.err
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/l/tests_all.log 2>&1
tail -3 gpurun_out/l/tests_mp.log; cat gpurun_out/l/bench.json; tail -c 1500 gpurun_out/l/bench_2rank_gloo.json; tail -3 gpurun_out/l/tests_all.log
