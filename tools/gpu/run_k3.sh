#!/bin/bash
# round 5: `--nlayers 3`: middle-hop sampler, engine vs the reference golden G24, engine vs oracle on its own three-hop blocks, the
# three-layer evaluation GCN (G25) -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-k3}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sampler" > $O/tests_sampler.log 2>&1; tail -15 $O/tests_sampler.log
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "train_class_against_reference_golden" > $O/tests_g24.log 2>&1; tail -15 $O/tests_g24.log
timeout 900 python -m pytest tests/test_gpu_reduce.py -x -q -m gpu -k "eval_gcn_against_reference_golden_g9" > $O/tests_g25.log 2>&1; tail -15 $O/tests_g25.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k "three_hops" > $O/tests_k3_oracle.log 2>&1; tail -25 $O/tests_k3_oracle.log
