#!/bin/bash
# round 6: GCN condensation model with dropout at full size + through the reference's entry points, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py -x -q -m gpu -k "gcn_condensation_model_with_dropout or dropout_reduce_api or dropout_train_class" 2>&1 | tail -n 30 > $O/tests_gcn_dropout_fullsize.log
tail -n 12 $O/tests_gcn_dropout_fullsize.log
timeout 3000 python3 -m pytest tests -q -m gpu --durations=15 2>&1 | tail -n 60 > $O/tests_gpu_all.log
tail -n 30 $O/tests_gpu_all.log
