#!/bin/bash
# round 6: GCN condensation model with dropout (golden G29 cases), the sibling methods' nlayers API tests, the headline-route quiet job
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "gcn_dropout or dropout_train_class" 2>&1 | tail -n 30 > $O/tests_gcn_dropout.log
tail -n 15 $O/tests_gcn_dropout.log
timeout 900 python3 -m pytest tests/test_gpu_msgc.py tests/test_gpu_sgdd.py -x -q -m gpu -k "reduce_api" 2>&1 | tail -n 15 > $O/tests_siblings_api.log
tail -n 8 $O/tests_siblings_api.log
timeout 1500 python3 -m pytest tests/test_gpu_reduce.py -x -q -m gpu -s -k "quiet_job_headline_routes" > $O/test_job_level_accuracy_quiet_job_headline_routes_40seeds.log 2>&1
tail -n 6 $O/test_job_level_accuracy_quiet_job_headline_routes_40seeds.log
