#!/bin/bash
# round 5: dropout in the condensation model (gs_gcond_config.dropout): parity tests -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-dropout}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "dropout" > $O/tests_dropout.log 2>&1; tail -40 $O/tests_dropout.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "dropout" > $O/tests_dropout_fullsize.log 2>&1; tail -30 $O/tests_dropout_fullsize.log
