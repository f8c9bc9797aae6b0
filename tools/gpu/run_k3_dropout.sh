#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-k3drop}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -s -k "three_hops_with_dropout" > $O/tests.log 2>&1; tail -8 $O/tests.log
