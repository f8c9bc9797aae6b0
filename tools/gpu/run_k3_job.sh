#!/bin/bash
# round 5: the three-hop quiet job (gross-error tripwire, 13 seeds) and the dropout tests -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-k3job}
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reduce.py -q -m gpu -s -k "quiet_job_three_hops" > $O/test_job_level_three_hops_13seeds.log 2>&1; tail -5 $O/test_job_level_three_hops_13seeds.log
timeout 900 python -m pytest tests/test_gpu_engine.py -q -m gpu -k "dropout" > $O/tests_dropout.log 2>&1; tail -5 $O/tests_dropout.log
