#!/bin/bash
# round 5: the tests added after the last full-suite run, then the same-blocks tripwire over 12 seeds (committed evidence) -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-chk5}
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "sharded_engines_reproduce or three_hops_reduce_api" > $O/tests_new_engine.log 2>&1; tail -6 $O/tests_new_engine.log
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s > $O/tests_fullsize.log 2>&1; tail -8 $O/tests_fullsize.log; grep -c "adj_self" $O/tests_fullsize.log
GS_JOB_SEEDS=12 timeout 2400 python -m pytest tests/test_gpu_reduce.py -x -q -m gpu -s -k "oracles_own_blocks" > $O/test_job_level_tripwire_same_blocks_12seeds.log 2>&1; tail -4 $O/test_job_level_tripwire_same_blocks_12seeds.log
