#!/bin/bash
# round 6: --dropout in the sibling methods (SGDD: the engine's dropout stage under the IGNR structure; MSGC: no effect, as upstream)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_sgdd.py tests/test_gpu_msgc.py -x -q -m gpu -k "dropout or one_outer_iteration" 2>&1 | tail -n 30 > $O/tests_siblings_dropout.log
tail -n 25 $O/tests_siblings_dropout.log
