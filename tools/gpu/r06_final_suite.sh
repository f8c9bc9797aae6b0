#!/bin/bash
# round 6: the whole GPU suite on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 3300 python3 -m pytest tests -q -m gpu --durations=10 2>&1 | tail -n 40 > $O/tests_gpu_all_final.log
tail -n 16 $O/tests_gpu_all_final.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
