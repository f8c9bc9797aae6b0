#!/bin/bash
# the GPU suite from a given test file on (after a fix), then smoke -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-rest}
mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu > $O/tests_gpu_all.log 2>&1
tail -12 $O/tests_gpu_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
