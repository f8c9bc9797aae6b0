#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-dropchk}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -q -m gpu -k "dropout" > $O/tests.log 2>&1; tail -3 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
