#!/bin/bash
# round 6: the bench line with H1's persistent grid on 14/16 of the compute units (the new default) against the full grid, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
for sh in 16 14 16 14; do
  python3 -c "
from graphslim_amd import _lib
_lib.lib().gs_panel_set_plain_share($sh)
import sys, bench
sys.argv = ['bench.py', '--no-cpu-baseline', '--eval-epochs', '0', '--no-modes', '--steps', '50', '--warmup', '5']
bench.main()" 2>/dev/null | tail -n 1 > $O/bench_plain_share_$sh.json
  python3 -c "
import json,sys
b=json.loads(open('$O/bench_plain_share_$sh.json').read())
print('share $sh: value', b['value'], 'ms_per_step', b['ms_per_step'], 'sustained', b.get('sustained',{}).get('value'), 'reuse', b['config'].get('iters_per_sec_with_forward_reuse'))" | tee -a $O/bench_plain_share.txt
done
