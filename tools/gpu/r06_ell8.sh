#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -x -q -m gpu -k "ogbn_arxiv_r0005 or reddit or sampler or ell or dropout_full_size or three_hops_against" 2>&1 | tail -n 3
for i in 1 2; do
python3 bench.py --no-cpu-baseline --eval-epochs 0 --no-modes --steps 50 --warmup 5 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
b=json.loads(sys.stdin.read()); h=b['roofline_hbm']
print('value', b['value'], 'hop2 in situ', h['avg_launch_us'], 'frac', h['frac'], 'isolated', h['isolated_launch_us'], 'isolated GB/s', h['isolated_achieved'])"
done
python3 bench.py --workload reddit --no-cpu-baseline --eval-epochs 0 --no-modes 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
b=json.loads(sys.stdin.read()); print('reddit', b['value'], json.dumps(b.get('roofline'))[:400])"
