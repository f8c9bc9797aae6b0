#!/bin/bash
# refresh of the HBM-byte counters of the bench command (separate FETCH_SIZE / WRITE_SIZE passes) on the final build -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-refpmc}
mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_bench_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes --no-prerun > /dev/null 2>&1
done
python3 tools/make_pmc_summary.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_bench_hbm_per_kernel.json > /dev/null
python3 tools/pmc_hop2.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_hop2_dispatches.json
rm -rf $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE
python3 -c "
import json
d=json.load(open('$O/pmc_bench_hbm_per_kernel.json'))
for k in d:
    if 'ell' in k or 'edge_gemm' in k or 'csc' in k or 'tall' in k: print(k[:60], d[k]['hbm_bytes_per_launch'])
print(json.load(open('$O/pmc_hop2_dispatches.json')))"
