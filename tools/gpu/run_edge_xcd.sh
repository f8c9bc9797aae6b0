#!/bin/bash
# round 5: class -> XCD layout of the balanced edge slabs (gs_gcond_set_edge_xcd): tests, stage times A/B, HBM bytes of the edge kernel, bench -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-edgexcd}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "balanced_edge or every_knob or per_edge_sign or sharded_engines_reproduce" > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in 0 1 0 1; do
  echo "GS_EDGE_XCD=$v" >> $O/stage_times_edge_xcd_ab.txt
  GS_EDGE_XCD=$v timeout 300 python tools/stage_times.py 2>/dev/null | grep -E "real-side|sum of|outer_step|reuse" >> $O/stage_times_edge_xcd_ab.txt
done
cat $O/stage_times_edge_xcd_ab.txt
timeout 600 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_arxiv.json 2> $O/bench_arxiv.err; python3 -c "
import json; d=json.load(open('$O/bench_arxiv.json')); print(d['value'], d['ms_per_step'], d['config']['iters_per_sec_with_forward_reuse'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd $R && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_bench_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes > /dev/null 2>&1 )
done
cd $R
python3 tools/make_pmc_summary.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_bench_hbm_per_kernel.json | head -30
rm -rf $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE
