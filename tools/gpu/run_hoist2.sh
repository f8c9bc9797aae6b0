#!/bin/bash
# round 5: hoisted training forward — A/B of the edge plan beside it (GS_HOIST_FWD 1 | 2) and one iteration's timeline -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-hoist2}
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "hoisted" > $O/tests.log 2>&1; tail -3 $O/tests.log
for v in 0 1 2 0 1 2; do
  echo "GS_HOIST_FWD=$v" >> $O/stage_times_hoist_ab.txt
  GS_HOIST_FWD=$v timeout 300 python tools/stage_times.py 2>/dev/null | grep -E "outer_step \(" >> $O/stage_times_hoist_ab.txt
done
cat $O/stage_times_hoist_ab.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes --no-prerun > $O/bench_under_rocprof.json 2>/dev/null )
cd $R
f=$(ls $O/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 70 > $O/timeline_hoisted.txt 2>/dev/null
rm -rf $O/kt
tail -3 $O/timeline_hoisted.txt
