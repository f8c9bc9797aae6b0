#!/bin/bash
# round 6: Amdahl tables of configurations where sharding could pay (ogbn-arxiv r = 1 %: n' = 909; reddit r = 0.1 %) + a fresh mode-F record for reddit
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
for wl in ogbn-arxiv-r0.01 reddit; do
  : > $O/amdahl_stage_times_$wl.txt
  for sh in "" "0,2" "1,4" "3,8"; do
    echo "== $wl GS_SHARD=$sh" >> $O/amdahl_stage_times_$wl.txt
    GS_SHARD=$sh timeout 600 python tools/stage_times.py $wl 2>$O/amdahl.err | grep -v "^gs_" >> $O/amdahl_stage_times_$wl.txt; tail -2 $O/amdahl.err | grep -v amdgpu >> $O/amdahl_stage_times_$wl.txt
  done
  grep "==\|sum of\|outer_step (feature" $O/amdahl_stage_times_$wl.txt
done
timeout 900 python bench.py --workload reddit --full-neighbour --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_reddit_full_neighbour.json 2> $O/bench_reddit_full_neighbour.err
tail -c 700 $O/bench_reddit_full_neighbour.json
timeout 900 python bench.py --workload reddit --steps 30 --warmup 5 --no-cpu-baseline --no-modes > $O/bench_reddit.json 2> $O/bench_reddit.err
python -c "
import json; d=json.load(open('$O/bench_reddit.json')); print('reddit', d['value'], d['ms_per_step'], d.get('roofline_hbm',{}).get('frac'))"
