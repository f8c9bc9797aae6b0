#!/bin/bash
# round 5: stage times of one virtual rank at world 1 / 2 / 4 / 8 (rows and replicated layouts): the Amdahl table of DESIGN section 6 -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-amdahl}
mkdir -p $O
: > $O/amdahl_stage_times.txt
for sh in "" "0,2" "1,4" "3,8" "3,8,replicated"; do
  echo "== GS_SHARD=$sh" >> $O/amdahl_stage_times.txt
  GS_SHARD=$sh timeout 300 python tools/stage_times.py 2>$O/amdahl.err | grep -v "^gs_" >> $O/amdahl_stage_times.txt; tail -3 $O/amdahl.err >> $O/amdahl_stage_times.txt
done
cat $O/amdahl_stage_times.txt
