#!/bin/bash
# round 5: compute units (sixteenths) of the hoisted training forward's persistent grid -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-hshare}
mkdir -p $O
for v in ${2:-14 12 13 15 16 14 12 13 15 16}; do
  echo "GS_HOIST_SHARE=$v" >> $O/stage_times_hoist_share.txt
  GS_HOIST_SHARE=$v timeout 300 python tools/stage_times.py 2>/dev/null | grep -E "outer_step \(" >> $O/stage_times_hoist_share.txt
done
cat $O/stage_times_hoist_share.txt
