#!/bin/bash
# static issue priority for one half of the persistent kernels' workgroups (gs_panel_tall_debug triage bits 8 / 9), stage times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-prio1}
mkdir -p $O
for t in 0 256 512 0 256 512; do
  echo "== triage $t" | tee -a $O/prio.txt
  GS_TALL=1,0,$t timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -E "PGE forward \(inference\)|PGE backward \(feature|real-side|outer_step" | tee -a $O/prio.txt
done
