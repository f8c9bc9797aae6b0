#!/bin/bash
# which stage of the PGE backward loses precision in the exact-split mode at cora r = 0.5 (Ph = 128, n' = 70)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/diag2
mkdir -p $O
for k in "" "GS_DW1_GEN=0" "GS_PANEL_MARG=1" "GS_PANEL_MARG=0" "GS_PANEL_A1GEN=0" "GS_PANEL=0"; do
  env GS_MODES=bf16x6 $k timeout 200 python tools/diag_fullsize.py cora 0 2>/dev/null | grep -E "^knobs|^cora" | cut -c1-330
done | tee $O/diag.txt
