#!/bin/bash
# SQ / GRBM counters of the 128-row and the persistent 256-row panel kernels (tools/probe_tall.py) -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-pmc_tall}
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/sq1 -- python3 tools/probe_tall.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq2 -- python3 tools/probe_tall.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/grbm -- python3 tools/probe_tall.py > /dev/null 2>&1
python3 tools/pmc_summary.py $O/sq1 $O/sq2 $O/grbm | grep -E "panel_gemm|tall" > $O/summary.txt
find $O -name "*.csv" -delete; find $O -type d -empty -delete
cat $O/summary.txt
