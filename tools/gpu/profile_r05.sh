#!/bin/bash
# round-5 committed evidence: bench lines (default = exact bf16 split, and the fp32 pipe), kernel-trace stats of the SAME command,
# PMC (FETCH / WRITE per kernel of the bench, SQ counters of the dominant kernel), stage times, virtual scaling through the library's
# one-call sharded iteration, other workloads, whole jobs.  Outputs -> gpurun_out/prof5 (copied to profiles/r05).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-prof5}
mkdir -p $O
nproc > $O/host.txt; lscpu | grep "Model name" >> $O/host.txt
timeout 600 python bench.py > $O/bench_arxiv.json 2> $O/bench_arxiv.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_arxiv_driver_args.json 2>/dev/null
timeout 600 python bench.py --operand-mode f32 --no-cpu-baseline --eval-epochs 0 > $O/bench_arxiv_f32.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes --no-prerun > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/bench_arxiv_kernel_stats.csv
f=$(ls $O/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 70 > $O/timeline_noreuse.txt 2>/dev/null; python3 tools/timeline_csv.py $f 12 > $O/timeline_reuse.txt 2>/dev/null
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_bench_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes --no-prerun > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_panel_$c -- python3 tools/probe_panel.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_dw1_$c -- python3 tools/probe_dw1.py > /dev/null 2>&1
done
python3 tools/make_pmc_summary.py $O/pmc_dw1_FETCH_SIZE $O/pmc_dw1_WRITE_SIZE $O/pmc_pge_epoch_backward_hbm_per_kernel.json > /dev/null
tools/gpu/prof_kernels.sh "dw1|scale_uv|reduce_slabs_v4|marg|bn2_bwd|make_a1|gemm_f32_kernel" tools/probe_dw1.py > $O/pge_epoch_backward_kernels.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_panel_sq -- python3 tools/probe_panel.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_panel_grbm -- python3 tools/probe_panel.py > /dev/null 2>&1
python3 tools/make_pmc_summary.py $O/pmc_panel_FETCH_SIZE $O/pmc_panel_WRITE_SIZE $O/pmc_panel_summary.json > /dev/null
python3 tools/make_pmc_summary.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_bench_hbm_per_kernel.json > /dev/null
python3 tools/pmc_hop2.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_hop2_dispatches.json > /dev/null
python3 tools/pmc_summary.py $O/pmc_panel_sq $O/pmc_panel_grbm > $O/pmc_panel_sq_summary.txt
timeout 300 python tools/trace_tall.py > $O/trace_tall_steps.txt 2>&1
timeout 300 python tools/ab_tall.py 20 > $O/ab_tall.txt 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -v "^/opt" > $O/stage_times.txt
timeout 300 python tools/stage_times.py ogbn-arxiv f32 2>/dev/null | grep -v "^/opt" >> $O/stage_times.txt
timeout 600 python tools/virtual_scale.py > $O/virtual_scale.txt 2>&1
timeout 600 python tools/virtual_scale.py ogbn-arxiv 30 1,2,4,8 replicated > $O/virtual_scale_replicated.txt 2>&1
timeout 300 python tools/host_issue.py > $O/host_issue.txt 2>&1
timeout 600 python bench.py --workload reddit --no-cpu-baseline --eval-epochs 0 > $O/bench_reddit.json 2>/dev/null
timeout 600 python bench.py --workload cora --no-cpu-baseline --eval-epochs 0 > $O/bench_cora.json 2>/dev/null
# `--nlayers 3` at the full ogbn-arxiv shape (sizes [15, 10, 5]: 45 056-row frontiers, 225 280-slot edge lists per class): not a bench line of
# the headline metric — how long an iteration of that configuration takes
timeout 900 python bench.py --workload arxiv-k3 --steps 20 --warmup 3 --no-cpu-baseline --eval-epochs 0 --no-modes --sustain-s 0 > $O/bench_arxiv_three_hops.json 2> $O/bench_arxiv_three_hops.err
(cd tools/micro && ./issue_order 1 && ./issue_order 0) > $O/issue_order_micro.txt 2>&1
for sh in "" "0,2" "1,4" "3,8" "3,8,replicated"; do echo "== GS_SHARD=$sh" >> $O/amdahl_stage_times.txt; GS_SHARD=$sh timeout 300 python tools/stage_times.py 2>/dev/null | grep -v "^gs_\|^/opt" >> $O/amdahl_stage_times.txt; done
GRAPHSLIM_AMD_SYNTHETIC=1 timeout 1500 python tools/full_runs.py ogbn-arxiv:0.005 ogbn-arxiv:0.005:f32 cora:0.5 reddit:0.001 2>/dev/null | grep FULL > $O/full_runs.txt
ls $O; cat $O/full_runs.txt; tail -n 5 $O/virtual_scale.txt
