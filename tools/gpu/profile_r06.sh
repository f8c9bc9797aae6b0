#!/bin/bash
# round-6 committed evidence: bench lines (default and the driver's arguments), kernel-trace stats of the SAME command, PMC (FETCH / WRITE per
# kernel of the bench and of the dominant product alone, SQ counters), stage times, the other workloads.  Outputs -> gpurun_out/prof6
# (copied to profiles/r06).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-prof6}
mkdir -p $O
nproc > $O/host.txt; lscpu | grep "Model name" >> $O/host.txt
timeout 600 python bench.py > $O/bench_arxiv.json 2> $O/bench_arxiv.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_arxiv_driver_args.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes --no-prerun > $O/bench_under_rocprof.json 2>/dev/null
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/bench_arxiv_kernel_stats.csv
f=$(ls $O/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 70 > $O/timeline_noreuse.txt 2>/dev/null
rm -rf $O/kt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_bench_$c -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes --no-prerun > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_panel_$c -- python3 tools/probe_panel.py > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_panel_sq -- python3 tools/probe_panel.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_panel_grbm -- python3 tools/probe_panel.py > /dev/null 2>&1
python3 tools/make_pmc_summary.py $O/pmc_panel_FETCH_SIZE $O/pmc_panel_WRITE_SIZE $O/pmc_panel_summary.json > /dev/null
python3 tools/make_pmc_summary.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_bench_hbm_per_kernel.json > /dev/null
python3 tools/pmc_hop2.py $O/pmc_bench_FETCH_SIZE $O/pmc_bench_WRITE_SIZE $O/pmc_hop2_dispatches.json > /dev/null
python3 tools/pmc_summary.py $O/pmc_panel_sq $O/pmc_panel_grbm > $O/pmc_panel_sq_summary.txt
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
timeout 300 python tools/stage_times.py ogbn-arxiv 2>/dev/null | grep -v "^/opt" > $O/stage_times.txt
timeout 300 python tools/host_issue.py > $O/host_issue.txt 2>&1
timeout 600 python bench.py --workload reddit --no-cpu-baseline --eval-epochs 0 > $O/bench_reddit_final.json 2>/dev/null
timeout 600 python bench.py --workload cora --no-cpu-baseline --eval-epochs 0 > $O/bench_cora.json 2>/dev/null
timeout 900 python bench.py --workload arxiv-k3 --steps 20 --warmup 3 --no-cpu-baseline --eval-epochs 0 --no-modes --sustain-s 0 > $O/bench_arxiv_three_hops.json 2> $O/bench_arxiv_three_hops.err
for w in arxiv-t3 arxiv-gcn3 arxiv-gcn2; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --eval-epochs 0 --no-modes --sustain-s 0 > $O/bench_$w.json 2> $O/bench_$w.err
done
ls $O; head -c 600 $O/bench_arxiv.json; echo; head -c 300 $O/bench_arxiv_driver_args.json; echo; for w in reddit_final cora arxiv_three_hops arxiv-t3 arxiv-gcn3 arxiv-gcn2; do head -c 200 $O/bench_$w.json; echo; done
