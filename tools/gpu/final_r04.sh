#!/bin/bash
# final checks of the round on the final build: whole GPU suite, timelines of one iteration (two streams / solo)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-final4}
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests_gpu_all.log 2>&1
tail -3 $O/tests_gpu_all.log
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --sustain-s 0 --eval-epochs 0 --no-modes > /dev/null 2>&1
f=$(ls $O/kt/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 70 > $O/timeline_noreuse.txt 2>&1; python3 tools/timeline_csv.py $f 12 > $O/timeline_reuse.txt 2>&1
rm -rf $O/kt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt2 -- python3 tools/trace_solo.py > /dev/null 2>&1
f=$(ls $O/kt2/*/*kernel_trace.csv | head -1); python3 tools/timeline_csv.py $f 8 > $O/timeline_solo_no_overlap.txt 2>&1
rm -rf $O/kt2
tail -1 $O/timeline_noreuse.txt; tail -1 $O/timeline_reuse.txt; tail -1 $O/timeline_solo_no_overlap.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
