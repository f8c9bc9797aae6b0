set -x
mkdir -p gpurun_out/r06
python tools/trace_inner_persist.py 48 > gpurun_out/r06/trace_inner_persist_48.txt 2>&1; cat gpurun_out/r06/trace_inner_persist_48.txt | tail -20
python tools/trace_inner_persist.py 96 > gpurun_out/r06/trace_inner_persist_96.txt 2>&1; cat gpurun_out/r06/trace_inner_persist_96.txt | tail -18
