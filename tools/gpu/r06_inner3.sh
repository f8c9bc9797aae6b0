set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "persistent_inner or fused_chains_against" 2>&1 | tail -8
python tools/trace_inner_persist.py 48 > gpurun_out/r06/trace_inner_persist_48.txt 2>&1; cat gpurun_out/r06/trace_inner_persist_48.txt | tail -17
GS_INNER_PERSIST=1 python tools/stage_times.py > gpurun_out/r06/stage_times_inner_persist_1.txt 2>&1; tail -6 gpurun_out/r06/stage_times_inner_persist_1.txt
