set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "persistent_inner or fused_chains_against or knob" 2>&1 | tail -15
for ip in 0 1; do GS_INNER_PERSIST=$ip python tools/stage_times.py > gpurun_out/r06/stage_times_inner_persist_$ip.txt 2>&1; tail -25 gpurun_out/r06/stage_times_inner_persist_$ip.txt; done
