set -x
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_engine.py -q -k "persistent_inner or fused_chains_against or knob" 2>&1 | tail -8
python tools/diag_inner_persist.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06/diag_inner_persist.txt
python tools/trace_inner_persist.py 48 > gpurun_out/r06/trace_inner_persist_48.txt 2>&1; tail -4 gpurun_out/r06/trace_inner_persist_48.txt
for cfg in "0 48" "2 48" "1 48" "1 64" "1 96"; do set -- $cfg; GS_INNER_PERSIST=$1 GS_INNER_GRID=$2 python tools/stage_times.py > gpurun_out/r06/stage_times_inner_persist_$1_grid$2.txt 2>&1; echo "persist $1 grid $2"; tail -6 gpurun_out/r06/stage_times_inner_persist_$1_grid$2.txt | grep -v "sum of\|PGE forward"; done
