set -x
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_engine.py -q -k "distinct_sources or balanced_edge or knob" 2>&1 | tail -12
for v in 0 1; do GS_EDGE_SRC=$v python tools/stage_times.py > gpurun_out/r06/stage_times_edge_src_$v.txt 2>&1; echo "edge_src $v"; grep -v amdgpu gpurun_out/r06/stage_times_edge_src_$v.txt | grep "real-side\|outer_step"; done
