set -x
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_engine.py -q -k "train_class_against_reference_golden or gcn_condense_model" 2>&1 | tail -6
( time timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -x -k "three_layer_gcn" -s 2>&1 | grep -v "^|" | cut -c1-500 | tail -14 ) > gpurun_out/r06/tests_gcn3_fullsize.log 2>&1
tail -16 gpurun_out/r06/tests_gcn3_fullsize.log
