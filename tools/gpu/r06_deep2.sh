set -x
mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_engine.py -q -k "train_class_against_reference_golden" 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_reduce.py -q -k "three_transforms or reduce_and_evaluate_api" 2>&1 | tail -12
( time timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -k "three_and_four_transforms" 2>&1 | tail -5 ) 2>&1 | tail -9
