timeout 1500 python -m pytest tests/test_gpu_msgc.py tests/test_gpu_sgdd.py -q -k "one_outer_iteration or reduce_api" 2>&1 | grep -v "^|" | tail -25
