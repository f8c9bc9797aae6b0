set -x
timeout 1500 python -m pytest tests/test_gpu_engine.py -q 2>&1 | tail -6
