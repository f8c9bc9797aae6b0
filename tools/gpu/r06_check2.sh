timeout 600 python -m pytest tests/test_gpu_engine.py -q -k "persistent_inner or distinct_sources" 2>&1 | tail -5
