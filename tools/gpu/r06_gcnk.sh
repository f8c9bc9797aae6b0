timeout 1200 python -m pytest tests/test_gpu_engine.py -q -x -k "gcn_condense_model_follows" 2>&1 | grep -v "^|" | tail -30
