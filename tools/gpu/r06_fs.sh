set -x
mkdir -p gpurun_out/r06
( time timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -k "cora_r05 or three_hops_full_scale" -s 2>&1 | grep -v "^|" | tail -30 ) > gpurun_out/r06/tests_fullsize_new.log 2>&1
tail -32 gpurun_out/r06/tests_fullsize_new.log | cut -c1-300
