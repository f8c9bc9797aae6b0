set -x
mkdir -p gpurun_out/r06
( time timeout 2400 python -m pytest tests/test_gpu_fullsize.py -q -x -k "three_and_four_transforms" -s 2>&1 | grep -v "^|" | tail -40 ) > gpurun_out/r06/tests_deep.log 2>&1
tail -42 gpurun_out/r06/tests_deep.log | cut -c1-400
