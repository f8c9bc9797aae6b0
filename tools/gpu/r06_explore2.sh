set -x
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_engine.py -q -x -k "c_caller or hoisted_training" 2>&1 | tail -5
python tools/explore_jobs.py 8 \
 '{"workload":"arxiv-gate","scale":0.05,"sep":2.0,"epochs":60,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate","scale":0.05,"sep":3.0,"epochs":60,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate","scale":0.05,"sep":4.0,"epochs":60,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate","scale":0.05,"sep":3.0,"epochs":60,"outer":10,"inner":3,"lr_feat":0.001,"lr_adj":0.001}' \
 '{"workload":"arxiv-gate","scale":0.05,"sep":4.0,"epochs":60,"outer":10,"inner":3,"lr_feat":0.001,"lr_adj":0.001}' \
 '{"workload":"arxiv-gate","scale":0.05,"sep":3.0,"epochs":120,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate","scale":0.05,"sep":6.0,"epochs":60,"outer":10,"inner":3}' \
  > gpurun_out/r06/explore_jobs2.txt 2>&1
grep -v "^|" gpurun_out/r06/explore_jobs2.txt | tail -30
