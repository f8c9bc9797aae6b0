set -x
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_engine.py -q -x -k "c_caller or hoisted_training" 2>&1 | tail -5
python tools/explore_jobs.py 8 \
 '{"workload":"arxiv-gate-c8","scale":0.05,"sep":1.0,"epochs":60,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate-c8","scale":0.05,"sep":1.5,"epochs":60,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate-c8","scale":0.05,"sep":2.0,"epochs":60,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate-c8","scale":0.05,"sep":1.5,"epochs":120,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate-c8","scale":0.05,"sep":1.5,"epochs":60,"outer":10,"inner":3,"lr_feat":0.003,"lr_adj":0.003}' \
 '{"workload":"arxiv-gate-c8","scale":0.05,"sep":1.0,"epochs":120,"outer":10,"inner":3,"lr_feat":0.003,"lr_adj":0.003}' \
  > gpurun_out/r06/explore_jobs3.txt 2>&1
grep -v "^|" gpurun_out/r06/explore_jobs3.txt | tail -30
