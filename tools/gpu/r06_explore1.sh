set -x
mkdir -p gpurun_out/r06
python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_baseline.json 2> gpurun_out/r06/bench_baseline.err
tail -c 600 gpurun_out/r06/bench_baseline.json
python tools/explore_jobs.py 8 \
 '{"workload":"arxiv-gate","scale":0.05,"epochs":100,"outer":10,"inner":3,"lr_feat":0.001,"lr_adj":0.001}' \
 '{"workload":"arxiv-gate","scale":0.05,"epochs":100,"outer":10,"inner":3,"lr_feat":0.0003,"lr_adj":0.0003}' \
 '{"workload":"arxiv-gate","scale":0.05,"epochs":100,"outer":10,"inner":3,"lr_feat":0.0001,"lr_adj":0.0001}' \
 '{"workload":"arxiv-gate","scale":0.05,"epochs":200,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate","scale":0.1,"epochs":100,"outer":10,"inner":3,"lr_feat":0.001,"lr_adj":0.001}' \
 '{"workload":"arxiv-gate","scale":0.05,"epochs":200,"outer":10,"inner":3,"lr_feat":0.001,"lr_adj":0.001,"eval_epochs":300}' \
  > gpurun_out/r06/explore_jobs1.txt 2>&1
grep -v "^|" gpurun_out/r06/explore_jobs1.txt | tail -30
