#!/bin/bash
# round 5: which whole-job configuration is quiet enough for the +-0.3 pp gate (tools/explore_jobs.py) -> gpurun_out/$1/explore_jobs.txt
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-explore}
mkdir -p $O
N=${2:-10}
timeout 1500 python tools/explore_jobs.py $N \
 '{"workload":"cora","epochs":50,"outer":20,"inner":15,"threshold":0.05}' \
 '{"workload":"cora","epochs":50,"outer":20,"inner":15,"threshold":0.05,"lr_feat":0.01,"lr_adj":0.01}' \
 '{"workload":"cora","epochs":30,"outer":10,"inner":5,"threshold":0.05,"lr_feat":0.01,"lr_adj":0.01}' \
 '{"workload":"cora-r0.25","epochs":50,"outer":20,"inner":15,"threshold":0.05}' \
 '{"workload":"cora-r0.25","epochs":50,"outer":20,"inner":15,"threshold":0.05,"lr_feat":0.01,"lr_adj":0.01}' \
 '{"workload":"arxiv-gate","scale":0.05,"epochs":60,"outer":10,"inner":3}' \
 '{"workload":"arxiv-gate","scale":0.05,"epochs":16,"outer":10,"inner":3,"eval_epochs":600}' \
 > $O/explore_jobs.txt 2> $O/explore_jobs.err
tail -30 $O/explore_jobs.txt; tail -5 $O/explore_jobs.err
