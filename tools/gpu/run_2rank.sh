#!/bin/bash
# functional record of the N > 1 bench path on the 1-GPU box: two ranks share the device, gloo carries the library's exchanges
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-tworank}
mkdir -p $O
for pge in rows replicated; do
  GS_DIST_BACKEND=gloo GS_SHARE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --pge-sharding $pge > $O/bench_2ranks_gloo_shared_gpu_$pge.json 2> $O/err_$pge.txt
  python3 - $O/bench_2ranks_gloo_shared_gpu_$pge.json <<'PY'
import json, sys
ok = False
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); ok = True
        print(sys.argv[1], d["value"], d["n_gpus"], d["config"].get("parallelism"), d.get("comm", {}).get("exchanges_per_step"), d.get("comm", {}).get("driver"))
if not ok:
    print("NO JSON LINE", sys.argv[1])
PY
  tail -3 $O/err_$pge.txt
done
