#!/bin/bash
# round 5: the job-level gates (quiet job: 48 oracle jobs; arxiv-gate: 30 jobs; same-blocks tripwire) + the new engine tests + a bench line -> gpurun_out/$1
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-gate5}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_reduce.py -x -q -m gpu -s -k "quiet_job" > $O/test_job_level_accuracy_quiet_job_48seeds.log 2>&1; tail -4 $O/test_job_level_accuracy_quiet_job_48seeds.log
timeout 900 python -m pytest tests/test_gpu_reduce.py -x -q -m gpu -s -k "job_level_accuracy_against" > $O/test_job_level_accuracy_own_sampler_30seeds.log 2>&1; tail -3 $O/test_job_level_accuracy_own_sampler_30seeds.log
GS_JOB_SEEDS=${2:-2} timeout 2400 python -m pytest tests/test_gpu_reduce.py -x -q -m gpu -s -k "oracles_own_blocks" > $O/test_job_level_tripwire_same_blocks.log 2>&1; tail -3 $O/test_job_level_tripwire_same_blocks.log
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "balanced_edge or every_knob" > $O/tests_edge.log 2>&1; tail -2 $O/tests_edge.log
timeout 900 python bench.py > $O/bench_arxiv.json 2> $O/bench_arxiv.err; python3 -c "
import json; d=json.load(open('$O/bench_arxiv.json')); print(d['value'], d['ms_per_step'], d['config']['iters_per_sec_with_forward_reuse']); print(json.dumps(d['roofline'])[:1500]); print(d['cpu_baseline'])"
