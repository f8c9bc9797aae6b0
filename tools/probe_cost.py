import os, sys, torch
sys.path.insert(0, "/root/repo")
os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.getcwd())
import bench
p = bench.build_problem("ogbn-arxiv", seed=1)
eng = bench.make_engine(p, "cuda:0")
bench.init_state(p, eng); bench.new_model(p, eng)
for _ in range(30): eng.outer_step(10)
torch.cuda.synchronize()
def run(probe):
    out = []
    for g in range(12):
        eng.probe(4 if probe else False)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(20): eng.outer_step(10)
        b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 50.0)
    eng.probe(False)
    out.sort(); return out[len(out)//2]
for rep in range(3):
    print("probe off: %.0f us/step   probe on: %.0f us/step" % (run(False), run(True)))
