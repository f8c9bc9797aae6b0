#!/usr/bin/env python3
"""The three persistent 256-row products of the headline iteration (FGEN = PGE layer-2 forward, inference pass; MARG = its fused
backward marginals; PLAIN = H1) launched through the engine's own stages with one gs_panel_tall_debug triage mask (phases of the
kernel switched off: 1 no epilogue, 16 no MFMAs, 32 no operand preparation, ...; results are garbage then) — the subject of a
rocprofv3 --kernel-trace --stats run per mask (tools/gpu/triage_forms.sh): skeleton time next to the full kernel (VERDICT r04 "next" 2).
usage: triage_forms.py <triage mask> [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphslim_amd import _lib  # noqa: E402
from graphslim_amd import workloads as W  # noqa: E402

tri, reps = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = _lib.lib()
p = W.build_problem("ogbn-arxiv", seed=1)
eng = W.make_engine(p, "cuda:0")
W.init_state(p, eng)
W.new_model(p, eng)
eng.set_overlap(0)
eng.outer_step(10)            # a whole clean iteration: every buffer the stages read is in place
torch.cuda.synchronize()
L.gs_panel_tall_debug(0, 1, tri)
for _ in range(reps):
    eng.stage_real_grads()                 # H1 (PLAIN)
    eng.stage_pge_forward(True)            # inference pass: FGEN on the persistent kernel
    eng.stage_pge_backward(True, True)     # MARG
torch.cuda.synchronize()
L.gs_panel_tall_debug(0, 1, 0)
print("triage", tri, "done")
