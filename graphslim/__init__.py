"""Drop-in alias: `import graphslim...` resolves to graphslim_amd for the modules on the GCond path, so that
GraphSlim's examples/train_gcond.py runs unchanged (`from graphslim.config import *`,
`from graphslim.evaluation.eval_agent import Evaluator`, `from graphslim.condensation import GCond`,
`from graphslim.dataset import *`)."""
import importlib
import sys

import graphslim_amd

__version__ = graphslim_amd.__version__
for _name in ("config", "utils", "host", "dataset", "dataset.loader", "dataset.utils", "dataset.synthetic", "condensation",
              "condensation.gcond", "condensation.gcond_base", "condensation.doscond", "condensation.gcondx", "condensation.msgc", "condensation.sgdd", "coarsening", "coarsening.averaging", "coarsening.clustering", "structure", "evaluation", "evaluation.eval_agent", "evaluation.utils",
              "models", "models.gcn", "sparsification", "sparsification.random", "reduction"):
    sys.modules[f"graphslim.{_name}"] = importlib.import_module(f"graphslim_amd.{_name}")
