"""GCond on the MI355X engine, end to end: parse the command line, build the graph, condense it, evaluate the condensed graph
with a GCN.  The public names are GraphSlim's (`graphslim/` in this repository is an alias package onto `graphslim_amd`), so
upstream's own examples/train_gcond.py also runs unchanged with this repository first on PYTHONPATH.

    python examples/train_gcond.py -D ogbn-arxiv -M gcond -R 0.005 -G 0
    python examples/train_gcond.py -D cora -M gcond -R 0.5 -G 0 --load_path /data/pyg      # raw Planetoid files, if present
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import graphslim.config as gs_config                      # noqa: E402
import graphslim.dataset as gs_dataset                    # noqa: E402
from graphslim.condensation import GCond                  # noqa: E402
from graphslim.evaluation.eval_agent import Evaluator     # noqa: E402


def main(argv=None):
    options = gs_config.cli(argv, standalone_mode=False)
    data = gs_dataset.get_dataset(options.dataset, options)
    condensed = GCond(setting=options.setting, data=data, args=options).reduce(data, verbose=options.verbose)
    mean, std = Evaluator(options).evaluate(condensed, model_type="GCN")
    return mean, std


if __name__ == "__main__":
    main()
