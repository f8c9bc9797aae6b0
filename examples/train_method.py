"""Any of the gradient-matching methods built here through GraphSlim's reducer registry (graphslim/reduction/registry.py:136-142):

    python examples/train_method.py -D cora -M msgc -R 0.5 -G 0          # MSGC: 16 sparse structures (configs/msgc/cora.json)
    python examples/train_method.py -D cora -M sgdd -R 0.5 -G 0          # SGDD: IGNR + optimal-transport regulariser
    python examples/train_method.py -D ogbn-arxiv -M doscond -R 0.005 -G 0
    python examples/train_method.py -D cora -M gcondx -R 0.5 -G 0

`-M` is one of gcond, doscond, gcondx, doscondx, msgc, sgdd; per-method hyper-parameters come from
graphslim_amd/configs/<method>/<dataset>.json exactly as upstream's configs do."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import graphslim.config as gs_config                      # noqa: E402
import graphslim.dataset as gs_dataset                    # noqa: E402
from graphslim.evaluation.eval_agent import Evaluator     # noqa: E402
from graphslim.reduction import create_reducer            # noqa: E402


def main(argv=None):
    options = gs_config.cli(argv, standalone_mode=False)
    data = gs_dataset.get_dataset(options.dataset, options)
    agent = create_reducer(options.method, setting=options.setting, data=data, args=options)
    condensed = agent.reduce(data, verbose=options.verbose)
    return Evaluator(options).evaluate(condensed, model_type="GCN")


if __name__ == "__main__":
    main()
