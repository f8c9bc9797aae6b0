"""Evaluator of the reference (graphslim/evaluation/eval_agent.py:159-291) for the GCN evaluation model: `run_eval`
seeded runs, each trains the GCN on the condensed graph (HIP ops) and tests on the full graph."""
import numpy as np
from tqdm import trange

from ..dataset.utils import get_syn_data
from ..host import seed_everything
from ..models.gcn import GCN, evaluation_args
from ..parallel import RowShard, RunShard, active_dist

_MODELS = {"GCN": GCN}


class Evaluator:
    def __init__(self, args, **kwargs):
        self.args = args
        self.device = args.device
        self.metric = args.metric

    def reset_parameters(self):
        pass

    def test(self, data, model_type, verbose=True, reduced=True, mode="eval", MIA=False):
        """eval_agent.py:159-232 (no attack / MIA branches: outside the GCond path)."""
        args = self.args
        if model_type not in _MODELS:
            raise NotImplementedError(f"graphslim_amd evaluates with GCN only (got {model_type})")
        if verbose:
            print(f"======= testing {model_type}")
        model = _MODELS[model_type](data.feat_full.shape[1], args.hidden, data.nclass, evaluation_args(args), mode=mode)
        best_acc_val = model.fit_with_val(data, train_iters=args.eval_epochs, normadj=True, verbose=verbose,
                                          setting=args.setting, reduced=reduced)
        acc_test = model.test(data, setting=args.setting, verbose=False)
        if verbose:
            print("Test full set results:", f"accuracy= {acc_test:.4f}")
        return best_acc_val, acc_test

    def evaluate(self, data, model_type, verbose=True, reduced=True, mode="eval"):
        """eval_agent.py:234-291: mean / std of the test accuracy over args.run_eval runs seeded seed+i."""
        args = self.args
        if reduced:
            data.feat_syn, data.adj_syn, data.labels_syn = get_syn_data(data, args, model_type=model_type, verbose=verbose)
        # multi-GPU (one process per GPU under torch.distributed): the seeded runs are sharded over the ranks
        # (args.eval_parallel = "runs", default) or every run row-shards its full-graph passes (= "rows")
        dist = active_dist()
        by_rows = dist is not None and getattr(args, "eval_parallel", "runs") == "rows"
        if by_rows:
            args.eval_row_shard = RowShard(dist)
        shard = RunShard(None if by_rows else dist)
        mine = shard.mine(args.run_eval)
        if verbose:
            print(f"Evaluating reduced data using {model_type}")
            runs = trange(len(mine))
        else:
            runs = range(len(mine))
        res = {}
        for k in runs:
            i = mine[k]
            seed_everything(args.seed + i)
            _, acc = self.test(data, model_type=model_type, verbose=args.verbose, reduced=reduced, mode=mode)
            res[i] = acc
            if verbose:
                runs.set_postfix(test_acc=acc)
        res = shard.combine(args.run_eval, res, self.device if dist is not None and str(self.device).startswith("cuda") else "cpu")
        msg = f"Seed:{args.seed}, Test Mean Accuracy: {100 * res.mean():.2f} +/- {100 * res.std():.2f}"
        if getattr(data, "synthetic", False):     # never report a synthetic stand-in under the real dataset's name alone
            msg += f"  [SYNTHETIC {getattr(args, 'dataset', '?')}-shaped graph, not the real dataset]"
        if hasattr(args, "logger"):
            args.logger.info(msg)
        if verbose:
            print(msg)
        return res.mean(), res.std()
