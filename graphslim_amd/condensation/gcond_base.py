"""Host-side mirror of graphslim/condensation/gcond_base.py for the GCond path: label allocation, feature
initialisation, intermediate evaluation.  The arithmetic of train_class / match_loss lives in libgslim_hip.so."""
import numpy as np
import torch

from ..dataset.utils import save_reduced
from ..host import generate_labels_syn, init_pge_params, pge_hidden
from ..models.gcn import evaluation_args
from ..models.gcn import GCN
from ..reduction import create_reducer

_EVAL_MODELS = {"GCN": GCN}


class GCondBase:
    def __init__(self, setting, data, args, **kwargs):
        """gcond_base.py:30-70."""
        self.data, self.args, self.setting = data, args, setting
        self.device = args.device
        if not str(self.device).startswith("cuda"):
            raise RuntimeError("graphslim_amd has no CPU path (the reference's `-G -1` mode): run with a GPU, e.g. `-G 0`")
        self.labels_syn = self.data.labels_syn = self.generate_labels_syn(data)
        n = self.nnodes_syn = self.data.labels_syn.shape[0]
        self.d = d = data.feat_train.shape[1]
        print(f"target reduced size:{int(data.feat_train.shape[0] * args.reduction_rate)}")
        print(f"actual reduced size:{n}")
        self.feat_syn = torch.empty(n, d, device=self.device)
        self.pge_hidden = pge_hidden(args.dataset, args.reduction_rate)
        # parametrized_adj.py:41-55: reddit with r >= 0.01 runs the pair list through the PGE in 5 chunks (per-chunk BatchNorm)
        self.pge_chunks = 5 if (args.dataset == "reddit" and args.reduction_rate >= 0.01) else 1
        self.pge_init = init_pge_params(d, self.pge_hidden)          # nn.Linear draws, constructor + reset order
        self.adj_syn = None
        print("adj_syn:", (n, n), "feat_syn:", self.feat_syn.shape)

    def generate_labels_syn(self, data):
        """gcond_base.py:79-115."""
        labels_syn, ncd, idx = generate_labels_syn(np.asarray(data.labels_train), self.args.reduction_rate)
        self.syn_class_indices = idx
        self.data.num_class_dict = self.num_class_dict = ncd
        if self.args.verbose:
            print(ncd)
        return labels_syn

    def init(self, with_adj=False, reuse_init=False):
        """gcond_base.py:117-151: features of randomly selected real nodes (saved under reduced_graph/<init>/)."""
        args = self.args
        agent = create_reducer(args.init, setting=args.setting, data=self.data, args=args)
        keep = args.method
        args.method = args.init
        reduced = agent.reduce(self.data, verbose=True, save=True)
        args.method = keep
        return (reduced.feat_syn, reduced.adj_syn) if with_adj else reduced.feat_syn

    def get_loops(self, args):
        return args.outer_loop, args.inner_loop

    def intermediate_evaluation(self, best_val, loss_avg=None, save=True, save_valid_acc=False):
        """gcond_base.py:287-324."""
        data, args = self.data, self.args
        if args.verbose:
            print("loss_avg: {}".format(loss_avg))
        res = np.array([self.test_with_val(verbose=False, setting=args.setting, iters=args.eval_epochs)
                        for _ in range(args.run_inter_eval)]).T
        current_val = res[0].mean()
        args.logger.info("\nVal:  {:.4f} +/- {:.4f}".format(100 * current_val, 100 * res[0].std()))
        args.logger.info("Test: {:.4f} +/- {:.4f}".format(100 * res[1].mean(), 100 * res[1].std()))
        if save and current_val > best_val:
            best_val = current_val
            save_reduced(data.adj_syn, data.feat_syn, data.labels_syn, args)
        return best_val

    def test_with_val(self, verbose=False, setting="trans", iters=200, best_val=None):
        """gcond_base.py:326-358."""
        args, data = self.args, self.data
        if args.eval_model not in _EVAL_MODELS:
            raise NotImplementedError(f"graphslim_amd evaluates with GCN only (got {args.eval_model})")
        model = _EVAL_MODELS[args.eval_model](data.feat_syn.shape[1], args.hidden, data.nclass, evaluation_args(args), mode="eval")
        acc_val = model.fit_with_val(data, train_iters=iters, normadj=True, verbose=False, setting=setting, reduced=True,
                                     best_val=best_val)
        acc_test = model.test(data, setting=setting, verbose=False)
        return [acc_val, acc_test]
