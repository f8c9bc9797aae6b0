"""GCond(setting, data, args).reduce(data) — host-side mirror of graphslim/condensation/gcond.py:18-81.  The loop
structure, the RNG draw order and the bookkeeping are the reference's; every outer iteration is one call into the HIP
engine (gs_gcond_outer_step)."""
import numpy as np
import torch
from tqdm import trange

from ..engine import EngineShapes, GCondEngine
from ..evaluation.utils import verbose_time_memory
from ..host import init_sgc_params, normalize_adj_sparse
from .gcond_base import GCondBase


class GCond(GCondBase):
    """"Graph Condensation for Graph Neural Networks" (Jin et al.) — gradient matching with a learned structure."""

    def __init__(self, setting, data, args, **kwargs):
        super().__init__(setting, data, args, **kwargs)

    @verbose_time_memory
    def reduce(self, data, verbose=True):
        args = self.args
        if args.condense_model != "SGC":
            raise NotImplementedError("graphslim_amd: condense_model must be SGC (GCN as condense model is a 'next' row, SURVEY §8f)")
        if args.nlayers != 2:
            raise NotImplementedError("graphslim_amd: nlayers (propagations) must be 2")
        if getattr(args, "with_bn", False):
            raise NotImplementedError("graphslim_amd: with_bn is outside the built path")
        labels_syn = np.asarray(data.labels_syn)
        if args.setting == "trans":
            features, adj = data.feat_full, data.adj_full
        else:
            features, adj = data.feat_train, data.adj_train
        feat_init = self.init()                                   # gcond.py:29 (np.random draws)
        csr = normalize_adj_sparse(adj)                           # gcond.py:33
        pools = data.class_pools(args.setting)
        sizes = (15, 8) if args.dataset in ("reddit", "flickr") else (10, 5)     # loader.py:197-210
        nclass = data.nclass
        ranges = [tuple(self.syn_class_indices.get(c, (0, 0))) for c in range(nclass)]
        shapes = EngineShapes(features.shape[0], self.d, nclass, self.nnodes_syn, args.ntrans, args.hidden, self.pge_hidden,
                              256, sizes[0], sizes[1])
        cls_ptr = np.concatenate([[0], np.cumsum([len(p) for p in pools])])
        eng = self.engine = GCondEngine(shapes, features, csr, cls_ptr, np.concatenate(pools), labels_syn, ranges,
                                        lr_feat=args.lr_feat, lr_adj=args.lr_adj, lr_model=args.lr,
                                        inner_loop=args.inner_loop, dis_metric=args.dis_metric, seed=args.seed,
                                        device=self.device)
        eng.set_feat(torch.as_tensor(feat_init))
        eng.set_pge(self.pge_init)
        eng.set_forward_reuse(True)      # gcond.py:48 recomputes what gcond.py:64 just produced; same values, one pass
        self.feat_syn = eng.feat_syn
        outer_loop, inner_loop = self.get_loops(args)
        loss_avg, best_val = 0, 0
        init_sgc_params(self.d, args.hidden, nclass, args.ntrans)           # model constructor draw (gcond.py:38)
        labels_t = torch.from_numpy(labels_syn).to(self.device)
        for it in trange(args.epochs):
            eng.set_model(init_sgc_params(self.d, args.hidden, nclass, args.ntrans))   # gcond.py:42-45
            for ol in range(outer_loop):
                eng.outer_step(it)
            loss_sum, _ = eng.read_loss()            # one host sync per epoch instead of one .item() per iteration
            loss_avg = (loss_avg + loss_sum) / (nclass * outer_loop)        # gcond.py:52,74 (running value is not reset)
            if it in args.checkpoints:
                self.adj_syn = eng.adj_syn.clone()
                data.adj_syn, data.feat_syn, data.labels_syn = self.adj_syn, eng.feat_syn.clone(), labels_t
                best_val = self.intermediate_evaluation(best_val, loss_avg)
        return data
