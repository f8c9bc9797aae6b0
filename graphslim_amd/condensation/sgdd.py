"""SGDD — "Does Graph Distillation See Like Vision Dataset Counterpart?" (graphslim/condensation/sgdd.py:14-105): GCond's gradient
matching with the IGNR graphon learner (graphslim/models/ignr.py) in place of the PGE and an optimal-transport regulariser between
the learned structure and a corner of the real adjacency.

train_class / match_loss, sampling and the model steps run in the HIP engine (structure supplied by the caller,
gs_gcond_set_structure); the IGNR towers, both BatchNorm modes and the regulariser run in csrc/gs_structgen.hip + gs_gemm_f32 +
rocSOLVER (graphslim_amd/structure.py).  This file is the host loop: RNG draw order, update rule, bookkeeping."""
import numpy as np
import torch
from tqdm import trange

from ..engine import EngineShapes, GCondEngine, adam_step
from ..evaluation.utils import verbose_time_memory
from ..host import generate_labels_syn, init_ignr_params, init_sgc_params, normalize_adj_sparse
from ..structure import IGNRGenerator, TransportRegulariser, pinv_corner
from .gcond_base import GCondBase


class SGDDLoop:
    """One outer iteration of sgdd.py:64-95.  X~ lives in the engine (eng.feat_syn); its Adam, the IGNR and the regulariser here."""

    def __init__(self, eng, ignr_params, Lx_inv, ep_ratio, opt_scale, lr_feat, lr_adj):
        n, d = eng.shapes.n_syn, eng.shapes.d
        self.eng, self.lr_feat, self.opt_scale = eng, lr_feat, float(opt_scale)
        self.gen = IGNRGenerator(ignr_params, d, n, ep_ratio, lr_adj, eng.dev)
        # sgdd.py:70-74: the regulariser only enters the loss for opt_scale > 0 (upstream still evaluates it; its value is unused then)
        self.reg = TransportRegulariser(Lx_inv, n, eng.dev) if self.opt_scale > 0 else None
        self.x = eng.feat_syn
        self.m_x, self.v_x, self.step_x = torch.zeros_like(self.x), torch.zeros_like(self.x), 0
        self.d_adj, self.d_feat = eng.view("d_adj").view(n, n), eng.view("d_feat").view(n, d)
        self.iterations = 0

    def step(self, it):
        eng, gen, x = self.eng, self.gen, self.x
        if eng.shapes.dropout:        # the masks of this iteration's train_class and inner steps (the staged calls do not advance the counter)
            eng.set_dropout_iter(self.iterations)
        self.iterations += 1
        eng.stage_real_async()
        adj = gen.forward(x, train=True)                                          # :65 self.pge(self.feat_syn, Lx=...)
        eng.set_structure(adj, inference=False)                                   # :66 normalize_adj_tensor
        eng.stage_syn_match()                                                     # :69 train_class
        eng.syn_finish()
        if self.reg is not None:                                                  # :70-74 loss += opt_scale * opt_loss
            # (run on a second stream beside train_class it gains nothing: the library's eigen-decomposition blocks the host, measured)
            self.reg.forward_backward(adj, gen.views()["P"], self.opt_scale, self.d_adj, gen.views(gen.grad)["P"])
        if it % 50 < 10:                                                          # :80-83
            gen.backward(self.d_adj)
            gen.adam_step()
        else:
            gen.dX.copy_(self.d_feat)                                             # the direct feature gradient (device copy) ...
            gen.backward(self.d_adj, accumulate_dx=True, param_grads=False)                          # ... plus the one through the structure
            self.step_x += 1
            adam_step(x, gen.dX, self.m_x, self.v_x, self.lr_feat, self.step_x)
        self.adj_inner = gen.forward(x, train=False)                              # :85 self.pge.inference (eval-mode BatchNorm)
        eng.set_structure(self.adj_inner, inference=True)
        eng.inner_steps()                                                         # :88-93

    def read_regulariser(self):
        """Sum of opt_loss over the iterations since the last call (one host sync)."""
        if self.reg is None:
            return 0.0
        v = float(self.reg.acc)
        self.reg.acc.zero_()
        bad_eig, clamped = self.reg.read_health()
        if (bad_eig or clamped) and not getattr(self, "_warned", False):
            self._warned = True
            import warnings
            warnings.warn(f"graphslim_amd SGDD: the optimal-transport regulariser left the fp32 range of upstream's formula "
                          f"({bad_eig} eigen-decomposition(s) not converged, {clamped} entries replaced by the range guard since the last "
                          f"epoch): with a rank-deficient adjacency corner the pseudo-inverse holds rounding-level reciprocals "
                          f"(ignr.py:13-27) — upstream's own run turns NaN here; consider opt_scale = 0 or a smaller mx_size")
        self.health = (bad_eig, clamped)
        return v


class SGDD(GCondBase):
    nfeat = 128                                                                    # sgdd.py:20

    def __init__(self, setting, data, args, **kwargs):
        """gcond_base.py:45-60 (labels, X~; no PGE for `sgdd`) + sgdd.py:17-28."""
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
        for key in ("opt_scale", "ep_ratio", "mx_size"):
            if not hasattr(args, key):
                raise AttributeError(f"SGDD needs args.{key} (configs/sgdd/<dataset>.json sets it; upstream has no command-line flag)")
        # upstream builds P with args.mx_size rows and only afterwards (sgdd.py:33-36) shrinks mx_size to the graph size when the
        # graph is smaller — leaving P and the corner inconsistent (AttributeError on data.adj_mx); the corner size is fixed here
        self.mx = int(min(args.mx_size, data.adj_full.shape[0]))
        self.ignr_init = init_ignr_params(d, n, self.mx, self.nfeat)
        self.adj_syn = None
        print("adj_syn:", (n, n), "feat_syn:", self.feat_syn.shape)

    def prepare(self, data):
        """Everything of reduce() in front of its loop (sgdd.py:30-54): engine, IGNR, regulariser.  Returns (engine, loop)."""
        args = self.args
        if args.condense_model != "SGC":
            raise NotImplementedError("graphslim_amd: SGDD is built for condense_model SGC (every configs/sgdd/*.json)")
        if args.nlayers not in (2, 3, 4, 5):
            # (`--nlayers` K: K sampled hops, K propagations on both sides — models/sgc.py:45-51, dataset/loader.py:205-210; the engine's
            # stages take the structure from the IGNR whatever K is: Q = A^^K with the product rule for d A^)
            raise NotImplementedError(f"graphslim_amd: nlayers must be 2 .. 5 (dataset/loader.py:197-210 knows no other), got {args.nlayers}")
        if getattr(args, "with_bn", False):
            raise NotImplementedError("graphslim_amd: SGDD with with_bn in the condensation model is not built")
        # `--dropout`: sgdd.py:62 puts the model into training mode like gcond.py:45 — masks in train_class and the inner steps (golden
        # G30 counts them); the engine's dropout stage for SGC with two transforms (one transform has no dropout layer, as upstream)
        dropout = float(getattr(args, "dropout", 0.0) or 0.0)
        if not 0.0 <= dropout < 1.0:
            raise ValueError(f"graphslim_amd: dropout must be in [0, 1), got {dropout}")
        if dropout != 0.0 and args.ntrans > 2:
            raise NotImplementedError("graphslim_amd: dropout != 0 with ntrans > 2 is not built")
        args.mx_size = self.mx                                                            # sgdd.py:33-36
        data.adj_mx = data.adj_full[:self.mx, :self.mx]
        if args.setting == "trans":
            features, adj = data.feat_full, data.adj_full
        else:
            features, adj = data.feat_train, data.adj_train
        feat_init = self.init()                                                           # :45
        csr = normalize_adj_sparse(adj)                                                   # :48
        nclass, d, n = data.nclass, self.d, self.nnodes_syn
        pools = data.class_pools(args.setting)
        sizes = {2: (15, 8) if args.dataset in ("reddit", "flickr") else (10, 5), 3: (15, 10, 5), 4: (15, 10, 5, 5),
                 5: (15, 10, 5, 5, 5)}[args.nlayers]                                           # loader.py:197-210
        ranges = [tuple(self.syn_class_indices.get(c, (0, 0))) for c in range(nclass)]
        # the engine's dense PGE is not used (pge_hidden = 4 keeps its buffers small); the structure comes from the IGNR
        shapes = EngineShapes(features.shape[0], d, nclass, n, args.ntrans, args.hidden, 4, 256, sizes[0], sizes[-1], fmid=tuple(sizes[1:-1]),
                              dropout=dropout if args.ntrans == 2 else 0.0)
        cls_ptr = np.concatenate([[0], np.cumsum([len(p) for p in pools])])
        eng = self.engine = GCondEngine(shapes, features, csr, cls_ptr, np.concatenate(pools), np.asarray(data.labels_syn), ranges,
                                        lr_feat=args.lr_feat, lr_adj=args.lr_adj, lr_model=args.lr, inner_loop=args.inner_loop,
                                        dis_metric=args.dis_metric, seed=args.seed, device=self.device)
        eng.set_feat(torch.as_tensor(feat_init))
        loop = self.loop = SGDDLoop(eng, self.ignr_init, pinv_corner(data.adj_mx), args.ep_ratio, args.opt_scale, args.lr_feat,
                                    args.lr_adj)
        self.feat_syn = eng.feat_syn
        init_sgc_params(d, args.hidden, nclass, args.ntrans)                              # model constructor draw (:54)
        return eng, loop

    @verbose_time_memory
    def reduce(self, data, verbose=True):
        args = self.args
        eng, loop = self.prepare(data)
        nclass, d = data.nclass, self.d
        outer_loop, inner_loop = self.get_loops(args)
        loss_avg, best_val = 0, 0
        labels_t = torch.from_numpy(np.asarray(data.labels_syn)).to(self.device)
        for it in trange(args.epochs):
            eng.set_model(init_sgc_params(d, args.hidden, nclass, args.ntrans))           # :58-60 initialize() + a fresh Adam
            for _ in range(outer_loop):
                loop.step(it)
            loss_sum, _ = eng.read_loss()
            loss_sum += loop.opt_scale * loop.read_regulariser()                           # :74 loss = loss + loss_opt
            loss_avg = (loss_avg + loss_sum) / (nclass * outer_loop)                      # :75,97 (the running value is not reset)
            if it in args.checkpoints:                                                    # :99-102
                self.adj_syn = loop.adj_inner.clone()
                data.adj_syn, data.feat_syn, data.labels_syn = self.adj_syn, eng.feat_syn.clone(), labels_t
                best_val = self.intermediate_evaluation(best_val, loss_avg)
        return data
