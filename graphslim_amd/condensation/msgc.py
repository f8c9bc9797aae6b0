"""MSGC — "Multiple sparse graphs condensation" (graphslim/condensation/msgc.py:14-236): GCond's gradient matching with
`batch_adj` sparse learnable structures over the same synthetic nodes instead of one dense PGE structure.

train_class / match_loss, the model steps and the real-side sampling run in the HIP engine on the block-diagonal graph of the
B structures (graphslim_amd/structure.py lays it out); the edge MLP with its training-mode BatchNorm runs forward and backward
in csrc/gs_structgen.hip + gs_gemm_f32.  This file is the host loop: RNG draw order, update rule, window bookkeeping."""
from collections import deque

import numpy as np
import torch
from tqdm import trange

from ..engine import EngineShapes, GCondEngine, adam_step
from ..evaluation.utils import verbose_time_memory
from ..host import init_edge_mlp_params, init_sgc_params, msgc_edges, msgc_labels, normalize_adj_sparse
from ..structure import BatchLayout, EdgeStructureGenerator
from .gcond_base import GCondBase


class MSGCLoop:
    """One outer iteration of msgc.py:95-118 on the HIP engine (a block-diagonal graph of B n' rows) and the edge-list structure
    generator.  Owns X~ [n', d] and its Adam; the engine holds the B copies."""

    def __init__(self, eng, edge_mlp_params, rows, cols, batch, lay, feat_init, lr_feat, lr_adj, hidden=256):
        self.eng, self.lay, self.lr_feat = eng, lay, lr_feat
        d, nb = eng.shapes.d, lay.n_big
        self.gen = EdgeStructureGenerator(edge_mlp_params, d, rows, cols, batch, lay, lr_adj, eng.dev, hidden=hidden)
        self.x = torch.as_tensor(feat_init).to(eng.dev, torch.float32).contiguous()
        self.m_x, self.v_x, self.step_x = torch.zeros_like(self.x), torch.zeros_like(self.x), 0
        self.gen.tile(self.x, eng.feat_syn)
        eng.set_block_structure(lay.pos)          # the B structures are disconnected blocks: per-block A^ A^ and dA^ products
        self.d_adj, self.d_feat_big = eng.view("d_adj").view(nb, nb), eng.view("d_feat").view(nb, d)
        self.adj_inner_norm = eng.view("adj_norm_inner").view(nb, nb)
        self.fwd_valid = False        # the generator holds a forward pass of the current (X~, theta)

    def step(self, it):
        eng, gen, x = self.eng, self.gen, self.x
        eng.stage_real_async()
        # :98 get_adj_t_syn — the same function of the same (X~, theta) as the previous iteration's :111 call: its structure and
        # activations are still in place (bitwise the values a second pass would produce)
        adj = gen.adj if self.fwd_valid else gen.forward(x)
        eng.set_structure(adj, inference=False)
        eng.stage_syn_match()                                                     # :100 train_class
        eng.syn_finish()
        if it % 50 < 10:                                                          # :105-108
            gen.backward(self.d_adj)
            gen.adam_step()
        else:
            gen.fold(self.d_feat_big, gen.dX)                                     # through the B copies of X~ ...
            gen.backward(self.d_adj, accumulate_dx=True, param_grads=False)                          # ... plus through the edge MLP
            self.step_x += 1
            adam_step(x, gen.dX, self.m_x, self.v_x, self.lr_feat, self.step_x)
            gen.tile(x, eng.feat_syn)
        eng.set_structure(gen.forward(x), inference=True)                         # :111
        self.fwd_valid = True
        eng.inner_steps()                                                         # :113-118

    def inner_structure(self):
        """The normalised [B, n', n'] batch of the last inner steps (what get_adj_t_syn().detach() returns, :111)."""
        return self.gen.batched(self.adj_inner_norm)


class MSGC(GCondBase):
    edge_hidden_channels = 256                                        # msgc.py:21

    def __init__(self, setting, data, args, **kwargs):
        """gcond_base.py:45-60 (the `msgc` branch: n' = int(|train| rate), no dense PGE) + msgc.py:18-39."""
        self.data, self.args, self.setting = data, args, setting
        self.device = args.device
        if not str(self.device).startswith("cuda"):
            raise RuntimeError("graphslim_amd has no CPU path (the reference's `-G -1` mode): run with a GPU, e.g. `-G 0`")
        n = self.nnodes_syn = self.n_syn = int(data.feat_train.shape[0] * args.reduction_rate)
        self.d = d = data.feat_train.shape[1]
        print(f"target reduced size:{n}")
        print(f"actual reduced size:{n}")
        self.feat_syn = torch.empty(n, d, device=self.device)
        n_each_y = self.generate_labels_syn(data)
        self.labels_syn = data.labels_syn = self.y_syn
        self.batch_size = int(args.batch_adj)
        self.n_classes = data.nclass
        self.num_class_dict = self.data.num_class_dict = {c: int(k) for c, k in enumerate(n_each_y)}
        self.pge_init = init_edge_mlp_params(d, self.edge_hidden_channels)        # nn.Sequential constructor draws
        self.adj_syn = None

    def generate_labels_syn(self, data):
        """msgc.py:41-64."""
        y_syn, n_each = msgc_labels(np.asarray(data.labels_train), self.n_syn, data.nclass)
        self.y_syn = torch.from_numpy(y_syn)
        if self.args.verbose:
            print(f"num_class:{n_each}")
        return n_each

    def reset_adj_batch(self):
        """msgc.py:178-222."""
        self.rows, self.cols, self.batch = msgc_edges(self.y_syn.numpy(), self.n_classes, self.batch_size)
        if self.args.verbose:
            print(f"n_edge:{len(self.rows) / self.batch_size / 2}")

    @verbose_time_memory
    def reduce(self, data, verbose=True):
        args = self.args
        if args.condense_model != "SGC":
            raise NotImplementedError("graphslim_amd: MSGC is built for condense_model SGC (every configs/msgc/*.json)")
        if args.nlayers not in (2, 3, 4, 5):
            raise NotImplementedError(f"graphslim_amd: nlayers must be 2 .. 5 (dataset/loader.py:197-210 knows no other), got {args.nlayers}")
        if getattr(args, "with_bn", False):
            raise NotImplementedError("graphslim_amd: MSGC with with_bn in the condensation model is not built")
        # `--dropout` is accepted and — exactly as upstream — has NO effect: msgc.py:99 puts the model into eval mode before every
        # train_class ("fix basic_model while optimizing graphsyner") and nothing puts it back, so F.dropout(training=False) is the
        # identity in train_class and in the inner steps (golden G30: the run with dropout 0.4 reproduces dropout 0 bit for bit)
        if args.setting == "trans":
            features, adj = data.feat_full, data.adj_full
        else:
            features, adj = data.feat_train, data.adj_train
        csr = normalize_adj_sparse(adj)                                                   # msgc.py:75
        nclass, d, B = data.nclass, self.d, self.batch_size
        init_sgc_params(d, args.hidden, nclass, args.ntrans)                              # basic_model constructor draw (:79)
        self.reset_adj_batch()                                                            # :81
        feat_init = self.init()                                                           # :82
        lay = BatchLayout(self.y_syn.numpy(), nclass, B)
        pools = data.class_pools(args.setting)
        sizes = {2: (15, 8) if args.dataset in ("reddit", "flickr") else (10, 5), 3: (15, 10, 5), 4: (15, 10, 5, 5),
                 5: (15, 10, 5, 5, 5)}[args.nlayers]                                           # loader.py:197-210
        # the engine's dense PGE is not used (pge_hidden = 4 keeps its buffers small); the structure comes from the generator
        shapes = EngineShapes(features.shape[0], d, nclass, lay.n_big, args.ntrans, args.hidden, 4, 256, sizes[0], sizes[-1],
                              fmid=tuple(sizes[1:-1]))
        cls_ptr = np.concatenate([[0], np.cumsum([len(p) for p in pools])])
        eng = self.engine = GCondEngine(shapes, features, csr, cls_ptr, np.concatenate(pools), lay.labels_big, lay.ranges,
                                        lr_feat=args.lr_feat, lr_adj=args.lr_adj, lr_model=args.lr, inner_loop=args.inner_loop,
                                        dis_metric=args.dis_metric, seed=args.seed, device=self.device)
        loop = self.loop = MSGCLoop(eng, self.pge_init, self.rows, self.cols, self.batch, lay, feat_init, args.lr_feat, args.lr_adj,
                                    hidden=self.edge_hidden_channels)
        dev = eng.dev
        y_rep = self.y_syn.repeat(B).to(dev)
        best_val = 0
        args.window = args.patience = 20                                                  # :87
        # the last `window` epochs' loss / X~ / structure (msgc.py:238-247 keeps a bounded list): a bounded deque
        losses, x_syns, adj_t_syns = (deque(maxlen=args.window) for _ in range(3))
        for it in trange(args.epochs):
            # basic_model.initialize() re-draws the weights in place; optimizer_basic_model (built once, :90) keeps its moments
            eng.set_model(init_sgc_params(d, args.hidden, nclass, args.ntrans), keep_optimizer=it > 0)
            for _ in range(args.outer_loop):
                loop.step(it)
            loss_sum, _ = eng.read_loss()
            loss_avg = loss_sum / (nclass * args.outer_loop)                              # :119 (reset every epoch, :94)
            losses.append(loss_avg)
            x_syns.append(loop.x.cpu())
            adj_t_syns.append(loop.inner_structure().cpu())                          # the NORMALISED batch, as upstream keeps it
            loss_window = sum(losses) / len(losses)
            if args.verbose:
                print(f"loss:{loss_window:.4f} feat:{float(x_syns[-1].sum()):.4f} adj:{float(adj_t_syns[-1].sum()):.4f}")
            if it in args.checkpoints:                                                    # :126-130
                best_x = sum(x_syns) / len(x_syns)
                best_adj = sum(adj_t_syns) / len(adj_t_syns)
                data.feat_syn, data.adj_syn, data.labels_syn = best_x.to(dev), best_adj.to(dev), y_rep
                best_val = self.intermediate_evaluation(best_val, loss_window)
        self.feat_syn = loop.x
        self.adj_syn = loop.inner_structure()
        return data
