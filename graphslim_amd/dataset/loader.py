"""GraphData contract of the reference (`TransAndInd`, graphslim/dataset/loader.py:100-229) over dataset-shaped
synthetic graphs (or real arrays handed in by the caller).  Host-side preparation only: numpy / scipy / torch-CPU,
exactly as the reference does it on the host before anything touches the device.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.sparse as sp
import torch

from .synthetic import SHAPES, make_raw


def ei2csr(edge_index, n):
    """graphslim/dataset/convertor.py:71-75: int64 ones, duplicates summed."""
    ei = edge_index.numpy() if isinstance(edge_index, torch.Tensor) else np.asarray(edge_index)
    return sp.csr_matrix((np.ones(ei.shape[1], dtype=np.int64), (ei[0], ei[1])), shape=(n, n))


_STANDARD = ["flickr", "reddit", "cora", "citeseer", "pubmed", "ogbn-arxiv"]


def get_dataset(name="cora", args=None, load_path=None):
    """graphslim/dataset/loader.py:39-97.  Resolution order for a dataset name (spelling normalised like :41-50):
      1. `<load_path>/<name>.npz` with arrays x, y, edge_index, train_mask, val_mask, test_mask;
      2. the raw files upstream GraphSlim / PyTorch-Geometric keep under `load_path` (Planetoid `ind.<name>.*`, GraphSAINT
         `adj_full.npz, feats.npy, class_map.json, role.json`) — graphslim_amd/dataset/readers.py;
      3. the dataset-shaped synthetic generator (graphslim_amd/dataset/synthetic.py) — there is no network on the target
         boxes, so nothing is ever downloaded."""
    key = name.lower().replace("-", "").replace("_", "")
    table = {n.replace("-", ""): n for n in _STANDARD}
    table["arxiv"] = "ogbn-arxiv"
    name = table.get(key, name)
    load_path = load_path or getattr(args, "load_path", None)
    raw = None
    if load_path and os.path.exists(os.path.join(load_path, f"{name}.npz")):
        z = np.load(os.path.join(load_path, f"{name}.npz"))
        from types import SimpleNamespace
        raw = SimpleNamespace(x=torch.from_numpy(z["x"]).float(), y=torch.from_numpy(z["y"]).long(),
                              edge_index=torch.from_numpy(z["edge_index"]).long(), num_nodes=int(z["x"].shape[0]),
                              train_mask=torch.from_numpy(z["train_mask"]), val_mask=torch.from_numpy(z["val_mask"]),
                              test_mask=torch.from_numpy(z["test_mask"]), name=name, synthetic=False)
    if raw is None and load_path:
        from .readers import find_and_read
        raw = find_and_read(load_path, name)
    if raw is None:
        if name not in SHAPES:
            raise ValueError(f"dataset '{name}' is not available offline; known shapes: {sorted(SHAPES)}")
        # No real files: the dataset-SHAPED synthetic generator.  Never silent: unless it was asked for
        # (GRAPHSLIM_AMD_SYNTHETIC=1, or args.synthetic), say so loudly on stderr, in the warnings stream and in the run's
        # log file; the returned GraphData carries .synthetic = True and the Evaluator tags its result lines with it.
        asked = os.environ.get("GRAPHSLIM_AMD_SYNTHETIC", "") in ("1", "true", "yes") or bool(getattr(args, "synthetic", False))
        if not asked:
            import sys
            import warnings
            msg = (f"graphslim_amd: no data files for '{name}' under load_path={load_path!r}: condensing a SYNTHETIC "
                   f"{name}-shaped graph (graphslim_amd/dataset/synthetic.py).  Accuracies are NOT {name} accuracies.  "
                   "Set GRAPHSLIM_AMD_SYNTHETIC=1 to acknowledge, or put <name>.npz / the raw Planetoid / GraphSAINT files "
                   "under --load_path.")
            print("WARNING: " + msg, file=sys.stderr)
            warnings.warn(msg, RuntimeWarning, stacklevel=2)
            logger = getattr(args, "logger", None)
            if logger is not None:
                logger.warning(msg)
        raw = make_raw(name, seed=getattr(args, "seed", 1), scale=float(os.environ.get("GRAPHSLIM_AMD_SCALE", "1")))
    data = splits(raw, getattr(args, "split", "fixed"))
    out = TransAndInd(data, name, bool(getattr(args, "pre_norm", True)))      # loader.py:85-87
    out.synthetic = bool(getattr(raw, "synthetic", False))
    out.nclass = int(raw.y.max()) + 1
    if args is not None:
        args.nclass = out.nclass
        args.synthetic_data = out.synthetic
    return out


def splits(data, exp="fixed"):
    """graphslim/dataset/utils.py:97-134.  A record that already carries masks keeps them whatever `exp` says (the
    reference only looks at `exp` when there is no train_mask); otherwise class-wise random splits: 'random' = 20 / 30 /
    rest per class, 'few' = 5 / 5 / rest, anything else = 80 % / 10 % / 10 % per class (torch.randperm per class, in
    class order, like the reference)."""
    if getattr(data, "train_mask", None) is None:
        y = data.y.reshape(-1)
        n = int(y.shape[0])
        per = []
        for c in range(int(y.max()) + 1):
            idx = (y == c).nonzero().view(-1)
            per.append(idx[torch.randperm(idx.size(0))])
        if exp == "random":
            cut = [(i[:20], i[20:50], i[50:]) for i in per]
        elif exp == "few":
            cut = [(i[:5], i[5:10], i[10:]) for i in per]
        else:
            cut = [(i[:int(i.shape[0] * 0.8)], i[int(i.shape[0] * 0.8):int(i.shape[0] * 0.9)], i[int(i.shape[0] * 0.9):]) for i in per]
        for k, name in enumerate(("train_mask", "val_mask", "test_mask")):
            m = torch.zeros(n, dtype=torch.bool)
            m[torch.cat([c[k] for c in cut])] = True
            setattr(data, name, m)
    data.idx_train = data.train_mask.nonzero().view(-1)
    data.idx_val = data.val_mask.nonzero().view(-1)
    data.idx_test = data.test_mask.nonzero().view(-1)
    return data


class SampledBlock:
    """One bipartite hop of a sampled batch: CSR [n_dst x n_src] with the full-graph normalised weights as values
    (the role torch_sparse.SparseTensor plays in PyG's `Adj.adj_t`)."""

    def __init__(self, rowptr, col, val, shape):
        self.rowptr, self.col, self.val, self.shape = rowptr, col, val, tuple(shape)

    def to(self, device):
        return SampledBlock(self.rowptr.to(device), self.col.to(device), self.val.to(device), self.shape)

    def sparse_sizes(self):
        return self.shape

    def csr(self):
        return self.rowptr, self.col, self.val

    def to_dense(self):
        out = torch.zeros(self.shape, device=self.val.device)
        rows = torch.arange(self.shape[0], device=self.val.device).repeat_interleave(self.rowptr[1:] - self.rowptr[:-1])
        out[rows, self.col] = self.val
        return out


class TransAndInd:
    """Attribute-for-attribute mirror of graphslim/dataset/loader.py:100-175."""

    def __init__(self, data, dataset, norm=True):
        self.class_dict = None
        self.samplers = None
        self.class_dict2 = None
        self.num_nodes = data.num_nodes
        self.train_mask, self.val_mask, self.test_mask = data.train_mask, data.val_mask, data.test_mask
        self.x, self.y = data.x, data.y
        self.feat_full, self.labels_full = data.x, data.y
        self.edge_index = data.edge_index
        self.adj_full = ei2csr(data.edge_index, data.x.shape[0])
        if dataset in ("flickr", "reddit", "ogbn-arxiv"):
            # loader.py:111-119: symmetrise, then StandardScaler fitted on the TRAIN rows (float64 like sklearn)
            a = self.adj_full
            self.adj_full = ((a + a.T) > 0).astype(np.int64).tocsr()
            coo = self.adj_full.tocoo()
            self.edge_index = torch.from_numpy(np.vstack([coo.row, coo.col]).astype(np.int64))
            xt = self.x[data.idx_train].double()
            mean, std = xt.mean(0), xt.std(0, unbiased=False)
            std[std == 0] = 1.0
            self.feat_full = ((self.x.double() - mean) / std).float()
        if norm and dataset in ("cora", "citeseer", "pubmed"):
            self.feat_full = torch.nn.functional.normalize(self.feat_full, p=1, dim=1)      # loader.py:120-121
        self.idx_train, self.idx_val, self.idx_test = data.idx_train, data.idx_val, data.idx_test
        it, iv, ie = (i.numpy() for i in (self.idx_train, self.idx_val, self.idx_test))
        self.adj_train = self.adj_full[np.ix_(it, it)]
        self.adj_val = self.adj_full[np.ix_(iv, iv)]
        self.adj_test = self.adj_full[np.ix_(ie, ie)]
        self.labels_train, self.labels_val, self.labels_test = (self.labels_full[i] for i in (self.idx_train, self.idx_val, self.idx_test))
        self.feat_train, self.feat_val, self.feat_test = (self.feat_full[i] for i in (self.idx_train, self.idx_val, self.idx_test))
        self.labels_syn = self.feat_syn = self.adj_syn = None
        self.nclass = int(self.labels_full.max()) + 1

    def to(self, device):
        """loader.py:141-158."""
        for k in ("feat_full", "labels_full", "x", "y", "edge_index", "feat_train", "feat_val", "feat_test"):
            setattr(self, k, getattr(self, k).to(device))
        return self

    def class_pools(self, setting):
        """loader.py:188-195: per-class seed pools — global train ids (trans) or local ids of the train sub-graph (ind)."""
        lt = self.labels_train.cpu().numpy()
        if setting == "trans":
            ids = self.idx_train.cpu().numpy()
        else:
            ids = np.arange(len(lt))
        return [ids[lt == c] for c in range(self.nclass)]

    def retrieve_class_sampler(self, c, adj, args, num=256):
        """dataset/loader.py:187-224 on the HIP sampler: (batch_size, n_id, adjs) for class `c`, `adjs` outermost hop
        first, each `(adj_t, e_id, size)` with `adj_t` a `SampledBlock` (CSR rowptr / col / val on the device, targets
        first) — what PyG's NeighborSampler.sample returns with return_e_id=False.  `adj` = (rowptr, col, val) of the
        normalised adjacency (host.normalize_adj_sparse) or None to build it from adj_full / adj_train.
        GCond.reduce does not go through this method (the engine samples all classes at once); it is here for callers of
        the GraphData contract.  Index shuffling below is host-side plumbing on torch tensors; the sampling is HIP."""
        from ..engine import BlockSet, sample_blocks
        from ..host import normalize_adj_sparse
        dev = torch.device(args.device)
        if not str(dev).startswith("cuda"):
            raise RuntimeError("graphslim_amd has no CPU path: run with a GPU (-G 0)")
        st = self.__dict__.setdefault("_gs_sampler", {})
        if "csr" not in st:
            if adj is None:
                adj = normalize_adj_sparse(self.adj_full if args.setting == "trans" else self.adj_train)
            pools = self.class_pools(args.setting)
            i32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32).to(dev)
            st["csr"] = (i32(adj[0]), i32(adj[1]), torch.as_tensor(np.asarray(adj[2]), dtype=torch.float32).to(dev))
            st["n"] = len(adj[0]) - 1
            st["ptr"] = i32(np.concatenate([[0], np.cumsum([len(p) for p in pools])]))
            st["nodes"] = i32(np.concatenate(pools))
            st["marker"] = torch.full((self.nclass * st["n"],), 0x7fffffff, dtype=torch.int32, device=dev)
            st["step"] = 0
        if args.nlayers != 2:
            raise NotImplementedError("graphslim_amd: the sampler is built for two hops (nlayers = 2)")
        f1, f2 = (15, 8) if args.dataset in ("reddit", "flickr") else (10, 5)
        blk = BlockSet(self.nclass, min(num, 256), f1, f2, dev)
        sample_blocks(*st["csr"], st["n"], st["ptr"], st["nodes"], st["marker"], blk, int(getattr(args, "seed", 1)), st["step"])
        st["step"] += 1
        B, n1 = int(blk.n_seed[c]), int(blk.n1[c])
        nid1 = blk.nid1[c, :n1].long()
        # hop-2 block: frontier (targets) <- sources; relabel the global sources targets-first like sample_adj
        v0, c0 = blk.val0[c, :n1].reshape(-1), blk.col0[c, :n1].reshape(-1).long()
        keep = v0 != 0
        rows0 = torch.arange(n1, device=dev).repeat_interleave(f2)[keep]
        src = c0[keep]
        pos = torch.full((st["n"],), -1, dtype=torch.long, device=dev)
        pos[nid1] = torch.arange(n1, device=dev)
        new = torch.unique(src[pos[src] < 0])
        pos[new] = n1 + torch.arange(new.numel(), device=dev)
        n_id = torch.cat([nid1, new])
        rp0 = torch.zeros(n1 + 1, dtype=torch.long, device=dev)
        rp0[1:] = torch.cumsum(torch.bincount(rows0, minlength=n1), 0)
        blk0 = SampledBlock(rp0, pos[src], v0[keep], (n1, n_id.numel()))
        v1, c1 = blk.val1[c, :B].reshape(-1), blk.col1[c, :B].reshape(-1).long()
        k1 = v1 != 0
        rows1 = torch.arange(B, device=dev).repeat_interleave(f1)[k1]
        rp1 = torch.zeros(B + 1, dtype=torch.long, device=dev)
        rp1[1:] = torch.cumsum(torch.bincount(rows1, minlength=B), 0)
        blk1 = SampledBlock(rp1, c1[k1], v1[k1], (B, n1))
        return B, n_id, [(blk0, None, (n_id.numel(), n1)), (blk1, None, (n1, B))]

    def retrieve_class(self, c, num=256):
        """loader.py:177-185."""
        lt = self.labels_train.cpu().numpy()
        idx = np.arange(len(lt))[lt == c]
        return np.random.permutation(idx)[:num]

    def reset(self):
        self.samplers = None
        self.class_dict2 = None
        self.labels_syn, self.feat_syn, self.adj_syn = None, None, None
