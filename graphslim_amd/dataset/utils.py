"""On-disk format of reduced graphs and the evaluation-time sparsification — graphslim/dataset/utils.py:8-66,136-296.
File names and tensor formats match the reference so condensed graphs interchange with upstream evaluators:
<save_path>/reduced_graph/<method>/{adj,feat,label}_<dataset>_<rate>_<seed>.pt holding a dense fp32 [n',n'] adjacency
(raw PGE output), fp32 [n',d] features and int64 [n'] labels."""
import os

import torch


def _dir(args):
    base = os.path.abspath(os.path.expanduser(args.save_path))
    if getattr(args, "attack", None) is not None and args.dataset in ["flickr"]:
        return os.path.join(base, "corrupt_graph", args.attack, "reduced_graph", args.method)
    return os.path.join(base, "reduced_graph", args.method)


def _name(kind, args):
    return f"{kind}_{args.dataset}_{args.reduction_rate}_{args.seed}.pt"


def save_reduced(adj_syn=None, feat_syn=None, labels_syn=None, args=None):
    """dataset/utils.py:136-152."""
    d = _dir(args)
    os.makedirs(d, exist_ok=True)
    for kind, t in (("adj", adj_syn), ("feat", feat_syn), ("label", labels_syn)):
        if t is not None:
            torch.save(t.detach().cpu() if isinstance(t, torch.Tensor) else t, os.path.join(d, _name(kind, args)))
    if getattr(args, "logger", None) is not None:
        args.logger.info(f"Saved {os.path.join(d, _name('adj', args))}")


def load_reduced(args, data=None):
    """dataset/utils.py:155-258: missing files fall back to the original features / labels / identity adjacency."""
    d = _dir(args)
    dev = args.device if torch.cuda.is_available() else "cpu"
    missing = 0

    def _load(kind):
        return torch.load(os.path.join(d, _name(kind, args)), map_location=dev)

    try:
        feat_syn = _load("feat")
    except Exception as e:
        print(f"find no feat at {os.path.join(d, _name('feat', args))}, use original feature matrix instead. Error: {e}")
        missing += 1
        feat_syn = data.feat_full if args.setting == "trans" else data.feat_train
    try:
        labels_syn = _load("label")
    except Exception as e:
        print(f"find no label at {os.path.join(d, _name('label', args))}, use original label matrix instead. Error: {e}")
        missing += 1
        labels_syn = data.labels_train
    try:
        adj_syn = _load("adj")
    except Exception as e:
        print(f"find no adj at {os.path.join(d, _name('adj', args))}, use identity matrix instead. Error: {e}")
        missing += 1
        adj_syn = torch.eye(feat_syn.size(0), device=dev)
    if missing == 3 and getattr(args, "logger", None) is not None:
        args.logger.info("no file found, use original graph instead")
    return adj_syn, feat_syn, labels_syn


def sparsify(model_type, adj_syn, args, verbose=False):
    """dataset/utils.py:8-66."""
    threshold = 0
    if model_type == "MLP":
        adj_syn = adj_syn - adj_syn
        torch.diagonal(adj_syn).fill_(1)
    elif model_type == "GAT":
        if args.method in ["gcond", "doscond"]:
            threshold = 0.5 if args.dataset in ["cora", "citeseer"] else 0.1
        elif args.method in ["msgc"]:
            threshold = args.threshold
        else:
            threshold = 0.5
    elif args.method in ["gcond", "doscond"]:
        threshold = args.threshold
    if threshold > 0:
        adj_syn[adj_syn < threshold] = 0
        if verbose:
            print("Sparsity after truncating:", adj_syn.nonzero().shape[0] / adj_syn.numel())
    return adj_syn


def get_syn_data(data, args, model_type, verbose=False):
    """dataset/utils.py:261-296."""
    adj_syn, feat_syn, labels_syn = load_reduced(args, data)
    if labels_syn.shape[0] == data.labels_train.shape[0]:
        return feat_syn, adj_syn, labels_syn
    if adj_syn.layout != torch.strided:
        adj_syn = adj_syn.to_dense()
    adj_syn = sparsify(model_type, adj_syn, args, verbose=verbose)
    return feat_syn, adj_syn, labels_syn
