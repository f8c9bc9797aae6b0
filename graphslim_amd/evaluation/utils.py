"""graphslim/evaluation/utils.py:115-175 — the timing / size decorator wrapped around reduce()."""
import time
from functools import wraps

import torch


def getsize_mb(elements):
    size = 0
    for e in elements:
        if isinstance(e, torch.Tensor):
            size += e.element_size() * e.nelement()
        elif hasattr(e, "data") and hasattr(e, "indices"):      # scipy sparse
            size += e.data.nbytes + e.indices.nbytes + e.indptr.nbytes
    return size / 1024 / 1024


def verbose_time_memory(func):
    @wraps(func)
    def wrapper(*args, **kwargs):
        verbose = kwargs.get("verbose", False)
        if verbose:
            start = time.perf_counter()
        result = func(*args, **kwargs)
        if verbose:
            run = time.perf_counter() - start
            print("Function Time:", run, "s")
            print("Function Time:", run * 1000, "ms")
            data = kwargs.get("data", None)
            if data is None:
                for a in args:
                    if hasattr(a, "feat_train") or hasattr(a, "x"):
                        data = a
                        break
                if data is None:
                    raise ValueError("The function must be called with 'data' as an argument.")
            origin = getsize_mb([data.feat_train, data.adj_train, data.labels_train])
            condensed = getsize_mb([t for t in (data.feat_syn, data.adj_syn, data.labels_syn) if t is not None])
            print(f"Original graph:{origin:.2f} Mb  Condensed graph:{condensed:.2f} Mb")
        return result

    return wrapper
