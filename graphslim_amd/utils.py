"""The few helpers of graphslim/utils.py the GCond path exposes to callers."""
import torch

from .host import seed_everything  # noqa: F401


def accuracy(output, labels):
    """graphslim/utils.py:516-538."""
    if not hasattr(labels, "__len__"):
        labels = [labels]
    if type(labels) is not torch.Tensor:
        labels = torch.LongTensor(labels)
    preds = output.max(1)[1].type_as(labels)
    return preds.eq(labels).double().sum() / len(labels)
