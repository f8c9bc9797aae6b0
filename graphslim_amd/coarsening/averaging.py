"""`averaging` initialiser (graphslim/coarsening/averaging.py:12-71): every synthetic node of class c starts at the mean of the
train features of class c; the initial structure is the identity.  Used by configs/msgc/{cora,reddit}.json.  The class means
are one gs_gather_rows_f32 + gs_colsum_f32 (fp64 accumulation) per class on the GPU."""
import numpy as np
import torch

from .._lib import check, lib, ptr, stream
from ..dataset.utils import save_reduced
from ..evaluation.utils import verbose_time_memory
from ..host import generate_labels_syn


class Average:
    def __init__(self, setting, data, args, **kwargs):
        self.data, self.args, self.setting = data, args, setting
        # upstream's constructor also sets args.method = "averaging" (averaging.py:30) BEFORE GCondBase.init saves the method name
        # it means to restore (gcond_base.py:146-149), so a run with `--init averaging` files its condensed graph under
        # reduced_graph/averaging/ for the rest of the job.  Not mirrored: init() already switches the name for the one save.

    @verbose_time_memory
    def reduce(self, data, verbose=True, save=True):
        args = self.args
        if not str(args.device).startswith("cuda"):
            raise RuntimeError("graphslim_amd has no CPU path: run with a GPU (-G 0)")
        dev = torch.device(args.device)
        y_train = np.asarray(data.labels_train)
        if getattr(data, "labels_syn", None) is not None:                 # averaging.py:53-58
            y_syn = torch.as_tensor(np.asarray(data.labels_syn.cpu() if isinstance(data.labels_syn, torch.Tensor) else data.labels_syn))
        else:                                                             # prepare_select, averaging.py:73-113
            y_syn = torch.from_numpy(generate_labels_syn(y_train, args.reduction_rate)[0])
        x_train = torch.as_tensor(data.feat_train).to(dev, torch.float32).contiguous()
        d = x_train.shape[1]
        x_syn = torch.zeros(y_syn.shape[0], d)
        mean = torch.empty(d, dtype=torch.float32, device=dev)
        for c in range(data.nclass):
            ids = np.nonzero(y_train == c)[0]
            if len(ids) == 0 or not bool((y_syn == c).any()):
                continue
            idx = torch.as_tensor(ids, dtype=torch.int32).to(dev)
            rows = torch.empty(len(ids), d, dtype=torch.float32, device=dev)
            check(lib().gs_gather_rows_f32(ptr(x_train), d, ptr(idx), len(ids), d, ptr(rows), d, stream()))
            check(lib().gs_colsum_f32(ptr(rows), d, len(ids), d, ptr(mean), stream()))
            x_syn[y_syn == c] = (mean.cpu().double() / len(ids)).float()
        data.feat_syn, data.labels_syn = x_syn, y_syn
        data.adj_syn = torch.eye(x_syn.shape[0])
        if save:
            save_reduced(data.adj_syn, data.feat_syn, data.labels_syn, args)
        return data
