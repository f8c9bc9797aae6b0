from .loader import TransAndInd, get_dataset, splits  # noqa: F401
from .utils import get_syn_data, load_reduced, save_reduced, sparsify  # noqa: F401
