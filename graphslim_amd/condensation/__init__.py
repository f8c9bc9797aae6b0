from .gcond import GCond  # noqa: F401
from .gcond_base import GCondBase  # noqa: F401
