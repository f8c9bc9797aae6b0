from .doscond import DosCond  # noqa: F401
from .gcond import GCond  # noqa: F401
from .gcond_base import GCondBase  # noqa: F401
from .gcondx import DosCondX, GCondX  # noqa: F401
from .msgc import MSGC  # noqa: F401
from .sgdd import SGDD  # noqa: F401
