from .averaging import Average  # noqa: F401
from .clustering import Cluster  # noqa: F401
