from .random import Random  # noqa: F401
