from .averaging import Average  # noqa: F401
