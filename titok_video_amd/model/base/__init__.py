from . import blocks, utils  # noqa: F401
