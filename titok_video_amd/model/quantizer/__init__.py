from . import fsq  # noqa: F401
