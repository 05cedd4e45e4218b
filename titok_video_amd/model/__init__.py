"""Mirror of the reference's `model` package for the tokenizer path (titok, base.blocks, base.utils, quantizer.fsq,
losses.loss_module)."""
from . import base, quantizer  # noqa: F401
