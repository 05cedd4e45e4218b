from .loss_module import ReconstructionLoss  # noqa: F401
