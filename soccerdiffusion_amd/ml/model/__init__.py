"""Host-side mirror of the reference's ``soccer_diffusion.ml.model`` package: same class
names, constructor arguments, methods and ``state_dict`` keys; computation is HIP."""

from .decoder import DiffusionActionGenerator
from .misc import PositionalEncoding, StepToken
from .model import End2EndDiffusionTransformer

__all__ = ["DiffusionActionGenerator", "End2EndDiffusionTransformer", "PositionalEncoding", "StepToken"]
