"""Import-path parity with the reference (soccer_diffusion/ml/model/encoder/base.py)."""

from .encoders import BaseEncoder  # noqa: F401
