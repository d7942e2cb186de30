"""Import-path parity with the reference (soccer_diffusion/ml/model/encoder/joint.py)."""

from .encoders import JointEncoder  # noqa: F401
