"""Import-path parity with the reference (soccer_diffusion/ml/model/encoder/imu.py)."""

from .encoders import IMUEncoder  # noqa: F401
