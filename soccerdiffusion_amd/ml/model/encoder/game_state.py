"""Import-path parity with the reference (soccer_diffusion/ml/model/encoder/game_state.py)."""

from .encoders import GameStateEncoder  # noqa: F401
