from .encoders import BaseEncoder, GameStateEncoder, IMUEncoder, JointEncoder

__all__ = ["BaseEncoder", "GameStateEncoder", "IMUEncoder", "JointEncoder"]
