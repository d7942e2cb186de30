"""soccerdiffusion_amd — MI355X-native denoiser hot path of bit-bots/SoccerDiffusion.

Python mirrors the reference's module API; all computation is hand-written HIP for gfx950
behind the C ABI in ``include/soccerdiffusion_hip.h``.  There is no CPU fallback.
"""

__version__ = "0.3.0"
