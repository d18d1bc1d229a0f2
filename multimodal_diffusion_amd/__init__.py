"""MI355X-native denoising hot path of mauruszach/multimodal_diffusion (MMDiT x2 CFG + noise head + DDIM).

Host-side mirrors of the reference's Python interface for this path; all compute is hand-written HIP for
gfx950 behind the C ABI in ``include/avdiff_hip.h`` (``multimodal_diffusion_amd/csrc`` → ``libavdiff_hip.so``).
There is no CPU or eager-PyTorch fallback: CPU tensors or a missing library raise.
"""
from . import adapters, ops, schedule_utils, schedules   # noqa: F401
from .adapters import TimestepCfg, TimestepEmbedder      # noqa: F401
from .mmdt import MMDiT, Block, MHA, MLP, RMSNorm      # noqa: F401
from .noise_heads import MultiModalNoiseHead           # noqa: F401
from .sampler import (DenoiseEngine, LinearAdapter, add_sinusoidal_timestep, build_components,   # noqa: F401
                      latents_to_tokens_audio, latents_to_tokens_video, sample_one_direction,
                      tokens_to_latents_audio)
from .schedules import ModalitySchedule, build_schedules_from_config   # noqa: F401
from .vae_video3d import VideoVAE, VideoVAEConfig                      # noqa: F401
from .audio_codec import AudioCodec, AudioCodecConfig                  # noqa: F401

__version__ = "0.1.0"
