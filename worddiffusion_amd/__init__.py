"""worddiffusion_amd - MI355X-native UNet denoising hot path of WordDiffusion."""
from .unet import UNetModel
from .unetPhosc import UNetModelPhosc
from .diffusion import EMA, Diffusion, label_padding
from .vae import AutoencoderKL

__all__ = ["UNetModel", "UNetModelPhosc", "Diffusion", "EMA", "label_padding", "AutoencoderKL"]
