"""Replaces Coupled-VAE/Modules/vae_couple.py (class VAE_COUPLE, Decoder, Gaussian)."""
from Modules import _common
_common.sync_settings()
from cvae_amd.modules import VAE_COUPLE, Decoder, Gaussian  # noqa: E402,F401
