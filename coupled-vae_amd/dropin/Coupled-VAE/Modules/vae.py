"""Replaces Coupled-VAE/Modules/vae.py (class VAE, Gaussian)."""
from Modules import _common
_common.sync_settings()
from cvae_amd.modules import VAE, Gaussian  # noqa: E402,F401
