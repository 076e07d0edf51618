"""Replaces Coupled-CVAE/Modules/vae.py (class VAE = conditional VAE, Gaussian)."""
from Modules import _common
_common.sync_settings()
from cvae_amd.modules import CVAE as VAE, Gaussian  # noqa: E402,F401
