"""Replaces Coupled-VAE/Modules/dae.py (class DAE)."""
from Modules import _common
_common.sync_settings()
from cvae_amd.modules import DAE  # noqa: E402,F401
