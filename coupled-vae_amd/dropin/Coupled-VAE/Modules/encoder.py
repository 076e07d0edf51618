"""Replaces Coupled-VAE/Modules/encoder.py."""
from Modules import _common
_common.sync_settings()
from cvae_amd.modules import Encoder  # noqa: E402,F401
