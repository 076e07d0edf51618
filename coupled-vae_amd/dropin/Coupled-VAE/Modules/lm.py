"""Replaces Coupled-VAE/Modules/lm.py (class LM and its own Decoder)."""
from Modules import _common
_common.sync_settings()
from cvae_amd.modules import LM, LMDecoder as Decoder  # noqa: E402,F401
