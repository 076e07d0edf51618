"""Replaces Coupled-VAE/Modules/Losses/SeqLoss.py."""
from Modules import _common
_common.sync_settings()
from cvae_amd.modules import SeqLoss  # noqa: E402,F401
