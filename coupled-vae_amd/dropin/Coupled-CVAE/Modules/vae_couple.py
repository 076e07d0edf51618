"""Replaces Coupled-CVAE/Modules/vae_couple.py (class VAE_COUPLE = coupled conditional VAE)."""
from Modules import _common
_common.sync_settings()
from cvae_amd.modules import CVAE_COUPLE as VAE_COUPLE, Decoder, Gaussian  # noqa: E402,F401
from cvae_amd.settings import settings as _settings  # noqa: E402

_n_sample = _settings.n_sample_chunk   # CC/Modules/vae_couple.py:20 (imported by the reference's seq2seq.py / vae.py)
