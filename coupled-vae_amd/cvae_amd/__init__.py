"""cvae_amd: MI355X-native hot path of Coupled-VAE (train step + IW-NLL / MI evaluation).

Importing this package loads libcvae_hip.so and fails loudly if it is missing."""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is absent)
from .settings import settings  # noqa: F401
