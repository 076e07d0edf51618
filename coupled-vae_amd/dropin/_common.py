"""Shared by the drop-in `Modules` packages: makes cvae_amd importable, mirrors config.use_lstm
into cvae_amd.settings, and lets the reference's own files serve every module the accelerated
path does not replace (lm, dae, wae, ... are resolved from $COUPLED_VAE_REFERENCE)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)


def extend_path(path_list, tree, *sub):
    ref = os.environ.get('COUPLED_VAE_REFERENCE')
    if ref:
        cand = os.path.join(ref, tree, *sub)
        if os.path.isdir(cand) and cand not in path_list:
            path_list.append(cand)


_synced = None


def sync_settings():
    """Mirror config.use_lstm into cvae_amd.settings -- once per process: Config() has directory side effects
    (SURVEY fact 4), so the shims share one instantiation instead of one each (the reference's own modules
    instantiate it at import too)."""
    global _synced
    from cvae_amd.settings import settings
    if _synced is None:
        try:
            from config import Config          # the reference's config.py, if the caller has one
            settings.use_lstm = bool(Config().use_lstm)
            _synced = True
        except ImportError:
            _synced = False
    return settings
