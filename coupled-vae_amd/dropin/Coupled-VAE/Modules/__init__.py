"""Drop-in `Modules` package for Coupled-VAE: put this directory's parent on sys.path AHEAD of the
reference tree.  The classes of the hot path come from cvae_amd (MI355X kernels); everything
else falls through to the reference's own Modules directory ($COUPLED_VAE_REFERENCE/Coupled-VAE/Modules)."""
import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    '_cvae_dropin_common', os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), '_common.py'))
_common = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_common)
_common.extend_path(__path__, 'Coupled-VAE', 'Modules')
