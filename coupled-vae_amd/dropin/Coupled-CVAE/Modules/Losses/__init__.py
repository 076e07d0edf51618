import importlib.util
import os

_spec = importlib.util.spec_from_file_location(
    '_cvae_dropin_common', os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), '_common.py'))
_common = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_common)
_common.extend_path(__path__, 'Coupled-CVAE', 'Modules', 'Losses')
