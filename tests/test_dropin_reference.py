"""Drop-in check against the reference's own harness (build container only: the reference does
not travel, so this is skipped on the GPU box).  The unmodified Experiment.py is imported with
coupled-vae_amd/dropin/<tree> ahead of the reference tree on sys.path; it must construct our
classes (not the reference's) and see the reference's state_dict contract.  Training itself
needs the GPU and is covered by tests/test_gpu_parity.py against recorded vectors."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'

CHILD = r'''
import io, os, sys, json, contextlib, tempfile, types
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import make_golden as mg
tree, dataset, model, cell = TREE, DATASET, MODEL, CELL
shp = dict(mg.TINY)
os.chdir(tempfile.mkdtemp(prefix='cvae_dropin_'))
for d in ('logs', 'saved_model', 'sampled_results', 'temp_results'):
    os.makedirs(os.path.join('outputs', d))
mg._write_corpus(tree, dataset, shp)
mg._install_stubs()
del sys.modules['Modules.vae_nf_geo']        # the reference's own files may import it: keep the stub
nf = types.ModuleType('Modules.vae_nf_geo'); nf._n_sample = 2; sys.modules['Modules.vae_nf_geo'] = nf
cfg = types.ModuleType('config')
exec(compile(mg._patched_config_source(tree, dataset, model, cell, shp), 'config.py', 'exec'), cfg.__dict__)
sys.modules['config'] = cfg
os.environ['COUPLED_VAE_REFERENCE'] = '/root/reference'
sys.path.insert(0, os.path.join('/root/reference', tree))
sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd', 'dropin', tree))
with contextlib.redirect_stdout(io.StringIO()):
    import Experiment as E
    exp = E.Experiment()
ae = exp.AutoEncoder
import cvae_amd.modules as M
print(json.dumps(dict(cls=type(ae).__module__ + '.' + type(ae).__name__, keys=list(ae.state_dict().keys()),
                      ours=isinstance(ae, (M.VAE, M.VAE_COUPLE, M.CVAE, M.CVAE_COUPLE)),
                      lstm=hasattr(ae.Decoder, 'LSTM'))))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present (GPU box)')
@pytest.mark.parametrize('tree,dataset,model,cell,golden', [
    ('Coupled-VAE', 'ptb', 'vae-couple', 'lstm', 'text_vae-couple_lstm_tiny'),
    ('Coupled-CVAE', 'switchboard', 'cvae-couple', 'gru', 'dialog_cvae-couple_gru_tiny'),
])
def test_reference_experiment_constructs_our_modules(tree, dataset, model, cell, golden):
    from golden_util import Golden
    code = CHILD.replace('ROOT', repr(ROOT)).replace('TREE', repr(tree)).replace('DATASET', repr(dataset)) \
                .replace('MODEL', repr(model)).replace('CELL', repr(cell))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['ours'], out['cls']
    assert out['lstm'] == (cell == 'lstm')
    assert out['keys'] == Golden(golden).meta['state_keys']
