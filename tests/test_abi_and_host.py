"""CPU-only checks: the C-ABI library loads and exports every symbol include/cvae_hip.h
declares (with the arity the ctypes binding uses), and the host mirror exposes the
reference's state_dict contract.  No compute calls: there is no GPU here."""
import ctypes
import os
import re

import pytest
import torch

from golden_util import Golden, names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'cvae_hip.h')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|int64_t|const char\*)\s+(cvae_\w+)\s*\(([^;]*?)\)\s*;', src, re.S):
        args = m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    from cvae_amd import _lib
    decl = declared_functions()
    assert len(decl) >= 30
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decl.items():
        assert hasattr(so, name), 'libcvae_hip.so does not export %s' % name
        if name == 'cvae_last_error':
            continue
        assert name in _lib.SIGNATURES, 'binding missing for %s' % name
        assert len(_lib.SIGNATURES[name]) == nargs, (name, len(_lib.SIGNATURES[name]), nargs)
    assert set(_lib.SIGNATURES) <= set(decl)


def test_init_fails_loudly_without_gpu():
    from cvae_amd import _lib
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.CvaeHipError):
        _lib.call('cvae_init', 0)


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, 'coupled-vae_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.hpp', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert 'oracle' not in txt.replace('# oracle', ''), os.path.join(dp, f)


@pytest.mark.parametrize('name', names())
def test_state_dict_contract(name):
    """Keys, order and shapes equal the reference's state_dict, incl. alias count."""
    from cvae_amd import harness
    g = Golden(name)
    m = g.meta
    hp = harness.HParams(model=m['model'], cell=m['cell'], V=m['V'], E=m['E'], H=m['H'], Z=m['Z'])
    model = harness.build_model(hp, 'cpu', seed=0)
    sd = model.state_dict()
    assert list(sd.keys()) == m['state_keys']
    for k in m['state_keys']:
        assert tuple(sd[k].shape) == tuple(g.np('init/' + k).shape), k
    sdk = model.state_dict(keep_vars=True)
    trainable = [v for v in sdk.values() if v.requires_grad]
    assert len(trainable) == m['trainable_entries']
    assert len({id(v) for v in trainable}) == m['trainable_unique']
    model.load_state_dict({k: g.t('init/' + k) for k in m['state_keys']}, strict=True)


def test_lambda_kl_matches_reference_formula():
    from cvae_amd import harness
    hp = harness.HParams()
    assert harness.lambda_kl(1, hp) == 0.0
    assert harness.lambda_kl(22000, hp) == 0.5
    assert harness.lambda_kl(2001, hp) == 2.5e-05
    assert harness.lambda_kl(10 ** 7, hp) == 1.0


def test_epoch_batches_plain_and_bucketed():
    """8f.2 host logic: every sentence exactly once per epoch; bucketing groups similar lengths."""
    import numpy as np
    from cvae_amd import data
    s = data.SHAPES['yelp']
    L = np.array([len(x) for x in data.synthetic_sentences(1000, s['V'], s['mu'], s['sigma'], s['cap'], seed=3)])
    for bucket in (False, True):
        bs = data.epoch_batches(L, 32, seed=5, bucket=bucket)
        assert sorted(i for b in bs for i in b) == list(range(1000))
        assert bs == data.epoch_batches(L, 32, seed=5, bucket=bucket)           # deterministic
        assert all(len(b) == 32 for b in bs[:-1]) or bucket
        assert len(data.epoch_batches(L, 32, seed=5, bucket=bucket, drop_last=True)) == 1000 // 32
    plain = np.mean([L[b].max() for b in data.epoch_batches(L, 32, seed=5)])
    buck = np.mean([L[b].max() for b in data.epoch_batches(L, 32, seed=5, bucket=True)])
    assert buck < 0.6 * plain


def test_run_reference_launcher_puts_the_dropin_first(tmp_path):
    """scripts/run_reference.py: an unchanged script of a reference tree must resolve `Modules.*` to the drop-in
    classes (Python would otherwise put the script's own directory first), run as __main__ inside the tree, and
    see its own arguments."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = tmp_path / 'Coupled-VAE'
    tree.mkdir()
    (tree / 'probe.py').write_text(textwrap.dedent('''
        import os, sys
        from Modules.Losses.SeqLoss import SeqLoss
        from Modules.vae_couple import VAE_COUPLE
        print(SeqLoss.__module__, VAE_COUPLE.__module__, os.path.basename(os.getcwd()), sys.argv[1:], __name__)
        # the two names Experiment.py imports run per-tensor unless told otherwise (the alias list must not race)
        import torch
        from torch.optim import Adam
        from torch.nn.utils import clip_grad_norm_
        w = torch.nn.Parameter(torch.ones(3))
        w.grad = torch.full((3,), 100.0)
        import warnings
        warnings.simplefilter('ignore')
        opt = Adam(params=[w, w], lr=0.5, betas=[0.9, 0.999], weight_decay=0)
        assert opt.defaults['foreach'] is False and Adam(params=[w], foreach=True).defaults['foreach'] is True
        clip_grad_norm_(parameters=[w, w], max_norm=1.0)     # norm counts the alias twice, scales it twice
        c = 1.0 / ((2 * 3 * 100.0 ** 2) ** 0.5 + 1e-6)
        assert abs(float(w.grad[0]) - 100.0 * c * c) < 1e-9, float(w.grad[0])
        opt.step()                                           # two sequential Adam steps on the alias
        assert int(opt.state[w]['step']) == 2
    '''))
    r = subprocess.run([sys.executable, os.path.join(root, 'scripts', 'run_reference.py'), str(tree), 'probe.py', 'x'],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    assert r.stdout.split() == ['cvae_amd.modules', 'cvae_amd.modules', 'Coupled-VAE', "['x']", '__main__']


def test_library_is_not_a_timing_build():
    """A library compiled with any *_EXP_* timing switch (scripts/build_variant.sh; kernels with work removed)
    reports a negative version; the product build must not be one."""
    from cvae_amd import _lib
    assert _lib.query('cvae_version') >= 100
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    guard = open(os.path.join(root, 'coupled-vae_amd', 'csrc', 'common.hpp')).read()
    for hook in ('CE_EXP_NOLOAD', 'CE_EXP_NOLDS', 'CE_EXP_NOEPI', 'GW_EXP_NOSPLIT', 'GW_EXP_NOMFMA', 'GW_EXP_NOLOAD',
                 'GW_EXP_NOLDS', 'RB_EXP_NOSPLIT', 'RB_EXP_NOEPI', 'RB_EXP_NOMFMA'):
        assert hook in guard, 'timing switch %s is not covered by the CVAE_TIMING_BUILD guard' % hook
    # every *_EXP_* name used anywhere in the kernels is one of the guarded ones
    import glob, re
    used = set()
    for f in glob.glob(os.path.join(root, 'coupled-vae_amd', 'csrc', '*.hip')):
        used |= set(re.findall(r'\b[A-Z]{2}_EXP_[A-Z]+\b', open(f).read()))
    assert used and all(u in guard for u in used), sorted(used)
