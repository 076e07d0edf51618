"""Probe: per-sentence IW-NLL of the same batches with the round-3 evaluation kernels on and off (the weights-in-registers
decoder step; the evaluation form of the CE core is switched by CVAE_CE_FAST_EVAL in the environment)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
cell = sys.argv[1] if len(sys.argv) > 1 else 'lstm'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
shape = data.SHAPES[os.environ.get('SHAPE', 'yelp')]
hp = harness.HParams(model='vae-couple', cell=cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'])
tr = harness.Trainer(hp, dev, seed=0)
B, n = 32, 5
corpus = data.Corpus(data.synthetic_sentences(B * n, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=123))
batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev) for i in range(n)]
res = {}
for name, mn in (('on', 64), ('off', 1 << 30)):
    _lib.query('cvae_rnn_large_wreg', mn)
    torch.manual_seed(7)
    with torch.no_grad():
        m, per = tr.iw_eval(batches, K)
    res[name] = (m, per)
    print(name, {k: round(float(v), 6) for k, v in m.items()}, flush=True)
a, b = res['on'][1], res['off'][1]
for i, name in enumerate(('xent', 'nll', 'kl')):
    x = torch.cat([t[i].double().flatten() for t in a]); y = torch.cat([t[i].double().flatten() for t in b])
    print('%s: max rel diff per sentence %.3e (%d sentences)' % (name, float(((x - y).abs() / y.abs().clamp_min(1e-9)).max()), x.numel()))
