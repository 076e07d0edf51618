"""Diagnostic: phase cycle stamps of the whole-sequence recurrent kernels (CVAE_PERSIST_STAMPS=1)."""
import os, sys, time
os.environ['CVAE_PERSIST_STAMPS'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
shape = dict(data.SHAPES['yelp'])
for cell in ('lstm',):
    hp = harness.HParams(model='vae-couple', cell=cell, V=2000, E=shape['E'], H=shape['H'], Z=shape['Z'], dropout=0.0)
    tr = harness.Trainer(hp, dev, seed=0)
    m = tr.model
    B, T = 32, 200
    bare = torch.randint(4, 2000, (B, T), device=dev); ln = torch.full((B,), T, device=dev, dtype=torch.int64)
    for which in ('enc', 'encbwd'):
        def run():
            h = m.Encoder.last_flat(bare, ln)
            if which == 'encbwd':
                h.sum().backward()
        with torch.set_grad_enabled(which == 'encbwd'):
            run(); run(); torch.cuda.synchronize(); _lib.query('cvae_persist_status')
            t0 = time.perf_counter(); run(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print('%s %s T=%d: wall %.2f ms' % (cell, which, T, (t2 - t0) * 1e3), flush=True)
        sys.stderr.flush()
        _lib.query('cvae_persist_status')
    # decoder pair through a train step (Hd = 512)
    corpus = data.Corpus(data.synthetic_sentences(64, 2000, shape['mu'], shape['sigma'], shape['cap'], seed=3))
    b = harness.text_batch(corpus, range(32), dev)
    tr.train_step(b); torch.cuda.synchronize(); _lib.query('cvae_persist_status')
    print('train step (enc fwd, dec-pair fwd, dec-pair bwd, enc bwd accumulated):', flush=True)
    tr.train_step(b); torch.cuda.synchronize(); _lib.query('cvae_persist_status')
