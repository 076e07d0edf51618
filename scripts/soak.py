"""Soak: many training steps over varied batches (both cells), checking that every loss is finite, the
whole-sequence kernels never time out (cvae_persist_status) and IW-NLL evaluations interleave cleanly."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for cell in ('lstm', 'gru'):
    shape = data.SHAPES['yelp']
    hp = harness.HParams(model='vae-couple', cell=cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'])
    tr = harness.Trainer(hp, dev, seed=1)
    B = 32
    corpus = data.Corpus(data.synthetic_sentences(B * 64, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=77))
    batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev, sort_rows=(i % 2 == 0)) for i in range(64)]
    ev = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev, truncate=False) for i in range(2)]
    t0 = time.perf_counter()
    tots = []
    for i in range(n):
        tots.append(tr.train_step(batches[i % 64])['tot'].detach())
        if i % 100 == 99:
            m, _ = tr.iw_eval(ev[:1], 20)
            vals = torch.stack(tots).cpu(); tots = []
            st = _lib.query('cvae_persist_status')
            assert st == 0, 'whole-sequence kernel timed out'
            assert bool(torch.isfinite(vals).all()) and m['nll'] == m['nll'], 'non-finite loss'
            print('%s step %d: loss %.3f  iw-nll %.3f  (%.1f s)' % (cell, i + 1, float(vals[-1]), m['nll'], time.perf_counter() - t0), flush=True)
print('soak ok')
