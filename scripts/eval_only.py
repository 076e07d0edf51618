"""Diagnostic: IW-NLL evaluation only (Yelp shape, or SHAPE=ptb ...; K samples per sentence), for rocprofv3.
NBATCH = timed batches (default 2)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness, settings
settings.iw_max_rows = int(os.environ.get('IW_MAX_ROWS', settings.iw_max_rows))
dev = torch.device('cuda:0'); _lib.init(0)
cell = sys.argv[1] if len(sys.argv) > 1 else 'lstm'
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
shape = data.SHAPES[os.environ.get('SHAPE', 'yelp')]
hp = harness.HParams(model='vae-couple', cell=cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'])
tr = harness.Trainer(hp, dev, seed=0)
B, n = 32, 1 + int(os.environ.get('NBATCH', '2'))
corpus = data.Corpus(data.synthetic_sentences(B * n, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000))
batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev) for i in range(n)]
with torch.no_grad():
    tr.iw_eval(batches[:1], K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m, _ = tr.iw_eval(batches[1:], K)
    torch.cuda.synchronize(); t1 = time.perf_counter()
print('IW-NLL K=%d: %.1f sentences/s (%.3f s for %d sentences) nll=%.3f' % (K, B * (n - 1) / (t1 - t0), t1 - t0, B * (n - 1), m['nll']))
