"""Diagnostic A/B of cvae_amd.settings flags on the training step: python scripts/ab_settings.py flag=value ... (SHAPE env)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness, settings
for a in sys.argv[1:]:
    k, v = a.split('=')
    setattr(settings, k, type(getattr(settings, k))(eval(v)))
dev = torch.device('cuda:0'); _lib.init(0)
shape = data.SHAPES[os.environ.get('SHAPE', 'yelp')]
hp = harness.HParams(model='vae-couple', cell='lstm', V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'], batch_size=32)
tr = harness.Trainer(hp, dev, seed=0)
corpus = data.Corpus(data.synthetic_sentences(256, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000))
batches = [harness.text_batch(corpus, range(i * 32, (i + 1) * 32), dev) for i in range(8)]
for i in range(5):
    tr.train_step(batches[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 60
for i in range(n):
    tr.train_step(batches[i % 8])
torch.cuda.synchronize()
print(sys.argv[1:], '%.3f ms/step' % ((time.perf_counter() - t0) / n * 1e3))
