"""Diagnostic: training steps only (Yelp shape; SHAPE=scaleup BATCH=64 for one GPU's share of the scale-up shape),
for rocprofv3 --kernel-trace."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
cell = sys.argv[1] if len(sys.argv) > 1 else 'lstm'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shape = data.SHAPES[os.environ.get('SHAPE', 'yelp')]
hp = harness.HParams(model=os.environ.get('MODEL', 'vae-couple'), cell=cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'], batch_size=int(os.environ.get('BATCH', '32')))
tr = harness.Trainer(hp, dev, seed=0)
B = int(os.environ.get('BATCH', '32'))
corpus = data.Corpus(data.synthetic_sentences(B * 8, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000))
sort_rows = os.environ.get('SORT_ROWS', '0') == '1'
if hp.is_dialog():       # MODEL=cvae-couple: posts and responses from two synthetic corpora
    resps = data.Corpus(data.synthetic_sentences(B * 8, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1001))
    batches = [harness.dialog_batch(corpus, resps, range(i * B, (i + 1) * B), dev) for i in range(8)]
else:
    batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev, sort_rows=sort_rows) for i in range(8)]
for i in range(5):
    tr.train_step(batches[i % 8])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n):
    tr.train_step(batches[i % 8])
torch.cuda.synchronize(); t1 = time.perf_counter()
print('%s: %.3f ms/step' % (cell, (t1 - t0) / n * 1e3))
