"""Diagnostic: is the training step host-bound?  Prints host issue time vs synchronized wall time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
cell = sys.argv[1] if len(sys.argv) > 1 else 'lstm'
shape = dict(data.SHAPES[os.environ.get('SHAPE', 'yelp')])
if len(sys.argv) > 2:
    shape['V'] = int(sys.argv[2])       # smaller vocab = less GPU work at equal launch count
hp = harness.HParams(model='vae-couple', cell=cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'])
tr = harness.Trainer(hp, dev, seed=0); tr.iter_num = 20000
B = 32; n = 13
corpus = data.Corpus(data.synthetic_sentences(B * n, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000))
batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev) for i in range(n)]
for i in range(3): tr.train_step(batches[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3, 13): tr.train_step(batches[i])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host issue %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) * 100, (t2 - t0) * 100))
# phases (synchronised)
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3
b = batches[5]
tr.opt.zero_grad()
res, t_f = timed(lambda: tr.losses(b))
_, t_b = timed(lambda: res['tot'].backward())
_, t_o = timed(lambda: tr.opt.step())
print('forward %.2f ms, backward %.2f ms, optimizer %.2f ms; T=%d' % (t_f, t_b, t_o, b['go'].shape[1]))
