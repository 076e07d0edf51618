"""Diagnostic: where does the HOST time of a training step go?  Accumulates wall time spent inside
each C-ABI call (launch issue cost, no sync) and prints a cProfile summary of the Python side."""
import cProfile, collections, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness, ops
dev = torch.device('cuda:0'); _lib.init(0)
cell = sys.argv[1] if len(sys.argv) > 1 else 'lstm'
shape = dict(data.SHAPES['yelp'])
if len(sys.argv) > 2:
    shape['V'] = int(sys.argv[2])
hp = harness.HParams(model='vae-couple', cell=cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'])
tr = harness.Trainer(hp, dev, seed=0); tr.iter_num = 20000
B = 32; n = 13
corpus = data.Corpus(data.synthetic_sentences(B * n, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000))
batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev) for i in range(n)]
for i in range(3): tr.train_step(batches[i])
torch.cuda.synchronize()

acc = collections.defaultdict(lambda: [0, 0.0])
real_call = ops.call
def timed_call(name, *a):
    t = time.perf_counter()
    r = real_call(name, *a)
    e = acc[name]; e[0] += 1; e[1] += time.perf_counter() - t
    return r
ops.call = timed_call
t0 = time.perf_counter()
for i in range(3, 13): tr.train_step(batches[i])
t1 = time.perf_counter()
torch.cuda.synchronize()
print('host issue %.2f ms/step' % ((t1 - t0) * 100))
tot = 0.0
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print('  %-34s %5.1f calls/step  %8.3f ms/step  %7.1f us/call' % (k, c / 10, t * 100, t / c * 1e6)); tot += t
print('  total inside C ABI: %.2f ms/step' % (tot * 100))
ops.call = real_call
pr = cProfile.Profile(); pr.enable()
for i in range(3, 13): tr.train_step(batches[i])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(25)
