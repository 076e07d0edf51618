"""Diagnostic: how long the host takes to ISSUE one training step (no synchronisation) against the time the
GPU takes to finish it, and the same for a tiny model (GPU work negligible: the step is then host-bound)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
for name, shape in (('yelp', data.SHAPES['yelp']), ('tiny', dict(V=200, E=16, H=8, Z=4, mu=4.4, sigma=0.5, cap=200))):
    hp = harness.HParams(model='vae-couple', cell='lstm', V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'])
    tr = harness.Trainer(hp, dev, seed=0)
    B = 32
    corpus = data.Corpus(data.synthetic_sentences(B * 8, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000))
    batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev, sort_rows=True) for i in range(8)]
    for i in range(5):
        tr.train_step(batches[i % 8])
    iss, tot = [], []
    for i in range(16):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.train_step(batches[i % 8])
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        iss.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print('%s: issue %.2f ms, complete %.2f ms (mean of 16 single steps)' % (name, sum(iss) / 16, sum(tot) / 16), flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(32):
        tr.train_step(batches[i % 8])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%s: 32 steps back to back: issue %.2f ms/step, complete %.2f ms/step' % (name, (t1 - t0) / 32 * 1e3, (t2 - t0) / 32 * 1e3), flush=True)

# ---- which library calls block the host when the queue is deep (Yelp shape, back to back)
import collections
from cvae_amd import ops
shape = data.SHAPES['yelp']
hp = harness.HParams(model='vae-couple', cell='lstm', V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'])
tr = harness.Trainer(hp, dev, seed=0)
corpus = data.Corpus(data.synthetic_sentences(B * 8, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000))
batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev, sort_rows=True) for i in range(8)]
for i in range(5):
    tr.train_step(batches[i % 8])
acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
orig = _lib.call
def timed(name, *a):
    t = time.perf_counter(); r = orig(name, *a); d = time.perf_counter() - t
    e = acc[name]; e[0] += d; e[1] += 1; e[2] = max(e[2], d)
    return r
ops.call = timed
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(32):
    tr.train_step(batches[i % 8])
t1 = time.perf_counter(); torch.cuda.synchronize()
print('instrumented: issue %.2f ms/step; time inside library calls per step:' % ((t1 - t0) / 32 * 1e3))
for k, (t, c, mx) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]:
    print('  %-34s %8.3f ms/step  %5.1f calls/step  max %.3f ms' % (k, t / 32 * 1e3, c / 32, mx * 1e3))
