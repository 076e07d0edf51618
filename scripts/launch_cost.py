"""Diagnostic: host cost of one kernel launch through the C ABI, and host-vs-GPU time of the
recurrent entry points in isolation (queue empty at start)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops, data, harness
from cvae_amd._lib import call, ptr, stream
dev = torch.device('cuda:0'); _lib.init(0)
x = torch.zeros(64, device=dev); y = torch.zeros(64, device=dev)
st = stream()
for n in (200, 2000):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): call('cvae_axpy', ptr(x), 1.0, 64, ptr(y), st)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('axpy x%d: host %.2f us/launch, wall %.2f us/launch' % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))

shape = dict(data.SHAPES['yelp'])
for cell in ('lstm', 'gru'):
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
            run(); run(); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print('%s %s T=%d: host %.2f ms, wall %.2f ms' % (cell, which, T, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
