"""Diagnostic: greedy decoding (sample_from_prior path) at the Yelp shape: 32 sentences x 200 steps."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
shape = data.SHAPES['yelp']
for cell in ('gru', 'lstm'):
    hp = harness.HParams(model='vae-couple', cell=cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'], max_len=200)
    tr = harness.Trainer(hp, dev, seed=0)
    tr.model.train(False)
    B = 32
    go = torch.full((B, 1), 1, dtype=torch.int64, device=dev)
    z = torch.randn(B, shape['Z'], device=dev)
    tr.model.decode_ids(z, go); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        ids = tr.model.decode_ids(z, go)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print('%s greedy: %.2f ms per batch (32 x 200 steps) = %.0f tokens/s, %.1f us/step' % (cell, dt * 1e3, B * 200 / dt, dt / 200 * 1e6))
