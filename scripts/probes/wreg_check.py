"""Probe: the weights-in-registers many-row step kernel against the 128-row-tile kernel on the same decoder call
(cvae_rnn_large_wreg switches between them at run time), values and time.  usage: wreg_check.py [Hd] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops
dev = torch.device('cuda:0'); _lib.init(0)
Hd = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
B, E, V, T = 32, 200, 2000, 24
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)
CELL = ops.GRU if os.environ.get('CELL', 'lstm') == 'gru' else ops.LSTM
G = 3 if CELL == ops.GRU else 4
w = [r(G * Hd, 2 * E), r(G * Hd, Hd), r(G * Hd), r(G * Hd), r(E, Hd), r(E)]
emb, init = r(V, E), r(B * K, E)
go = torch.randint(4, V, (B, T), generator=g).to(dev)
for name, live in (('all live', torch.full((T,), B, dtype=torch.int32)),
                   ('ragged', torch.tensor([max(1, B - (3 * t) // 2) for t in range(T)], dtype=torch.int32))):
    out = {}
    for mode, mn in (('tile128', 1 << 30), ('wreg', 64)):
        _lib.query('cvae_rnn_large_wreg', mn)
        for _ in range(2):
            f = ops.decoder_sorted(CELL, init, go, live, emb, w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            f = ops.decoder_sorted(CELL, init, go, live, emb, w)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        out[mode] = f
        print('%-9s %-8s decoder %.2f ms (%.1f us per step)' % (name, mode, dt * 1e3, dt / T * 1e6), flush=True)
    a, b = out['tile128'], out['wreg']
    print('%-9s max |diff| %.3e  (max |value| %.3f), finite %s' % (name, float((a - b).abs().max()), float(a.abs().max()),
                                                                  bool(torch.isfinite(b).all())), flush=True)
