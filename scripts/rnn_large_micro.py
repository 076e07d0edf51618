"""Diagnostic: the many-row recurrent step (rnn_large_step_kernel) at the Yelp IW-eval shape, every row live:
B = 32 sentences x K samples, Hd = 512, T steps.  For rocprofv3 --kernel-trace / --pmc.
usage: python scripts/rnn_large_micro.py [cell] [K] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops
dev = torch.device('cuda:0'); _lib.init(0)
cell = ops.LSTM if (sys.argv[1] if len(sys.argv) > 1 else 'lstm') == 'lstm' else ops.GRU
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
B, E, Hd, V = 32, 200, 512, 2000
G = 4 if cell == ops.LSTM else 3
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)
w = [r(G * Hd, 2 * E), r(G * Hd, Hd), r(G * Hd), r(G * Hd), r(E, Hd), r(E)]
emb, init = r(V, E), r(B * K, E)
go = torch.randint(4, V, (B, T), generator=g).to(dev)
live = torch.full((T,), B, dtype=torch.int32)
for _ in range(3):
    ops.decoder_sorted(cell, init, go, live, emb, w)
torch.cuda.synchronize()
