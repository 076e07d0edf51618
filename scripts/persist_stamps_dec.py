"""Diagnostic: phase stamps of the decoder pair's whole-sequence kernels, forward and backward separately
(CVAE_PERSIST_STAMPS=1; cycles of block (0,0), accumulated over the T steps of one launch)."""
import os, sys
os.environ['CVAE_PERSIST_STAMPS'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops
dev = torch.device('cuda:0'); _lib.init(0)
cellname = sys.argv[1] if len(sys.argv) > 1 else 'lstm'
cell = ops.LSTM if cellname == 'lstm' else ops.GRU
G = 4 if cell == ops.LSTM else 3
B, T, E, Hd, V = 32, 200, 200, 512, 2000
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev).requires_grad_(True)
emb = r(V, E)
go = torch.randint(4, V, (B, T), generator=g).to(dev)
specs = [(r(B, E), None, None, [r(G * Hd, 2 * E), r(G * Hd, Hd), r(G * Hd), r(G * Hd), r(E, Hd), r(E)]) for _ in range(2)]
for it in range(3):
    outs = ops.decoder_pair(cell, go, emb, specs)
    torch.cuda.synchronize()
    if it == 2: print('forward (T=%d):' % T, flush=True)
    _lib.query('cvae_persist_status')
    (outs[0].sum() + outs[1].sum()).backward()
    torch.cuda.synchronize()
    if it == 2: print('backward:', flush=True)
    _lib.query('cvae_persist_status')
