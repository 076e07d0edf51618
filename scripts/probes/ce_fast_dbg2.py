import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops, settings
dev = torch.device('cuda:0'); _lib.init(0)
settings.ce_save_logits = False
for (T, R, E, V, lo) in ((1, 4, 200, 40, 1),):
    g = torch.Generator().manual_seed(5)
    final = torch.tanh(torch.randn(T, R, E, generator=g)).to(dev)
    W = (torch.randn(V, E, generator=g) * 0.1).to(dev)
    b = (torch.randn(V, generator=g) * 0.1).to(dev)
    eos = torch.randint(lo, V, (R, T), generator=g).to(dev)
    with torch.no_grad():
        got = ops.proj_ce(final, W, b, eos, 0)
        logits = final.double() @ W.double().t() + b.double()
        lse = torch.logsumexp(logits, -1)
        tl = logits.gather(-1, eos.t().unsqueeze(-1)).squeeze(-1)
    print((T, R, E, V), 'targets', eos.flatten().tolist(), 'got', got.tolist(), 'ref', (lse - tl).sum(0).tolist(), flush=True)
