import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, ops, settings
dev = torch.device('cuda:0'); _lib.init(0)
settings.ce_save_logits = False
for (T, R, E, V) in ((3, 100, 200, 20001), (3, 100, 200, 20002), (3, 100, 200, 20016), (3, 100, 200, 20031), (3, 100, 200, 20033), (3, 100, 200, 33), (3, 100, 200, 1025), (3, 100, 200, 10241)):
    g = torch.Generator().manual_seed(5)
    final = torch.tanh(torch.randn(T, R, E, generator=g)).to(dev)
    W = (torch.randn(V, E, generator=g) * 0.1).to(dev)
    b = (torch.randn(V, generator=g) * 0.1).to(dev)
    eos = torch.randint(1, V, (R, T), generator=g).to(dev)
    with torch.no_grad():
        got = ops.proj_ce(final, W, b, eos, 0)
        logits = final.double() @ W.double().t() + b.double()
        ref = (torch.logsumexp(logits, -1) - logits.gather(-1, eos.t().unsqueeze(-1)).squeeze(-1)).sum(0)
    bad = ~torch.isfinite(got)
    print((T, R, E, V), 'non-finite', int(bad.sum()), 'of', got.numel(), 'max rel err (finite)', float(((got - ref).abs() / ref.abs())[~bad].max()) if (~bad).any() else None, flush=True)
