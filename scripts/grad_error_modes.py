"""Diagnostic: gradient / post-step error against the reference fixtures with the split-bf16 products on and off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'coupled-vae_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
from golden_util import Golden
from test_gpu_parity import make_trainer, rel
from cvae_amd import _lib, modules
dev = torch.device('cuda:0'); _lib.init(0)
for name in sys.argv[1:] or ['dialog_cvae-couple_lstm_tiny', 'text_vae-couple_lstm_mid', 'text_vae-couple_gru_mid']:
    for mode in (1, 0):
        _lib.query('cvae_split_bf16', mode)
        g = Golden(name); m = g.meta
        tr = make_trainer(g, dev)
        sd = tr.model.state_dict(keep_vars=True)
        worst_g, worst_p = (0, ''), (0, '')
        for it in range(m['n_train_steps']):
            p = 'train/%d/' % it
            batch = {k: v.to(dev) for k, v in g.batch(p).items()}
            modules.set_noise_source(modules.ReplayNoise(g.group(p + 'noise/')))
            res = tr.forward_backward(batch)
            for k in m['state_keys']:
                if k not in m['none_grad_keys']:
                    e = rel(sd[k].grad, g.t(p + 'grad/' + k))
                    if e > worst_g[0]: worst_g = (e, k)
            tr.opt.step()
            for k in m['state_keys']:
                if (p + 'post/' + k) in g:
                    e = rel(sd[k], g.t(p + 'post/' + k))
                    if e > worst_p[0]: worst_p = (e, k)
        modules.set_noise_source(None)
        print('%s split_bf16=%d  worst grad rel %.2e (%s)  worst post-step rel %.2e (%s)' % (name, mode, worst_g[0], worst_g[1], worst_p[0], worst_p[1]))
