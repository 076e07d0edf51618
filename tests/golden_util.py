"""Loader for tests/golden/*.npz (written by oracle/make_golden.py from the reference)."""
import glob
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def names(prefix=''):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, prefix + '*.npz')))


class Golden(object):
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)
        self.meta = json.loads(bytes(self.z['meta']).decode())
        self.name = name

    def __contains__(self, k):
        return k in self.z.files

    def np(self, k):
        return self.z[k]

    def t(self, k, dtype=None):
        a = torch.from_numpy(np.ascontiguousarray(self.z[k]))
        if dtype is not None and a.is_floating_point():
            a = a.to(dtype)
        return a

    def params(self, prefix, dtype=torch.float32):
        """State dict under `prefix` ('init/' or 'eval/'), embedding aliases tied."""
        P = {}
        for k in self.meta['state_keys']:
            P[k] = self.t(prefix + k, dtype).clone()
        emb = [k for k in P if k.endswith('Embedding.weight')]
        for k in emb[1:]:
            P[k] = P[emb[0]]
        return P

    def group(self, prefix, dtype=None):
        n = len(prefix)
        return {k[n:]: self.t(k, dtype) for k in self.z.files if k.startswith(prefix)}

    def batch(self, prefix):
        keys = ('bare', 'go', 'eos', 'len', 'post_bare', 'post_len', 'resp_bare', 'resp_go',
                'resp_eos', 'resp_len')
        return {k: self.t(prefix + k) for k in keys if (prefix + k) in self}

    def cfg(self, lambda_kl=0.0):
        m = self.meta
        return dict(latent_dim=m['Z'], lambda_kl=lambda_kl, lambda_couple=m['lambda_couple'],
                    lambda_seq_couple=m['lambda_seq_couple'], euclidean_dist=m['euclidean_dist'],
                    lr=m['lr'], clip_norm=m['clip_norm'], lambda_fb=m.get('lambda_fb', 0.0),
                    lambda_bow=m.get('lambda_BoW', 0.0))

    def none_grad_keys(self, it):
        """state_dict keys whose gradient is None in the reference at recorded iteration `it` (the BoW head while the
        bag-of-words term is left out of the total, CV/Experiment.py:711-713)."""
        per = self.meta.get('none_grad_keys_per_step')
        return per[it] if per else self.meta['none_grad_keys']

    def kl_per_sentence(self, prefix):
        """The KL term of every sentence as the recorded model's training loop forms it from the recorded
        kl_divergence(q, p) elements: their sum, or for surprising-fix* the sum of the elements floored at
        lambda_fb / Z (free bits, CV/Experiment.py:1141-1142)."""
        k = self.t(prefix + 'kl_elem')
        if self.meta['model'].startswith('surprising-fix'):
            k = torch.clamp(k, min=self.meta['lambda_fb'] / self.meta['Z'])
        return k.sum(1)
