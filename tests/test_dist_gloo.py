"""world_size-2 gloo tests of the data-parallel design (CPU, no GPU): the product's
dist_utils collectives around the CPU oracle's arithmetic.

1. mean of per-rank gradients on half batches == full-batch gradient (what Trainer does with
   the flat gradient arena before the clip);
2. sharded IW-NLL evaluation + one reduction of five sums == single-process corpus metrics;
3. sample-sharded IW-NLL (eps broadcast, K/world samples per rank, all-gather of the cross-entropy
   rows) == the single-process estimate, identical on every rank."""
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup():
    for p in (ROOT, os.path.join(ROOT, 'coupled-vae_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)


def _problem():
    from oracle import cvae_oracle as O
    torch.manual_seed(0)
    V, E, H, Z, B = 60, 16, 8, 4, 8
    P = O.tie_aliases(O.init_params('vae-couple', 'gru', V, E, H, Z, seed=3))
    sents = O.synthetic_sentences(B, V, 1.8, 0.4, 12, seed=4)
    return O, P, sents, (V, E, H, Z, B)


def _grads(O, P, sents, eps, cfg):
    import numpy as np
    bare, go, eos, ln = (torch.from_numpy(x) for x in O.make_batch(sents, 12, truncate=True))
    Q = O.tie_aliases({k: v.clone() for k, v in P.items()})
    uniq = {id(v): v for v in Q.values()}
    for v in uniq.values():
        v.requires_grad_(True)
    out = O.train_forward(Q, 'vae-couple', 'gru', dict(bare=bare, go=go, eos=eos, len=ln), {'eps': eps}, cfg)
    out['tot'].backward()
    keys = sorted(k for k in Q if Q[k].grad is not None and not k.endswith('Embedding.weight')) + ['Encoder.Embedding.weight']
    return torch.cat([Q[k].grad.reshape(-1) for k in keys]), float(out['tot'].detach())


def _worker(rank, world, port, q):
    _setup()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            'cvae_dist_utils', os.path.join(ROOT, 'coupled-vae_amd', 'cvae_amd', 'dist_utils.py'))
        du = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(du)
        O, P, sents, (V, E, H, Z, B) = _problem()
        cfg = dict(latent_dim=Z, lambda_kl=0.5, lambda_couple=0.1, lambda_seq_couple=1.0, euclidean_dist=False)
        eps = torch.randn(B, Z, generator=torch.Generator().manual_seed(1))
        half = B // world
        sl = slice(rank * half, (rank + 1) * half)
        g_local, _ = _grads(O, P, sents[sl], eps[sl], cfg)
        flat = g_local.clone()
        du.allreduce_mean_(flat, n_buckets=3)
        g_full, _ = _grads(O, P, sents, eps, cfg)
        err = float((flat - g_full).abs().max() / g_full.abs().max())
        # sharded evaluation: 4 batches of 2 sentences, K = 4
        K, Bb = 4, 2
        nb = B // Bb
        eps_k = torch.randn(nb, K, Bb, Z, generator=torch.Generator().manual_seed(2))

        def eval_batches(idx):
            acc = torch.zeros(5, dtype=torch.float64)
            for b in idx:
                bare, go, eos, ln = (torch.from_numpy(x) for x in O.make_batch(sents[b * Bb:(b + 1) * Bb], 12, True))
                with torch.no_grad():
                    xent, nll, kl, _ = O.iw_nll(P, 'vae-couple', 'gru', dict(bare=bare, go=go, eos=eos, len=ln), eps_k[b])
                acc += torch.tensor([nll.sum(), xent.sum(), kl.sum(), (ln + 1).sum(), nll.numel()], dtype=torch.float64)
            return acc

        mine = du.shard_batches(nb)
        assert mine == list(range(rank, nb, world))
        acc = du.reduce_sums_(eval_batches(mine))
        ref = eval_batches(range(nb))
        err2 = float((acc - ref).abs().max() / ref.abs().max())
        ppl = math.exp(acc[0] / acc[3])
        # sample-sharded IW-NLL (preferred): same batch and eps (broadcast from rank 0)
        # everywhere, K/world samples decoded per rank, (K/world, B) rows all-gathered
        K3 = 8
        bare, go, eos, ln = (torch.from_numpy(x) for x in O.make_batch(sents[:4], 12, True))
        batch = dict(bare=bare, go=go, eos=eos, len=ln)
        eps3 = torch.randn(K3, 4, Z, generator=torch.Generator().manual_seed(10 + rank))   # differs per rank
        du.broadcast_(eps3)
        lo, hi = du.sample_slice(K3)
        with torch.no_grad():
            rows = du.allgather_rows(O.iw_xent_rows(P, 'vae-couple', 'gru', batch, eps3, lo, hi))
            got = O.iw_from_rows(P, 'vae-couple', 'gru', batch, eps3, rows)
            eps_ref = torch.randn(K3, 4, Z, generator=torch.Generator().manual_seed(10))
            want = O.iw_nll(P, 'vae-couple', 'gru', batch, eps_ref)
        err3 = max(float((a - b).abs().max()) for a, b in zip(got[:3], want[:3]))
        q.put((rank, err, err2, ppl, err3))
    finally:
        dist.destroy_process_group()


def test_dp_gradient_mean_and_sharded_eval_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, err2, ppl, err3 in res:
        assert err < 1e-5, ('gradient mean', rank, err)
        assert err2 < 1e-12, ('eval sums', rank, err2)
        assert err3 < 1e-12, ('sample-sharded IW-NLL', rank, err3)
    assert abs(res[0][3] - res[1][3]) < 1e-9       # every rank ends with identical corpus metrics
