"""Two ranks on ONE GPU (gloo carries the CUDA tensors through the host): the product's
sample-sharded IW-NLL (settings.shard_samples) equals the single-process estimate
and is identical on both ranks.  RCCL itself needs one GPU per rank and is exercised by
`bench.py --gpus N` on the multi-GPU node."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'coupled-vae_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cvae_amd import _lib, data, harness, modules, settings
        dev = torch.device('cuda:0')
        _lib.init(0)
        s = dict(data.SHAPES['ptb'])
        s['V'] = 2000
        out = {}
        for cell in ('gru', 'lstm'):
            hp = harness.HParams(model='vae-couple', cell=cell, V=s['V'], E=s['E'], H=s['H'], Z=s['Z'])
            tr = harness.Trainer(hp, dev, seed=4)                       # same weights on both ranks
            B, K = 16, 40                                               # 20 samples x 16 sentences = 320 rows per rank
            c = data.Corpus(data.synthetic_sentences(B, s['V'], s['mu'], s['sigma'], 30, seed=6))
            batch = harness.text_batch(c, range(B), dev, truncate=False)
            tr.model.train(False)
            res = {}
            for mode in ('sharded', 'single'):
                settings.shard_samples = (mode == 'sharded')
                # rank 1 draws a different eps sequence; sharded mode must broadcast rank 0's
                torch.manual_seed(100 if (mode == 'single' or rank == 0) else 101)
                torch.cuda.manual_seed(100 if (mode == 'single' or rank == 0) else 101)
                xent, nll, kl, _ = tr.model.test_lm(bare=batch['bare'], sent_len=batch['len'], go=batch['go'],
                                                    eos=batch['eos'], n_sample=K)
                res[mode] = torch.stack([xent, nll, kl]).cpu()
            settings.shard_samples = False
            assert _lib.query('cvae_persist_status') == 0
            out[cell] = (res['sharded'], res['single'])
        q.put((rank, out))
    except Exception as e:                                  # surface the failure instead of a silent hang
        import traceback
        q.put((rank, 'ERROR: %s\n%s' % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_sample_sharded_iw_nll_two_ranks():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert not isinstance(res[r], str), res[r]
    for cell in ('gru', 'lstm'):
        s0, one0 = res[0][cell]
        s1, _ = res[1][cell]
        assert torch.equal(s0, s1), cell                                   # identical on every rank
        rel = float((s0 - one0).abs().max() / one0.abs().max())
        assert rel < 1e-5, (cell, rel)                                     # == single-process estimate


# ----------------------------------------------------------------------------- data-parallel TRAINING step
def _train_worker(rank, world, port, q):
    os.environ['CVAE_PERSISTENT_RNN'] = '0'       # two processes share one GPU here: the whole-sequence kernels of
    for p in (ROOT, os.path.join(ROOT, 'coupled-vae_amd'), os.path.join(ROOT, 'tests')):   # both cannot be co-resident
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cvae_amd import _lib, data, harness, modules, settings
        settings.dp_bucket_bytes = 48 << 10          # several buckets at this small shape (default: 16 MB)
        dev = torch.device('cuda:0')
        _lib.init(0)
        V, E, H, Z, Bl = _DP_SHAPE
        out = {}
        for cell in ('gru', 'lstm'):
            hp = harness.HParams(model='vae-couple', cell=cell, V=V, E=E, H=H, Z=Z, dropout=0.0)
            tr = harness.Trainer(hp, dev, seed=11)                  # identical initial weights on both ranks
            assert tr.world == 2
            sents = data.synthetic_sentences(world * Bl, V, 2.6, 0.5, 40, seed=12)
            c = data.Corpus(sents)
            batch = harness.text_batch(c, range(rank * Bl, (rank + 1) * Bl), dev)      # this rank's sentences
            eps = torch.randn(world * Bl, Z, generator=torch.Generator().manual_seed(13))
            mine = eps[rank * Bl:(rank + 1) * Bl].unsqueeze(0)
            # pass 0: one blocking all-reduce, learns which parameters receive gradients and arms the buckets;
            # pass 1 (same inputs, weights untouched in between): the bucketed form, collectives issued from the
            # autograd hooks while the backward pass is still running.  Both must give the same reduced gradient.
            seen = []
            for it in range(2):
                tr.iter_num = int(0.37 * hp.annealing_step) + hp.annealing_start - 1
                modules.set_noise_source(modules.ReplayNoise({'eps': [mine], 'eps_couple': [mine]}))
                try:
                    res = tr.forward_backward(batch)                # includes the all-reduce (mean) of the arena
                finally:
                    modules.set_noise_source(None)
                seen.append(tr.opt.grads.detach().clone())
            assert tr.opt.buckets is not None and len(tr.opt.buckets) >= 3, 'buckets not armed'
            assert tr.opt.issued_early >= 2, 'no bucket was issued from inside the backward pass'
            # (not bit-equal: the embedding gradient is scattered with float atomics, whose order varies run to run)
            err = float((seen[0] - seen[1]).abs().max() / seen[0].abs().max())
            assert err < 2e-6, 'bucketed all-reduce differs from the single collective: %.3g' % err
            sd = tr.model.state_dict(keep_vars=True)
            grads = {k: v.grad.detach().cpu().numpy().copy() for k, v in sd.items() if v.grad is not None}   # numpy: pickled by value
            tr.opt.step()
            torch.cuda.synchronize()
            post = {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}
            out[cell] = (float(res['tot'].detach()), grads, post)
        assert _lib.query('cvae_persist_status') == 0
        q.put((rank, out))
    except Exception as e:
        import traceback
        q.put((rank, 'ERROR: %s\n%s' % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


_DP_SHAPE = (500, 32, 64, 8, 8)       # V, E, H, Z, sentences per rank


def test_data_parallel_train_step_two_ranks():
    """The product's DP training step (harness.Trainer with world size 2: per-rank forward + backward, mean
    all-reduce of the flat gradient arena BEFORE the clip, then clip + Adam): the all-reduced gradient equals the
    oracle's gradient of the FULL 16-sentence batch, both ranks end the step with bit-identical weights, and those
    equal the oracle's clip + Adam on the full-batch gradient (CV/Experiment.py:509-522 clips the full batch)."""
    from oracle import cvae_oracle as O
    from cvae_amd import data, harness
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=200) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert not isinstance(res[r], str), res[r]
    V, E, H, Z, Bl = _DP_SHAPE
    sents = data.synthetic_sentences(2 * Bl, V, 2.6, 0.5, 40, seed=12)
    eps = torch.randn(2 * Bl, Z, generator=torch.Generator().manual_seed(13))
    bare, go, eos, ln = (torch.from_numpy(x) for x in O.make_batch([s.tolist() for s in sents], 200, truncate=True))
    full = dict(bare=bare, go=go, eos=eos, len=ln)
    for cell in ('gru', 'lstm'):
        hp = harness.HParams(model='vae-couple', cell=cell, V=V, E=E, H=H, Z=Z, dropout=0.0)
        model = harness.build_model(hp, 'cpu', seed=11)             # the workers' initial weights
        P = O.tie_aliases({k: v.detach().clone() for k, v in model.state_dict().items()})
        keys = list(model.state_dict().keys())
        for p_ in {id(v): v for v in P.values()}.values():
            p_.requires_grad_(True)
        cfg = dict(latent_dim=Z, lambda_kl=0.37, lambda_couple=hp.lambda_couple, lambda_seq_couple=1.0,
                   euclidean_dist=False)
        out = O.train_forward(P, 'vae-couple', cell, full, {'eps': eps}, cfg)
        out['tot'].backward()
        tot0, g0, post0 = res[0][cell]
        tot1, g1, post1 = res[1][cell]
        g0, g1, post0, post1 = ({k: torch.from_numpy(v) for k, v in d.items()} for d in (g0, g1, post0, post1))
        # the mean of the two ranks' totals is the full-batch total (every term is a batch mean)
        assert abs(0.5 * (tot0 + tot1) - float(out['tot'])) < 1e-4 * abs(float(out['tot']))
        for k in keys:
            if P[k].grad is None:
                continue
            ref = P[k].grad
            err = float((g0[k] - ref).abs().max() / (ref.abs().max() + 1e-30))
            assert err < 1e-3, (cell, k, err)
            assert torch.equal(g0[k], g1[k]), (cell, k)             # the same reduced gradient on both ranks
        for k in keys:
            assert torch.equal(post0[k], post1[k]), (cell, k)       # bit-identical weights after the step
        grads = {k: P[k].grad.detach() for k in keys if P[k].grad is not None}
        with torch.no_grad():
            Q = O.tie_aliases({k: v.detach().clone() for k, v in P.items()})
            O.clip_and_adam(Q, grads, {}, [k for k in keys if k in grads])
        for k in keys:
            if k in grads:
                # Adam's first step is lr * g / (|g| + eps): where |g| is at the fp32 noise floor the step is noise
                # too, so the 2e-4 bar is for elements with a resolved gradient; the rest moves by at most lr
                diff = (post0[k] - Q[k]).abs()
                solid = P[k].grad.abs() > 1e-3 * P[k].grad.abs().max()
                assert float(diff[solid].max() / (Q[k].abs().max() + 1e-30)) < 2e-4, (cell, k)
                assert float(diff.max()) <= 2.001e-3, (cell, k)


# ----------------------------------------------------------------------------- a time-out in a data-parallel job
class _IterNoise(object):
    """eps that depends only on (iteration, rank): a re-run of an iteration sees the noise of its first attempt."""

    def __new__(cls, tr, rank):
        from cvae_amd import modules

        class N(modules.NoiseSource):
            def normal(self, shape, device, name):
                g = torch.Generator().manual_seed(1000 * tr.iter_num + 10 * rank + (1 if name == 'eps' else 2))
                return torch.randn(shape, generator=g).to(device)
        return N()


def _fallback_worker(rank, world, port, q):
    os.environ['CVAE_PERSISTENT_RNN'] = '0'       # two processes share one GPU (see _train_worker)
    for p in (ROOT, os.path.join(ROOT, 'coupled-vae_amd'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cvae_amd import _lib, data, harness, modules, settings
        settings.dp_bucket_bytes = 48 << 10
        settings.dp_fallback_steps = 2
        dev = torch.device('cuda:0')
        _lib.init(0)
        _lib.query('cvae_deterministic', 1)           # bit-reproducible steps: the two runs below must agree exactly
        V, E, H, Z, Bl = _DP_SHAPE
        hp = harness.HParams(model='vae-couple', cell='lstm', V=V, E=E, H=H, Z=Z, dropout=0.0, annealing_start=0,
                             annealing_step=10)
        n_steps = 8
        c = data.Corpus(data.synthetic_sentences(world * Bl * n_steps, V, 2.6, 0.5, 30, seed=21))
        batches = [harness.text_batch(c, range((world * i + rank) * Bl, (world * i + rank + 1) * Bl), dev)
                   for i in range(n_steps)]
        out = {}
        for tag, fail_at in (('clean', None), ('timeout', 3)):
            tr = harness.Trainer(hp, dev, seed=11)
            assert _lib.query('cvae_persist_headroom', -1) == settings.dp_persist_headroom
            modules.set_noise_source(_IterNoise(tr, rank))
            try:
                for i, b in enumerate(batches):
                    if fail_at is not None and i + 1 == fail_at and rank == 1:
                        torch.cuda.synchronize()
                        _lib.call('cvae_persist_status_set', 1)      # what a timed-out kernel of THIS rank would do
                    tr.train_step(b)
                tr.dp_flush()
            finally:
                modules.set_noise_source(None)
            torch.cuda.synchronize()
            assert _lib.query('cvae_persist_status') == 0
            out[tag] = (list(tr.fallbacks), tr.iter_num, tr.opt.steps,
                        tr.opt.params.detach().cpu().numpy().copy(), (tr._cooldown, _lib.query('cvae_persistent_rnn', 0)))
        q.put((rank, out))
    except Exception as e:
        import traceback
        q.put((rank, 'ERROR: %s\n%s' % (e, traceback.format_exc())))
    finally:
        dist.destroy_process_group()


def test_data_parallel_timeout_falls_back_on_every_rank():
    """A whole-sequence kernel times out on ONE rank of a data-parallel job (its status word is raised by hand before
    iteration 3): the word travels through the gradient all-reduce, so both ranks skip that step on the device, both
    notice it at the same point two steps later, switch to the per-timestep kernels at run time and re-run the skipped
    iterations; they report the same fallback iteration, end bit-identical, and -- the host's counters having been
    rolled back -- end exactly where a run without the time-out ends (deterministic mode)."""
    import numpy as np
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=200) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert not isinstance(res[r], str), res[r]
    for r in (0, 1):
        fb, it, steps, _, persist_on = res[r]['timeout']
        assert fb == [3], (r, fb)                               # the same iteration on both ranks
        assert it == 8 and steps == 8, (r, it, steps)           # counters as after 8 applied iterations
        assert persist_on == (0, 0)                             # cool-down over, the switch back at its previous setting
        #                                                         (off in this test: CVAE_PERSISTENT_RNN=0, one GPU for two ranks)
        assert res[r]['clean'][0] == []
    assert np.array_equal(res[0]['timeout'][3], res[1]['timeout'][3])          # replicas identical
    assert np.array_equal(res[0]['clean'][3], res[1]['clean'][3])
    assert np.array_equal(res[0]['timeout'][3], res[0]['clean'][3])            # == the run without the time-out
