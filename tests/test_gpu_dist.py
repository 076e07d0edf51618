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
