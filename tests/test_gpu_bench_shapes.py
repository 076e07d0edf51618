"""Parity at the shapes bench.py times and BASELINE.json's configs name (V cut so the CPU oracle stays
in seconds; the vocabulary only scales the CE kernel, which has its own V = 20000 case below):

* configs[1] Yelp shape (H=256 -> Hd=512, E=200, B=32): the two lock-stepped 512-unit decoders on the
  whole-sequence kernels (2 x 128 forward blocks, 2 x 64 backward blocks) over T = 201 steps, i.e. the
  double-buffered sc1 exchange re-read 100 times per buffer; rows unsorted (the reference's order) and
  sorted by length (what bench.py used to feed);
* IW-NLL at Hd=512 with B*K > 256: rnn_large_step_kernel / gemm_nt_batched_kernel at H=512;
* CE forward_train / backward_saved at V=20000 with ragged live rows;
* configs[2] Yahoo shape with the Euclidean coupling loss;
* configs[4] scale-up share: H=1024 (Hd=2048), B=64 per GPU.

Same bars as test_gpu_parity.py: losses 1e-4, gradients 1e-3 relative."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cvae_oracle as O                                             # noqa: E402  (checker only)
from test_gpu_parity import _train_step_vs_oracle, _chunks, rel, LOSS_TOL       # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from cvae_amd import _lib
    _lib.init(0)
    return torch.device('cuda:0')


def _bench_sentences(shape, B, V, seed, with_200=True):
    """The bench's length distribution, plus one sentence at the cap and one of a single token."""
    from cvae_amd import data
    s = data.SHAPES[shape]
    sents = data.synthetic_sentences(2 * B, V, s['mu'], s['sigma'], s['cap'], seed=seed)
    if with_200:
        rs = np.random.RandomState(seed + 1)
        sents[3] = rs.randint(4, V, size=200).astype(np.int64)
        sents[7] = rs.randint(4, V, size=1).astype(np.int64)
        sents[B + 5] = rs.randint(4, V, size=200).astype(np.int64)
    return sents


@pytest.mark.parametrize('sort_rows', [False, True])
@pytest.mark.parametrize('cell', ['lstm', 'gru'])
def test_train_step_vs_oracle_yelp_shape(dev, cell, sort_rows):
    from cvae_amd import data
    s = dict(data.SHAPES['yelp'])
    s['V'] = 2000
    B = 32
    sents = _bench_sentences('yelp', B, s['V'], seed=21)
    _train_step_vs_oracle(dev, 'vae-couple', cell, s, sents[:B], sents[B:], sort_rows=sort_rows)


def test_train_step_vs_oracle_yahoo_shape_euclid(dev):
    from cvae_amd import data
    s = dict(data.SHAPES['yahoo'])
    s['V'] = 2001                       # odd, as Yahoo's 20001
    B = 32
    sents = _bench_sentences('yahoo', B, s['V'], seed=22)
    _train_step_vs_oracle(dev, 'vae-couple', 'gru', s, sents[:B], sents[B:], euclid=True)


def test_train_step_vs_oracle_switchboard_lstm(dev):
    """configs[3] with the LSTM cell (test_gpu_parity covers the GRU)."""
    from cvae_amd import data
    s = dict(data.SHAPES['switchboard'])
    s['V'] = 2000
    B = 32
    sents = data.synthetic_sentences(2 * B, s['V'], s['mu'], s['sigma'], 60, seed=23)
    _train_step_vs_oracle(dev, 'cvae-couple', 'lstm', s, sents[:B], sents[B:])


@pytest.mark.parametrize('cell', ['lstm', 'gru'])
def test_train_step_vs_oracle_scaleup_share(dev, cell):
    """One GPU's share of configs[4]: Hd = 2048, 64 rows (outside the whole-sequence kernels' range)."""
    from cvae_amd import data
    s = dict(data.SHAPES['scaleup'])
    s['V'] = 1000
    B = 64
    sents = data.synthetic_sentences(2 * B, s['V'], 3.0, 0.4, 30, seed=24)
    _train_step_vs_oracle(dev, 'vae-couple', cell, s, sents[:B], sents[B:])


@pytest.mark.parametrize('shape,cell,B,K', [('yelp', 'lstm', 16, 20), ('yelp', 'gru', 16, 20), ('yelp', 'lstm', 32, 40),
                                            ('ptb', 'lstm', 16, 20), ('ptb', 'gru', 16, 20), ('scaleup', 'lstm', 8, 40)])
def test_iw_nll_vs_oracle_bench_shapes(dev, shape, cell, B, K):
    """IW-NLL with B*K > 256 rows at Hd = 512 (and 2048): the many-row recurrence (rnn_large_wreg_kernel at Hd = 512,
    both cells -- the default from 64 live rows; rnn_large_step_kernel below that and at Hd = 2048), the batched W_c
    product and the CE kernel's evaluation form on sentence-major sorted rows, against the oracle."""
    from cvae_amd import modules, data, harness
    s = dict(data.SHAPES[shape])
    s['V'] = 2000
    hp = harness.HParams(model='vae-couple', cell=cell, V=s['V'], E=s['E'], H=s['H'], Z=s['Z'])
    tr = harness.Trainer(hp, dev, seed=4)
    cap = 200 if shape == 'yelp' else 40
    sents = data.synthetic_sentences(B, s['V'], s['mu'] if shape == 'yelp' else 3.0, s['sigma'], cap, seed=31)    # ('ptb': Hd = 256)
    if shape == 'yelp':
        sents[2] = np.random.RandomState(5).randint(4, s['V'], size=200).astype(np.int64)
    c = data.Corpus(sents)
    batch = harness.text_batch(c, range(B), dev, truncate=False)
    eps = torch.randn(K, B, s['Z'], generator=torch.Generator().manual_seed(2))
    P = O.tie_aliases({k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()})
    cb = {k: v.cpu() for k, v in batch.items()}
    m = int(cb['len'].max())
    cb_t = dict(bare=cb['bare'][:, :m], go=cb['go'][:, :m + 1], eos=cb['eos'][:, :m + 1], len=cb['len'])
    with torch.no_grad():
        xent, nll, kl, _ = O.iw_nll(P, 'vae-couple', cell, cb_t, eps)
    tr.model.train(False)
    modules.set_noise_source(modules.ReplayNoise({'iw_eps': _chunks(eps)}))
    try:
        got = tr.model.test_lm(bare=batch['bare'], sent_len=batch['len'], go=batch['go'], eos=batch['eos'], n_sample=K)
    finally:
        modules.set_noise_source(None)
    assert rel(got[0], xent) < LOSS_TOL and rel(got[1], nll) < LOSS_TOL
    assert rel(got[2], kl) < LOSS_TOL


def test_projce_saved_logits_v20000_ragged(dev):
    """ce_core_kernel<SAVE> + ce_grad_from_logits_kernel at the bench's vocabulary (V = 20000, E = 200) with a
    realistic live-row pattern (ragged lengths, about half of the T*R rows are <pad>), forward and every
    gradient against the oracle in fp64; and the recompute variant must agree with it."""
    from cvae_amd import ops, settings
    T, R, E, V = 64, 32, 200, 20000
    g = torch.Generator().manual_seed(5)
    rs = np.random.RandomState(3)
    lens = np.clip(np.floor(rs.lognormal(3.3, 0.5, size=R)), 1, T).astype(np.int64)
    lens[0], lens[1] = T, 1
    final = torch.tanh(torch.randn(T, R, E, generator=g))
    W = torch.randn(V, E, generator=g) * 0.1
    b = torch.randn(V, generator=g) * 0.1
    eos = torch.randint(4, V, (R, T), generator=g)
    eos[torch.arange(T)[None, :] >= torch.from_numpy(lens)[:, None]] = 0
    gs = torch.randn(R, generator=g)
    f64 = final.double().requires_grad_(True)
    P = {'D.ToVocab.weight': W.double().requires_grad_(True), 'D.ToVocab.bias': b.double().requires_grad_(True)}
    ref = O.seq_xent(P, 'D.', f64, eos)
    (ref * gs.double()).sum().backward()
    try:
        for save in (True, False):
            settings.ce_save_logits = save
            fd, Wd, bd = (t.to(dev).requires_grad_(True) for t in (final, W, b))
            got = ops.proj_ce(fd, Wd, bd, eos.to(dev), 0)
            assert rel(got, ref) < 2e-5
            (got * gs.to(dev)).sum().backward()
            assert rel(fd.grad, f64.grad) < 1e-4
            assert rel(Wd.grad, P['D.ToVocab.weight'].grad) < 1e-4
            assert rel(bd.grad, P['D.ToVocab.bias'].grad) < 1e-4
    finally:
        settings.ce_save_logits = True


def test_resident_batches_back_to_back_behind_a_long_kernel(dev):
    """Corpus.batch_resident stages the batch's indices through pinned memory with an asynchronous copy: many
    batches issued back-to-back while the stream is busy must each be packed from their OWN indices."""
    from cvae_amd import data
    s = data.SHAPES['ptb']
    c = data.Corpus(data.synthetic_sentences(640, s['V'], s['mu'], s['sigma'], 40, seed=2), max_len=30).to_device(dev)
    want = [c.batch(range(i * 32, (i + 1) * 32), dev, True) for i in range(20)]
    torch.cuda.synchronize()
    a = torch.randn(4096, 4096, device=dev)
    for _ in range(30):                        # ~30 x 137 GFLOP of queued work: the host runs far ahead
        a = torch.mm(a, a) * 1e-3
    got = [c.batch_resident(range(i * 32, (i + 1) * 32), True) for i in range(20)]
    torch.cuda.synchronize()
    for w, g in zip(want, got):
        for x, y in zip(w, g):
            assert torch.equal(x, y)


def test_persist_timeout_is_loud_and_the_step_is_not_applied(dev):
    """A timed-out whole-sequence kernel raises the status word (host-mapped).  With the word set -- here from the
    host, as the kernel would -- Trainer.train_step must raise, cvae_clip_adam must leave every parameter as it was,
    and cvae_persist_status must report and clear it; the next step then runs and updates the weights."""
    from cvae_amd import data, harness, _lib
    hp = harness.HParams(model='vae-couple', cell='gru', V=300, E=24, H=128, Z=8)
    tr = harness.Trainer(hp, dev, seed=1)
    c = data.Corpus(data.synthetic_sentences(16, hp.V, 2.5, 0.4, 20, seed=3))
    batch = harness.text_batch(c, range(16), dev)
    tr.train_step(batch)
    before = tr.opt.params.clone()
    _lib.call('cvae_persist_status_set', 1)
    try:
        assert _lib.query('cvae_persist_status_peek') == 1
        with pytest.raises(_lib.CvaeHipError, match='timed out'):
            tr.train_step(batch)
        torch.cuda.synchronize()
        assert torch.equal(tr.opt.params, before), 'a step was applied although the status word was set'
    finally:
        _lib.query('cvae_persist_status')                 # report + clear (check_kernels has already done so)
    assert _lib.query('cvae_persist_status_peek') == 0
    tr.train_step(batch)
    torch.cuda.synchronize()
    assert not torch.equal(tr.opt.params, before)


@pytest.mark.gpu
def test_deterministic_mode_is_bit_reproducible(dev):
    """cvae_deterministic(1): three training steps from the same seed give bit-identical parameters, losses and clip
    norms on two runs (the embedding-gradient scatter, the only float-atomic accumulation of a training step, goes
    through fixed point), and agree with the default mode to fp32 rounding.  PTB shape, vocabulary cut to 600 so that
    tokens repeat many times within a batch."""
    from cvae_amd import data, harness, _lib
    s = data.SHAPES['ptb']
    hp = harness.HParams(model='vae-couple', cell='lstm', V=600, E=s['E'], H=s['H'], Z=s['Z'], batch_size=32)
    c = data.Corpus(data.synthetic_sentences(96, 600, s['mu'], s['sigma'], s['cap'], seed=41))

    def run():
        tr = harness.Trainer(hp, dev, seed=9)
        torch.manual_seed(123)
        out = []
        for i in range(3):
            res = tr.train_step(harness.text_batch(c, range(32 * i, 32 * i + 32), dev))
            out.append((float(res['tot']), float(tr.opt.norm)))
        return out, tr.opt.params.detach().cpu().clone()

    prev = _lib.query('cvae_deterministic', 1)
    try:
        a, pa = run()
        b, pb = run()
        assert a == b and torch.equal(pa, pb)
        _lib.query('cvae_deterministic', 0)
        d, pd = run()
    finally:
        _lib.query('cvae_deterministic', prev)
    assert all(abs(x[0] - y[0]) < 1e-5 * abs(y[0]) for x, y in zip(a, d))
    assert float((pa - pd).abs().max()) < 1e-5


@pytest.mark.gpu
def test_deterministic_mode_propagates_non_finite_gradients(dev):
    """A diverged step must stay visible: with a NaN weight in the encoder the embedding gradient and the clip norm are
    NaN in the deterministic mode exactly as in the default one (the fixed-point scatter takes finite addends within
    its range only; anything else is added as a float)."""
    from cvae_amd import data, harness, _lib
    hp = harness.HParams(model='vae-couple', cell='lstm', V=300, E=32, H=128, Z=8, batch_size=16, dropout=0.0)
    c = data.Corpus(data.synthetic_sentences(16, 300, 2.3, 0.4, 20, seed=3))
    prev = _lib.query('cvae_deterministic', 0)
    try:
        for det in (0, 1):
            _lib.query('cvae_deterministic', det)
            tr = harness.Trainer(hp, dev, seed=1)
            tr.model.Encoder.LSTM.weight_ih_l0.data[0, 0] = float('nan')
            tr.forward_backward(harness.text_batch(c, range(16), dev))
            emb_grad = tr.model.Encoder.Embedding.weight.grad
            assert bool(torch.isnan(emb_grad).any()), det
            tr.opt.step()
            assert bool(torch.isnan(tr.opt.norm).all()), det
    finally:
        _lib.query('cvae_deterministic', prev)
        _lib.query('cvae_persist_status')


@pytest.mark.gpu
def test_release_stream_frees_the_library_buffers(dev):
    """cvae_release_stream: the (device, stream) buffers go back to the driver and the next call on that stream
    allocates fresh ones and computes the same values."""
    from cvae_amd import _lib, ops
    side = torch.cuda.Stream()
    n = 1 << 20
    g = torch.randn(n, device=dev)

    def step():
        p, m, v = torch.ones(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        sumsq, norm = torch.zeros(2, device=dev), torch.zeros(1, device=dev)
        with torch.cuda.stream(side):
            ops.clip_adam(p, g.clone(), m, v, 0, 0, 1, 5.0, 1e-3, (0.9, 0.999), 1e-8, 0, 0, sumsq, norm)
        side.synchronize()
        return p, float(norm)

    torch.cuda.synchronize()
    side.wait_stream(torch.cuda.current_stream())
    p1, n1 = step()
    assert _lib.query('cvae_release_stream', side.cuda_stream) == 0
    assert _lib.query('cvae_release_stream', side.cuda_stream) == 0          # nothing left: still fine
    p2, n2 = step()
    assert n1 == n2 and torch.equal(p1, p2)
    assert _lib.query('cvae_release_stream', side.cuda_stream) == 0
