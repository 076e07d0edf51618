"""Per-kernel numerics on a real MI355X, through the C ABI (ctypes), against fp64 torch /
the CPU oracle.  Tolerances: fp32 contractions 2e-5 relative to max|ref| (exact-f32 MFMA
chains), everything integer bit-exact."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cvae_oracle as O   # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from cvae_amd import _lib
    _lib.init(0)
    return torch.device('cuda:0')


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rnd(*shape, dev, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dev)


GEMM_CASES = [
    # M, N, K, transA, transB, bias, act, beta
    (5, 50, 16, 0, 1, 1, 0, 0),            # tiny, scalar tails
    (7, 13, 9, 0, 0, 0, 0, 0),             # unaligned everything
    (32, 1536, 512, 0, 1, 1, 0, 0),        # skinny (M<=64), K split over waves
    (32, 512, 1536, 0, 0, 0, 0, 1),        # skinny, B not transposed, accumulate (RNN backward shape)
    (64, 200, 32, 0, 1, 1, 2, 0),          # toInit shape: K < 64 -> tiled 64x64, relu
    (300, 200, 400, 0, 1, 1, 1, 0),        # tiled 64x64, tanh
    (2912, 1536, 400, 0, 1, 1, 0, 0),      # decoder input projection (Yelp train): 128x128 tiles
    (1536, 400, 2912, 1, 0, 0, 0, 0),      # weight gradient dW = dG^T X
    (2912, 400, 1536, 0, 0, 0, 0, 0),      # input gradient dX = dG W
    (100, 64, 5000, 0, 0, 0, 0, 0),        # few tiles, long K -> split-K with atomics
    (100, 64, 5000, 0, 0, 0, 0, 1),        # split-K accumulate
    (130, 70, 33, 1, 1, 1, 3, 0),          # both transposed, exp
    (1, 1, 1, 0, 1, 0, 0, 0),
    # long-K, few-hundred-column products -> persistent split-K kernel (gemm_wide.hip)
    (1024, 256, 3001, 1, 0, 0, 0, 0),      # dW_hh: two 128-column tiles, ragged K
    (1024, 200, 6432, 1, 0, 0, 0, 1),      # dW_ih accumulate: one 224-wide tile
    (333, 131, 517, 0, 0, 0, 0, 0),        # ragged M/N/K, 224-wide tile with 131 live columns
    (70, 447, 600, 1, 0, 0, 0, 0),         # two 224-wide tiles, unaligned N
    (6432, 200, 1024, 0, 0, 0, 0, 0),      # dX = dG W_ih
    (2048, 512, 3000, 1, 0, 0, 0, 0),      # decoder dW_hh: four 128-column tiles
    (200, 512, 3000, 1, 0, 0, 0, 1),       # dW_c accumulate
    (1334, 200, 1500, 1, 0, 0, 0, 0),      # k-major A, last row tile partial (54 rows) and M % 4 == 2: clamped groups + a straddling one
    (20004, 200, 700, 1, 0, 0, 0, 1),      # ToVocab-like: 157 row tiles, the last one 36 rows wide, accumulate
    # x W^T with >= 192 output tiles (128x128 tiles): K tails, ragged edges, epilogues
    (6400, 2048, 400, 0, 1, 1, 0, 0),      # decoder input projection: K tail of 16
    (2100, 1540, 200, 0, 1, 1, 1, 0),      # ragged M / N, bias + tanh, K tail of 8
    (3000, 1100, 36, 0, 1, 0, 0, 1),       # short K with a 4-wide tail, accumulate
    (1600, 2048, 512, 0, 1, 0, 2, 0),      # relu
    # K % 16 == 0: the k-contiguous kernel (gemm_nt_kernel); ragged M / N, epilogues, accumulate
    (2100, 1540, 208, 0, 1, 1, 2, 0),
    (3000, 1100, 48, 0, 1, 0, 0, 1),
    (2049, 1537, 16, 0, 1, 1, 3, 0),
]


@pytest.mark.parametrize('case', GEMM_CASES)
def test_gemm(dev, case):
    from cvae_amd import ops
    M, N, K, tA, tB, bias, act, beta = case
    A = rnd(*((K, M) if tA else (M, K)), dev=dev, seed=1)
    B = rnd(*((N, K) if tB else (K, N)), dev=dev, seed=2)
    if act in (1, 3):
        # pre-activations of the size the model has (|x| of a few units): the split-bf16 x W^T product is accurate to
        # ~5e-6 of the pre-activation's magnitude (2^-17 per product), which a tanh over |x| ~ 50 would turn into
        # an absolute 3e-4 around its zero crossing
        A = A * 0.1
    b = rnd(N, dev=dev, seed=3) if bias else None
    C0 = rnd(M, N, dev=dev, seed=4)
    out = C0.clone()
    ops.gemm(A, B, bool(tA), bool(tB), b, act, out=out, beta=beta)
    Ad, Bd = A.double(), B.double()
    ref = (Ad.t() if tA else Ad) @ (Bd.t() if tB else Bd)
    if bias:
        ref = ref + b.double()
    ref = {0: lambda x: x, 1: torch.tanh, 2: torch.relu, 3: torch.exp}[act](ref)
    if beta:
        ref = ref + C0.double()
    # x W^T products run as split-bf16 (bf16x3) by default: 2^-17 per product, i.e. a few 1e-6 of the
    # pre-activation's magnitude (fp32-input MFMA: ~1e-7); every other form is fp32-input or bf16x6
    tol = 6e-5 if (tB and not tA) else 2e-5
    assert rel(out, ref) < tol, case


def test_gemm_strided_operands(dev):
    """Sub-matrix views (the decoder reads the two column halves of W_ih)."""
    from cvae_amd import ops
    W = rnd(96, 80, dev=dev)
    x = rnd(70, 40, dev=dev, seed=5)
    out = ops.gemm(x, W[:, 40:], False, True)
    assert rel(out, x.double() @ W[:, 40:].double().t()) < 2e-5


def test_pack_batch_and_sort_bit_exact(dev):
    from cvae_amd import ops
    rs = np.random.RandomState(0)
    sents = [rs.randint(4, 50, size=n).tolist() for n in (3, 12, 15, 1, 7, 12, 20)]
    max_len = 12
    for truncate in (False, True):
        bare, go, eos, ln = O.make_batch(sents, max_len, truncate=truncate)
        flat = torch.tensor([t for s in sents for t in s], dtype=torch.int64, device=dev)
        off = torch.tensor(np.cumsum([0] + [len(s) for s in sents]), dtype=torch.int64, device=dev)
        width = bare.shape[1]
        got = ops.pack_batch(flat, off, max_len, width)
        for g, r in zip(got, (bare, go, eos, ln)):
            assert g.dtype == torch.int64 and np.array_equal(g.cpu().numpy(), r)
    for lens in ([7, 3, 5, 3, 1], [2, 2, 2], [1], [4, 9, 9, 1, 4, 9, 2], list(rs.randint(1, 9, size=300))):
        s, r = ops.sort_resort(torch.tensor(lens, dtype=torch.int64, device=dev))
        s_ref, r_ref = O.sort_resort(lens)
        assert s.tolist() == s_ref and r.tolist() == r_ref


@pytest.mark.parametrize('M,N,K,act', [(32, 32, 512, 0), (32, 32, 512, 3), (32, 200, 32, 2), (32, 200, 200, 0),
                                        (5, 16, 4, 2), (64, 77, 131, 1), (1, 1, 1, 0), (33, 513, 70, 3),
                                        (100, 200, 32, 2), (65, 32, 512, 3)])
def test_linear_forward_backward(dev, M, N, K, act):
    """y = act(x W^T + b) and its backward (a5 / a6 heads): M <= 64 takes the one-launch backward
    (linear_bwd_small_kernel), larger M the GEMM path; both against fp64 torch."""
    from cvae_amd import ops
    x, W, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, scale=0.3, seed=2), rnd(N, dev=dev, seed=3)
    g = rnd(M, N, dev=dev, seed=4)
    f = {0: lambda t: t, 1: torch.tanh, 2: torch.relu, 3: torch.exp}[act]
    xr, Wr, br = (t.detach().double().cpu().requires_grad_(True) for t in (x, W, b))
    ref = f(xr @ Wr.t() + br)
    (ref * g.double().cpu()).sum().backward()
    for need_dx in (True, False):
        xd, Wd, bd = x.clone().requires_grad_(need_dx), W.clone().requires_grad_(True), b.clone().requires_grad_(True)
        got = ops.linear(xd, Wd, bd, act)
        assert rel(got, ref) < 2e-5
        (got * g).sum().backward()
        assert rel(Wd.grad, Wr.grad) < 2e-5 and rel(bd.grad, br.grad) < 2e-5
        if need_dx:
            assert rel(xd.grad, xr.grad) < 2e-5


def _rnn_params(cell, inp, hid, bi, seed):
    G = 3 if cell == 'gru' else 4
    g = torch.Generator().manual_seed(seed)
    k = 1.0 / math.sqrt(hid)
    ws = []
    for _ in range(2 if bi else 1):
        ws += [(torch.rand(G * hid, inp, generator=g) * 2 - 1) * k, (torch.rand(G * hid, hid, generator=g) * 2 - 1) * k,
               (torch.rand(G * hid, generator=g) * 2 - 1) * k, (torch.rand(G * hid, generator=g) * 2 - 1) * k]
    return ws


@pytest.mark.parametrize('cell', ['gru', 'lstm'])
# H = 8: per-step kernels; H = 128 / 256 with B <= 32: whole-sequence kernels (rnn_persist.hip), ragged rows
@pytest.mark.parametrize('shape', [(5, 7, 16, 8, 50), (32, 40, 200, 128, 1000), (19, 23, 40, 256, 300)])
def test_encoder_forward_backward(dev, cell, shape):
    """Packed bi-RNN last states + every gradient vs the oracle in fp64."""
    from cvae_amd import ops
    B, L, E, H, V = shape
    g = torch.Generator().manual_seed(1)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    if B > 3:
        lens[1] = lens[2]                                    # ties
    ids = torch.randint(0, V, (B, L), generator=g)
    emb = torch.randn(V, E, generator=g)
    ws = _rnn_params(cell, E, H, True, 2)
    mask = (torch.rand(L, B, E, generator=g) > 0.2).float() / 0.8
    dlast = torch.randn(B, 2 * H, generator=g)
    # oracle, fp64
    cn = 'GRU' if cell == 'gru' else 'LSTM'
    P = {'Enc.Embedding.weight': emb.double().requires_grad_(True)}
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    for d, suf in enumerate(['', '_reverse']):
        for j, n in enumerate(names):
            P['Enc.%s.%s%s' % (cn, n, suf)] = ws[4 * d + j].double().requires_grad_(True)
    ref = O.encoder(P, 'Enc.', cell, ids, lens, mask.double())
    (ref * dlast.double()).sum().backward()
    # device
    c = ops.GRU if cell == 'gru' else ops.LSTM
    emb_d = emb.to(dev).requires_grad_(True)
    ws_d = [w.to(dev).requires_grad_(True) for w in ws]
    _, res_idx = ops.sort_resort(lens.to(dev))
    got = ops.encoder(c, ids.to(dev), lens.to(dev), mask.to(dev), res_idx, emb_d, ws_d)
    assert rel(got, ref) < 2e-5
    (got * dlast.to(dev)).sum().backward()
    assert rel(emb_d.grad, P['Enc.Embedding.weight'].grad) < 1e-4
    for d, suf in enumerate(['', '_reverse']):
        for j, n in enumerate(names):
            assert rel(ws_d[4 * d + j].grad, P['Enc.%s.%s%s' % (cn, n, suf)].grad) < 1e-4, (n, suf)
    # eval mode (no mask, nothing saved)
    with torch.no_grad():
        got_e = ops.encoder(c, ids.to(dev), lens.to(dev), None, None, emb_d, ws_d)
        ref_e = O.encoder(P, 'Enc.', cell, ids, lens, None)
    assert rel(got_e, ref_e) < 2e-5


@pytest.mark.parametrize('cell', ['gru', 'lstm'])
# (B, L_a, L_b, E, H, V): per-step kernels, whole-sequence kernels with the longer batch first / second
@pytest.mark.parametrize('shape', [(5, 7, 4, 16, 8, 50), (32, 40, 25, 200, 128, 1000), (19, 12, 23, 40, 256, 300)])
def test_encoder_pair_matches_separate(dev, cell, shape):
    """The dialogue model's two encoders in lock-step (cvae_encoder_forward_n / backward_n) against the two
    single-encoder calls: same last states, same gradients (shared embedding gradient = the sum)."""
    from cvae_amd import ops
    B, La, Lb, E, H, V = shape
    g = torch.Generator().manual_seed(5)
    c = ops.GRU if cell == 'gru' else ops.LSTM
    emb = torch.randn(V, E, generator=g).to(dev)
    side = []
    for L, seed in ((La, 3), (Lb, 4)):
        lens = torch.randint(1, L + 1, (B,), generator=g)
        lens[0] = L
        side.append(dict(ids=torch.randint(0, V, (B, L), generator=g).to(dev), lens=lens.to(dev),
                         mask=((torch.rand(L, B, E, generator=g) > 0.2).float() / 0.8).to(dev),
                         rows=ops.sort_resort(lens.to(dev))[1], ws=[w.to(dev) for w in _rnn_params(cell, E, H, True, seed)],
                         dlast=torch.randn(B, 2 * H, generator=g).to(dev)))
    for train in (True, False):
        res = []
        for paired in (False, True):
            e = emb.clone().requires_grad_(train)
            ws = [[w.clone().requires_grad_(train) for w in q['ws']] for q in side]
            m = [(q['mask'], q['rows']) if train else (None, None) for q in side]
            a, b = side
            if paired:
                la, lb = ops.encoder_pair(c, a['ids'], a['lens'], m[0][0], m[0][1], ws[0],
                                          b['ids'], b['lens'], m[1][0], m[1][1], ws[1], e)
            else:
                la = ops.encoder(c, a['ids'], a['lens'], m[0][0], m[0][1], e, ws[0])
                lb = ops.encoder(c, b['ids'], b['lens'], m[1][0], m[1][1], e, ws[1])
            out = [la.detach(), lb.detach()]
            if train:
                ((la * a['dlast']).sum() + (lb * b['dlast']).sum()).backward()
                out += [e.grad] + [w.grad for w in ws[0] + ws[1]]
            res.append(out)
        for x, y in zip(*res):
            assert rel(y, x) < 2e-5
    # one of the two outputs unused: its gradient arrives as None
    e = emb.clone().requires_grad_(True)
    ws = [[w.clone().requires_grad_(True) for w in q['ws']] for q in side]
    a, b = side
    la, lb = ops.encoder_pair(c, a['ids'], a['lens'], None, None, ws[0], b['ids'], b['lens'], None, None, ws[1], e)
    (la * a['dlast']).sum().backward()
    assert float(ws[1][0].grad.abs().max()) == 0.0 and float(ws[0][0].grad.abs().max()) > 0


@pytest.mark.parametrize('cell', ['gru', 'lstm'])
@pytest.mark.parametrize('shape', [(5, 8, 16, 16, 50), (32, 30, 200, 256, 1000), (13, 17, 24, 512, 200)])
def test_decoder_forward_backward(dev, cell, shape):
    from cvae_amd import ops
    B, T, E, Hd, V = shape
    g = torch.Generator().manual_seed(3)
    go = torch.randint(0, V, (B, T), generator=g)
    emb = torch.randn(V, E, generator=g)
    init = torch.randn(B, E, generator=g)
    ws = _rnn_params(cell, 2 * E, Hd, False, 4)
    k = 1.0 / math.sqrt(Hd)
    W_c, b_c = (torch.rand(E, Hd, generator=g) * 2 - 1) * k, (torch.rand(E, generator=g) * 2 - 1) * k
    m_in = (torch.rand(T, B, 2 * E, generator=g) > 0.2).float() / 0.8
    m_out = (torch.rand(T, B, Hd, generator=g) > 0.2).float() / 0.8
    dfinal = torch.randn(T, B, E, generator=g)
    cn = 'GRU' if cell == 'gru' else 'LSTM'
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    P = {'D.Embedding.weight': emb.double().requires_grad_(True), 'D.W_c.weight': W_c.double().requires_grad_(True),
         'D.W_c.bias': b_c.double().requires_grad_(True)}
    for j, n in enumerate(names):
        P['D.%s.%s' % (cn, n)] = ws[j].double().requires_grad_(True)
    init64 = init.double().requires_grad_(True)
    ref = O.decoder_final(P, 'D.', cell, init64, go, m_in.double(), m_out.double())
    (ref * dfinal.double()).sum().backward()
    c = ops.GRU if cell == 'gru' else ops.LSTM
    emb_d = emb.to(dev).requires_grad_(True)
    init_d = init.to(dev).requires_grad_(True)
    w_d = [w.to(dev).requires_grad_(True) for w in ws + [W_c, b_c]]
    got = ops.decoder(c, True, init_d, go.to(dev), m_in.to(dev), m_out.to(dev), emb_d, w_d)
    assert rel(got, ref) < 2e-5
    (got * dfinal.to(dev)).sum().backward()
    assert rel(init_d.grad, init64.grad) < 1e-4
    assert rel(emb_d.grad, P['D.Embedding.weight'].grad) < 1e-4
    refs = [P['D.%s.%s' % (cn, n)].grad for n in names] + [P['D.W_c.weight'].grad, P['D.W_c.bias'].grad]
    for a, b in zip(w_d, refs):
        assert rel(a.grad, b) < 1e-4
    # eval path with the de-duplicated input projection, R = n*B rows (small and > 256 rows)
    for n_rep in (2, 60 if B == 5 else 10):
        init_r = torch.randn(n_rep * B, E, generator=g)
        with torch.no_grad():
            ref_e = O.decoder_final(P, 'D.', cell, init_r.double(), go.repeat(n_rep, 1))
            got_e = ops.decoder(c, False, init_r.to(dev), go.to(dev), None, None, emb_d, w_d)
        assert rel(got_e, ref_e) < 2e-5, n_rep


@pytest.mark.parametrize('cell', ['gru', 'lstm'])
def test_stream_rnn_kernels(dev, cell):
    """The weight-streaming split-bf16 step kernels (skinny_stream_kernel: forward with the fused cell, backward as
    K-split partial products + cell_bwd_parts_kernel), which by default only take recurrent weights of 8 MB and more
    (the scale-up shape), forced onto small shapes the whole-sequence kernels do not take (more than 32 rows, or H not
    a multiple of 128): encoder (packed, two directions in lock-step) and decoder, every gradient vs the oracle."""
    from cvae_amd import _lib
    prev = _lib.query('cvae_stream_rnn_min_bytes', 0)
    try:
        for shape in ((48, 13, 40, 64, 100), (21, 9, 24, 192, 60), (64, 6, 16, 256, 50)):
            test_encoder_forward_backward(dev, cell, shape)
        for shape in ((40, 11, 32, 320, 80), (64, 7, 16, 128, 50), (17, 5, 24, 64, 40)):
            test_decoder_forward_backward(dev, cell, shape)
    finally:
        _lib.query('cvae_stream_rnn_min_bytes', prev)


@pytest.mark.parametrize('cell', ['gru', 'lstm'])
@pytest.mark.parametrize('shape', [(6, 9, 16, 32, 40, 50), (12, 21, 40, 128, 90, 30)])
def test_decoder_forward_sorted(dev, cell, shape):
    """cvae_decoder_forward_sorted (sentence-major rows, finished rows skipped) against the plain
    eval branch on the same rows: live rows equal, skipped rows untouched (zero)."""
    from cvae_amd import ops
    B, T, E, Hd, V, K = shape
    g = torch.Generator().manual_seed(8)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    lens = torch.sort(lens, descending=True).values
    go = torch.randint(1, V, (B, T), generator=g)
    emb = torch.randn(V, E, generator=g)
    ws = _rnn_params(cell, 2 * E, Hd, False, 4)
    k = 1.0 / math.sqrt(Hd)
    W_c, b_c = (torch.rand(E, Hd, generator=g) * 2 - 1) * k, (torch.rand(E, generator=g) * 2 - 1) * k
    init = torch.randn(K, B, E, generator=g)                               # sample-major (k, b)
    c = ops.GRU if cell == 'gru' else ops.LSTM
    w_d = [w.to(dev) for w in ws + [W_c, b_c]]
    plain = ops.decoder(c, False, init.reshape(K * B, E).to(dev), go.to(dev), None, None, emb.to(dev), w_d)
    plain = plain.view(T, K, B, E).transpose(1, 2)                         # (T, B, K, E)
    live_sent = (lens[None, :] > torch.arange(T)[:, None]).sum(1).to(torch.int32)
    init_s = init.transpose(0, 1).reshape(B * K, E).to(dev)               # row b*K + k
    got = ops.decoder_sorted(c, init_s, go.to(dev), live_sent, emb.to(dev), w_d).view(T, B, K, E)
    for t in range(T):
        n = int(live_sent[t])
        # small live prefixes take another GEMM kernel (different summation order): fp32 tolerance
        assert (got[t, :n] - plain[t, :n]).abs().max().item() < 2e-5, t
        assert not got[t, n:].any(), t
    # sentence-major targets
    eos = torch.where(torch.arange(T)[None, :] < lens[:, None], go, torch.zeros_like(go)).to(dev)
    tgt = ops.targets_sentence_major(eos, K).view(T, B, K)
    assert torch.equal(tgt, eos.t()[:, :, None].expand(T, B, K))


@pytest.mark.parametrize('cell', ['lstm', 'gru'])
@pytest.mark.parametrize('shape', [(5, 7, 24, 256, 50, 53), (9, 6, 40, 512, 60, 37), (3, 4, 16, 512, 30, 700)])
def test_decoder_sorted_weights_in_registers(dev, shape, cell):
    """The many-row LSTM / GRU step with W_hh resident in registers (rnn_large_wreg_kernel: h read as a bf16 (hi, lo) image
    by LDS-DMA, the cell one tile behind) against the 128-row-tile kernel on the same call -- cvae_rnn_large_wreg()
    switches between them at run time -- with live prefixes that are not multiples of the 16-row tile, at both hidden
    sizes the kernel takes; skipped rows stay zero.  (Both kernels form bf16x3 products, in different summation
    orders; the 128-row-tile kernel is pinned against the oracle by the IW-NLL tests, which run this kernel by
    default at their Hd = 256 / 512 shapes.)"""
    from cvae_amd import _lib, ops
    B, T, E, Hd, V, K = shape
    g = torch.Generator().manual_seed(21)
    lens = torch.randint(1, T + 1, (B,), generator=g)
    lens[0] = T
    lens = torch.sort(lens, descending=True).values
    go = torch.randint(1, V, (B, T), generator=g).to(dev)
    emb = torch.randn(V, E, generator=g).to(dev)
    ws = _rnn_params(cell, 2 * E, Hd, False, 4)
    c = ops.GRU if cell == 'gru' else ops.LSTM
    k = 1.0 / math.sqrt(Hd)
    W_c, b_c = (torch.rand(E, Hd, generator=g) * 2 - 1) * k, (torch.rand(E, generator=g) * 2 - 1) * k
    w_d = [w.to(dev) for w in ws + [W_c, b_c]]
    init_s = torch.randn(B * K, E, generator=g).to(dev)
    live_sent = (lens[None, :] > torch.arange(T)[:, None]).sum(1).to(torch.int32)
    prev = _lib.query('cvae_rnn_large_wreg', 1 << 30)
    try:
        want = ops.decoder_sorted(c, init_s, go, live_sent, emb, w_d)
        _lib.query('cvae_rnn_large_wreg', 1)
        got = ops.decoder_sorted(c, init_s, go, live_sent, emb, w_d)
        again = ops.decoder_sorted(c, init_s, go, live_sent, emb, w_d)
    finally:
        _lib.query('cvae_rnn_large_wreg', prev)
    assert torch.equal(got, again)                                        # no dependence on what the scratch images held
    assert torch.isfinite(got).all()
    for t in range(T):
        n = int(live_sent[t]) * K
        assert (got[t, :n] - want[t, :n]).abs().max().item() < 3e-5, t
        assert not got[t, n:].any(), t


# E = 16/32/200: exact-K register path; 30/104: zero-padded K; 208: generic fallback (E > 200);
# V off the 32-column tile; R*T above and below one 128-row item
@pytest.mark.parametrize('shape', [(7, 5, 16, 50), (40, 32, 200, 10002), (3, 700, 200, 20001), (6, 9, 30, 77),
                                   (5, 33, 104, 333), (9, 40, 32, 160), (4, 6, 208, 100)])
def test_projce_forward_backward(dev, shape):
    from cvae_amd import ops
    T, R, E, V = shape
    g = torch.Generator().manual_seed(5)
    final = torch.tanh(torch.randn(T, R, E, generator=g))
    W = torch.randn(V, E, generator=g) * 0.1
    b = torch.randn(V, generator=g) * 0.1
    eos = torch.randint(0, V, (R, T), generator=g)
    eos[:, -1] = 0
    eos[0, :] = 0                                   # an all-pad sentence
    eos[1, 0] = V - 1
    gs = torch.randn(R, generator=g)
    f64 = final.double().requires_grad_(True)
    P = {'D.ToVocab.weight': W.double().requires_grad_(True), 'D.ToVocab.bias': b.double().requires_grad_(True)}
    ref = O.seq_xent(P, 'D.', f64, eos)
    (ref * gs.double()).sum().backward()
    from cvae_amd import settings
    # both backward variants: logits saved by the forward pass and turned into the gradient in place
    # (default), and the projection recomputed in backward
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
    lg = ops.logits_bt(fd, Wd, bd)
    assert rel(lg, O.decoder_logits(P, 'D.', f64)) < 2e-5


@pytest.mark.parametrize('V', [33, 34, 40, 47, 63, 64, 65, 1025])
def test_projce_eval_form_pad_columns(dev, V):
    """The evaluation form of the CE core (ce_core_bf3_kernel<.., FAST>: base-2 logits, the bias riding in the free K
    slot of W's image, columns past V at -1e30 instead of a mask, its maximum taken by inline asm on the MFMA result)
    at vocabulary sizes that leave 0 .. 31 padded columns in the last 32-column tile, against fp64: a last tile with
    few real columns is where a too-early read of the accumulators showed."""
    from cvae_amd import ops
    T, R, E = 2, 37, 200
    g = torch.Generator().manual_seed(V)
    final = torch.tanh(torch.randn(T, R, E, generator=g))
    W, b = torch.randn(V, E, generator=g) * 0.1, torch.randn(V, generator=g) * 0.1
    eos = torch.randint(1, V, (R, T), generator=g)
    eos[0, 0] = V - 1
    logits = final.double() @ W.double().t() + b.double()
    ref = (torch.logsumexp(logits, -1) - logits.gather(-1, eos.t().unsqueeze(-1)).squeeze(-1)).sum(0)
    with torch.no_grad():
        got = ops.proj_ce(final.to(dev), W.to(dev), b.to(dev), eos.to(dev), 0)
    assert torch.isfinite(got).all()
    assert rel(got, ref) < 2e-5


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(9, 5, 16, 50), (30, 32, 200, 2100), (41, 32, 200, 10002), (64, 32, 200, 20000)])
def test_projce_pair_matches_oracle_and_single(dev, shape):
    """The two decoders' losses over one set of targets in one call (cvae_projce_forward_train_n /
    cvae_projce_backward_saved_n: shared row list, one zero-fill, batched gradient products where they are small):
    against the fp64 oracle and against two separate calls; only one of the two losses carrying a gradient too."""
    from cvae_amd import ops
    T, R, E, V = shape
    g = torch.Generator().manual_seed(11)
    finals = [torch.tanh(torch.randn(T, R, E, generator=g)) for _ in range(2)]
    Ws = [torch.randn(V, E, generator=g) * 0.1 for _ in range(2)]
    bs = [torch.randn(V, generator=g) * 0.1 for _ in range(2)]
    eos = torch.randint(1, V, (R, T), generator=g)
    for r in range(R):                                   # ragged targets, one all-pad sentence
        eos[r, max(1, T - r):] = 0
    eos[R - 1, :] = 0
    gs = [torch.randn(R, generator=g) for _ in range(2)]
    refs = []
    for d in range(2):
        f64 = finals[d].double().requires_grad_(True)
        P = {'D.ToVocab.weight': Ws[d].double().requires_grad_(True), 'D.ToVocab.bias': bs[d].double().requires_grad_(True)}
        out = O.seq_xent(P, 'D.', f64, eos)
        (out * gs[d].double()).sum().backward()
        refs.append((out, f64.grad, P['D.ToVocab.weight'].grad, P['D.ToVocab.bias'].grad))
    dv = [[t.to(dev).requires_grad_(True) for t in (finals[d], Ws[d], bs[d])] for d in range(2)]
    assert ops.proj_ce_pair_ok([dv[0][0], dv[1][0]], [dv[0][1], dv[1][1]])
    outs = ops.proj_ce_pair([dv[0][0], dv[1][0]], [dv[0][1], dv[1][1]], [dv[0][2], dv[1][2]], eos.to(dev), 0)
    sum((outs[d] * gs[d].to(dev)).sum() for d in range(2)).backward()
    for d in range(2):
        assert rel(outs[d], refs[d][0]) < 2e-5
        for got, want in zip([t.grad for t in dv[d]], refs[d][1:]):
            assert rel(got, want) < 1e-4
        # two separate calls
        sv = [t.detach().clone().requires_grad_(True) for t in dv[d]]
        single = ops.proj_ce(sv[0], sv[1], sv[2], eos.to(dev), 0)
        (single * gs[d].to(dev)).sum().backward()
        assert torch.equal(single, outs[d])
        for got, want in zip([t.grad for t in dv[d]], [t.grad for t in sv]):
            assert rel(got, want) < 2e-6
    # only the second loss is used: the first one's gradients are exactly zero, the second one's unchanged
    dv2 = [[t.detach().clone().requires_grad_(True) for t in dv[d]] for d in range(2)]
    outs2 = ops.proj_ce_pair([dv2[0][0], dv2[1][0]], [dv2[0][1], dv2[1][1]], [dv2[0][2], dv2[1][2]], eos.to(dev), 0)
    (outs2[1] * gs[1].to(dev)).sum().backward()
    assert all(float(t.grad.abs().max()) == 0.0 for t in dv2[0])
    for got, want in zip([t.grad for t in dv2[1]], [t.grad for t in dv[1]]):
        assert rel(got, want) < 2e-6


def test_latent_kernels(dev):
    from cvae_amd import ops
    g = torch.Generator().manual_seed(7)
    B, Z, E = 33, 32, 200
    mu, mu_p = torch.randn(B, Z, generator=g), torch.randn(B, Z, generator=g)
    sg, sg_p = torch.rand(B, Z, generator=g) + 0.3, torch.rand(B, Z, generator=g) + 0.3
    gk = torch.randn(B, generator=g)
    for prior in (False, True):
        a = [t.double().requires_grad_(True) for t in ((mu, sg, mu_p, sg_p) if prior else (mu, sg))]
        ref = O.kl_pair(*a) if prior else O.kl_std(*a)
        (ref * gk.double()).sum().backward()
        d = [t.to(dev).requires_grad_(True) for t in ((mu, sg, mu_p, sg_p) if prior else (mu, sg))]
        got = ops.kl_div(*d)
        assert rel(got, ref) < 1e-5
        (got * gk.to(dev)).sum().backward()
        for x, y in zip(d, a):
            assert rel(x.grad, y.grad) < 1e-5
    init, init_c = torch.randn(B, E, generator=g) * 0.3, torch.randn(B, E, generator=g) * 0.3
    for euclid in (False, True):
        i64 = init.double().requires_grad_(True)
        ref = O.couple_loss(i64, init_c.double(), Z, euclid)
        (ref * gk.double()).sum().backward()
        idv = init.to(dev).requires_grad_(True)
        got = ops.couple_loss(idv, init_c.to(dev), Z, euclid)
        assert rel(got, ref) < 1e-5
        (got * gk.to(dev)).sum().backward()
        assert rel(idv.grad, i64.grad) < 1e-5
    eps = torch.randn(B, Z, generator=g)
    m64, s64 = mu.double().requires_grad_(True), sg.double().requires_grad_(True)
    ((m64 + s64 * eps.double()) ** 2).sum().backward()
    md, sd = mu.to(dev).requires_grad_(True), sg.to(dev).requires_grad_(True)
    z = ops.ReparamFn.apply(md, sd, eps.to(dev))
    (z ** 2).sum().backward()
    assert rel(md.grad, m64.grad) < 1e-5 and rel(sd.grad, s64.grad) < 1e-5


def test_iw_reduce_and_mi(dev):
    from cvae_amd import ops
    g = torch.Generator().manual_seed(9)
    for (K, B, Z) in ((10, 5, 4), (500, 32, 32), (100, 513, 32)):
        mu, sg = torch.randn(B, Z, generator=g), torch.rand(B, Z, generator=g) + 0.3
        mu_p, sg_p = torch.randn(B, Z, generator=g), torch.rand(B, Z, generator=g) + 0.3
        eps = torch.randn(K, B, Z, generator=g)
        z = mu + sg * eps
        xent = torch.rand(K, B, generator=g) * 300
        for prior in (False, True):
            mp = mu_p.double() if prior else torch.zeros(B, Z).double()
            sp = sg_p.double() if prior else torch.ones(B, Z).double()
            lpz = O.normal_log_prob(z.double(), mp, sp).sum(2)
            lqz = O.normal_log_prob(z.double(), mu.double(), sg.double()).sum(2)
            nll = -(O.log_sum_exp(lpz - xent.double() - lqz, 0) - math.log(K))
            kl = (lqz - lpz).mean(0)
            xm, n2, k2 = ops.iw_reduce(z.to(dev), xent.to(dev), mu.to(dev), sg.to(dev),
                                       mu_p.to(dev) if prior else None, sg_p.to(dev) if prior else None)
            assert rel(n2, nll) < 1e-5 and rel(k2, kl) < 1e-4 and rel(xm, xent.double().mean(0)) < 1e-5
        if K <= 100:
            pair = O.normal_log_prob(z.double().unsqueeze(2), mu.double()[None, None], sg.double()[None, None]).sum(3)
            pair = pair.masked_fill(torch.eye(B, dtype=torch.bool).unsqueeze(0), -float('inf'))
            ref = O.normal_log_prob(z.double(), mu.double(), sg.double()).sum(2).mean(0) - \
                (O.log_sum_exp(pair, 2) - math.log(B - 1)).mean(0)
            got = ops.mi(z.to(dev), mu.to(dev), sg.to(dev))
            assert float((got.double().cpu() - ref).abs().max()) < 1e-4 * float(ref.abs().max()) + 1e-4


def test_clip_adam_alias_semantics(dev):
    """Flat clip + Adam vs the oracle's per-listing loop, 3 steps, both the
    clipped and the un-clipped regime."""
    from cvae_amd import ops
    g = torch.Generator().manual_seed(11)
    n_emb, n_rest = 5000, 12345
    for scale in (0.01, 50.0):
        p = torch.randn(n_emb + n_rest, generator=g)
        P = {'A.Embedding.weight': p[:n_emb].clone(), 'rest': p[n_emb:].clone()}
        P['B.Embedding.weight'] = P['C.Embedding.weight'] = P['A.Embedding.weight']
        keys = ['A.Embedding.weight', 'rest', 'B.Embedding.weight', 'C.Embedding.weight']
        state = {}
        pd = p.to(dev)
        m, v = torch.zeros_like(pd), torch.zeros_like(pd)
        sumsq, norm = torch.zeros(2, device=dev), torch.zeros(1, device=dev)
        for step in range(3):
            gr = torch.randn(n_emb + n_rest, generator=g) * scale
            ge = gr[:n_emb].clone()
            want_sq = (float((gr.double() ** 2).sum()), float((ge.double() ** 2).sum()))   # the oracle scales in place
            grads = {'A.Embedding.weight': ge, 'B.Embedding.weight': ge, 'C.Embedding.weight': ge,
                     'rest': gr[n_emb:].clone()}
            total = O.clip_and_adam(P, grads, state, keys)
            gd = gr.to(dev)
            ops.clip_adam(pd, gd, m, v, 0, n_emb, 3, 30.0, 1e-3, (0.9, 0.999), 1e-8, step, step * 3, sumsq, norm)
            assert abs(float(norm) - total) < 1e-4 * total
            ref = torch.cat([P['A.Embedding.weight'], P['rest']])
            assert float((pd.cpu() - ref).abs().max()) < 2e-6, (scale, step)
            # sumsq receives the two squared norms (all gradients, alias segment)
            assert abs(float(sumsq[0]) - want_sq[0]) < 1e-5 * want_sq[0]
            assert abs(float(sumsq[1]) - want_sq[1]) < 1e-5 * want_sq[1]
    # the norm is summed without float atomics: the same bits on every run, at a size that takes every block
    big = torch.randn(3_000_000, generator=g).to(dev)
    outs = []
    for _ in range(4):
        pd, m, v = torch.zeros_like(big), torch.zeros_like(big), torch.zeros_like(big)
        ops.clip_adam(pd, big.clone(), m, v, 0, 100_000, 2, 5.0, 1e-3, (0.9, 0.999), 1e-8, 0, 0, sumsq, norm)
        outs.append((float(norm), sumsq.cpu().clone(), pd.cpu()))
    for o in outs[1:]:
        assert o[0] == outs[0][0] and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])


def test_kl_free_bits(dev):
    """surprising-fix* KL: clamp(kl_divergence(q, N(0,1)), min=lambda_fb/Z).sum(1), forward and gradient,
    against torch.distributions in fp64 (the expression of CV/Experiment.py:1137-1142)."""
    from cvae_amd import ops
    g = torch.Generator().manual_seed(11)
    B, Z, floor = 37, 32, 6.0 / 32
    mu = (torch.randn(B, Z, generator=g) * 0.8)
    sigma = torch.rand(B, Z, generator=g) * 0.8 + 0.5
    gk = torch.randn(B, generator=g)
    m64, s64 = mu.double().requires_grad_(True), sigma.double().requires_grad_(True)
    kl_e = torch.distributions.kl_divergence(torch.distributions.Normal(m64, s64),
                                             torch.distributions.Normal(torch.zeros(Z, dtype=torch.float64),
                                                                        torch.ones(Z, dtype=torch.float64)))
    assert (kl_e < floor).any() and (kl_e > floor).any()
    ref = torch.clamp(kl_e, min=floor).sum(1)
    (ref * gk.double()).sum().backward()
    md, sd = mu.to(dev).requires_grad_(True), sigma.to(dev).requires_grad_(True)
    got = ops.kl_free_bits(md, sd, floor)
    assert rel(got, ref) < 1e-5
    (got * gk.to(dev)).sum().backward()
    assert rel(md.grad, m64.grad) < 1e-5 and rel(sd.grad, s64.grad) < 1e-5


def test_pack_batch_from_resident_corpus(dev):
    """8f.2: batches gathered from the device-resident CSR corpus by index == the a1 layout built from the
    batch's own CSR arrays (bit-exact), training (truncated) and evaluation widths, repeated indices."""
    from cvae_amd import data
    s = data.SHAPES['ptb']
    c = data.Corpus(data.synthetic_sentences(300, s['V'], s['mu'], s['sigma'], 40, seed=2), max_len=30).to_device(dev)
    for idx in ([5, 17, 3, 299, 0, 17], list(range(40, 72)), [123]):
        for trunc in (True, False):
            a = c.batch(idx, dev, trunc)
            b = c.batch_resident(idx, trunc)
            for x, y in zip(a, b):
                assert x.dtype == torch.int64 and torch.equal(x, y)


def test_decode_entry_point_limits(dev):
    """Greedy at the row maximum (B = 256), beam at K*B = 256, and the refusals just past them."""
    from cvae_amd import ops, _lib
    g = torch.Generator().manual_seed(3)
    E, Hd, V, T = 8, 16, 40, 5
    r = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev)
    w = [r(3 * Hd, 2 * E), r(3 * Hd, Hd), r(3 * Hd), r(3 * Hd), r(E, Hd), r(E)]
    emb, Wv, bv = r(V, E), r(V, E), r(V)
    P = {'D.GRU.weight_ih_l0': w[0], 'D.GRU.weight_hh_l0': w[1], 'D.GRU.bias_ih_l0': w[2], 'D.GRU.bias_hh_l0': w[3],
         'D.W_c.weight': w[4], 'D.W_c.bias': w[5], 'D.Embedding.weight': emb, 'D.ToVocab.weight': Wv, 'D.ToVocab.bias': bv}
    P = {k: v.double().cpu() for k, v in P.items()}
    for B, K in ((256, 1), (32, 8), (128, 2)):
        init, first = r(B, E), torch.randint(4, V, (B,), generator=g).to(dev)
        if K == 1:
            ids = ops.decoder_greedy(ops.GRU, init, first, T, emb, w, Wv, bv)
            ref = O.decoder_greedy(P, 'D.', 'gru', init.double().cpu(), first.cpu(), T)
        else:
            ids = ops.decoder_beam(ops.GRU, init, first, T, K, emb, w, Wv, bv)
            ref = O.decoder_beam(P, 'D.', 'gru', init.double().cpu(), first.cpu(), T, K)[0]
        assert int((ids.cpu() != ref).any(1).sum()) == 0, (B, K)
    init, first = r(257, E), torch.randint(4, V, (257,), generator=g).to(dev)
    with pytest.raises(_lib.CvaeHipError):
        ops.decoder_greedy(ops.GRU, init, first, T, emb, w, Wv, bv)
    init, first = r(33, E), torch.randint(4, V, (33,), generator=g).to(dev)
    with pytest.raises(_lib.CvaeHipError):
        ops.decoder_beam(ops.GRU, init, first, T, 8, emb, w, Wv, bv)          # K*B = 264
    with pytest.raises(_lib.CvaeHipError):
        ops.decoder_beam(ops.GRU, init[:4], first[:4], T, 9, emb, w, Wv, bv)  # K > 8


def test_projce_bf16x3_error(dev):
    """The hidden x vocab product of the CE kernels runs as error-compensated split-bf16 (three bf16 MFMA passes,
    fp32 accumulation) by default.  Measured here against fp64 at the bench's shape (E = 200, V = 20000), beside the
    fp32-input MFMA kernel: materialised-free quantities only -- per-row loss, per-sentence sums and d loss / d final.
    Bars: per-row loss within 2e-5 absolute (the north star's 1e-4 is RELATIVE on sums of ~100 such rows of
    magnitude ~10), sentence sums 2e-6 relative, gradients 1e-4 relative; and the split-bf16 error stays within
    20x of the fp32-input kernel's own rounding error."""
    from cvae_amd import ops, _lib
    T, R, E, V = 40, 24, 200, 20000
    g = torch.Generator().manual_seed(17)
    final = torch.tanh(torch.randn(T, R, E, generator=g))
    W = torch.randn(V, E, generator=g) * 0.1
    b = torch.randn(V, generator=g) * 0.1
    eos = torch.randint(4, V, (R, T), generator=g)
    gs = torch.randn(R, generator=g)
    f64 = final.double().requires_grad_(True)
    P = {'D.ToVocab.weight': W.double(), 'D.ToVocab.bias': b.double()}
    ref = O.seq_xent(P, 'D.', f64, eos)
    (ref * gs.double()).sum().backward()
    errs = {}
    prev = _lib.query('cvae_split_bf16', 1)
    try:
        for mode in (1, 0):
            _lib.query('cvae_split_bf16', mode)
            fd = final.to(dev).requires_grad_(True)
            got = ops.proj_ce(fd, W.to(dev), b.to(dev), eos.to(dev), 0)
            (got * gs.to(dev)).sum().backward()
            errs[mode] = (float((got.double().cpu() - ref).abs().max()), rel(got, ref), rel(fd.grad, f64.grad))
    finally:
        _lib.query('cvae_split_bf16', prev)
    print('CE product error vs fp64 (sentence-sum abs, sentence-sum rel, grad rel): bf16x3 %s | fp32 MFMA %s' % (errs[1], errs[0]))
    assert errs[1][1] < 2e-6 and errs[1][2] < 1e-4
    assert errs[1][1] < 20 * max(errs[0][1], 1e-8)
