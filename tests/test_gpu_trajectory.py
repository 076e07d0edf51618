"""Drift over a training horizon: 30 optimizer steps of the product against the CPU oracle on identical weights,
batches and noise, with the split-bf16 products (the default) and with the fp32-input MFMA kernels.

The golden fixtures pin 2-4 steps; this measures what the narrower forward products (bf16x3: ~2^-17 per product)
do over a trajectory.  Two fp32 implementations of the same training run separate step by step anyway (different
summation orders; Adam's early updates are ~lr * sign(g), so an element whose gradient is noise-sized flips by a
whole lr): the fp32-input kernels give the size of that unavoidable separation, and the split-bf16 kernels must stay
within a small factor of it.  Measured (printed with -s): every step's total loss within 2.2e-7 of the oracle's in both
modes; worst final weight 1.2e-4 (split-bf16) / 1.8e-4 (fp32-input) of its tensor's largest entry at H = 128, 2e-6 /
4e-6 at the mid fixture's shape.  Bars: loss within 1e-5 at every step (a tenth of north_star's loss bar, held over the
whole horizon), final weights within 1e-3, and the split-bf16 run no further from the oracle than 3x the fp32-input
run (+ 2e-5 / 1e-4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cvae_oracle as O     # noqa: E402  (checker only)

N_STEPS = 30


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from cvae_amd import _lib
    _lib.init(0)
    return torch.device('cuda:0')


def _noise(gen, B, T, L, E, Hd, Z):
    drop = lambda *s: (torch.rand(*s, generator=gen) > 0.2).float() / 0.8
    return dict(eps=torch.randn(B, Z, generator=gen), enc_drop=drop(L, B, E), dec_in_drop=drop(T, B, 2 * E),
                dec_out_drop=drop(T, B, Hd), decc_in_drop=drop(T, B, 2 * E), decc_out_drop=drop(T, B, Hd))


def _run_product(dev, hp, state0, batches, noises, split):
    from cvae_amd import _lib, harness, modules
    prev = _lib.query('cvae_split_bf16', 1 if split else 0)
    try:
        model = harness.build_model(hp, 'cpu', seed=0)
        model.load_state_dict(state0, strict=True)
        tr = harness.Trainer(hp, dev, model=model.to(dev))
        tots = []
        for b, nz in zip(batches, noises):
            replay = {k: (v.unsqueeze(0) if k == 'eps' else v) for k, v in nz.items()}
            modules.set_noise_source(modules.ReplayNoise(replay))
            res = tr.train_step({k: v.to(dev) for k, v in b.items()})
            tots.append(res['tot'].detach())
        modules.set_noise_source(None)
        tots = [float(t) for t in torch.stack(tots).cpu()]
        return tots, {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    finally:
        modules.set_noise_source(None)
        _lib.query('cvae_split_bf16', prev)


@pytest.mark.parametrize('cell,H,E,V,Z,B,cap', [('lstm', 128, 32, 500, 8, 16, 24), ('gru', 24, 32, 164, 8, 8, 40)])
def test_thirty_step_trajectory_vs_oracle(dev, cell, H, E, V, Z, B, cap):
    """(lstm, H 128): the whole-sequence recurrent kernels, split-K products and the fused CE all take part;
    (gru, H 24 ...): the mid fixture's shape (per-step recurrent kernels)."""
    from cvae_amd import data, harness
    hp = harness.HParams(model='vae-couple', cell=cell, V=V, E=E, H=H, Z=Z, batch_size=B, annealing_start=5,
                         annealing_step=20)                     # lambda_KL ramps 0 -> 1 inside the horizon
    model0 = harness.build_model(hp, 'cpu', seed=11)
    state0 = {k: v.detach().clone() for k, v in model0.state_dict().items()}
    keys = list(state0.keys())
    corpus = data.Corpus(data.synthetic_sentences(B * 6, V, 2.4, 0.5, cap, seed=17))
    gen = torch.Generator().manual_seed(5)
    batches, noises = [], []
    for i in range(N_STEPS):
        idx = [(i * B + j) % len(corpus) for j in range(B)]
        bare, go, eos, ln = (t.cpu() for t in corpus.batch(idx, dev, True, False))     # a1 layout, packed on the device
        batches.append(dict(bare=bare, go=go, eos=eos, len=ln))
        noises.append(_noise(gen, B, go.shape[1], bare.shape[1], E, 2 * H, Z))
    # ---- the oracle's trajectory (fp32 CPU, per-parameter Adam over the alias list)
    P = O.tie_aliases({k: v.clone() for k, v in state0.items()})
    none_grad = ('BoW.', 'PosteriorGaussianCouple.toSigma.')
    okeys = [k for k in keys if not k.startswith(none_grad)]
    opt_state, ref_tot = {}, []
    for i in range(N_STEPS):
        cfg = dict(latent_dim=Z, lambda_kl=harness.lambda_kl(i + 1, hp), lambda_couple=hp.lambda_couple,
                   lambda_seq_couple=hp.lambda_seq_couple, euclidean_dist=False, lr=hp.lr, clip_norm=hp.clip_norm)
        losses, _, _ = O.train_step(P, 'vae-couple', cell, batches[i], noises[i], cfg, opt_state, okeys)
        ref_tot.append(losses['tot'])
    assert harness.lambda_kl(N_STEPS, hp) == 1.0 and harness.lambda_kl(6, hp) > 0
    ref_w = {k: P[k].detach() for k in keys}
    # ---- the product, both arithmetic modes
    report = {}
    for name, split in (('split-bf16', True), ('fp32-input', False)):
        tots, w = _run_product(dev, hp, state0, batches, noises, split)
        loss_err = [abs(a - b) / abs(b) for a, b in zip(tots, ref_tot)]
        w_err = {k: float((w[k] - ref_w[k]).abs().max() / (ref_w[k].abs().max() + 1e-30)) for k in keys}
        report[name] = (loss_err, w_err)
        print('%s %s H=%d: loss rel err by step  %s' % (name, cell, H, ' '.join('%.1e' % e for e in loss_err)))
        print('%s %s H=%d: worst final weight err %.2e (%s)' % (name, cell, H, max(w_err.values()),
                                                                 max(w_err, key=w_err.get)))
    for name, (loss_err, w_err) in report.items():
        assert max(loss_err) < 1e-5, (name, max(loss_err))
        assert max(w_err.values()) < 1e-3, (name, max(w_err, key=w_err.get), max(w_err.values()))
    s_loss, f_loss = max(report['split-bf16'][0]), max(report['fp32-input'][0])
    s_w, f_w = max(report['split-bf16'][1].values()), max(report['fp32-input'][1].values())
    assert s_loss <= 3 * f_loss + 2e-5 and s_w <= 3 * f_w + 1e-4, (s_loss, f_loss, s_w, f_w)
