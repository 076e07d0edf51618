"""The path a `scripts/run_reference.py` user executes: the mirror classes driven the way the reference's own
training loop drives them (CV/Experiment.py:589-600 optimizer construction, :509-522 step, :694-757 loss assembly;
CC/Experiment.py:386-428), i.e. WITHOUT cvae_amd.harness:

    model(go=, sent_len=, bare=)  ->  criterionSeq(LazyLogits, eos, keep_batch=True).mean(0)
    torch.distributions.kl_divergence(Normal, Normal).sum(1).mean(0)
    torch arithmetic on init / init_couple.detach() (Euclidean distance, IMQ kernel sum)
    trainable = [v for k, v in state_dict(keep_vars=True).items() if v.requires_grad]   (embedding listed per alias)
    zero_grad; tot.backward(); torch clip_grad_norm_(trainable, 30); torch.optim.Adam(trainable).step()

against the fixtures recorded from the reference (tests/golden): loss terms, every gradient and the post-step
weights of both recorded iterations -- the alias behaviour has to come out of torch's own
clip + Adam here, not out of cvae_clip_adam.

Both torch calls are made with foreach=False, which is what `scripts/run_reference.py` pins:
on a GPU torch >= 2 defaults them to multi-tensor kernels, which apply one in-place update concurrently to every
listing of the shared embedding (a data race, result undefined); the per-tensor loops are the semantics the reference
was written against and the ones its CPU run -- the fixtures -- records."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import Golden          # noqa: E402

LOSS_TOL, GRAD_TOL, POST_TOL = 1e-4, 1e-3, 2e-4
IMQ_SCALES = (.1, .2, .5, 1., 2., 5., 10.)

CASES = ['text_vae-couple_gru_tiny', 'text_vae-couple_lstm_tiny', 'text_vae-couple_lstm_mid',
         'text_vae-couple-euclid_gru_tiny', 'text_vae_gru_tiny', 'dialog_cvae-couple_gru_tiny',
         'dialog_cvae-couple_lstm_tiny', 'dialog_cvae_gru_tiny', 'text_vae-couple-bow_gru_tiny', 'text_vae-bow_lstm_tiny']


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'GPU tests need an MI355X'
    from cvae_amd import _lib
    _lib.init(0)
    return torch.device('cuda:0')


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def close(a, b, tol):
    a = a.detach() if torch.is_tensor(a) else a
    return abs(float(a) - float(b)) <= tol * abs(float(b)) + 1e-7


def _model(g, dev):
    from cvae_amd import harness
    m = g.meta
    hp = harness.HParams(model=m['model'], cell=m['cell'], V=m['V'], E=m['E'], H=m['H'], Z=m['Z'], dropout=m['dropout'],
                         max_len=m['max_len'])
    model = harness.build_model(hp, 'cpu', seed=0)
    model.load_state_dict({k: g.t('init/' + k) for k in m['state_keys']}, strict=True)
    return model.to(dev)


@pytest.mark.parametrize('name', CASES)
def test_reference_training_loop_pattern(dev, name):
    from torch.nn.utils import clip_grad_norm_
    from torch.optim import Adam
    from cvae_amd import modules
    from cvae_amd.settings import settings
    g = Golden(name)
    m = g.meta
    dialog, couple = m['model'].startswith('cvae'), m['model'].endswith('couple')
    model = _model(g, dev)
    assert settings.restore_row_order          # the caller's row order: what an Experiment.py user gets
    # Experiment.train: the alias list, and torch's own Adam over it
    trainable = [v for _, v in model.state_dict(keep_vars=True).items() if v.requires_grad]
    assert len(trainable) == m['trainable_entries'] and len({id(t) for t in trainable}) == m['trainable_unique']
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')        # "duplicate parameters": the reference's list, on purpose
        optim = Adam(params=trainable, lr=m['lr'], betas=[0.9, 0.999], weight_decay=0, foreach=False)
    sd = model.state_dict(keep_vars=True)
    model.train(True)
    try:
        for it in range(m['n_train_steps']):
            p = 'train/%d/' % it
            b = {k: v.to(dev) for k, v in g.batch(p).items()}
            modules.set_noise_source(modules.ReplayNoise(g.group(p + 'noise/')))
            if dialog:
                out = model(post_bare=b['post_bare'], post_len=b['post_len'], resp_go=b['resp_go'],
                            resp_len=b['resp_len'], resp_bare=b['resp_bare'])
                eos = b['resp_eos']
                if couple:
                    logits, prior, post, logits_c, init, init_c = out
                else:
                    logits, prior, post = out
            else:
                out = model(go=b['go'], sent_len=b['len'], bare=b['bare'])
                eos = b['eos']
                if couple:
                    logits, post, _, bow_logits, logits_c, init, init_c = out
                else:
                    logits, post, _, bow_logits = out
                prior = torch.distributions.Normal(torch.zeros(m['Z'], device=dev), torch.ones(m['Z'], device=dev))
            seq = model.criterionSeq(logits, eos, keep_batch=True).mean(0)
            kl_elem = torch.distributions.kl_divergence(post, prior)
            kl = kl_elem.sum(1).mean(0)
            lam = float(g.np(p + 'lambda_kl'))
            tot = seq + lam * kl
            if couple:
                seq_c = model.criterionSeq(logits_c, eos, keep_batch=True).mean(0)
                euc = ((init - init_c.detach()) ** 2).sum(-1)
                if m['euclidean_dist']:
                    cpl = euc
                else:
                    cpl = 0.
                    for s in IMQ_SCALES:
                        C = 2. * m['Z'] * s
                        cpl = cpl - C / (C + euc)
                tot = tot + seq_c * m['lambda_seq_couple'] + m['lambda_couple'] * cpl.mean(0)
            if m.get('lambda_BoW', 0.0) > 0:          # CV/Experiment.py:703-706, 711-713
                bow_logits = bow_logits.unsqueeze(1).expand(-1, b['bare'].shape[1], -1)
                bow_sum = model.criterionSeq(bow_logits, b['bare'], keep_batch=True)
                assert rel(bow_sum, g.t(p + 'bow_sum_per_sent')) < LOSS_TOL
                if lam > 1e-5:
                    tot = tot + m['lambda_BoW'] * (bow_sum / b['len'].float()).mean(0)
            optim.zero_grad()
            tot.backward()
            # ---- against the reference's recorded values
            assert close(seq, g.t(p + 'seq_per_sent').mean(), LOSS_TOL)
            assert rel(kl_elem, g.t(p + 'kl_elem')) < LOSS_TOL
            if couple:
                assert close(seq_c, g.t(p + 'seq_c_per_sent').mean(), LOSS_TOL)
                assert rel(euc, g.t(p + 'euc')) < LOSS_TOL
            assert close(tot, g.np(p + 'tot'), LOSS_TOL)
            for k in m['state_keys']:
                if k in g.none_grad_keys(it):
                    assert sd[k].grad is None or float(sd[k].grad.abs().max()) == 0.0, k
                else:
                    assert rel(sd[k].grad, g.t(p + 'grad/' + k)) < GRAD_TOL, (it, k)
            clip_grad_norm_(parameters=trainable, max_norm=m['clip_norm'], foreach=False)
            optim.step()
            for k in m['state_keys']:
                if (p + 'post/' + k) in g:
                    assert rel(sd[k], g.t(p + 'post/' + k)) < POST_TOL, (it, k)
    finally:
        modules.set_noise_source(None)
    # the aliases still are one tensor (Dec_Emb = Enc_Emb, CV/Experiment.py:93-94)
    emb = [v for k, v in sd.items() if k.endswith('Embedding.weight')]
    assert all(e.data_ptr() == emb[0].data_ptr() for e in emb)
