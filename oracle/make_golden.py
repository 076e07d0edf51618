"""Generate golden vectors by running the REFERENCE itself (CPU) in the build container.

    python oracle/make_golden.py            # regenerates every tests/golden/*.npz

The reference (/root/reference, read-only) is imported unmodified, one child process per
(tree, model, cell, variant) because both trees use the package name `Modules`.  Shims
(nothing is written under /root/reference):
  * sys.modules stubs for the absent h5py / nltk / seaborn and the missing
    `Modules.vae_nf_geo` (only `_n_sample = 2` is read from it);
  * scratch CWD with outputs/* and a synthetic corpus under data/<dataset>/;
  * config.py is read as text, the dataset / model selectors and the size constants are
    substituted (tiny shapes), and it is exec'd as module `config`;
  * recording wrappers around torch-level functions the reference calls
    (F.dropout -> explicit recorded mask, distributions' _standard_normal -> recorded eps,
    kl_divergence, Tensor.backward) and around the reference's own
    SeqLoss.forward / euclidean / Experiment.step, so that every noise draw, loss term,
    gradient and post-step weight of `Experiment.train()` is captured.

What is recorded per fixture: see `tests/golden/README.md`.
This script never runs on the GPU box (the reference does not travel).
"""
import io
import json
import os
import re
import subprocess
import sys
import tempfile
import types
import contextlib

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, 'tests', 'golden')
REF = '/root/reference'

TINY = dict(E=16, H=8, Z=4, B=5, max_len=12, n_train=5, n_val=7, n_test=33, vocab_words=46)
MID = dict(E=32, H=24, Z=8, B=8, max_len=40, n_train=11, n_val=9, n_test=20, vocab_words=160, lean=True)

JOBS = [
    # name, tree, dataset, model, cell, shape
    ('text_vae-couple_gru_tiny', 'Coupled-VAE', 'ptb', 'vae-couple', 'gru', TINY),
    ('text_vae-couple_lstm_tiny', 'Coupled-VAE', 'ptb', 'vae-couple', 'lstm', TINY),
    ('text_vae_gru_tiny', 'Coupled-VAE', 'ptb', 'vae', 'gru', TINY),
    ('text_vae_lstm_tiny', 'Coupled-VAE', 'ptb', 'vae', 'lstm', TINY),
    ('text_vae-couple-euclid_gru_tiny', 'Coupled-VAE', 'yahoo', 'vae-couple', 'gru', TINY),
    ('text_vae-couple_gru_mid', 'Coupled-VAE', 'ptb', 'vae-couple', 'gru', MID),
    ('text_vae-couple_lstm_mid', 'Coupled-VAE', 'ptb', 'vae-couple', 'lstm', MID),
    ('dialog_cvae-couple_gru_tiny', 'Coupled-CVAE', 'switchboard', 'cvae-couple', 'gru', TINY),
    ('dialog_cvae-couple_lstm_tiny', 'Coupled-CVAE', 'switchboard', 'cvae-couple', 'lstm', TINY),
    ('dialog_cvae_gru_tiny', 'Coupled-CVAE', 'switchboard', 'cvae', 'gru', TINY),
    # sibling variants on the same encoder / decoder.  The annealing constants are
    # substituted so that lambda_KL is live in the two recorded iterations: beta-vae sits at its
    # cap (config.beta), the cyclic schedule is inside its first ramp.
    ('text_beta-vae_gru_tiny', 'Coupled-VAE', 'ptb', 'beta-vae', 'gru',
     dict(TINY, annealing_start=-10000000, annealing_step=10)),
    ('text_cyc-anneal-vae-couple_lstm_tiny', 'Coupled-VAE', 'ptb', 'cyc-anneal-vae-couple', 'lstm',
     dict(TINY, annealing_start=0, annealing_step=80)),
    # free bits (CV/Experiment.py:1131-1165).  lambda_fb is substituted (shipped: 6, i.e. 1.5 nats per latent
    # dimension at Z = 4, which would clamp EVERY dimension of a fresh model) so that clamped and un-clamped
    # dimensions both occur; lambda_KL = 1 as for beta-vae.  The model restores a pretrained DAE at construction
    # (Experiment.py:384-410): torch.load of that checkpoint and DAE.load_state_dict are shimmed to no-ops, i.e. the
    # "pretrained" encoder is the freshly initialised DAE's, which the reference then copies as it always does.
    ('text_surprising-fix_gru_tiny', 'Coupled-VAE', 'ptb', 'surprising-fix', 'gru',
     dict(TINY, annealing_start=-10000000, annealing_step=10, lambda_fb=0.3)),
    ('text_dae_gru_tiny', 'Coupled-VAE', 'ptb', 'dae', 'gru', TINY),
    ('text_lm_gru_tiny', 'Coupled-VAE', 'ptb', 'lm', 'gru', TINY),
    # bag-of-words loss (CV/Experiment.py:703-715, 737-740, 759-761; lambda_BoW is 0 in every shipped config and is
    # substituted here).  Four recorded iterations (two epochs of two batches) across the reference's `lambda_KL > 1e-5` gate: with
    # annealing_start = annealing_step = 1 lambda_KL is 0 at iteration 1 (the BoW term is left out of the total: the
    # BoW head's gradient is None and torch's clip / Adam skip it) and 1 from iteration 2 on (the term enters, and
    # the BoW head takes ITS first Adam step at the model's second).
    ('text_vae-couple-bow_gru_tiny', 'Coupled-VAE', 'ptb', 'vae-couple', 'gru',
     dict(TINY, annealing_start=1, annealing_step=1, lambda_BoW=0.5, num_iters=4)),
    ('text_vae-bow_lstm_tiny', 'Coupled-VAE', 'ptb', 'vae', 'lstm',
     dict(TINY, annealing_start=1, annealing_step=1, lambda_BoW=0.5, num_iters=4)),
]
BASE_MODEL = {'beta-vae': 'vae', 'cyc-anneal-vae': 'vae', 'cyc-anneal-vae-couple': 'vae-couple', 'surprising-fix': 'vae'}

NOISE_ORDER = {
    # training-forward draw order (SURVEY 8a "RNG draw order")
    'vae': ['enc_drop', 'eps', 'dec_in_drop', 'dec_out_drop'],
    'vae-couple': ['enc_drop', 'eps', 'eps_couple', 'dec_in_drop', 'dec_out_drop',
                   'decc_in_drop', 'decc_out_drop'],
    'dae': ['enc_drop', 'dec_in_drop', 'dec_out_drop'],
    'lm': ['dec_in_drop', 'dec_out_drop'],
    'cvae': ['post_enc_drop', 'resp_enc_drop', 'eps_prior', 'eps', 'dec_in_drop', 'dec_out_drop'],
    'cvae-couple': ['post_enc_drop', 'resp_enc_drop', 'eps_prior', 'eps', 'eps_couple',
                    'dec_in_drop', 'dec_out_drop', 'decc_in_drop', 'decc_out_drop'],
}


# ----------------------------------------------------------------------------- child
def _install_stubs():
    for name in ('h5py', 'seaborn'):
        sys.modules[name] = types.ModuleType(name)
    sys.modules['seaborn'].set = lambda *a, **k: None
    nltk = types.ModuleType('nltk')
    nltk.bigrams = lambda x: list(zip(x, x[1:]))
    nltk.word_tokenize = lambda s: s.split()
    tr = types.ModuleType('nltk.translate')
    bs = types.ModuleType('nltk.translate.bleu_score')

    class SmoothingFunction(object):
        method1 = None

    bs.SmoothingFunction = SmoothingFunction
    bs.corpus_bleu = lambda *a, **k: 0.0
    bs.sentence_bleu = lambda *a, **k: 0.0
    nltk.translate = tr
    tr.bleu_score = bs
    sys.modules.update({'nltk': nltk, 'nltk.translate': tr, 'nltk.translate.bleu_score': bs})
    nf = types.ModuleType('Modules.vae_nf_geo')
    nf._n_sample = 2
    sys.modules['Modules.vae_nf_geo'] = nf


def _patched_config_source(tree, dataset, model, cell, shp):
    src = open(os.path.join(REF, tree, 'config.py')).read()
    if tree == 'Coupled-VAE':
        ds_list = ['ptb', 'bc', 'yelp', 'yahoo']
        src = src.replace("['ptb', 'bc', 'yelp', 'yahoo'][2]",
                          "['ptb', 'bc', 'yelp', 'yahoo'][%d]" % ds_list.index(dataset))
        models = re.search(r"self\.model = \[(.*?)\]\[0\]", src, re.S)
        names = re.findall(r"'([^']+)'", models.group(1))
        src = src.replace(models.group(0), models.group(0)[:-3] + '[%d]' % names.index(model))
        if dataset == 'yahoo':   # shipped-config bug, SURVEY fact 5
            src = src.replace("self.num_workers = 8", "self.num_workers = 8\n"
                              "        if not hasattr(self, 'lambda_seq_couple'):\n"
                              "            self.lambda_seq_couple = 1", 1)
    else:
        names = ['seq2seq', 'cvae', 'cvae-couple']
        src = src.replace("][2]  # TODO\n        self.train_mode",
                          "][%d]  # TODO\n        self.train_mode" % names.index(model))
    src = src.replace('self.use_lstm = False', 'self.use_lstm = %s' % (cell == 'lstm'))
    src = re.sub(r'self\.hid_dim = \d+', 'self.hid_dim = %d' % shp['H'], src)
    src = re.sub(r'self\.emb_dim = \d+', 'self.emb_dim = %d' % shp['E'], src)
    src = re.sub(r'self\.latent_dim = \d+', 'self.latent_dim = %d' % shp['Z'], src)
    src = src.replace('self.max_len = 200', 'self.max_len = %d' % shp['max_len'])
    src = src.replace('self.batch_size = 32', 'self.batch_size = %d' % shp['B'])
    src = src.replace('self.num_workers = 8', 'self.num_workers = 0')
    src = src.replace('self.use_tensorboard = True', 'self.use_tensorboard = False')
    src = src.replace('self.num_iters = 60000', 'self.num_iters = %d' % shp.get('num_iters', 2))
    if 'lambda_BoW' in shp:
        src = src.replace('self.lambda_BoW = 0', 'self.lambda_BoW = %r' % shp['lambda_BoW'])
    if 'annealing_start' in shp:
        src = re.sub(r'self\.annealing_start = \d+', 'self.annealing_start = %d' % shp['annealing_start'], src)
        src = re.sub(r'self\.annealing_step = \d+', 'self.annealing_step = %d' % shp['annealing_step'], src)
    if 'lambda_fb' in shp:
        src = re.sub(r'self\.lambda_fb = \d+', 'self.lambda_fb = %r' % shp['lambda_fb'], src)
    return src


def _write_corpus(tree, dataset, shp):
    rs = np.random.RandomState(1234)
    words = ['w%d' % i for i in range(shp['vocab_words'])]
    root = os.path.join('data', dataset)
    os.makedirs(root)
    cap = shp['max_len'] + 3       # some sentences exceed max_len -> a1 truncation

    def sent(first=None):
        n = int(rs.randint(1, cap + 1))
        s = [words[int(i)] for i in rs.randint(0, len(words), size=n)]
        return s

    for split, n in (('train', shp['n_train']), ('valid', shp['n_val']), ('test', shp['n_test'])):
        with open(os.path.join(root, split + '.txt'), 'w') as f:
            if split == 'train':
                # every word appears >= 5 times in train so that the dialog vocab (min_occur=5)
                # and the text vocab (min_occur=1; lookups raise KeyError on OOV) cover it.
                for rep in range(5):
                    line = ' '.join(words[rep::5] if False else words)
                    if tree == 'Coupled-CVAE':
                        f.write(line + '\t' + line + '\n')
                    else:
                        f.write(line + '\n')
            for i in range(n):
                if tree == 'Coupled-CVAE':
                    f.write(' '.join(sent()) + '\t' + ' '.join(sent()) + '\n')
                else:
                    f.write(' '.join(sent()) + '\n')


def _analysis_fixtures(exp, ae, E, rec, out, meta, shp, base, tb):
    """gen_interps / sample_from_posterior / saliency of the text models (CV/Modules/vae_couple.py:583-633, 657-679,
    743-778; vae.py:230-278, 348-374) on the first sentences of the test set, with the weights under 'eval/'.
    Recorded: inputs, every noise draw, and what the reference's callers compute from the results
    (Experiment.interpolation: decoded ids; Experiment.analyze_saliency, CV/Experiment.py:1543-1619: the gradient of
    each loss term with respect to the encoder state, and dhde)."""
    import torch
    from Modules.Losses.SeqLoss import SeqLoss
    bare, go, eos, ln = tb
    B, Z = shp['B'], shp['Z']
    go1 = torch.zeros(B, 1, dtype=torch.long) + exp.test_set.go
    # ---- gen_interps: evaluation mode (greedy decode of n + 2 points between the two posterior means)
    ae.train(False)
    A, Bs = slice(0, B), slice(B, 2 * B)
    n_interps = 3
    with torch.no_grad():
        interps = ae.gen_interps(bareA=bare[A], sent_lenA=ln[A], bareB=bare[Bs], sent_lenB=ln[Bs], go=go1,
                                 n_interps=n_interps)
    out['interp/bareA'], out['interp/lenA'] = bare[A].numpy(), ln[A].numpy()
    out['interp/bareB'], out['interp/lenB'] = bare[Bs].numpy(), ln[Bs].numpy()
    out['interp/go'] = go1.numpy()
    out['interp/ids'] = np.asarray([[[int(t.item()) for t in sent] for sent in per_b] for per_b in interps],
                                   dtype=np.int64)                       # (B, n_interps + 2, max_len)
    meta['n_interps'] = n_interps
    # ---- sample_from_posterior: Normal.sample draws through torch.normal(mean, std), which computes
    # mean + std * N(0,1): the shim below does exactly that with a recorded standard-normal draw
    n_sample = 5
    _orig_normal = torch.normal
    drawn = []

    def rec_normal(mean, std, *a, **k):
        e = torch.randn(mean.shape)
        drawn.append(e.clone())
        return mean + std * e

    torch.normal = rec_normal
    try:
        with torch.no_grad():
            smp = ae.sample_from_posterior(bare=bare[A], sent_len=ln[A], n_sample=n_sample)
    finally:
        torch.normal = _orig_normal
    assert len(drawn) == 1 and tuple(drawn[0].shape) == (n_sample, B, Z)
    out['postsample/eps'], out['postsample/samples'] = drawn[0].numpy(), smp.numpy()   # samples (B, n_sample, Z)
    # ---- saliency: training mode (analyze_saliency calls set_training(True): dropout is live)
    ae.train(True)
    crit = SeqLoss(voc_size=exp.test_set.vocab.size, pad=exp.test_set.pad, end=exp.test_set.eos, unk=exp.test_set.unk)
    rec['noise'].clear()
    res = ae.saliency(bare=bare[A], sent_len=ln[A], go=go[A])
    names = NOISE_ORDER[base]
    assert len(rec['noise']) == len(names)
    for nm, (kind, t) in zip(names, rec['noise']):
        assert (kind == 'eps') == nm.startswith('eps'), (nm, kind)
        out['saliency/noise/' + nm] = t.squeeze(0).numpy() if kind == 'eps' else t.numpy()
    if base == 'vae':
        logits, dist, init, last = res
        logits_c = None
    else:
        logits, logits_c, dist, init, init_c, last = res
    grad = lambda y: torch.autograd.grad(outputs=y, inputs=last, create_graph=False, retain_graph=True,
                                         only_inputs=True)[0]
    seq = crit(logits, eos[A], keep_batch=True).sum(0)
    std = torch.distributions.Normal(torch.zeros(Z), torch.ones(Z))
    kl = torch.distributions.kl_divergence(dist, std).sum(1).sum(0)
    out['saliency/bare'], out['saliency/go'] = bare[A].numpy(), go[A].numpy()
    out['saliency/eos'], out['saliency/len'] = eos[A].numpy(), ln[A].numpy()
    out['saliency/last'], out['saliency/init'] = last.detach().numpy(), init.detach().numpy()
    out['saliency/seq_sum'] = np.float64(seq.item())
    out['saliency/d_seq'] = grad(seq).numpy()
    out['saliency/d_kl'] = grad(kl).numpy()
    if logits_c is not None:
        seq_c = crit(logits_c, eos[A], keep_batch=True).sum(0)
        out['saliency/init_couple'] = init_c.detach().numpy()
        out['saliency/seq_c_sum'] = np.float64(seq_c.item())
        out['saliency/d_seq_c'] = grad(seq_c).numpy()
        out['saliency/d_seq_tot'] = grad(seq + seq_c).numpy()
    dhde = 0
    for _init in init.sum(0):
        dhde = dhde + (grad(_init) ** 2).sum(1)
    out['saliency/dhde'] = (dhde / torch.sum(init ** 2, dim=1)).detach().numpy()
    rec['noise'].clear()
    ae.train(False)


def child(name, tree, dataset, model, cell, shp, out_path):
    sys.dont_write_bytecode = True
    scratch = tempfile.mkdtemp(prefix='cvae_golden_')
    os.chdir(scratch)
    for d in ('logs', 'saved_model', 'sampled_results', 'temp_results'):
        os.makedirs(os.path.join('outputs', d))
    _write_corpus(tree, dataset, shp)
    _install_stubs()
    cfg_mod = types.ModuleType('config')
    exec(compile(_patched_config_source(tree, dataset, model, cell, shp), 'config.py', 'exec'),
         cfg_mod.__dict__)
    sys.modules['config'] = cfg_mod
    sys.path.insert(0, os.path.join(REF, tree))

    import torch
    import torch.nn.functional as F
    torch.set_num_threads(1)

    rec = {'noise': [], 'seq': [], 'kl': [], 'euc': [], 'tot': [], 'fwd': []}

    def rec_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        mask = torch.bernoulli(torch.full_like(x, 1 - p)) / (1 - p)
        rec['noise'].append(('drop', mask.detach().clone()))
        return x * mask

    F.dropout = rec_dropout
    import torch.distributions.normal as tdn
    _orig_sn = tdn._standard_normal

    def rec_standard_normal(shape, dtype, device):
        e = _orig_sn(shape, dtype=dtype, device=device)
        rec['noise'].append(('eps', e.detach().clone()))
        return e

    tdn._standard_normal = rec_standard_normal
    _orig_kl = torch.distributions.kl_divergence

    def rec_kl(p, q):
        r = _orig_kl(p, q)
        rec['kl'].append(r.detach().clone())
        return r

    torch.distributions.kl_divergence = rec_kl
    _orig_bw = torch.Tensor.backward

    def rec_backward(self, *a, **k):
        rec['tot'].append(float(self.detach()))
        return _orig_bw(self, *a, **k)

    torch.Tensor.backward = rec_backward

    with contextlib.redirect_stdout(io.StringIO()):
        import Experiment as E
    from Modules.Losses.SeqLoss import SeqLoss
    _orig_seq = SeqLoss.forward

    def rec_seq(self, logits, gts, keep_batch=False):
        r = _orig_seq(self, logits, gts, keep_batch=keep_batch)
        rec['seq'].append(r.detach().clone())
        return r

    SeqLoss.forward = rec_seq
    _orig_euc = E.euclidean

    def rec_euc(a, b, clip=None):
        r = _orig_euc(a, b, clip=clip)
        rec['euc'].append(r.detach().clone())
        return r

    E.euclidean = rec_euc

    if model.startswith('surprising-fix'):
        # no pretrained DAE checkpoint exists here: the freshly initialised DAE stands in for it (see JOBS)
        _orig_load = torch.load
        torch.load = lambda *a, **k: {}
        E.DAE.load_state_dict = lambda self, sd, strict=True: None
    with contextlib.redirect_stdout(io.StringIO()):
        exp = E.Experiment()
    if model.startswith('surprising-fix'):
        torch.load = _orig_load
    ae = exp.AutoEncoder
    keys_all = list(ae.state_dict().keys())
    out = {}
    meta = dict(name=name, tree=tree, dataset=dataset, model=model, cell=cell,
                E=shp['E'], H=shp['H'], Z=shp['Z'], B=shp['B'], max_len=shp['max_len'],
                V=int(exp.vocab.size), state_keys=keys_all,
                lambda_couple=getattr(E.config, 'lambda_couple', 0.0),
                lambda_seq_couple=getattr(E.config, 'lambda_seq_couple', 1),
                euclidean_dist=bool(getattr(E.config, 'euclidean_dist', False)),
                annealing_start=getattr(E.config, 'annealing_start', 0),
                annealing_step=getattr(E.config, 'annealing_step', 1),
                beta=getattr(E.config, 'beta', 1.0), cyc_M=getattr(E.config, 'M', 0), cyc_R=getattr(E.config, 'R', 0.0),
                lambda_fb=float(getattr(E.config, 'lambda_fb', 0.0)),
                dropout=E.config.dropout, clip_norm=E.config.clip_norm, lr=E.config.gen_lr,
                torch_version=torch.__version__)

    def sd_np(prefix):
        for k, v in ae.state_dict().items():
            out[prefix + k] = v.detach().cpu().numpy().copy()

    # a1 fixtures: raw sentences as ids + what the reference's Dataset emits.
    ds = exp.test_set
    raw_ids, rows = [], []
    sents = ds.get_sentences()
    if tree == 'Coupled-CVAE':
        sents = sents[1]
    for i in range(min(len(ds), 12)):
        ids = [ds.vocab.word2id[w] for w in sents[i]]
        raw_ids.append(ids)
        b, g, e, l = ds.process_raw_sent(sents[i], ds.max_len)
        rows.append((b.numpy(), g.numpy(), e.numpy(), int(l)))
    out['a1/raw_flat'] = np.asarray([t for ids in raw_ids for t in ids], dtype=np.int64)
    out['a1/raw_offsets'] = np.cumsum([0] + [len(ids) for ids in raw_ids]).astype(np.int64)
    out['a1/bare'] = np.stack([r[0] for r in rows])
    out['a1/go'] = np.stack([r[1] for r in rows])
    out['a1/eos'] = np.stack([r[2] for r in rows])
    out['a1/len'] = np.asarray([r[3] for r in rows], dtype=np.int64)
    # a2 fixtures
    from utils.utils import sort_resort
    for j, lens in enumerate([[7, 3, 5, 3, 1], [2, 2, 2], [1], [4, 9, 9, 1, 4, 9, 2]]):
        s, r = sort_resort(torch.LongTensor(lens))
        out['a2/%d/len' % j] = np.asarray(lens, dtype=np.int64)
        out['a2/%d/s_idx' % j] = np.asarray(s, dtype=np.int64)
        out['a2/%d/res_idx' % j] = np.asarray(r, dtype=np.int64)
    meta['a2_cases'] = 4

    sd_np('init/')

    # ---- Experiment.train(): 2 iterations, everything recorded -------------------
    batches = []
    _orig_pre = exp.preprocess_data

    def rec_pre(data):
        batches.append([d.clone() for d in data])
        return _orig_pre(data)

    exp.preprocess_data = rec_pre
    steps = []
    _orig_step = exp.step

    def rec_step(scopes):
        g = {}
        for k, v in ae.state_dict(keep_vars=True).items():
            g[k] = None if v.grad is None else v.grad.detach().clone()
        trainable = list(exp.trainable_gen)
        _orig_step(scopes)
        w = {k: v.detach().clone() for k, v in ae.state_dict().items()}
        steps.append((g, w, len(trainable), len({id(t) for t in trainable}), float(getattr(exp, 'lambda_KL', 0.0))))

    exp.step = rec_step

    def fwd_hook(mod, inp, outp):
        rec['fwd'].append(outp)

    h = ae.register_forward_hook(fwd_hook)
    with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        exp.train()
    h.remove()
    exp.step = _orig_step
    exp.preprocess_data = _orig_pre
    n_it = shp.get('num_iters', 2)
    assert len(steps) == n_it, len(steps)
    meta['n_train_steps'] = n_it
    meta['lambda_BoW'] = float(shp.get('lambda_BoW', 0.0))
    meta['trainable_entries'] = steps[0][2]
    meta['trainable_unique'] = steps[0][3]
    base = BASE_MODEL.get(model, model)
    names = NOISE_ORDER[base]
    per = len(names)
    assert len(rec['noise']) == n_it * per, (len(rec['noise']), per)
    coupled = model.endswith('couple')
    lean = bool(shp.get('lean'))
    with_bow = shp.get('lambda_BoW', 0.0) > 0
    meta['none_grad_keys_per_step'] = []
    for it in range(n_it):
        p = 'train/%d/' % it
        data = batches[it]
        if tree == 'Coupled-VAE':
            bare, go, eos, ln = data
            m = int(ln.max())
            out[p + 'bare'], out[p + 'go'] = bare[:, :m].numpy(), go[:, :m + 1].numpy()
            out[p + 'eos'], out[p + 'len'] = eos[:, :m + 1].numpy(), ln.numpy()
        else:
            post_bare, post_len, resp_bare, resp_go, resp_eos, resp_len = data
            m = int(post_len.max())
            out[p + 'post_bare'], out[p + 'post_len'] = post_bare[:, :m].numpy(), post_len.numpy()
            m = int(resp_len.max())
            out[p + 'resp_bare'], out[p + 'resp_len'] = resp_bare[:, :m].numpy(), resp_len.numpy()
            out[p + 'resp_go'], out[p + 'resp_eos'] = resp_go[:, :m + 1].numpy(), resp_eos[:, :m + 1].numpy()
        for j, nm in enumerate(names):
            kind, t = rec['noise'][it * per + j]
            assert (kind == 'eps') == nm.startswith('eps'), (nm, kind)
            out[p + 'noise/' + nm] = t.squeeze(0).numpy() if kind == 'eps' else t.numpy()
        fw = rec['fwd'][it]
        if base == 'lm':
            logits, dist = fw, None                     # LM.forward returns the logits tensor itself
        elif base == 'dae':
            logits, dist = fw[0], None
            out[p + 'latent'] = fw[1].detach().numpy()
        elif tree == 'Coupled-VAE':
            logits, dist, z = fw[0], fw[1], fw[2]
            if coupled:
                if not lean:
                    out[p + 'logits_couple'] = fw[4].detach().numpy()
                out[p + 'init'], out[p + 'init_couple'] = fw[5].detach().numpy(), fw[6].detach().numpy()
        else:
            logits, prior, dist = fw[0], fw[1], fw[2]
            out[p + 'mu_prior'] = prior.loc.detach().numpy()
            out[p + 'sigma_prior'] = prior.scale.detach().numpy()
            if coupled:
                if not lean:
                    out[p + 'logits_couple'] = fw[3].detach().numpy()
                out[p + 'init'], out[p + 'init_couple'] = fw[4].detach().numpy(), fw[5].detach().numpy()
        if not lean:
            out[p + 'logits'] = logits.detach().numpy()
        if dist is not None:
            out[p + 'mu'], out[p + 'sigma'] = dist.loc.detach().numpy(), dist.scale.detach().numpy()
        nseq = (2 if coupled else 1) + (1 if with_bow else 0)   # criterionSeq calls per iteration: seq [, seq_c] [, bow]
        out[p + 'seq_per_sent'] = rec['seq'][it * nseq].numpy()
        if with_bow:
            # criterionSeq(BoW_logits expanded, bare, keep_batch=True): per-sentence sums BEFORE the division by sent_len
            out[p + 'bow_sum_per_sent'] = rec['seq'][it * nseq + nseq - 1].numpy()
        if coupled:
            out[p + 'seq_c_per_sent'] = rec['seq'][it * nseq + 1].numpy()
            out[p + 'euc'] = rec['euc'][it].numpy()
        if base not in ('dae', 'lm'):
            out[p + 'kl_elem'] = rec['kl'][it].numpy()
        out[p + 'tot'] = np.float64(rec['tot'][it])
        out[p + 'lambda_kl'] = np.float64(steps[it][4])   # 0 in the base fixtures: iter 1,2 << annealing_start
        g, w = steps[it][0], steps[it][1]
        for k in keys_all:
            if g[k] is not None:
                out[p + 'grad/' + k] = g[k].numpy()
            if not lean or it == n_it - 1:
                out[p + 'post/' + k] = w[k].numpy()
        meta.setdefault('none_grad_keys', [k for k in keys_all if g[k] is None])
        meta['none_grad_keys_per_step'].append([k for k in keys_all if g[k] is None])
    rec['noise'].clear(); rec['seq'].clear(); rec['kl'].clear(); rec['euc'].clear()

    def finish():
        # lambda_KL schedule of this model at the SHIPPED annealing constants (Experiment.update_lambda_kl)
        if tree == 'Coupled-VAE' and hasattr(E.config, 'kl_annealing'):
            E.config.annealing_start, E.config.annealing_step = 2000, 40000
            grid = [0, 1, 1999, 2000, 2001, 2500, 4000, 6999, 7000, 7001, 9500, 12000, 12001, 17000, 21999, 22000,
                    22001, 30000, 41999, 42000, 42001, 50000, 10 ** 6]
            vals = []
            for it in grid:
                exp.iter_num = it
                exp.update_lambda_kl()
                vals.append(float(exp.lambda_KL))
            out['sched/iters'] = np.asarray(grid, dtype=np.int64)
            out['sched/lambda'] = np.asarray(vals, dtype=np.float64)
        out['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        np.savez_compressed(out_path, **out)
        print('wrote', out_path, '%.1f KB' % (os.path.getsize(out_path) / 1024))

    from torch.utils.data import DataLoader
    if base == 'lm':
        # LM.test_lm (Modules/lm.py:218-235): one pass, kl = 0, nll = xent
        ae.train(False)
        vb = next(iter(DataLoader(exp.val_set, batch_size=shp['B'], shuffle=False)))
        bare, go, eos, ln = vb
        with torch.no_grad():
            xent, nll, kl, lat = ae.test_lm(bare=bare, sent_len=ln, go=go, eos=eos, n_sample=2)
        assert bool((kl == 0).all()) and bool((nll == xent).all())
        out['iw/lm/bare'], out['iw/lm/go'], out['iw/lm/eos'], out['iw/lm/len'] = \
            bare.numpy(), go.numpy(), eos.numpy(), ln.numpy()
        out['iw/lm/xent'] = xent.numpy()
        sd_np('eval/')
        finish()
        return
    if base == 'dae':
        # DAE.test_lm (Modules/dae.py): one deterministic decode, nll = kl = inf by construction
        ae.train(False)
        vb = next(iter(DataLoader(exp.val_set, batch_size=shp['B'], shuffle=False)))
        bare, go, eos, ln = vb
        with torch.no_grad():
            xent, nll, kl, lat = ae.test_lm(bare=bare, sent_len=ln, go=go, eos=eos, n_sample=2)
        assert bool(torch.isinf(nll).all()) and bool(torch.isinf(kl).all())
        out['iw/dae/bare'], out['iw/dae/go'], out['iw/dae/eos'], out['iw/dae/len'] = \
            bare.numpy(), go.numpy(), eos.numpy(), ln.numpy()
        out['iw/dae/xent'], out['iw/dae/latent'] = xent.numpy(), lat.numpy()
        sd_np('eval/')
        finish()
        return

    # ---- IW-NLL: test_lm on one val batch, K = 2 and K = 10 -----------------------
    ae.train(False)
    vb = next(iter(DataLoader(exp.val_set, batch_size=shp['B'], shuffle=False)))
    with torch.no_grad():
        for K in (2, 10):
            p = 'iw/K%d/' % K
            rec['noise'].clear()
            if tree == 'Coupled-VAE':
                bare, go, eos, ln = vb
                xent, nll, kl, zs = ae.test_lm(bare=bare, sent_len=ln, go=go, eos=eos, n_sample=K)
                out[p + 'bare'], out[p + 'go'], out[p + 'eos'], out[p + 'len'] = \
                    bare.numpy(), go.numpy(), eos.numpy(), ln.numpy()
                out[p + 'z'] = zs.numpy()
            else:
                post_bare, post_len, resp_bare, resp_go, resp_eos, resp_len = vb
                m = int(post_len.max()); post_bare = post_bare[:, :m].contiguous()
                m = int(resp_len.max()); resp_bare = resp_bare[:, :m].contiguous()
                resp_go = resp_go[:, :m + 1].contiguous(); resp_eos = resp_eos[:, :m + 1].contiguous()
                xent, nll, kl = ae.test_lm(post_bare=post_bare, post_len=post_len, resp_go=resp_go,
                                           resp_len=resp_len, resp_bare=resp_bare,
                                           resp_eos=resp_eos, n_sample=K)
                for nm, t in (('post_bare', post_bare), ('post_len', post_len), ('resp_bare', resp_bare),
                              ('resp_go', resp_go), ('resp_eos', resp_eos), ('resp_len', resp_len)):
                    out[p + nm] = t.numpy()
            eps = [t for kind, t in rec['noise'] if kind == 'eps']
            # dialog test_lm draws prior + posterior rsample((1,B,Z)) first (discarded)
            eps = [t for t in eps if t.shape[0] == 2]
            assert len(eps) == K // 2
            out[p + 'eps'] = torch.cat(eps, 0).numpy()
            out[p + 'xent'], out[p + 'nll'], out[p + 'kl'] = xent.numpy(), nll.numpy(), kl.numpy()

    # ---- corpus loops: language_modeling / evaluate_dialogue ('val', K=10) --------
    rec['noise'].clear()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
        if tree == 'Coupled-VAE':
            exp.language_modeling('val')
        else:
            exp.evaluate_dialogue('val')
    txt = buf.getvalue()
    for nm in ('nll', 'xent', 'kl', 'ppl'):
        out['corpus/' + nm] = np.float64(re.search(r'^%s = (\S+)' % nm, txt, re.M).group(1))
    eps = [t for kind, t in rec['noise'] if kind == 'eps' and t.shape[0] == 2]
    nb = (shp['n_val'] + shp['B'] - 1) // shp['B']
    assert len(eps) == nb * 5, (len(eps), nb)
    for b in range(nb):
        out['corpus/eps/%d' % b] = torch.cat(eps[b * 5:(b + 1) * 5], 0).numpy()
    meta['corpus_batches'] = nb
    meta['corpus_K'] = 10
    # the val set as raw id lists so that the build can re-batch it itself (a1 + a16)
    vs = exp.val_set
    if tree == 'Coupled-VAE':
        sv = vs.get_sentences()
        flat = [[vs.vocab.word2id[w] for w in s] for s in sv]
        out['corpus/raw_flat'] = np.asarray([t for s in flat for t in s], dtype=np.int64)
        out['corpus/raw_offsets'] = np.cumsum([0] + [len(s) for s in flat]).astype(np.int64)
    else:
        posts, resps = vs.get_sentences()
        for nm, ss in (('post', posts), ('resp', resps)):
            flat = [[vs.vocab.word2id.get(w, vs.unk) for w in s] for s in ss]
            out['corpus/%s_raw_flat' % nm] = np.asarray([t for s in flat for t in s], dtype=np.int64)
            out['corpus/%s_raw_offsets' % nm] = np.cumsum([0] + [len(s) for s in flat]).astype(np.int64)

    # ---- MI (text tree only): estimate_mi() over the 33-sentence test set ---------
    if tree == 'Coupled-VAE':
        sd_np('eval/')     # weights used by iw/corpus/mi (= after 2 train steps)
        rec['noise'].clear()
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            exp.estimate_mi()
        out['mi/avg'] = np.float64(re.search(r'Mutual Information = (\S+)', buf.getvalue()).group(1))
        eps = [t for kind, t in rec['noise'] if kind == 'eps' and t.shape[0] == 2]
        assert len(eps) == 50
        out['mi/eps'] = torch.cat(eps, 0).numpy()
        tb = next(iter(DataLoader(exp.test_set, batch_size=513, shuffle=False)))
        bare, go, eos, ln = tb
        out['mi/bare'], out['mi/go'], out['mi/len'] = bare.numpy(), go.numpy(), ln.numpy()
        # per-sentence values for B = 5 and B = n_test with K = 4
        ae.train(False)   # estimate_mi() above leaves the model in training mode
        for Bm in (5, shp['n_test']):
            rec['noise'].clear()
            with torch.no_grad():
                mi, _ = ae.estimate_mi(go=go[:Bm], sent_len=ln[:Bm], bare=bare[:Bm], n_sample=4)
            eps = [t for kind, t in rec['noise'] if kind == 'eps' and t.shape[0] == 2]
            out['mi/B%d/eps' % Bm] = torch.cat(eps, 0).numpy()
            out['mi/B%d/mi' % Bm] = mi.numpy()
        # ---- greedy decoding (8f.3): decode_from on fixed latents, max_len steps, <unk> masked ----
        ae.train(False)
        gen = torch.Generator().manual_seed(77)
        lat = torch.randn(shp['B'], shp['Z'], generator=gen)
        go1 = torch.zeros(shp['B'], 1, dtype=torch.long) + exp.test_set.go
        with torch.no_grad():
            sents_g = ae.decode_from(lat, go1)
        out['greedy/latents'], out['greedy/go'] = lat.numpy(), go1.numpy()
        out['greedy/ids'] = np.asarray([[int(t.item()) for t in srow] for srow in sents_g], dtype=np.int64)
        meta['greedy_steps'] = int(out['greedy/ids'].shape[1])
        # ---- beam branch (8f.3): same latents with Decoder.beam_size = 3 (an instance attribute read in
        # forward; every shipped config has 1).  GRU only: the branch indexes the state as one tensor.
        if cell == 'gru':
            ae.Decoder.beam_size = 3
            with torch.no_grad():
                sents_b = ae.decode_from(lat, go1)
            ae.Decoder.beam_size = 1
            out['beam/ids'] = np.asarray([[int(t.item()) for t in srow] for srow in sents_b], dtype=np.int64)
            meta['beam_size'] = 3
        if not lean:
            _analysis_fixtures(exp, ae, E, rec, out, meta, shp, base, tb)
    else:
        sd_np('eval/')
        # ---- generation (8f.3): sample_from_prior on one val batch -- post encoder, z from the prior
        # network (its eps recorded), greedy decode; and, GRU only, the beam branch with beam_size = 3 ----
        ae.train(False)
        post_bare, post_len, resp_bare, resp_go, resp_eos, resp_len = vb
        m = int(post_len.max()); post_bare = post_bare[:, :m].contiguous()
        out['gen/post_bare'], out['gen/post_len'], out['gen/resp_go'] = post_bare.numpy(), post_len.numpy(), resp_go.numpy()
        for tag, K in (('greedy', 1), ('beam', 3)):
            if K > 1 and cell != 'gru':
                continue
            rec['noise'].clear()
            ae.Decoder.beam_size = K
            with torch.no_grad():
                preds = ae.sample_from_prior(post_bare=post_bare, post_len=post_len, resp_go=resp_go)
            ae.Decoder.beam_size = 1
            eps = [t for kind, t in rec['noise'] if kind == 'eps']
            assert len(eps) == 1 and tuple(eps[0].shape) == (1, post_bare.shape[0], shp['Z'])
            out['gen/%s_eps' % tag] = eps[0].numpy()
            out['gen/%s_ids' % tag] = np.asarray([[int(t.item()) for t in srow] for srow in preds], dtype=np.int64)
        meta['gen_steps'] = int(out['gen/greedy_ids'].shape[1])
        meta['beam_size'] = 3

    finish()


def main():
    os.makedirs(GOLD, exist_ok=True)
    only = sys.argv[1:] if len(sys.argv) > 1 else None
    for job in JOBS:
        if only and job[0] not in only:
            continue
        out_path = os.path.join(GOLD, job[0] + '.npz')
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', json.dumps(job), out_path])
        if r.returncode != 0:
            raise SystemExit('golden job failed: %s' % job[0])


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        job = json.loads(sys.argv[2])
        child(*job, sys.argv[3])
    else:
        main()
