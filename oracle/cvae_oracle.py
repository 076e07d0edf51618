"""CPU oracle for the Coupled-VAE hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The shipped path
(``coupled-vae_amd/cvae_amd``) never imports anything under ``oracle/``.

It is a from-scratch restatement, in explicit torch-CPU arithmetic (own GRU/LSTM cell
loops, own log-softmax cross-entropy, own Gaussian log-densities, own clip + Adam), of the
reference's teacher-forced training step and importance-weighted evaluation.  Integer /
layout work (a1, a2) is plain Python / numpy.  Floating-point work is fp32 by default and
can be run in fp64 (``dtype=torch.float64``) to arbitrate between two fp32 results.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference itself in the
build container (CPU) and records inputs, noise, outputs, losses, gradients and post-step
weights into ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
below against those vectors.

Reference citations (CV = Coupled-VAE, CC = Coupled-CVAE under /root/reference):
  process_raw_sent   CV/dataloaders/ptb.py:118-129, CC/dataloaders/switchboard.py:66-76
  sort_resort        CV/utils/utils.py:18-21
  encoder            CV/Modules/encoder.py:41-70
  gaussian           CV/Modules/vae_couple.py:791-803
  to_init            CV/Modules/vae_couple.py:429-434, CV/Modules/decoder.py:45-47,90
  decoder            CV/Modules/vae_couple.py:86-121, CV/Modules/decoder.py:85-113,
                     CC/Modules/decoder.py:86-119
  seq_xent           CV/Modules/Losses/SeqLoss.py:36-44
  kl / couple / loss CV/Experiment.py:730-757,1892-1923 ; CC/Experiment.py:395-416
  clip + Adam        CV/Experiment.py:509-522,589-600
  iw_nll             CV/Modules/vae_couple.py:526-578 ; CC/Modules/vae_couple.py:312-364
  mi                 CV/Modules/vae_couple.py:471-502
  corpus reductions  CV/Experiment.py:1382-1396,1502
"""
import math

import numpy as np
import torch

PAD, GO, EOS, UNK = 0, 1, 2, 3          # CV/utils/vocab.py:29-30
IMQ_SCALES = (.1, .2, .5, 1., 2., 5., 10.)  # CV/Experiment.py:1896
N_SAMPLE_CHUNK = 2                      # CC/Modules/vae_couple.py:20 (_n_sample)


# --------------------------------------------------------------------------- a1 / a2
def process_raw_sent(ids, max_len, pad=PAD, go=GO, eos=EOS):
    """CV/dataloaders/ptb.py:118-129.  ids: list[int] -> (bare[max_len], go[max_len+1],
    eos[max_len+1], len) as int64 numpy."""
    ids = list(ids)
    l = len(ids)
    if l > max_len:
        ids = ids[:max_len]
        l = max_len
    padding = [pad] * (max_len - l)
    return (np.asarray(ids + padding, dtype=np.int64),
            np.asarray([go] + ids + padding, dtype=np.int64),
            np.asarray(ids + [eos] + padding, dtype=np.int64),
            np.int64(l))


def make_batch(sentences, max_len, truncate=True):
    """Batch a1 rows and (train / CVAE eval) truncate to the batch max
    (CV/Experiment.py:659-664).  Returns int64 arrays bare(B,L) go(B,L+1) eos(B,L+1) len(B)."""
    rows = [process_raw_sent(s, max_len) for s in sentences]
    bare = np.stack([r[0] for r in rows])
    go = np.stack([r[1] for r in rows])
    eos = np.stack([r[2] for r in rows])
    ln = np.asarray([r[3] for r in rows], dtype=np.int64)
    if truncate:
        m = int(ln.max())
        bare, go, eos = bare[:, :m].copy(), go[:, :m + 1].copy(), eos[:, :m + 1].copy()
    return bare, go, eos, ln


def sort_resort(length):
    """CV/utils/utils.py:18-21: stable descending order (ties keep original order) and
    its inverse permutation."""
    length = [int(x) for x in length]
    s_idx = [ix for ix, _ in sorted(enumerate(length), key=lambda x: x[1], reverse=True)]
    res_idx = [a for a, _ in sorted(enumerate(s_idx), key=lambda x: x[1])]
    return s_idx, res_idx


# --------------------------------------------------------------------------- cells
def _gru_step(xp, h, W_hh, b_hh):
    H = h.shape[1]
    hp = h @ W_hh.t() + b_hh
    r = torch.sigmoid(xp[:, :H] + hp[:, :H])
    z = torch.sigmoid(xp[:, H:2 * H] + hp[:, H:2 * H])
    n = torch.tanh(xp[:, 2 * H:] + r * hp[:, 2 * H:])
    return (1 - z) * n + z * h


def _lstm_step(xp, h, c, W_hh, b_hh):
    H = h.shape[1]
    g = xp + h @ W_hh.t() + b_hh
    i = torch.sigmoid(g[:, :H])
    f = torch.sigmoid(g[:, H:2 * H])
    gg = torch.tanh(g[:, 2 * H:3 * H])
    o = torch.sigmoid(g[:, 3 * H:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def _cellname(cell):
    return {'gru': 'GRU', 'lstm': 'LSTM'}[cell]


def _rnn_dir(P, pre, suffix, cell, x, length, reverse):
    """One direction of a packed 1-layer RNN: rows are updated only while t < len, which
    is what pack_padded_sequence does (CV/Modules/encoder.py:56-60).  x: (L,B,E).
    Returns the final hidden state (B,H)."""
    W_ih, W_hh = P[pre + 'weight_ih_l0' + suffix], P[pre + 'weight_hh_l0' + suffix]
    b_ih, b_hh = P[pre + 'bias_ih_l0' + suffix], P[pre + 'bias_hh_l0' + suffix]
    L, B, _ = x.shape
    H = W_hh.shape[1]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    xp_all = x @ W_ih.t() + b_ih
    steps = range(L - 1, -1, -1) if reverse else range(L)
    for t in steps:
        live = (length > t).unsqueeze(1)
        if cell == 'gru':
            h2 = _gru_step(xp_all[t], h, W_hh, b_hh)
            h = torch.where(live, h2, h)
        else:
            h2, c2 = _lstm_step(xp_all[t], h, c, W_hh, b_hh)
            h = torch.where(live, h2, h)
            c = torch.where(live, c2, c)
    return h


def encoder(P, pre, cell, bare, length, drop_mask=None):
    """CV/Modules/encoder.py:41-70 -> flattened last states (B, 2H) = [h_fwd | h_bwd]
    (the only output any caller consumes; CV/Modules/vae_couple.py:716).

    drop_mask: (L,B,E) multiplicative mask *in the reference's draw layout*, i.e. in
    length-sorted batch order (the reference sorts, then applies dropout).  None = eval."""
    emb = P[pre + 'Embedding.weight']
    x = emb[bare].transpose(0, 1)                      # (L,B,E)
    if drop_mask is not None:
        _, res_idx = sort_resort(length.tolist())
        x = x * drop_mask[:, res_idx, :]
    rp = pre + _cellname(cell) + '.'
    hf = _rnn_dir(P, rp, '', cell, x, length, reverse=False)
    hb = _rnn_dir(P, rp, '_reverse', cell, x, length, reverse=True)
    return torch.cat([hf, hb], dim=1)


def gaussian(P, pre, x, eps=None):
    """CV/Modules/vae_couple.py:791-803.  eps: (B,Z) or None (-> z = None)."""
    mu = x @ P[pre + 'toMiu.0.weight'].t() + P[pre + 'toMiu.0.bias']
    sigma = torch.exp(x @ P[pre + 'toSigma.0.weight'].t() + P[pre + 'toSigma.0.bias'])
    z = None if eps is None else mu + sigma * eps
    return mu, sigma, z


def to_init(P, pre, z):
    """Linear(Z,E)-ReLU-Linear(E,E), CV/Modules/vae_couple.py:429-434."""
    h = torch.relu(z @ P[pre + '0.weight'].t() + P[pre + '0.bias'])
    return h @ P[pre + '2.weight'].t() + P[pre + '2.bias']


def decoder_final(P, pre, cell, init, go, mask_in=None, mask_out=None):
    """Teacher-forcing branch, CV/Modules/vae_couple.py:96-115, up to `final`
    = tanh(W_c drop(h_t)) (T,B,E).  init: (B,E); go: (B,T) int64."""
    rp = pre + _cellname(cell) + '.'
    W_ih, W_hh = P[rp + 'weight_ih_l0'], P[rp + 'weight_hh_l0']
    b_ih, b_hh = P[rp + 'bias_ih_l0'], P[rp + 'bias_hh_l0']
    B, T = go.shape
    Hd = W_hh.shape[1]
    x0 = torch.cat([init, init], dim=1)
    h = init.new_zeros(B, Hd)
    c = init.new_zeros(B, Hd)
    if cell == 'gru':
        h = _gru_step(x0 @ W_ih.t() + b_ih, h, W_hh, b_hh)
    else:
        h, c = _lstm_step(x0 @ W_ih.t() + b_ih, h, c, W_hh, b_hh)
    emb = P[pre + 'Embedding.weight'][go].transpose(0, 1)           # (T,B,E)
    x = torch.cat([emb, init.unsqueeze(0).expand(T, -1, -1)], dim=2)
    if mask_in is not None:
        x = x * mask_in
    xp = x @ W_ih.t() + b_ih
    outs = []
    for t in range(T):
        if cell == 'gru':
            h = _gru_step(xp[t], h, W_hh, b_hh)
        else:
            h, c = _lstm_step(xp[t], h, c, W_hh, b_hh)
        outs.append(h)
    outs = torch.stack(outs)                                       # (T,B,Hd)
    if mask_out is not None:
        outs = outs * mask_out
    return torch.tanh(outs @ P[pre + 'W_c.weight'].t() + P[pre + 'W_c.bias'])


def lm_decoder_final(P, pre, go, mask_in=None, mask_out=None):
    """The LM's decoder (CV/Modules/lm.py:78-100), GRU flavour: a GRU over the word embeddings alone,
    h_0 = 0, then tanh(W_c drop(h_t)).  go (B,T) -> final (T,B,E); mask_in (T,B,E)."""
    rp = pre + 'GRU.'
    W_ih, W_hh = P[rp + 'weight_ih_l0'], P[rp + 'weight_hh_l0']
    b_ih, b_hh = P[rp + 'bias_ih_l0'], P[rp + 'bias_hh_l0']
    B, T = go.shape
    x = P[pre + 'Embedding.weight'][go].transpose(0, 1)            # (T,B,E)
    if mask_in is not None:
        x = x * mask_in
    xp = x @ W_ih.t() + b_ih
    h = x.new_zeros(B, W_hh.shape[1])
    outs = []
    for t in range(T):
        h = _gru_step(xp[t], h, W_hh, b_hh)
        outs.append(h)
    outs = torch.stack(outs)
    if mask_out is not None:
        outs = outs * mask_out
    return torch.tanh(outs @ P[pre + 'W_c.weight'].t() + P[pre + 'W_c.bias'])


def seq_xent(P, pre, final, eos, pad=PAD):
    """ToVocab + SeqLoss(keep_batch=True): CV/Modules/vae_couple.py:119,
    CV/Modules/Losses/SeqLoss.py:36-44.  final (T,B,E), eos (B,T) -> (B,)."""
    logits = final @ P[pre + 'ToVocab.weight'].t() + P[pre + 'ToVocab.bias']   # (T,B,V)
    m = logits.max(dim=2, keepdim=True).values
    lse = m.squeeze(2) + torch.log(torch.exp(logits - m).sum(2))
    tgt = eos.t()                                                  # (T,B)
    picked = logits.gather(2, tgt.unsqueeze(2)).squeeze(2)
    w = (tgt != pad).to(final.dtype)
    return ((lse - picked) * w).sum(0)


def decoder_logits(P, pre, final):
    """(B,T,V) logits, as the reference returns them."""
    return (final @ P[pre + 'ToVocab.weight'].t() + P[pre + 'ToVocab.bias']).transpose(0, 1)


# --------------------------------------------------------------------------- losses
def kl_std(mu, sigma):
    """kl_divergence(N(mu,sigma), N(0,1)).sum(1)  (CV/Experiment.py:733-735)."""
    var = sigma * sigma
    return (0.5 * (var + mu * mu - 1.0 - torch.log(var))).sum(1)


def kl_pair(mu_q, sig_q, mu_p, sig_p):
    """kl_divergence(N(mu_q,sig_q), N(mu_p,sig_p)).sum(1)  (CC/Experiment.py:398-399)."""
    var_ratio = (sig_q / sig_p) ** 2
    t1 = ((mu_q - mu_p) / sig_p) ** 2
    return (0.5 * (var_ratio + t1 - 1.0 - torch.log(var_ratio))).sum(1)


def couple_loss(init, init_couple, latent_dim, euclidean_dist):
    """CV/Experiment.py:742-751,1892-1923 -> (B,) before the batch mean."""
    d = ((init - init_couple.detach()) ** 2).sum(-1)
    if euclidean_dist:
        return d
    cbase = 2.0 * latent_dim
    imq = 0.0
    for s in IMQ_SCALES:
        C = cbase * s
        imq = imq + C / (C + d)
    return -1 * imq


def lambda_kl(iter_num, annealing_start, annealing_step, max_val=1.0):
    """CV/Experiment.py:556."""
    return float(min(max(np.round((iter_num - annealing_start) / annealing_step, decimals=6), 0),
                     max_val))


def lambda_kl_model(iter_num, model, annealing_start, annealing_step, beta=1.0, M=4, R=0.5):
    """Experiment.update_lambda_kl (CV/Experiment.py:524-558) for the variants that share the VAE
    architecture: the linear ramp capped at 1 (vae*), at config.beta (beta-vae*), or the cyclic
    schedule of cyc-anneal-vae* (M cycles inside the annealing window, ramp over the first R of each)."""
    if model.startswith('cyc-anneal-vae'):
        if iter_num < annealing_start:
            return 0.0
        if iter_num > annealing_step + annealing_start:
            return 1.0
        tau = ((iter_num - annealing_start) % int(annealing_step / M)) / (annealing_step / M)
        return 1.0 if tau > R else tau / R
    return lambda_kl(iter_num, annealing_start, annealing_step, beta if model.startswith('beta-vae') else 1.0)


# Variants that reuse the VAE / VAE_COUPLE modules unchanged (CV/Experiment.py:147-173, 325-351):
# only the lambda_KL schedule (beta-vae*, cyc-anneal-vae*) or the KL term (surprising-fix*:
# free bits, Experiment.py:1141-1142) differs.
ARCH = {'beta-vae': 'vae', 'beta-vae-couple': 'vae-couple', 'cyc-anneal-vae': 'vae',
        'cyc-anneal-vae-couple': 'vae-couple', 'surprising-fix': 'vae', 'surprising-fix-couple': 'vae-couple'}


def arch_of(model):
    return ARCH.get(model, model)


def is_couple(model):
    return model.endswith('couple')


def kl_std_elem(mu, sigma):
    """KL(N(mu,sigma) || N(0,1)) per latent dimension (torch.distributions.kl_divergence)."""
    return 0.5 * (sigma * sigma + mu * mu - 1.0) - torch.log(sigma)


def is_dialog(model):
    return model.startswith('cvae')


def train_forward(P, model, cell, batch, noise, cfg):
    """One teacher-forced training forward + loss assembly.

    model: 'vae' | 'vae-couple' | 'cvae' | 'cvae-couple' | 'dae', or a variant listed in ARCH.
    batch (text):   dict bare, go, eos, len            (int64 torch tensors)
    batch (dialog): dict post_bare, post_len, resp_bare, resp_go, resp_eos, resp_len
    noise: dict of masks / eps in the reference's draw layout (see make_golden.NOISE_ORDER);
           missing dropout masks mean dropout 0.
    cfg: dict latent_dim, lambda_kl, lambda_couple, lambda_seq_couple, euclidean_dist; optional
         lambda_fb (free bits of surprising-fix*: each latent dimension's KL is floored at lambda_fb/Z).
    Returns dict with per-term losses, 'tot', and intermediate tensors."""
    out = {}
    g = noise.get
    free_bits = cfg.get('lambda_fb') if model.startswith('surprising-fix') else None
    model = arch_of(model)
    if model == 'lm':
        # LM.forward (CV/Modules/lm.py:237-256): the decoder alone, reconstruction loss only
        assert cell == 'gru', 'Modules/lm.py only runs with GRU cells'
        final = lm_decoder_final(P, 'Decoder.', batch['go'], g('dec_in_drop'), g('dec_out_drop'))
        seq = seq_xent(P, 'Decoder.', final, batch['eos'])
        out.update(final=final, seq_per_sent=seq, seq=seq.mean(0), tot=seq.mean(0))
        return out
    if model == 'dae':
        # DAE.forward (CV/Modules/dae.py): deterministic latent = toLatent(last), no KL term
        go, eos = batch['go'], batch['eos']
        last = encoder(P, 'Encoder.', cell, batch['bare'], batch['len'], g('enc_drop'))
        latent = last @ P['toLatent.weight'].t() + P['toLatent.bias']
        init = to_init(P, 'Decoder.toInit.', latent)
        final = decoder_final(P, 'Decoder.', cell, init, go, g('dec_in_drop'), g('dec_out_drop'))
        seq = seq_xent(P, 'Decoder.', final, eos)
        out.update(latent=latent, init=init, final=final, seq_per_sent=seq, seq=seq.mean(0), tot=seq.mean(0))
        return out
    if not is_dialog(model):
        go, eos = batch['go'], batch['eos']
        last = encoder(P, 'Encoder.', cell, batch['bare'], batch['len'], g('enc_drop'))
        mu, sigma, z = gaussian(P, 'PosteriorGaussian.', last, noise['eps'])
        if free_bits is None:
            kl = kl_std(mu, sigma)
        else:
            kl = torch.clamp(kl_std_elem(mu, sigma), min=free_bits / cfg['latent_dim']).sum(1)
        post_repr = 0.0
        couple_in = last
    else:
        go, eos = batch['resp_go'], batch['resp_eos']
        post = encoder(P, 'PostEncoder.', cell, batch['post_bare'], batch['post_len'], g('post_enc_drop'))
        post_repr = post @ P['PostRepr.weight'].t() + P['PostRepr.bias']
        resp = encoder(P, 'RespEncoder.', cell, batch['resp_bare'], batch['resp_len'], g('resp_enc_drop'))
        mu_p, sig_p, _ = gaussian(P, 'PriorGaussian.', post)
        couple_in = torch.cat([resp, post], dim=1)
        mu, sigma, z = gaussian(P, 'PosteriorGaussian.', couple_in, noise['eps'])
        kl = kl_pair(mu, sigma, mu_p, sig_p)
        out['mu_prior'], out['sigma_prior'] = mu_p, sig_p
    init_pre = 'toInit.' if is_couple(model) else 'Decoder.toInit.'
    init = to_init(P, init_pre, z) + post_repr
    final = decoder_final(P, 'Decoder.', cell, init, go, g('dec_in_drop'), g('dec_out_drop'))
    seq = seq_xent(P, 'Decoder.', final, eos)
    out.update(mu=mu, sigma=sigma, z=z, init=init, final=final, seq_per_sent=seq, kl_per_sent=kl)
    out['seq'] = seq.mean(0)
    out['kl'] = kl.mean(0)
    tot = out['seq'] + cfg['lambda_kl'] * out['kl']
    if is_couple(model):
        mu_c, _, _ = gaussian(P, 'PosteriorGaussianCouple.', couple_in)
        init_c = to_init(P, 'toInitCouple.', mu_c) + post_repr
        final_c = decoder_final(P, 'DecoderCouple.', cell, init_c, go,
                                g('decc_in_drop'), g('decc_out_drop'))
        seq_c = seq_xent(P, 'DecoderCouple.', final_c, eos)
        cpl = couple_loss(init, init_c, cfg['latent_dim'], cfg['euclidean_dist'])
        out.update(init_couple=init_c, final_couple=final_c, seq_c_per_sent=seq_c,
                   couple_per_sent=cpl)
        out['seq_c'] = seq_c.mean(0)
        out['couple'] = cpl.mean(0)
        tot = tot + out['seq_c'] * cfg['lambda_seq_couple'] + cfg['lambda_couple'] * out['couple']
    if cfg.get('lambda_bow', 0.0) > 0 and not is_dialog(model):
        # bag-of-words loss (CV/Experiment.py:703-706 / 737-740): BoW(z) scored against every token of the sentence,
        # per sentence divided by its length; added to the total only while lambda_KL > 1e-5 (:713-715)
        bow = bow_xent(P, z, batch['bare'], batch['len'])
        out['bow_per_sent'] = bow
        out['bow'] = bow.mean(0)
        if cfg['lambda_kl'] > 1e-5:
            tot = tot + cfg['lambda_bow'] * out['bow']
    out['tot'] = tot
    return out


def bow_xent(P, z, bare, length, pad=0):
    """criterionSeq(BoW_logits.unsqueeze(1).expand(-1, L, -1), bare, keep_batch=True) / sent_len
    (CV/Experiment.py:703-706; BoW = nn.Linear(latent_dim, V), CV/Modules/vae_couple.py:425, 724): for every
    sentence the sum over its non-pad tokens of -log softmax(BoW(z))[token], divided by its length."""
    logits = z @ P['BoW.weight'].t() + P['BoW.bias']              # (B, V)
    logp = torch.log_softmax(logits, dim=1)
    tok = logp.gather(1, bare)                                    # (B, L)
    keep = (bare != pad).to(logp.dtype)
    return -(tok * keep).sum(1) / length.to(logp.dtype)


# --------------------------------------------------------------------------- a12
def alias_groups(state_keys):
    """Keys that name the shared embedding (CV/Experiment.py:93-94,589-592)."""
    return [k for k in state_keys if k.endswith('Embedding.weight')]


def clip_and_adam(P, grads, state, keys, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=30.0):
    """clip_grad_norm_(30) then Adam over the *alias list* `keys` (state_dict order,
    the same tensor possibly listed several times; keys whose grad is None are skipped).
    P / grads: dict key -> tensor, aliases share the very same tensor object.
    state: dict id(tensor) -> {'step','m','v'}; updated in place.  Follows torch's
    foreach clip (each alias counted, and scaled, once per listing) and the
    single-tensor CPU Adam (one full step per listing).  Returns the pre-clip norm."""
    listed = [k for k in keys if grads.get(k) is not None]
    norms = torch.stack([torch.linalg.vector_norm(grads[k]) for k in listed])
    total = torch.linalg.vector_norm(norms)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for k in listed:                      # an aliased grad is scaled once per listing
        grads[k].mul_(coef)
    b1, b2 = betas
    for k in listed:                      # ... and stepped once per listing
        p, gr = P[k], grads[k]
        st = state.setdefault(id(p), {'step': 0, 'm': torch.zeros_like(p), 'v': torch.zeros_like(p)})
        st['step'] += 1
        st['m'].lerp_(gr, 1 - b1)
        st['v'].mul_(b2).addcmul_(gr, gr, value=1 - b2)
        bc1 = 1 - b1 ** st['step']
        bc2 = 1 - b2 ** st['step']
        denom = (st['v'].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st['m'], denom, value=-(lr / bc1))
    return float(total)


def tie_aliases(P):
    """Make every '*Embedding.weight' entry the same tensor object (the reference shares
    one nn.Embedding).  Returns P."""
    keys = alias_groups(P.keys())
    for k in keys[1:]:
        P[k] = P[keys[0]]
    return P


def train_step(P, model, cell, batch, noise, cfg, opt_state, keys):
    """forward + backward + clip + Adam, in place on P (leaf tensors).  Returns
    (losses dict of floats, grads dict (pre-clip clones), pre-clip norm)."""
    uniq = {}
    for k in keys:
        uniq.setdefault(id(P[k]), P[k])
    for p in uniq.values():
        p.requires_grad_(True)
        p.grad = None
    out = train_forward(P, model, cell, batch, noise, cfg)
    out['tot'].backward()
    grads = {k: P[k].grad for k in keys}
    pre = {k: (None if v is None else v.detach().clone()) for k, v in grads.items()}
    with torch.no_grad():
        norm = clip_and_adam(P, grads, opt_state, keys, lr=cfg.get('lr', 1e-3),
                             max_norm=cfg.get('clip_norm', 30.0))
    losses = {k: float(out[k].detach()) for k in ('seq', 'kl', 'seq_c', 'couple', 'tot') if k in out}
    return losses, pre, norm


# --------------------------------------------------------------------------- a13-a16
def log_sum_exp(value, dim):
    """CV/utils/utils.py:69-82."""
    m = value.max(dim=dim, keepdim=True).values
    return m.squeeze(dim) + torch.log(torch.exp(value - m).sum(dim))


def normal_log_prob(x, mu, sigma):
    return -((x - mu) ** 2) / (2 * sigma * sigma) - torch.log(sigma) - math.log(math.sqrt(2 * math.pi))


def _iw_setup(P, model, cell, batch):
    if not is_dialog(model):
        go, eos = batch['go'], batch['eos']
        last = encoder(P, 'Encoder.', cell, batch['bare'], batch['len'])
        mu, sigma, _ = gaussian(P, 'PosteriorGaussian.', last)
        mu_p, sig_p = torch.zeros_like(mu), torch.ones_like(sigma)
        post_repr = 0.0
    else:
        go, eos = batch['resp_go'], batch['resp_eos']
        post = encoder(P, 'PostEncoder.', cell, batch['post_bare'], batch['post_len'])
        post_repr = post @ P['PostRepr.weight'].t() + P['PostRepr.bias']
        resp = encoder(P, 'RespEncoder.', cell, batch['resp_bare'], batch['resp_len'])
        mu_p, sig_p, _ = gaussian(P, 'PriorGaussian.', post)
        mu, sigma, _ = gaussian(P, 'PosteriorGaussian.', torch.cat([resp, post], dim=1))
    init_pre = 'toInit.' if is_couple(model) else 'Decoder.toInit.'
    return go, eos, mu, sigma, mu_p, sig_p, post_repr, init_pre


def iw_xent_rows(P, model, cell, batch, eps, lo=0, hi=None):
    """Per-sample cross-entropy rows (hi-lo, B) of samples [lo, hi) of eps (K,B,Z): the decoder
    half of importance_sampling (CV/Modules/vae_couple.py:540-557), in the reference's chunks of 2."""
    go, eos, mu, sigma, _, _, post_repr, init_pre = _iw_setup(P, model, cell, batch)
    hi = eps.shape[0] if hi is None else hi
    xents = []
    for k0 in range(lo, hi, N_SAMPLE_CHUNK):
        z = mu + sigma * eps[k0:min(hi, k0 + N_SAMPLE_CHUNK)]             # (2,B,Z)
        n, B, Z = z.shape
        init = (to_init(P, init_pre, z) + post_repr).reshape(n * B, -1)
        final = decoder_final(P, 'Decoder.', cell, init, go.repeat(n, 1))
        xents.append(seq_xent(P, 'Decoder.', final, eos.repeat(n, 1)).view(n, B))
    return torch.cat(xents)


def iw_from_rows(P, model, cell, batch, eps, xent):
    """The estimator half (CV/Modules/vae_couple.py:558-578) from the full (K,B) table."""
    _, _, mu, sigma, mu_p, sig_p, _, _ = _iw_setup(P, model, cell, batch)
    K = eps.shape[0]
    z = mu + sigma * eps
    lpz = normal_log_prob(z, mu_p, sig_p).sum(2)
    lqz = normal_log_prob(z, mu, sigma).sum(2)
    nll = -(log_sum_exp(lpz - xent - lqz, 0) - math.log(K))
    kl = (lqz - lpz).mean(0)
    return xent.mean(0), nll, kl, z.transpose(0, 1)


def _decode_fns(P, prefix, cell, mask_id):
    """(step, logits_of) closures shared by the greedy and beam restatements."""
    cn = 'GRU' if cell == 'gru' else 'LSTM'
    W_ih, W_hh = P[prefix + cn + '.weight_ih_l0'], P[prefix + cn + '.weight_hh_l0']
    b_ih, b_hh = P[prefix + cn + '.bias_ih_l0'], P[prefix + cn + '.bias_hh_l0']

    def step(x, h, c):
        gi, gh = x @ W_ih.t() + b_ih, h @ W_hh.t() + b_hh
        if cell == 'gru':
            ir, iz, inn = gi.chunk(3, 1)
            hr, hz, hn = gh.chunk(3, 1)
            r, z = torch.sigmoid(ir + hr), torch.sigmoid(iz + hz)
            n = torch.tanh(inn + r * hn)
            return (1 - z) * n + z * h, c
        i, f, g, o = (gi + gh).chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c), c

    def logits_of(h):
        final = torch.tanh(h @ P[prefix + 'W_c.weight'].t() + P[prefix + 'W_c.bias'])
        logits = final @ P[prefix + 'ToVocab.weight'].t() + P[prefix + 'ToVocab.bias']
        if mask_id is not None and mask_id >= 0:
            logits[:, mask_id] = -float('inf')
        return logits

    return step, logits_of, W_hh.shape[1]


def decoder_beam(P, prefix, cell, init, first, T, K, mask_id=3):
    """Beam branch of Decoder.forward (CV/Modules/decoder.py:114-168; vae_couple.py:122-172), eval mode.
    State per sentence b: up to K beams (h, c, tokens so far, cost).  Each step every beam takes one cell
    step on [emb(last token), init], scores log_softmax(masked logits), proposes its K best tokens
    (decoder.py:142-143); the sentence keeps the K cheapest proposals by accumulated cost, ties in proposal
    order (beam index, then rank) -- Python's stable `sorted` (decoder.py:161).  T steps, no early stop.
    The reference cannot run this branch with the LSTM; (h, c) are carried alike here.
    init (B,E), first (B,) -> (ids (B,T) int64 of beam 0, its cost (B,))."""
    step, logits_of, H = _decode_fns(P, prefix, cell, mask_id)
    emb = P[prefix + 'Embedding.weight']
    B = init.shape[0]
    z = torch.zeros(B, H, dtype=init.dtype)
    h0, c0 = step(torch.cat([init, init], 1), z, z.clone())
    # beams[k] = (h (B,H), c (B,H), last token (B,), history: list of (B,) tensors, cost (B,))
    beams = [(h0, c0, first, [], torch.zeros(B, dtype=init.dtype))]
    for _ in range(T):
        props = []                           # per source beam: (h, c, top values (B,K), top ids (B,K), cost (B,))
        for h, c, last, hist, cost in beams:
            h, c = step(torch.cat([emb[last], init], 1), h, c)
            lv, li = torch.topk(torch.log_softmax(logits_of(h), dim=1), K, dim=1)
            props.append((h, c, lv, li, cost, hist))
        new = [(torch.empty_like(h0), torch.empty_like(c0), torch.empty(B, dtype=torch.int64),
                [torch.empty(B, dtype=torch.int64) for _ in range(len(beams[0][3]) + 1)],
                torch.empty(B, dtype=init.dtype)) for _ in range(K)]
        for b in range(B):
            cand = [(float(pr[4][b] - pr[2][b, l]), k, l) for k, pr in enumerate(props) for l in range(K)]
            order = sorted(range(len(cand)), key=lambda i: cand[i][0])          # stable
            for j in range(K):
                _, k, l = cand[order[j]]
                pr = props[k]
                new[j][0][b], new[j][1][b] = pr[0][b], pr[1][b]
                new[j][2][b] = pr[3][b, l]
                for tt, tokens in enumerate(pr[5]):
                    new[j][3][tt][b] = tokens[b]
                new[j][3][-1][b] = pr[3][b, l]
                new[j][4][b] = pr[4][b] - pr[2][b, l]
        beams = new
    return torch.stack(beams[0][3], 1), beams[0][4]


def dialog_sample_from_prior(P, model, cell, post_bare, post_len, resp_go, eps, T, K=1):
    """sample_from_prior of the dialogue models (CC/Modules/vae_couple.py:369-397; vae.py:157-184), eval
    mode: post encoder -> PostRepr; z = mu_p + sigma_p * eps from the prior network; init = toInit(z) +
    post_repr; greedy (K = 1) or beam decode of T steps from resp_go[:, 0].  eps (B,Z) -> ids (B,T)."""
    post = encoder(P, 'PostEncoder.', cell, post_bare, post_len)
    post_repr = post @ P['PostRepr.weight'].t() + P['PostRepr.bias']
    mu_p, sig_p, _ = gaussian(P, 'PriorGaussian.', post)
    init_pre = 'toInit.' if is_couple(model) else 'Decoder.toInit.'
    init = to_init(P, init_pre, mu_p + sig_p * eps) + post_repr
    if K > 1:
        return decoder_beam(P, 'Decoder.', cell, init, resp_go[:, 0], T, K)[0]
    return decoder_greedy(P, 'Decoder.', cell, init, resp_go[:, 0], T)


def decoder_greedy(P, prefix, cell, init, first, T, mask_id=3):
    """Greedy branch of Decoder.forward (CV/Modules/decoder.py:176-201; vae_couple.py:173-201), eval mode:
    h_0 from the prefix step on [init, init]; then T steps of [emb(y_{t-1}), init] -> cell -> tanh(W_c h)
    -> ToVocab -> logit[mask_id] = -inf -> argmax.  init (B,E), first (B,) -> ids (B,T) int64."""
    cn = 'GRU' if cell == 'gru' else 'LSTM'
    W_ih, W_hh = P[prefix + cn + '.weight_ih_l0'], P[prefix + cn + '.weight_hh_l0']
    b_ih, b_hh = P[prefix + cn + '.bias_ih_l0'], P[prefix + cn + '.bias_hh_l0']
    emb = P[prefix + 'Embedding.weight']
    B = init.shape[0]
    H = W_hh.shape[1]
    h = torch.zeros(B, H, dtype=init.dtype)
    c = torch.zeros(B, H, dtype=init.dtype)

    def step(x, h, c):
        gi, gh = x @ W_ih.t() + b_ih, h @ W_hh.t() + b_hh
        if cell == 'gru':
            ir, iz, inn = gi.chunk(3, 1)
            hr, hz, hn = gh.chunk(3, 1)
            r, z = torch.sigmoid(ir + hr), torch.sigmoid(iz + hz)
            n = torch.tanh(inn + r * hn)
            return (1 - z) * n + z * h, c
        i, f, g, o = (gi + gh).chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c), c

    h, c = step(torch.cat([init, init], 1), h, c)
    cur, out = first, []
    for _ in range(T):
        h, c = step(torch.cat([emb[cur], init], 1), h, c)
        final = torch.tanh(h @ P[prefix + 'W_c.weight'].t() + P[prefix + 'W_c.bias'])
        logits = final @ P[prefix + 'ToVocab.weight'].t() + P[prefix + 'ToVocab.bias']
        if mask_id is not None and mask_id >= 0:
            logits[:, mask_id] = -float('inf')
        cur = torch.argmax(logits, dim=1)
        out.append(cur)
    return torch.stack(out, 1)


def lm_test_lm(P, batch):
    """LM.test_lm (CV/Modules/lm.py:218-235): xent (B,) of one pass; the reference returns kl = 0, nll = xent."""
    return seq_xent(P, 'Decoder.', lm_decoder_final(P, 'Decoder.', batch['go']), batch['eos'])


def dae_test_lm(P, cell, batch):
    """DAE.test_lm (CV/Modules/dae.py): one deterministic teacher-forced decode -> xent (B,), latent (B,Z);
    the reference returns nll = kl = +inf for this model."""
    last = encoder(P, 'Encoder.', cell, batch['bare'], batch['len'])
    latent = last @ P['toLatent.weight'].t() + P['toLatent.bias']
    final = decoder_final(P, 'Decoder.', cell, to_init(P, 'Decoder.toInit.', latent), batch['go'])
    return seq_xent(P, 'Decoder.', final, batch['eos']), latent


def iw_nll(P, model, cell, batch, eps):
    """test_lm + importance_sampling (CV/Modules/vae_couple.py:504-578;
    CC/Modules/vae_couple.py:274-364).  eps: (K,B,Z) = the K/2 chunk draws concatenated.
    Returns xent, nll, kl (B,) and z (B,K,Z)."""
    return iw_from_rows(P, model, cell, batch, eps, iw_xent_rows(P, model, cell, batch, eps))


def estimate_mi(P, cell, batch, eps):
    """estimate_mi + importance_sampling_mi (CV/Modules/vae_couple.py:453-502).
    eps (K,B,Z) -> mi (B,)."""
    last = encoder(P, 'Encoder.', cell, batch['bare'], batch['len'])
    mu, sigma, _ = gaussian(P, 'PosteriorGaussian.', last)
    B = mu.shape[0]
    z = mu + sigma * eps                                              # (K,B,Z)
    lqzx = normal_log_prob(z, mu, sigma).sum(2)                        # (K,B)
    pair = normal_log_prob(z.unsqueeze(2), mu.unsqueeze(0).unsqueeze(0),
                           sigma.unsqueeze(0).unsqueeze(0)).sum(3)     # (K,B,B)
    pair = pair.masked_fill(torch.eye(B, dtype=torch.bool).unsqueeze(0), -float('inf'))
    lqz = log_sum_exp(pair, 2) - math.log(B - 1)
    return lqzx.mean(0) - lqz.mean(0)


def corpus_metrics(nll, xent, kl, lens):
    """CV/Experiment.py:1382-1396."""
    return dict(nll=float(nll.mean(0)), xent=float(xent.mean(0)), kl=float(kl.mean(0)),
                ppl=float(torch.exp(nll.sum(0) / (lens + 1).to(nll.dtype).sum(0))))


# --------------------------------------------------------------------------- helpers
def init_params(model, cell, V, E, H, Z, seed=0, dtype=torch.float32):
    """Random parameters with the reference's state_dict keys / shapes.
    Values: embedding N(0,1) (CV/utils/vocab.py:22-23), everything else U(-k,k)."""
    gen = torch.Generator().manual_seed(seed)
    G = 3 if cell == 'gru' else 4
    cn = _cellname(cell)
    Hd = 2 * H
    P = {}

    def u(*shape, k):
        return ((torch.rand(*shape, generator=gen, dtype=torch.float64) * 2 - 1) * k).to(dtype)

    def rnn(pre, inp, hid, bi):
        k = 1.0 / math.sqrt(hid)
        for suf in ([''] + (['_reverse'] if bi else [])):
            P[pre + 'weight_ih_l0' + suf] = u(G * hid, inp, k=k)
            P[pre + 'weight_hh_l0' + suf] = u(G * hid, hid, k=k)
            P[pre + 'bias_ih_l0' + suf] = u(G * hid, k=k)
            P[pre + 'bias_hh_l0' + suf] = u(G * hid, k=k)

    def lin(pre, inp, outp):
        k = 1.0 / math.sqrt(inp)
        P[pre + 'weight'] = u(outp, inp, k=k)
        P[pre + 'bias'] = u(outp, k=k)

    def gauss(pre, inp):
        lin(pre + 'toMiu.0.', inp, Z)
        lin(pre + 'toSigma.0.', inp, Z)

    def dec(pre, with_init):
        P[pre + 'Embedding.weight'] = emb
        if with_init:
            lin(pre + 'toInit.0.', Z, E)
            lin(pre + 'toInit.2.', E, E)
        rnn(pre + cn + '.', 2 * E, Hd, False)
        lin(pre + 'W_c.', Hd, E)
        lin(pre + 'ToVocab.', E, V)

    emb = torch.randn(V, E, generator=gen, dtype=torch.float64).to(dtype)
    encs = ['PostEncoder.', 'RespEncoder.'] if is_dialog(model) else ['Encoder.']
    for e in encs:
        P[e + 'Embedding.weight'] = emb
        rnn(e + cn + '.', E, H, True)
    if is_dialog(model):
        gauss('PriorGaussian.', Hd)
        gauss('PosteriorGaussian.', 2 * Hd)
        if is_couple(model):
            gauss('PosteriorGaussianCouple.', 2 * Hd)
    else:
        gauss('PosteriorGaussian.', Hd)
        if is_couple(model):
            gauss('PosteriorGaussianCouple.', Hd)
    dec('Decoder.', not is_couple(model))
    if is_couple(model):
        dec('DecoderCouple.', False)
    lin('BoW.', Z, V)
    if is_dialog(model):
        lin('PostRepr.', Hd, E)
    if is_couple(model):
        for pre in ('toInit.', 'toInitCouple.'):
            lin(pre + '0.', Z, E)
            lin(pre + '2.', E, E)
    return P


def synthetic_sentences(n, V, mu, sigma, cap, seed=0):
    """SURVEY 8d synthetic corpora: ids uniform in [4,V), len = clip(floor(lognormal), 1, cap)."""
    rs = np.random.RandomState(seed)
    lens = np.clip(np.floor(rs.lognormal(mu, sigma, size=n)), 1, cap).astype(np.int64)
    return [rs.randint(4, V, size=int(l)).tolist() for l in lens]
