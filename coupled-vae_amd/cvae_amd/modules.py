"""Host-side mirror of the reference's `Modules` package for the hot path.

Same class names, constructor signatures, method signatures, return tuples and state_dict
keys as CV/Modules/{encoder,decoder,vae,vae_couple}.py, CC/Modules/{vae,vae_couple}.py and
CV/Modules/Losses/SeqLoss.py, so that the reference's Experiment.py can drive them
(INTEGRATION.md).  nn.GRU / nn.LSTM / nn.Linear / nn.Embedding objects are used ONLY as
parameter containers (identical names, shapes and initialisation order => identical
state_dict and, for equal seeds, identical initial weights); their forward() is never called.
All arithmetic goes through libcvae_hip (ops.py).

The greedy and beam branches of Decoder.forward are the step-wise entry points cvae_decoder_greedy / _beam.
"""
import weakref

import torch
import torch.nn as nn

from . import dist_utils, ops
from .settings import settings


# --------------------------------------------------------------------------- noise plumbing
class NoiseSource(object):
    """Where dropout masks and Gaussian noise come from.  The default draws them with torch's
    device generator in exactly the order and shapes the reference does (SURVEY 8a 'RNG draw
    order'); tests replace it with a replay of recorded tensors."""

    _ones = {}                                   # device -> flat buffer of ones (input of the fused dropout kernel)

    def dropout_mask(self, shape, p, device, name):
        """Mask with values {0, 1/(1-p)}: one fused dropout kernel over a cached buffer of ones (what
        nn.Dropout itself launches in the reference), instead of fill + bernoulli + scale."""
        if p == 0.0:
            return None
        n = 1
        for d in shape:
            n *= int(d)
        buf = self._ones.get(device)
        if buf is None or buf.numel() < n:
            buf = torch.ones(max(n, 1 << 22), dtype=torch.float32, device=device)
            NoiseSource._ones[device] = buf
        return torch.nn.functional.dropout(buf[:n].view(shape), p, True)

    def normal(self, shape, device, name):
        return torch.empty(shape, dtype=torch.float32, device=device).normal_()

    def normal_chunks(self, n_chunks, shape, device, name):
        """n_chunks consecutive draws of `shape`, concatenated along dim 0 (the reference draws the importance
        samples' eps in K/2 chunks of (2,B,Z)).  Here: ONE launch for the same number of iid normals (K/2 launches of
        3 us each were 1.5 % of an evaluation batch); a replay hands back the recorded chunks one by one."""
        return torch.empty((n_chunks * shape[0],) + tuple(shape[1:]), dtype=torch.float32, device=device).normal_()


class ReplayNoise(NoiseSource):
    """Replays named tensors (golden fixtures): dict name -> tensor or list of tensors.  The recorded tensors are
    in the reference's layout (column b = batch row b of the caller); while a RowOrder is active the model
    processes the batch's rows in another order, so the replayed columns follow it.  (The default source draws
    i.i.d. values: which row gets which column is immaterial there, so it never permutes.)  The encoders' masks are
    exempt: the reference draws them in ITS length-sorted layout, which cvae_encoder_forward's mask_rows handles."""

    def __init__(self, table):
        self.table = {k: (list(v) if isinstance(v, (list, tuple)) else [v]) for k, v in table.items()}

    def _pop(self, name, shape, device):
        if name not in self.table or not self.table[name]:
            return None
        t = self.table[name].pop(0).to(device=device, dtype=torch.float32).contiguous()
        if t.dim() == len(shape) - 1 and shape[0] == 1:      # a (1,B,Z) draw recorded as (B,Z)
            t = t.unsqueeze(0)
        assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), tuple(shape))
        if _row_order is not None and not name.endswith('enc_drop'):
            t = t.index_select(1, _row_order.s_idx)          # every per-row draw has its rows in dim 1
        return t

    def dropout_mask(self, shape, p, device, name):
        return self._pop(name, shape, device)

    def normal(self, shape, device, name):
        t = self._pop(name, shape, device)
        if t is None:   # draws whose value is discarded by the model (e.g. eps_couple)
            return torch.zeros(shape, dtype=torch.float32, device=device)
        return t

    def normal_chunks(self, n_chunks, shape, device, name):
        return torch.cat([self.normal(shape, device, name) for _ in range(n_chunks)], 0)


_noise = NoiseSource()
_row_order = None        # the RowOrder of the forward pass in progress (None: rows in the caller's order)


class RowOrder(object):
    """Training forward passes process the batch's rows in order of non-increasing length (stable: the order of
    utils.sort_resort, CV/utils/utils.py:18-21, computed on the device).  Rows are independent and every loss term
    is a per-row value, so nothing changes numerically; what it buys: rows still inside their sentence at step t
    are a prefix, so the backward recurrence exchanges, and the weight-gradient products visit, live rows only,
    and the encoder's dropout mask (drawn by the reference in exactly this order, encoder.py:50-56) needs no
    indirection.  Per-row outputs are returned in the caller's order (`back`) unless settings.restore_row_order
    is off."""

    def __init__(self, length):
        self.s_idx, self.res_idx = ops.sort_resort(length)
        self.restore = settings.restore_row_order
        self._targets = {}

    def take(self, *ts):
        out = tuple(t.index_select(0, self.s_idx) for t in ts)
        return out if len(out) > 1 else out[0]

    def back(self, t):
        return t.index_select(0, self.res_idx) if self.restore else t

    def targets(self, gts):
        """The caller's (B,T) targets in the processing order (shared by the two decoders' losses)."""
        key = (gts.data_ptr(), tuple(gts.shape))
        if key not in self._targets:
            self._targets = {key: gts.index_select(0, self.s_idx)}
        return self._targets[key]


class _rows_in(object):
    """with _rows_in(order): the replayed noise follows `order` inside the block."""

    def __init__(self, order):
        self.order = order

    def __enter__(self):
        global _row_order
        self.prev, _row_order = _row_order, self.order
        return self.order

    def __exit__(self, *exc):
        global _row_order
        _row_order = self.prev
        return False


def _train_order(length):
    return RowOrder(length) if (settings.sort_train_rows and length.numel() > 1) else None


def _left_sorted(ro):
    """The RowOrder per-row outputs were left in (restore_row_order off), or None when they are in the caller's order."""
    return ro if (ro is not None and not ro.restore) else None


def set_noise_source(src):
    global _noise
    _noise = src if src is not None else NoiseSource()


def _cell():
    return ops.LSTM if settings.use_lstm else ops.GRU


def _rnn_weights(rnn, reverse=False):
    suf = '_reverse' if reverse else ''
    return [getattr(rnn, n + suf) for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]


class LazyLogits(object):
    """What Decoder.forward returns in place of the (R,T,V) logits tensor: the decoder's
    `final` states plus the ToVocab parameters.  SeqLoss consumes it through the fused
    projection + cross-entropy kernel; `.materialize()` builds the real tensor on demand."""

    def __init__(self, final, weight, bias, n_sent, order=None):
        self.final, self.weight, self.bias, self.n_sent = final, weight, bias, n_sent
        self.order = order            # RowOrder the rows of `final` are in (None: the caller's order)
        self.shared = None            # dict shared with the sibling decoder's logits (decode_pair): targets built once
        self.sibling, self.pair_index = None, 0    # the other decoder's logits of the same step (SeqLoss._paired)
        self.always_restore = False   # True: per-sentence sums go back to the caller's order whatever the settings say

    @property
    def shape(self):
        T, R, _ = self.final.shape
        return torch.Size([R, T, self.weight.shape[0]])

    def materialize(self):
        out = ops.logits_bt(self.final.detach(), self.weight.detach(), self.bias.detach())
        return out if self.order is None else out.index_select(0, self.order.res_idx)


class LazyBoW(object):
    """What forward() returns in place of BoW_logits = BoW(z) (CV/Modules/vae.py / vae_couple.py:724): the latent
    and the BoW head's parameters.  The reference's caller does
    `BoW_logits.unsqueeze(1).expand(-1, L, -1)` and hands the result to criterionSeq with the bare tokens as
    targets (CV/Experiment.py:703-706): those two calls turn this object into a LazyLogits whose rows are z repeated
    over the L positions, so the bag-of-words loss runs through the same fused projection + cross-entropy kernel
    and the (B, L, V) tensor is never formed."""

    def __init__(self, z, weight, bias, order=None):
        self.z, self.weight, self.bias = z, weight, bias
        self.order = order            # RowOrder the rows of z were LEFT in (settings.restore_row_order = False), else None

    @property
    def shape(self):
        return torch.Size([self.z.shape[0], self.weight.shape[0]])

    def unsqueeze(self, dim):
        assert dim == 1, 'LazyBoW supports the reference\'s unsqueeze(1).expand(-1, L, -1) only'
        return self

    def expand(self, *sizes):
        L = int(sizes[1])
        B, Z = self.z.shape
        final = self.z.unsqueeze(0).expand(L, B, Z).contiguous()       # time-major rows, as a decoder's `final`
        out = LazyLogits(final, self.weight, self.bias, B, order=self.order)
        out.always_restore = True     # the caller divides the per-sentence sums by ITS sent_len: caller's row order
        return out

    def materialize(self):
        return torch.nn.functional.linear(self.z.detach(), self.weight.detach(), self.bias.detach())


# --------------------------------------------------------------------------- SeqLoss
class SeqLoss(nn.Module):
    """CV/Modules/Losses/SeqLoss.py.  forward(logits, gts, keep_batch) with logits a LazyLogits."""

    def __init__(self, voc_size, pad, end, unk):
        super(SeqLoss, self).__init__()
        self.voc_size = voc_size
        self.pad = pad

    def forward(self, logits, gts, keep_batch=False):
        if not isinstance(logits, LazyLogits):
            raise NotImplementedError('SeqLoss takes the LazyLogits returned by Decoder.forward (or by '
                                      'BoW_logits.unsqueeze(1).expand(-1, L, -1)); dense logits are outside the '
                                      'accelerated path')
        if logits.shape[0] == 0:
            return torch.zeros((), device=gts.device)
        assert logits.shape[:-1] == gts.shape or logits.shape[0] % gts.shape[0] == 0
        tgt = gts[:logits.n_sent]
        if logits.order is not None:
            tgt = logits.order.targets(tgt)
        sent = self._paired(logits, tgt)
        if sent is None:
            sent = ops.proj_ce(logits.final, logits.weight, logits.bias, tgt, self.pad, cache=logits.shared)
        if logits.order is not None:
            sent = sent.index_select(0, logits.order.res_idx) if logits.always_restore else logits.order.back(sent)
        if keep_batch:
            return sent
        return sent.sum() / (gts != self.pad).sum().to(sent.dtype)


    def _paired(self, logits, tgt):
        """The reference takes the decoder's and the coupled decoder's losses one after the other on the same targets
        (Experiment.py:684-702).  The first of the two calls computes both (ops.proj_ce_pair), the second one finds
        its result here; a sibling that is asked about other targets is computed on its own."""
        sib = logits.sibling() if logits.sibling is not None else None
        if sib is None or logits.shared is None:
            return None
        key = ('pair', tgt.data_ptr(), tuple(tgt.shape), tuple(tgt.stride()), tgt._version, int(self.pad))
        hit = logits.shared.get(key)
        if hit is None:
            pair = (logits, sib) if logits.pair_index == 0 else (sib, logits)
            if not ops.proj_ce_pair_ok([q.final for q in pair], [q.weight for q in pair]):
                return None
            outs = ops.proj_ce_pair([q.final for q in pair], [q.weight for q in pair], [q.bias for q in pair], tgt,
                                    self.pad, cache=logits.shared)
            hit = logits.shared[key] = (outs, tgt)          # holds tgt: its storage stays pinned while cached
        return hit[0][logits.pair_index]


# --------------------------------------------------------------------------- Encoder
class Encoder(nn.Module):
    """CV/Modules/encoder.py.  Returns (None, last_states (n_dir, B, H)): no caller on the hot
    path reads the per-step outputs, so they are not produced."""

    def __init__(self, emb_dim, hid_dim, n_layer, dropout, bi, embedding):
        super(Encoder, self).__init__()
        assert n_layer == 1 and bi, 'the accelerated path covers the 1-layer bidirectional encoder of every config'
        self.emb_dim, self.hid_dim, self.n_layer, self.dropout, self.bi = emb_dim, hid_dim, n_layer, dropout, bi
        self.n_dir = 2
        self.drop = nn.Dropout(self.dropout)
        self.Embedding = embedding
        self.use_lstm = settings.use_lstm
        kw = dict(input_size=emb_dim, hidden_size=hid_dim, num_layers=n_layer, bidirectional=True)
        if self.use_lstm:
            self.LSTM = nn.LSTM(**kw)
        else:
            self.GRU = nn.GRU(**kw)

    def _operands(self, bare, length, noise_name, mask_rows='sort'):
        """mask_rows: 'sort' = column of the dropout mask that row b uses, from this batch's own length order (the
        reference draws the mask after sorting by length, encoder.py:50-56); None = the rows already are in that
        order; or the (B,) int64 column vector itself."""
        rnn = self.LSTM if self.use_lstm else self.GRU
        w = _rnn_weights(rnn) + _rnn_weights(rnn, True)
        B, L = bare.shape
        mask = None
        if self.training:
            mask = _noise.dropout_mask((L, B, self.emb_dim), self.dropout, bare.device, noise_name)
        if mask is None:
            mask_rows = None
        elif isinstance(mask_rows, str):
            _, mask_rows = ops.sort_resort(length)
        return w, mask, mask_rows

    def last_flat(self, bare, length, noise_name='enc_drop', mask_rows='sort'):
        """(B, 2H) = [h_fwd | h_bwd]."""
        w, mask, mask_rows = self._operands(bare, length, noise_name, mask_rows)
        cell = ops.LSTM if self.use_lstm else ops.GRU
        return ops.encoder(cell, bare, length, mask, mask_rows, self.Embedding.weight, w)

    @staticmethod
    def last_flat_pair(enc_a, bare_a, len_a, noise_a, enc_b, bare_b, len_b, noise_b, rows_a='sort', rows_b='sort'):
        """Both encoders of the dialogue model in the same launches when they share the embedding and shape;
        otherwise one after the other.  The noise tape is consumed in the same order either way."""
        same = (enc_a.Embedding is enc_b.Embedding and enc_a.hid_dim == enc_b.hid_dim
                and enc_a.use_lstm == enc_b.use_lstm and bare_a.shape[0] == bare_b.shape[0]
                and settings.pair_encoders)
        if not same:
            return enc_a.last_flat(bare_a, len_a, noise_a, rows_a), enc_b.last_flat(bare_b, len_b, noise_b, rows_b)
        w_a, mask_a, rows_a = enc_a._operands(bare_a, len_a, noise_a, rows_a)
        w_b, mask_b, rows_b = enc_b._operands(bare_b, len_b, noise_b, rows_b)
        cell = ops.LSTM if enc_a.use_lstm else ops.GRU
        return ops.encoder_pair(cell, bare_a, len_a, mask_a, rows_a, w_a, bare_b, len_b, mask_b, rows_b, w_b,
                                enc_a.Embedding.weight)

    def forward(self, bare, length):
        last = self.last_flat(bare, length)
        B = bare.shape[0]
        return None, last.view(B, 2, self.hid_dim).transpose(0, 1)


# --------------------------------------------------------------------------- Decoder
class Decoder(nn.Module):
    """Teacher-forcing branch of CV/Modules/vae_couple.py:40-121 (with_init=False) and of
    CV/Modules/decoder.py:28-113 / CC/Modules/decoder.py (with_init=True: owns toInit)."""

    def __init__(self, voc_size, latent_dim, emb_dim, hid_dim, n_layer, dropout, max_len, beam_size, WEAtt_type,
                 embedding, with_init=False):
        super(Decoder, self).__init__()
        assert n_layer == 1 and WEAtt_type == 'none', 'every shipped config uses 1 layer and WEAtt_type="none"'
        self.voc_size, self.latent_dim, self.emb_dim, self.hid_dim = voc_size, latent_dim, emb_dim, hid_dim
        self.n_layers, self.dropout, self.max_len, self.beam_size = n_layer, dropout, max_len, beam_size
        self.WEAtt_type = WEAtt_type
        self.drop = nn.Dropout(dropout)
        self.Embedding = embedding
        if with_init:
            self.toInit = nn.Sequential(nn.Linear(latent_dim, emb_dim), nn.ReLU(), nn.Linear(emb_dim, emb_dim))
        self.use_lstm = settings.use_lstm
        kw = dict(input_size=emb_dim * 2, hidden_size=hid_dim, num_layers=n_layer, bidirectional=False)
        if self.use_lstm:
            self.LSTM = nn.LSTM(**kw)
        else:
            self.GRU = nn.GRU(**kw)
        self.W_c = nn.Linear(hid_dim, emb_dim)
        self.ToVocab = nn.Linear(emb_dim, voc_size)

    def final_states(self, init_input, helper, train, noise_prefix='dec'):
        rnn = self.LSTM if self.use_lstm else self.GRU
        w = _rnn_weights(rnn) + [self.W_c.weight, self.W_c.bias]
        R = init_input.shape[0]
        B, T = helper.shape
        mask_in = mask_out = None
        if train:
            dev = helper.device
            mask_in = _noise.dropout_mask((T, R, 2 * self.emb_dim), self.dropout, dev, noise_prefix + '_in_drop')
            mask_out = _noise.dropout_mask((T, R, self.hid_dim), self.dropout, dev, noise_prefix + '_out_drop')
        cell = ops.LSTM if self.use_lstm else ops.GRU
        return ops.decoder(cell, train, init_input, helper, mask_in, mask_out, self.Embedding.weight, w)

    def final_states_sorted(self, init_input, helper, live_sent):
        rnn = self.LSTM if self.use_lstm else self.GRU
        w = _rnn_weights(rnn) + [self.W_c.weight, self.W_c.bias]
        cell = ops.LSTM if self.use_lstm else ops.GRU
        return ops.decoder_sorted(cell, init_input, helper, live_sent, self.Embedding.weight, w)

    def greedy(self, init_input, helper):
        """Greedy branch (CV/Modules/decoder.py:176-201; vae_couple.py:173-201): max_len steps from
        helper[:, 0], <unk> (id 3) masked as in the reference (test_non_unk).  -> ids (B, max_len)."""
        rnn = self.LSTM if self.use_lstm else self.GRU
        w = _rnn_weights(rnn) + [self.W_c.weight, self.W_c.bias]
        cell = ops.LSTM if self.use_lstm else ops.GRU
        return ops.decoder_greedy(cell, init_input, helper[:, 0], self.max_len, self.Embedding.weight, w,
                                  self.ToVocab.weight, self.ToVocab.bias, mask_id=3)

    def beam(self, init_input, helper, return_nll=False):
        """Beam branch (CV/Modules/decoder.py:114-168; vae_couple.py:122-172) with self.beam_size beams.
        -> ids (B, max_len) of beam[0]."""
        rnn = self.LSTM if self.use_lstm else self.GRU
        w = _rnn_weights(rnn) + [self.W_c.weight, self.W_c.bias]
        cell = ops.LSTM if self.use_lstm else ops.GRU
        return ops.decoder_beam(cell, init_input, helper[:, 0], self.max_len, self.beam_size, self.Embedding.weight,
                                w, self.ToVocab.weight, self.ToVocab.bias, mask_id=3, return_nll=return_nll)

    def forward(self, init_states, init_input, helper, test_lm=False, _noise_prefix='dec', _n_sent=None, _order=None):
        if not (self.training or test_lm):
            ids = self.beam(init_input, helper) if self.beam_size > 1 else self.greedy(init_input, helper)
            # the reference's return format: per sentence, a list of max_len tensors of shape (1,)
            return [[ids[b, t:t + 1] for t in range(ids.shape[1])] for b in range(ids.shape[0])]
        n_sent = helper.shape[0] if _n_sent is None else _n_sent
        final = self.final_states(init_input, helper[:n_sent], self.training and not test_lm, _noise_prefix)
        return LazyLogits(final, self.ToVocab.weight, self.ToVocab.bias, n_sent, _order)


def decode_pair(dec_a, dec_b, init_a, init_b, helper, order=None):
    """Decoder and DecoderCouple on the same `helper` in lock-step (training): same draws, in the
    reference's order (dec in/out masks, then decc in/out masks), one launch per timestep for both."""
    R = init_a.shape[0]
    B, T = helper.shape
    dev = helper.device
    specs = []
    for dec, init, pre in ((dec_a, init_a, 'dec'), (dec_b, init_b, 'decc')):
        rnn = dec.LSTM if dec.use_lstm else dec.GRU
        w = _rnn_weights(rnn) + [dec.W_c.weight, dec.W_c.bias]
        mi = _noise.dropout_mask((T, R, 2 * dec.emb_dim), dec.dropout, dev, pre + '_in_drop')
        mo = _noise.dropout_mask((T, R, dec.hid_dim), dec.dropout, dev, pre + '_out_drop')
        specs.append((init, mi, mo, w))
    cell = ops.LSTM if dec_a.use_lstm else ops.GRU
    fa, fb = ops.decoder_pair(cell, helper, dec_a.Embedding.weight, specs)
    la = LazyLogits(fa, dec_a.ToVocab.weight, dec_a.ToVocab.bias, B, order)
    lb = LazyLogits(fb, dec_b.ToVocab.weight, dec_b.ToVocab.bias, B, order)
    la.shared = lb.shared = {}
    la.sibling, lb.sibling, lb.pair_index = weakref.ref(lb), weakref.ref(la), 1     # weak: no reference cycle
    return la, lb


def to_init(seq, x):
    """Linear-ReLU-Linear (vae_couple.py:429-434)."""
    h = ops.linear(x, seq[0].weight, seq[0].bias, ops.ACT_RELU)
    return ops.linear(h, seq[2].weight, seq[2].bias, ops.ACT_NONE)


# --------------------------------------------------------------------------- Gaussian
class Gaussian(nn.Module):
    """CV/Modules/vae_couple.py:781-803."""

    def __init__(self, in_dim, out_dim):
        super(Gaussian, self).__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.toMiu = nn.Sequential(nn.Linear(in_dim, out_dim))
        self.toSigma = nn.Sequential(nn.Linear(in_dim, out_dim))

    def params(self, x, need_sigma=True):
        mu = ops.linear(x, self.toMiu[0].weight, self.toMiu[0].bias, ops.ACT_NONE)
        sigma = ops.linear(x, self.toSigma[0].weight, self.toSigma[0].bias, ops.ACT_EXP) if need_sigma else None
        return mu, sigma

    def forward(self, input, _noise_name='eps', _sample=True):
        mu, sigma = self.params(input)
        dist = torch.distributions.Normal(mu, sigma, validate_args=False)
        eps = _noise.normal((1,) + tuple(mu.shape), mu.device, _noise_name).squeeze(0)   # always drawn
        z = ops.ReparamFn.apply(mu, sigma, eps) if _sample else None
        return dist, z


# --------------------------------------------------------------------------- IW-NLL core
def _iw_nll(decoder, to_init_seq, go, eos, mu, sigma, mu_p, sigma_p, post_repr, n_sample, pad):
    """importance_sampling (CV/Modules/vae_couple.py:526-578; CC/...:312-364) on the GPU.
    eps is drawn in the reference's (2,B,Z) chunks; samples are then decoded `kc` at a time
    (rows kc*B), which changes nothing numerically per row."""
    n_chunk = settings.n_sample_chunk
    assert n_sample % n_chunk == 0
    B, T = go.shape
    Z = mu.shape[1]
    dev = go.device
    eps = _noise.normal_chunks(n_sample // n_chunk, (n_chunk, B, Z), dev, 'iw_eps')
    sharded = settings.shard_samples and dist_utils.world() > 1
    if sharded:
        dist_utils.broadcast_(eps)                                      # one eps sequence for every rank
    z = ops.rsample_n(mu, sigma, eps)                                   # (K,B,Z)
    k_lo, k_hi = dist_utils.sample_slice(n_sample) if sharded else (0, n_sample)
    kc = max(1, min(k_hi - k_lo, settings.iw_max_rows // B))
    xent = torch.empty((k_hi - k_lo, B), dtype=torch.float32, device=dev)
    plan = _sorted_plan(eos, pad) if settings.sort_eval and kc * B > ops.SKINNY_MAX_ROWS else None
    W, bias = decoder.ToVocab.weight, decoder.ToVocab.bias
    for k0 in range(k_lo, k_hi, kc):
        k1 = min(k_hi, k0 + kc)
        out = xent[k0 - k_lo:k1 - k_lo]
        if plan is not None and (k1 - k0) * B > ops.SKINNY_MAX_ROWS:
            order, live_sent = plan
            zz = z[k0:k1].index_select(1, order).transpose(0, 1).reshape(-1, Z)      # row b*(k1-k0) + k
            init = to_init(to_init_seq, zz)
            if post_repr is not None:
                init = init + post_repr.index_select(0, order).repeat_interleave(k1 - k0, 0)
            go_s, eos_s = go.index_select(0, order), eos.index_select(0, order)
            final = decoder.final_states_sorted(init, go_s, live_sent)
            xs = ops.proj_ce_sorted(final, W, bias, eos_s, pad).view(B, k1 - k0)
            out.index_copy_(1, order, xs.t())
        else:
            zz = z[k0:k1].reshape(-1, Z)
            init = to_init(to_init_seq, zz)
            if post_repr is not None:
                init = init + post_repr.repeat(k1 - k0, 1)
            final = decoder.final_states(init, go, False)
            out.copy_(ops.proj_ce(final, W, bias, eos, pad).view(k1 - k0, B))
    if sharded:
        xent = dist_utils.allgather_rows(xent)
    xm, nll, kl = ops.iw_reduce(z, xent, mu, sigma, mu_p, sigma_p)
    return xm, nll, kl, z.transpose(0, 1)


def _sorted_plan(eos, pad):
    """Order sentences by non-increasing target length so that the rows still decoding at step t
    are a prefix (ops.decoder_sorted).  One device->host copy of B lengths per batch.  Returns
    None when some sentence has <pad> inside its target (then no prefix exists)."""
    B, T = eos.shape
    live = eos != pad
    lens = live.sum(1)
    steps = torch.arange(T, device=eos.device)
    prefix_ok = (live == (steps[None, :] < lens[:, None])).all()
    host = torch.cat([lens, prefix_ok.view(1).to(lens.dtype)]).cpu()
    if int(host[-1]) == 0:
        return None
    order = torch.argsort(lens, descending=True, stable=True)
    live_sent = (host[:-1][None, :] > torch.arange(T)[:, None]).sum(1).to(torch.int32).contiguous()
    return order, live_sent


def _truncate_eval(go, eos, bare, sent_len):
    """Pad targets carry weight 0 (SeqLoss.py:15-16), so cutting T to max(len)+1 is exact."""
    if not settings.truncate_eval:
        return go, eos, bare
    m = int(sent_len.max())
    return go[:, :m + 1].contiguous(), eos[:, :m + 1].contiguous(), bare[:, :max(m, 1)].contiguous()


# --------------------------------------------------------------------------- text models
class _TextBase(nn.Module):
    def _common(self, hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                WEAtt_type, encoder_emb, decoder_emb):
        assert encoder_emb.num_embeddings == decoder_emb.num_embeddings
        assert encoder_emb.embedding_dim == decoder_emb.embedding_dim
        self.voc_size, self.emb_dim = encoder_emb.num_embeddings, encoder_emb.embedding_dim
        self.hid_dim, self.enc_layers, self.dec_layers, self.dropout = hid_dim, enc_layers, dec_layers, dropout
        self.enc_bi, self.n_dir = enc_bi, 2 if enc_bi else 1
        self.dec_max_len, self.beam_size, self.WEAtt_type, self.latent_dim = dec_max_len, beam_size, WEAtt_type, latent_dim

    def _dec_kwargs(self, decoder_emb):
        return dict(voc_size=self.voc_size, latent_dim=self.latent_dim, emb_dim=self.emb_dim,
                    hid_dim=self.hid_dim * self.n_dir, n_layer=self.dec_layers, dropout=self.dropout,
                    max_len=self.dec_max_len, beam_size=self.beam_size, WEAtt_type=self.WEAtt_type,
                    embedding=decoder_emb)

    def _to_init_seq(self):
        return self.toInit if hasattr(self, 'toInit') else self.Decoder.toInit

    def decode_ids(self, latents, go):
        """decode_from (CV/Modules/vae_couple.py:681-696) as an id tensor (B, max_len)."""
        with torch.no_grad():
            init = to_init(self._to_init_seq(), latents)
            return self.Decoder.beam(init, go) if self.Decoder.beam_size > 1 else self.Decoder.greedy(init, go)

    def decode_from(self, latents, go):
        """CV/Modules/vae_couple.py:681-696 / vae.py:299: greedy (or, with Decoder.beam_size > 1, beam) decode of given latents, reference format."""
        ids = self.decode_ids(latents, go)
        return [[ids[b, t:t + 1] for t in range(ids.shape[1])] for b in range(ids.shape[0])]

    def sample_from_prior(self, go):
        """CV/Modules/vae_couple.py:635-655: z ~ N(0, I) then greedy decode."""
        z = _noise.normal((go.shape[0], self.latent_dim), go.device, 'prior_z')
        return self.decode_from(z, go)

    def gen_interps(self, bareA, sent_lenA, bareB, sent_lenB, go, n_interps):
        """CV/Modules/vae_couple.py:583-633 (vae.py:230-278): decode n_interps + 2 points on the segment between the
        posterior means of A and B.  -> per sentence a list of n_interps + 2 decoded sentences (reference format)."""
        with torch.no_grad():
            zA, _ = self.PosteriorGaussian.params(self.Encoder.last_flat(bareA, sent_lenA), need_sigma=False)
            _noise.normal((1,) + tuple(zA.shape), zA.device, 'eps')            # Gaussian.forward always draws
            zB, _ = self.PosteriorGaussian.params(self.Encoder.last_flat(bareB, sent_lenB), need_sigma=False)
            _noise.normal((1,) + tuple(zB.shape), zB.device, 'eps')
        interps = [[] for _ in range(go.shape[0])]
        for i in range(n_interps + 2):
            zk = zA * ((n_interps - i + 1) / (n_interps + 1)) + zB * (i / (n_interps + 1))
            for b, sent in enumerate(self.decode_from(zk, go)):
                interps[b].append(sent)
        return interps

    def sample_from_posterior(self, bare, sent_len, n_sample):
        """CV/Modules/vae_couple.py:657-679: n_sample draws of q(z|x) -> (B, n_sample, Z).  (Normal.sample draws
        one (n_sample, B, Z) block after the (1, B, Z) draw of Gaussian.forward.)"""
        with torch.no_grad():
            mu, sigma = self.PosteriorGaussian.params(self.Encoder.last_flat(bare, sent_len))
            _noise.normal((1,) + tuple(mu.shape), mu.device, 'eps')
            eps = _noise.normal((n_sample,) + tuple(mu.shape), mu.device, 'posterior_eps')
            return ops.rsample_n(mu, sigma, eps).transpose(0, 1).contiguous()

    def estimate_mi(self, go, sent_len, bare, n_sample):
        """CV/Modules/vae_couple.py:453-502 -> (mi (B,), z (B,K,Z))."""
        with torch.no_grad():
            last = self.Encoder.last_flat(bare, sent_len)
            mu, sigma = self.PosteriorGaussian.params(last)
            _noise.normal((1,) + tuple(mu.shape), mu.device, 'eps')        # Gaussian.forward always draws
            n_chunk = settings.n_sample_chunk
            assert n_sample % n_chunk == 0
            B, Z = mu.shape
            eps = _noise.normal_chunks(n_sample // n_chunk, (n_chunk, B, Z), mu.device, 'mi_eps')
            z = ops.rsample_n(mu, sigma, eps)
            return ops.mi(z, mu, sigma), z.transpose(0, 1)

    def test_lm(self, go, sent_len, bare, eos, n_sample):
        """CV/Modules/vae_couple.py:504-524 -> (xent, nll, kl (B,), z (B,K,Z))."""
        with torch.no_grad():
            go, eos, bare = _truncate_eval(go, eos, bare, sent_len)
            last = self.Encoder.last_flat(bare, sent_len)
            mu, sigma = self.PosteriorGaussian.params(last)
            _noise.normal((1,) + tuple(mu.shape), mu.device, 'eps')
            return _iw_nll(self.Decoder, self._to_init_seq(), go, eos, mu, sigma, None, None, None, n_sample,
                           self.criterionSeq.pad)


class VAE(_TextBase):
    """CV/Modules/vae.py:17-346."""

    def __init__(self, hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                 WEAtt_type, encoder_emb, decoder_emb, pad_id):
        super(VAE, self).__init__()
        self._common(hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                     WEAtt_type, encoder_emb, decoder_emb)
        self.Encoder = Encoder(self.emb_dim, hid_dim, enc_layers, dropout, enc_bi, encoder_emb)
        self.PosteriorGaussian = Gaussian(hid_dim * self.n_dir * enc_layers, latent_dim)
        self.Decoder = Decoder(with_init=True, **self._dec_kwargs(decoder_emb))
        self.BoW = nn.Linear(latent_dim, self.voc_size)
        self.criterionSeq = SeqLoss(voc_size=self.voc_size, pad=pad_id, end=None, unk=None)

    def forward(self, go, sent_len=None, bare=None):
        if not self.training:
            raise NotImplementedError()
        ro = _train_order(sent_len)
        logits, dist, z, _, _ = self._run(go, sent_len, bare, ro)
        return logits, dist, z, LazyBoW(z, self.BoW.weight, self.BoW.bias, _left_sorted(ro))

    def _run(self, go, sent_len, bare, ro):
        if ro is not None:
            go, bare, sent_len = ro.take(go, bare, sent_len)
        with _rows_in(ro):
            last = self.Encoder.last_flat(bare, sent_len, mask_rows='sort' if ro is None else None)
            dist, z = self.PosteriorGaussian(last)
            init = to_init(self.Decoder.toInit, z)
            logits = self.Decoder(None, init, go, _order=ro)
        if ro is not None:
            dist = torch.distributions.Normal(ro.back(dist.loc), ro.back(dist.scale), validate_args=False)
            z, init = ro.back(z), ro.back(init)
        return logits, dist, z, init, last

    def saliency(self, go, sent_len=None, bare=None):
        """CV/Modules/vae.py:348-374: the training forward that also hands out the encoder state, for
        Experiment.analyze_saliency's torch.autograd.grad calls (rows stay in the caller's order)."""
        logits, dist, _, init, last = self._run(go, sent_len, bare, None)
        return logits, dist, init, last


class DAE(_TextBase):
    """CV/Modules/dae.py: deterministic auto-encoder on the same Encoder / Decoder (no posterior)."""

    def __init__(self, hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                 WEAtt_type, encoder_emb, decoder_emb, pad_id):
        super(DAE, self).__init__()
        self._common(hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                     WEAtt_type, encoder_emb, decoder_emb)
        self.Encoder = Encoder(self.emb_dim, hid_dim, enc_layers, dropout, enc_bi, encoder_emb)
        self.toLatent = nn.Linear(hid_dim * self.n_dir * enc_layers, latent_dim)
        self.Decoder = Decoder(with_init=True, **self._dec_kwargs(decoder_emb))
        self.criterionSeq = SeqLoss(voc_size=self.voc_size, pad=pad_id, end=None, unk=None)

    def _latent(self, bare, sent_len, mask_rows='sort'):
        last = self.Encoder.last_flat(bare, sent_len, mask_rows=mask_rows)
        return ops.linear(last, self.toLatent.weight, self.toLatent.bias, ops.ACT_NONE)

    def forward(self, go, sent_len=None, bare=None):
        if not self.training:                     # dae.py forward, eval branch: decode a prior sample greedily
            latent = _noise.normal((go.shape[0], self.latent_dim), go.device, 'prior_z')
            with torch.no_grad():
                return self.Decoder(None, to_init(self.Decoder.toInit, latent), go)
        ro = _train_order(sent_len)
        if ro is None:
            latent = self._latent(bare, sent_len)
            return self.Decoder(None, to_init(self.Decoder.toInit, latent), go), latent
        go, bare, sent_len = ro.take(go, bare, sent_len)
        with _rows_in(ro):
            latent = self._latent(bare, sent_len, mask_rows=None)
            logits = self.Decoder(None, to_init(self.Decoder.toInit, latent), go, _order=ro)
        return logits, ro.back(latent)

    def saliency(self, go, sent_len=None, bare=None):
        """CV/Modules/dae.py saliency: (logits, init_input, last_states), rows in the caller's order."""
        last = self.Encoder.last_flat(bare, sent_len)
        latent = ops.linear(last, self.toLatent.weight, self.toLatent.bias, ops.ACT_NONE)
        init = to_init(self.Decoder.toInit, latent)
        return self.Decoder(None, init, go), init, last

    def test_lm(self, go, sent_len, bare, eos, n_sample):
        """dae.py test_lm: one deterministic decode; nll = kl = +inf by the reference's definition."""
        with torch.no_grad():
            go, eos, bare = _truncate_eval(go, eos, bare, sent_len)
            latent = self._latent(bare, sent_len)
            final = self.Decoder.final_states(to_init(self.Decoder.toInit, latent), go, False)
            xent = ops.proj_ce(final, self.Decoder.ToVocab.weight, self.Decoder.ToVocab.bias, eos, self.criterionSeq.pad)
            inf = torch.full_like(xent, float('inf'))
            return xent, xent + inf, inf, latent


class LMDecoder(nn.Module):
    """CV/Modules/lm.py:28-100 (its own Decoder class): GRU over the word embeddings alone, h_0 = 0, no
    latent input; `toInit` exists in the state_dict but is never used.  The reference builds the LSTM
    flavour with input_size 2E and then feeds E-wide inputs (it cannot run), so only GRU is offered."""

    def __init__(self, voc_size, latent_dim, emb_dim, hid_dim, n_layer, dropout, max_len, beam_size, WEAtt_type,
                 embedding):
        super(LMDecoder, self).__init__()
        assert n_layer == 1 and WEAtt_type == 'none'
        assert not settings.use_lstm, 'Modules/lm.py cannot run with use_lstm (input width mismatch in the reference)'
        self.voc_size, self.latent_dim, self.emb_dim, self.hid_dim = voc_size, latent_dim, emb_dim, hid_dim
        self.n_layers, self.dropout, self.max_len, self.beam_size = n_layer, dropout, max_len, beam_size
        self.WEAtt_type, self.use_lstm = WEAtt_type, False
        self.drop = nn.Dropout(dropout)
        self.Embedding = embedding
        self.toInit = nn.Sequential(nn.Linear(latent_dim, emb_dim), nn.ReLU(), nn.Linear(emb_dim, emb_dim))
        self.GRU = nn.GRU(input_size=emb_dim, hidden_size=hid_dim, num_layers=n_layer, bidirectional=False)
        self.W_c = nn.Linear(hid_dim, emb_dim)
        self.ToVocab = nn.Linear(emb_dim, voc_size)

    def final_states(self, helper, train):
        w = _rnn_weights(self.GRU) + [self.W_c.weight, self.W_c.bias]
        B, T = helper.shape
        mask_in = mask_out = None
        if train:
            dev = helper.device
            mask_in = _noise.dropout_mask((T, B, self.emb_dim), self.dropout, dev, 'dec_in_drop')
            mask_out = _noise.dropout_mask((T, B, self.hid_dim), self.dropout, dev, 'dec_out_drop')
        return ops.decoder_noprefix(ops.GRU, helper, mask_in, mask_out, self.Embedding.weight, w)

    def forward(self, init_states, latent_vector, helper, test_lm=False):
        if not (self.training or test_lm):
            raise NotImplementedError('the LM does not decode (Experiment.sample_from_prior returns early for it)')
        final = self.final_states(helper, self.training and not test_lm)
        return LazyLogits(final, self.ToVocab.weight, self.ToVocab.bias, helper.shape[0])


class LM(nn.Module):
    """CV/Modules/lm.py:190-256."""

    def __init__(self, hid_dim, latent_dim, dec_layers, dropout, dec_max_len, beam_size, WEAtt_type, decoder_emb,
                 pad_id, **_unused):
        super(LM, self).__init__()
        self.voc_size, self.emb_dim = decoder_emb.num_embeddings, decoder_emb.embedding_dim
        self.hid_dim, self.dec_layers, self.dropout, self.n_dir = hid_dim, dec_layers, dropout, 2
        self.dec_max_len, self.beam_size, self.WEAtt_type, self.latent_dim = dec_max_len, beam_size, WEAtt_type, latent_dim
        self.Decoder = LMDecoder(voc_size=self.voc_size, latent_dim=latent_dim, emb_dim=self.emb_dim,
                                 hid_dim=hid_dim * self.n_dir, n_layer=dec_layers, dropout=dropout, max_len=dec_max_len,
                                 beam_size=beam_size, WEAtt_type=WEAtt_type, embedding=decoder_emb)
        self.criterionSeq = SeqLoss(voc_size=self.voc_size, pad=pad_id, end=None, unk=None)

    def forward(self, go, sent_len=None, bare=None):
        if not self.training:
            raise NotImplementedError()
        return self.Decoder(None, None, go)

    def test_lm(self, go, sent_len, bare, eos, n_sample):
        """lm.py:218-235: xent of one deterministic pass; kl = 0, nll = xent, latent = zeros (B, Z)."""
        with torch.no_grad():
            go, eos, bare = _truncate_eval(go, eos, bare, sent_len)
            final = self.Decoder.final_states(go, False)
            xent = ops.proj_ce(final, self.Decoder.ToVocab.weight, self.Decoder.ToVocab.bias, eos, self.criterionSeq.pad)
            kl = torch.zeros_like(xent)
            return xent, xent + kl, kl, torch.zeros(go.shape[0], self.latent_dim)


class VAE_COUPLE(_TextBase):
    """CV/Modules/vae_couple.py:374-741."""

    def __init__(self, hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                 WEAtt_type, encoder_emb, decoder_emb, pad_id):
        super(VAE_COUPLE, self).__init__()
        self._common(hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                     WEAtt_type, encoder_emb, decoder_emb)
        self.Encoder = Encoder(self.emb_dim, hid_dim, enc_layers, dropout, enc_bi, encoder_emb)
        self.PosteriorGaussian = Gaussian(hid_dim * self.n_dir * enc_layers, latent_dim)
        self.PosteriorGaussianCouple = Gaussian(hid_dim * self.n_dir * enc_layers, latent_dim)
        self.Decoder = Decoder(**self._dec_kwargs(decoder_emb))
        self.DecoderCouple = Decoder(**self._dec_kwargs(decoder_emb))
        self.BoW = nn.Linear(latent_dim, self.voc_size)
        self.criterionSeq = SeqLoss(voc_size=self.voc_size, pad=pad_id, end=None, unk=None)
        self.toInit = nn.Sequential(nn.Linear(latent_dim, self.emb_dim), nn.ReLU(), nn.Linear(self.emb_dim, self.emb_dim))
        self.toInitCouple = nn.Sequential(nn.Linear(latent_dim, self.emb_dim), nn.ReLU(),
                                          nn.Linear(self.emb_dim, self.emb_dim))

    def forward(self, go, sent_len=None, bare=None):
        if not self.training:
            raise NotImplementedError()
        ro = _train_order(sent_len)
        logits, dist, z, logits_c, init, init_c, _ = self._run(go, sent_len, bare, ro)
        return logits, dist, z, LazyBoW(z, self.BoW.weight, self.BoW.bias, _left_sorted(ro)), logits_c, init, init_c

    def _run(self, go, sent_len, bare, ro):
        if ro is not None:
            go, bare, sent_len = ro.take(go, bare, sent_len)
        with _rows_in(ro):
            last = self.Encoder.last_flat(bare, sent_len, mask_rows='sort' if ro is None else None)
            dist, z = self.PosteriorGaussian(last)
            # the coupled head only uses its mean, but the reference still draws its noise (vae_couple.py:721,801)
            mu_c, _ = self.PosteriorGaussianCouple.params(last, need_sigma=False)
            _noise.normal((1,) + tuple(mu_c.shape), mu_c.device, 'eps_couple')
            init = to_init(self.toInit, z)
            init_c = to_init(self.toInitCouple, mu_c)
            logits, logits_c = decode_pair(self.Decoder, self.DecoderCouple, init, init_c, go, ro)
        if ro is not None:
            dist = torch.distributions.Normal(ro.back(dist.loc), ro.back(dist.scale), validate_args=False)
            z, init, init_c = ro.back(z), ro.back(init), ro.back(init_c)
        return logits, dist, z, logits_c, init, init_c, last

    def saliency(self, go, sent_len=None, bare=None):
        """CV/Modules/vae_couple.py:743-778: the training forward that also hands out the encoder state (B, 2H),
        for Experiment.analyze_saliency's torch.autograd.grad calls (rows stay in the caller's order)."""
        logits, dist, _, logits_c, init, init_c, last = self._run(go, sent_len, bare, None)
        return logits, logits_c, dist, init, init_c, last


# --------------------------------------------------------------------------- dialogue models
class _DialogBase(nn.Module):
    _common = _TextBase._common
    _dec_kwargs = _TextBase._dec_kwargs
    _to_init_seq = _TextBase._to_init_seq

    def _encode(self, post_bare, post_len, resp_bare, resp_len, rows_post='sort', rows_resp='sort'):
        post, resp = Encoder.last_flat_pair(self.PostEncoder, post_bare, post_len, 'post_enc_drop',
                                            self.RespEncoder, resp_bare, resp_len, 'resp_enc_drop', rows_post, rows_resp)
        post_repr = ops.linear(post, self.PostRepr.weight, self.PostRepr.bias)
        return post, post_repr, torch.cat([resp, post], dim=1)

    def _train_rows(self, post_bare, post_len, resp_go, resp_len, resp_bare):
        """Rows in order of non-increasing RESPONSE length (the decoders' sequences).  The response encoder's mask
        is then in its own order; the post encoder's mask columns follow the reference's sort of the ORIGINAL
        rows by post length, looked up through the row order."""
        ro = _train_order(resp_len)
        if ro is None:
            return None, (post_bare, post_len, resp_go, resp_len, resp_bare), 'sort', 'sort'
        rows_post = 'sort'
        if self.training and self.dropout > 0.0:
            rows_post = ops.sort_resort(post_len)[1].index_select(0, ro.s_idx)
        return ro, ro.take(post_bare, post_len, resp_go, resp_len, resp_bare), rows_post, None

    def sample_from_prior(self, post_bare, post_len, resp_go):
        """CC/Modules/vae_couple.py:369-397 / vae.py:157-184: encode the post, z ~ prior network, decode
        (greedy or beam in eval mode; the reference's list-of-lists return format)."""
        post = self.PostEncoder.last_flat(post_bare, post_len, 'post_enc_drop')
        post_repr = ops.linear(post, self.PostRepr.weight, self.PostRepr.bias)
        _, z = self.PriorGaussian(post, _noise_name='eps_prior')
        init = to_init(self._to_init_seq(), z) + post_repr
        return self.Decoder(None, init, resp_go)

    def test_lm(self, post_bare, post_len, resp_go, resp_len, resp_bare, resp_eos, n_sample):
        """CC/Modules/vae_couple.py:274-364 -> (xent, nll, kl)."""
        with torch.no_grad():
            post, post_repr, both = self._encode(post_bare, post_len, resp_bare, resp_len)
            mu_p, sig_p = self.PriorGaussian.params(post)
            _noise.normal((1,) + tuple(mu_p.shape), post.device, 'eps_prior')
            mu, sigma = self.PosteriorGaussian.params(both)
            _noise.normal((1,) + tuple(mu.shape), post.device, 'eps')
            xm, nll, kl, _ = _iw_nll(self.Decoder, self._to_init_seq(), resp_go, resp_eos, mu, sigma, mu_p, sig_p,
                                     post_repr, n_sample, self.criterionSeq.pad)
            return xm, nll, kl


class CVAE(_DialogBase):
    """CC/Modules/vae.py (class VAE there)."""

    def __init__(self, hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                 WEAtt_type, encoder_emb, decoder_emb, pad_id):
        super(CVAE, self).__init__()
        self._common(hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                     WEAtt_type, encoder_emb, decoder_emb)
        self.PostEncoder = Encoder(self.emb_dim, hid_dim, enc_layers, dropout, enc_bi, encoder_emb)
        self.RespEncoder = Encoder(self.emb_dim, hid_dim, enc_layers, dropout, enc_bi, encoder_emb)
        d = hid_dim * self.n_dir * enc_layers
        self.PriorGaussian = Gaussian(d, latent_dim)
        self.PosteriorGaussian = Gaussian(2 * d, latent_dim)
        self.PostRepr = nn.Linear(d, self.emb_dim)
        self.Decoder = Decoder(with_init=True, **self._dec_kwargs(decoder_emb))
        self.BoW = nn.Linear(latent_dim, self.voc_size)
        self.criterionSeq = SeqLoss(voc_size=self.voc_size, pad=pad_id, end=None, unk=None)

    def forward(self, post_bare, post_len, resp_go, resp_len, resp_bare):
        if not self.training:
            raise NotImplementedError()
        ro, (post_bare, post_len, resp_go, resp_len, resp_bare), rows_post, rows_resp = self._train_rows(
            post_bare, post_len, resp_go, resp_len, resp_bare)
        with _rows_in(ro):
            post, post_repr, both = self._encode(post_bare, post_len, resp_bare, resp_len, rows_post, rows_resp)
            prior, _ = self.PriorGaussian(post, _noise_name='eps_prior', _sample=False)
            dist, z = self.PosteriorGaussian(both)
            init = to_init(self.Decoder.toInit, z) + post_repr
            logits = self.Decoder(None, init, resp_go, _order=ro)
        if ro is not None:
            prior = torch.distributions.Normal(ro.back(prior.loc), ro.back(prior.scale), validate_args=False)
            dist = torch.distributions.Normal(ro.back(dist.loc), ro.back(dist.scale), validate_args=False)
        return logits, prior, dist


class CVAE_COUPLE(_DialogBase):
    """CC/Modules/vae_couple.py:207-451 (class VAE_COUPLE there)."""

    def __init__(self, hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                 WEAtt_type, encoder_emb, decoder_emb, pad_id):
        super(CVAE_COUPLE, self).__init__()
        self._common(hid_dim, latent_dim, enc_layers, dec_layers, dropout, enc_bi, dec_max_len, beam_size,
                     WEAtt_type, encoder_emb, decoder_emb)
        self.PostEncoder = Encoder(self.emb_dim, hid_dim, enc_layers, dropout, enc_bi, encoder_emb)
        self.RespEncoder = Encoder(self.emb_dim, hid_dim, enc_layers, dropout, enc_bi, encoder_emb)
        d = hid_dim * self.n_dir * enc_layers
        self.PriorGaussian = Gaussian(d, latent_dim)
        self.PosteriorGaussian = Gaussian(2 * d, latent_dim)
        self.PosteriorGaussianCouple = Gaussian(2 * d, latent_dim)
        self.Decoder = Decoder(**self._dec_kwargs(decoder_emb))
        self.DecoderCouple = Decoder(**self._dec_kwargs(decoder_emb))
        self.BoW = nn.Linear(latent_dim, self.voc_size)
        self.PostRepr = nn.Linear(d, self.emb_dim)
        self.criterionSeq = SeqLoss(voc_size=self.voc_size, pad=pad_id, end=None, unk=None)
        self.toInit = nn.Sequential(nn.Linear(latent_dim, self.emb_dim), nn.ReLU(), nn.Linear(self.emb_dim, self.emb_dim))
        self.toInitCouple = nn.Sequential(nn.Linear(latent_dim, self.emb_dim), nn.ReLU(),
                                          nn.Linear(self.emb_dim, self.emb_dim))

    def forward(self, post_bare, post_len, resp_go, resp_len, resp_bare):
        if not self.training:
            raise NotImplementedError()
        ro, (post_bare, post_len, resp_go, resp_len, resp_bare), rows_post, rows_resp = self._train_rows(
            post_bare, post_len, resp_go, resp_len, resp_bare)
        with _rows_in(ro):
            post, post_repr, both = self._encode(post_bare, post_len, resp_bare, resp_len, rows_post, rows_resp)
            prior, _ = self.PriorGaussian(post, _noise_name='eps_prior', _sample=False)
            dist, z = self.PosteriorGaussian(both)
            mu_c, _ = self.PosteriorGaussianCouple.params(both, need_sigma=False)
            _noise.normal((1,) + tuple(mu_c.shape), mu_c.device, 'eps_couple')
            init = to_init(self.toInit, z) + post_repr
            init_c = to_init(self.toInitCouple, mu_c) + post_repr
            logits, logits_c = decode_pair(self.Decoder, self.DecoderCouple, init, init_c, resp_go, ro)
        if ro is not None:
            prior = torch.distributions.Normal(ro.back(prior.loc), ro.back(prior.scale), validate_args=False)
            dist = torch.distributions.Normal(ro.back(dist.loc), ro.back(dist.scale), validate_args=False)
            init, init_c = ro.back(init), ro.back(init_c)
        return logits, prior, dist, logits_c, init, init_c


# Experiment.__init__ (CV/Experiment.py:96-383) builds beta-vae*, cyc-anneal-vae* and surprising-fix* from the
# very same VAE / VAE_COUPLE classes; they differ in the lambda_KL schedule / KL term (harness).
MODEL_CLASSES = {'vae': VAE, 'vae-couple': VAE_COUPLE, 'cvae': CVAE, 'cvae-couple': CVAE_COUPLE, 'dae': DAE, 'lm': LM,
                 'beta-vae': VAE, 'beta-vae-couple': VAE_COUPLE, 'cyc-anneal-vae': VAE,
                 'cyc-anneal-vae-couple': VAE_COUPLE, 'surprising-fix': VAE, 'surprising-fix-couple': VAE_COUPLE}
