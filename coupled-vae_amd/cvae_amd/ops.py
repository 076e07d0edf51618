"""torch.autograd wrappers around the C ABI (one Function per reference function on the path).

torch is plumbing here: it owns device memory, the stream and the autograd tape; every
number is produced by libcvae_hip kernels.  All tensors are fp32 / int64 on the GPU.
"""
import ctypes

import torch

from . import _lib
from .settings import settings
from ._lib import call, ptr, ptr_array, query, stream

GRU, LSTM = 0, 1
SKINNY_MAX_ROWS = 256      # csrc/common.hpp: above this row count the recurrence runs as tiled GEMM + cell kernel
ACT_NONE, ACT_TANH, ACT_RELU, ACT_EXP = 0, 1, 2, 3


def _f(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _c(t):
    return t if t is None or t.is_contiguous() else t.contiguous()


def gates_of(cell):
    return 3 if cell == GRU else 4


# --------------------------------------------------------------------------- a1 / a2
def pack_batch(raw_flat, raw_offsets, max_len, width, pad=0, go=1, eos=2):
    """CV/dataloaders/ptb.py:118-129 + Experiment.py:659-664 on device."""
    B = raw_offsets.numel() - 1
    dev = raw_flat.device
    bare = torch.empty((B, width), dtype=torch.int64, device=dev)
    go_t = torch.empty((B, width + 1), dtype=torch.int64, device=dev)
    eos_t = torch.empty((B, width + 1), dtype=torch.int64, device=dev)
    ln = torch.empty((B,), dtype=torch.int64, device=dev)
    call('cvae_pack_batch', ptr(raw_flat), ptr(raw_offsets), B, max_len, width, pad, go, eos, ptr(bare), ptr(go_t),
         ptr(eos_t), ptr(ln), stream())
    return bare, go_t, eos_t, ln


def pack_batch_indexed(corpus_flat, corpus_offsets, sel, max_len, width, pad=0, go=1, eos=2):
    """pack_batch for sentences `sel` (B,) int64 of a device-resident CSR corpus."""
    B = sel.numel()
    dev = corpus_flat.device
    bare = torch.empty((B, width), dtype=torch.int64, device=dev)
    go_t = torch.empty((B, width + 1), dtype=torch.int64, device=dev)
    eos_t = torch.empty((B, width + 1), dtype=torch.int64, device=dev)
    ln = torch.empty((B,), dtype=torch.int64, device=dev)
    call('cvae_pack_batch_indexed', ptr(corpus_flat), ptr(corpus_offsets), ptr(_c(sel)), B, max_len, width, pad, go,
         eos, ptr(bare), ptr(go_t), ptr(eos_t), ptr(ln), stream())
    return bare, go_t, eos_t, ln


def sort_resort(length):
    """CV/utils/utils.py:18-21 on device -> (s_idx, res_idx) int64 tensors."""
    length = _c(length)
    B = length.numel()
    s_idx = torch.empty_like(length)
    res_idx = torch.empty_like(length)
    call('cvae_sort_resort', ptr(length), B, ptr(s_idx), ptr(res_idx), stream())
    return s_idx, res_idx


def gemm(A, B, transA=False, transB=False, bias=None, act=ACT_NONE, out=None, beta=0):
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = B.shape[0] if transB else B.shape[1]
    if out is None:
        out = _f((M, N), A)
    for t in (A, B, out):      # row-strided views are fine: the ABI takes leading dimensions
        assert t.dim() == 2 and t.stride(1) == 1
    call('cvae_gemm', int(transA), int(transB), M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0),
         out.data_ptr(), out.stride(0), ptr(bias), act, beta, stream())
    return out


# --------------------------------------------------------------------------- linear
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b)."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        x, W, b = _c(x), _c(W), _c(b)
        M, K = x.shape
        N = W.shape[0]
        y = _f((M, N), x)
        call('cvae_linear_forward', ptr(x), ptr(W), ptr(b), act, M, N, K, ptr(y), stream())
        ctx.act = act
        ctx.save_for_backward(x, W, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        dy = _c(dy)
        M, K = x.shape
        N = W.shape[0]
        dx = _f((M, K), x) if ctx.needs_input_grad[0] else None
        dW, db = torch.empty_like(W), _f((N,), x)
        scratch = _f((M, N), x) if ctx.act != ACT_NONE else None
        call('cvae_linear_backward', ptr(x), ptr(W), ptr(y), ptr(dy), ctx.act, M, N, K, ptr(dx), ptr(dW), ptr(db),
             ptr(scratch), stream())
        return dx, dW, db, None


def linear(x, W, b, act=ACT_NONE):
    lead = x.shape[:-1]
    y = LinearFn.apply(x.reshape(-1, x.shape[-1]), W, b, act)
    return y.reshape(*lead, W.shape[0])


# --------------------------------------------------------------------------- encoder
class EncoderFn(torch.autograd.Function):
    """CV/Modules/encoder.py:41-70 -> last states (B, 2H)."""

    @staticmethod
    def forward(ctx, cell, ids, length, mask, mask_rows, emb, *w):
        ids, length, emb = _c(ids), _c(length), _c(emb)
        w = [_c(t) for t in w]
        B, Lmax = ids.shape
        V, E = emb.shape
        H = w[1].shape[1]
        G = gates_of(cell)
        need_grad = any(ctx.needs_input_grad)
        X = _f((Lmax, B, E), emb)
        xproj = _f((2, Lmax, B, G * H), emb)
        hs = _f((2, Lmax + 1, B, H), emb)
        cs = _f((2, Lmax + 1, B, H), emb) if cell == LSTM else None
        gates = _f((2, Lmax, B, 4 * H), emb) if need_grad else None
        last = _f((B, 2 * H), emb)
        warr = ptr_array(w)
        call('cvae_encoder_forward', cell, B, Lmax, E, H, V, ptr(ids), ptr(length), ptr(emb), ptr(mask),
             ptr(mask_rows), warr, ptr(X), ptr(xproj), ptr(hs), ptr(cs), ptr(gates), ptr(last), stream())
        if need_grad:
            ctx.cell, ctx.dims = cell, (B, Lmax, E, H, V)
            ctx.save_for_backward(ids, length, mask, mask_rows, emb, X, hs, cs, gates, *w)
        return last

    @staticmethod
    def backward(ctx, dlast):
        ids, length, mask, mask_rows, emb, X, hs, cs, gates = ctx.saved_tensors[:9]
        w = list(ctx.saved_tensors[9:])
        B, Lmax, E, H, V = ctx.dims
        dlast = _c(dlast)
        dw = [torch.empty_like(t) for t in w]
        demb = torch.zeros_like(emb)
        scratch = _f((query('cvae_encoder_backward_scratch', ctx.cell, B, Lmax, E, H),), emb)
        warr, dwarr = ptr_array(w), ptr_array(dw)
        call('cvae_encoder_backward', ctx.cell, B, Lmax, E, H, V, ptr(ids), ptr(length), ptr(mask), ptr(mask_rows),
             warr, ptr(X), ptr(hs), ptr(cs), ptr(gates), ptr(dlast), dwarr, ptr(demb), ptr(scratch), stream())
        return (None, None, None, None, None, demb, *dw)


def encoder(cell, ids, length, mask, mask_rows, emb, w):
    """w: [W_ih, W_hh, b_ih, b_hh, W_ih_r, W_hh_r, b_ih_r, b_hh_r]."""
    return EncoderFn.apply(cell, ids, length, mask, mask_rows, emb, *w)


class EncoderPairFn(torch.autograd.Function):
    """Two encoders over the same embedding in lock-step (CC/Modules/cvae_couple.py's post and response
    encoders, CC/Modules/encoder.py:41-70 each): both are padded to the longer L -- steps past a row's
    length are carries, so the result does not change -- and all four directions share their launches."""

    @staticmethod
    def forward(ctx, cell, ids_a, len_a, mask_a, rows_a, ids_b, len_b, mask_b, rows_b, emb, *w):
        emb = _c(emb)
        w = [_c(t) for t in w]
        B = ids_a.shape[0]
        L = max(ids_a.shape[1], ids_b.shape[1])
        V, E = emb.shape
        H = w[1].shape[1]
        G = gates_of(cell)
        need_grad = any(ctx.needs_input_grad)

        def padded(ids, mask):
            l = ids.shape[1]
            if l < L:
                ids = torch.cat([ids, ids.new_zeros((B, L - l))], dim=1)
                if mask is not None:
                    mask = torch.cat([mask, mask.new_ones((L - l, B, E))], dim=0)
            return _c(ids), (_c(mask) if mask is not None else None)

        ids_a, mask_a = padded(ids_a, mask_a)
        ids_b, mask_b = padded(ids_b, mask_b)
        len_a, len_b = _c(len_a), _c(len_b)
        X = _f((2, L, B, E), emb)
        xproj = _f((2, 2, L, B, G * H), emb)
        hs = _f((2, 2, L + 1, B, H), emb)
        cs = _f((2, 2, L + 1, B, H), emb) if cell == LSTM else None
        gates = _f((2, 2, L, B, 4 * H), emb) if need_grad else None
        last = _f((2, B, 2 * H), emb)
        pair = lambda t: [t[0], t[1]] if t is not None else [None, None]
        keep = [ptr_array(x) for x in ([ids_a, ids_b], [len_a, len_b], [mask_a, mask_b], [rows_a, rows_b], w,
                                       pair(X), pair(xproj), pair(hs), pair(cs), pair(gates), pair(last))]
        call('cvae_encoder_forward_n', 2, cell, B, L, E, H, V, keep[0], keep[1], ptr(emb), keep[2], keep[3], keep[4],
             keep[5], keep[6], keep[7], keep[8], keep[9], keep[10], stream())
        if need_grad:
            ctx.cell, ctx.dims = cell, (B, L, E, H, V)
            ctx.save_for_backward(ids_a, ids_b, len_a, len_b, mask_a, mask_b, rows_a, rows_b, emb, X, hs, cs, gates, *w)
        return last[0], last[1]

    @staticmethod
    def backward(ctx, dlast_a, dlast_b):
        ids_a, ids_b, len_a, len_b, mask_a, mask_b, rows_a, rows_b, emb, X, hs, cs, gates = ctx.saved_tensors[:13]
        w = list(ctx.saved_tensors[13:])
        B, L, E, H, V = ctx.dims
        zero = lambda d: _c(d) if d is not None else torch.zeros((B, 2 * H), device=emb.device, dtype=emb.dtype)
        dlast = [zero(dlast_a), zero(dlast_b)]
        dw = [torch.empty_like(t) for t in w]
        demb = torch.zeros_like(emb)
        n = query('cvae_encoder_backward_scratch', ctx.cell, B, L, E, H)
        scratch = _f((2, n), emb)
        pair = lambda t: [t[0], t[1]] if t is not None else [None, None]
        keep = [ptr_array(x) for x in ([ids_a, ids_b], [len_a, len_b], [mask_a, mask_b], [rows_a, rows_b], w,
                                       pair(X), pair(hs), pair(cs), pair(gates), dlast, dw, pair(scratch))]
        call('cvae_encoder_backward_n', 2, ctx.cell, B, L, E, H, V, keep[0], keep[1], keep[2], keep[3], keep[4],
             keep[5], keep[6], keep[7], keep[8], keep[9], keep[10], ptr(demb), keep[11], stream())
        return (None,) * 9 + (demb,) + tuple(dw)


def encoder_pair(cell, ids_a, len_a, mask_a, rows_a, w_a, ids_b, len_b, mask_b, rows_b, w_b, emb):
    """-> (last_a, last_b), each (B, 2H); w_a / w_b as in `encoder`."""
    return EncoderPairFn.apply(cell, ids_a, len_a, mask_a, rows_a, ids_b, len_b, mask_b, rows_b, emb, *w_a, *w_b)


# --------------------------------------------------------------------------- decoder
class DecoderFn(torch.autograd.Function):
    """Teacher-forced decoder up to `final` (T,R,E): CV/Modules/vae_couple.py:86-115."""

    @staticmethod
    def forward(ctx, cell, train, init, go, mask_in, mask_out, emb, *w):
        init, go, emb = _c(init), _c(go), _c(emb)
        w = [_c(t) for t in w]
        R, E = init.shape
        B, T = go.shape
        V = emb.shape[0]
        Hd = w[1].shape[1]
        G = gates_of(cell)
        big = R > 256
        hs = _f((T + 2, R, Hd), emb)
        cs = _f((T + 2, R, Hd), emb) if cell == LSTM else None
        final = _f((T, R, E), emb)
        hp = _f((R, G * Hd), emb) if big else None
        if train:
            X = _f((T + 1, R, 2 * E), emb)
            xproj = _f((T + 1, R, G * Hd), emb)
            xadd = None
            gates = _f((T + 1, R, 4 * Hd), emb)
            hdrop = _f((T, R, Hd), emb) if mask_out is not None else None
        else:
            X = _f((T, B, E), emb)
            xproj = _f((T, B, G * Hd), emb)
            xadd = _f((2, R, G * Hd), emb)
            gates = hdrop = None
        warr = ptr_array(w)
        call('cvae_decoder_forward', cell, int(train), B, R, T, E, Hd, V, ptr(init), ptr(go), ptr(emb), ptr(mask_in),
             ptr(mask_out), warr, ptr(X), ptr(xproj), ptr(xadd), ptr(hp), ptr(hs), ptr(cs), ptr(gates), ptr(hdrop),
             ptr(final), stream())
        if train:
            ctx.cell, ctx.dims = cell, (B, T, E, Hd, V)
            ctx.save_for_backward(go, mask_in, mask_out, emb, X, hs, cs, gates, hdrop, final, *w)
        return final

    @staticmethod
    def backward(ctx, dfinal):
        go, mask_in, mask_out, emb, X, hs, cs, gates, hdrop, final = ctx.saved_tensors[:10]
        w = list(ctx.saved_tensors[10:])
        B, T, E, Hd, V = ctx.dims
        dfinal = _c(dfinal)
        dinit = _f((B, E), emb)
        dw = [torch.empty_like(t) for t in w]
        demb = torch.zeros_like(emb)
        scratch = _f((query('cvae_decoder_backward_scratch', ctx.cell, B, T, E, Hd),), emb)
        warr, dwarr = ptr_array(w), ptr_array(dw)
        call('cvae_decoder_backward', ctx.cell, B, T, E, Hd, V, ptr(go), ptr(mask_in), ptr(mask_out), warr, ptr(X),
             ptr(hs), ptr(cs), ptr(gates), ptr(hdrop), ptr(final), ptr(dfinal), ptr(dinit), dwarr, ptr(demb),
             ptr(scratch), stream())
        return (None, None, dinit, None, None, None, demb, *dw)


class DecoderNoPrefixFn(torch.autograd.Function):
    """The LM's decoder (CV/Modules/lm.py:78-100): the training branch without the prefix step, h_0 = 0.
    Same buffers as DecoderFn(train=True); masks may be None (evaluation)."""

    @staticmethod
    def forward(ctx, cell, init, go, mask_in, mask_out, emb, *w):
        init, go, emb = _c(init), _c(go), _c(emb)
        w = [_c(t) for t in w]
        B, E = init.shape
        T = go.shape[1]
        V, Hd, G = emb.shape[0], w[1].shape[1], gates_of(cell)
        hs = _f((T + 2, B, Hd), emb)
        cs = _f((T + 2, B, Hd), emb) if cell == LSTM else None
        final = _f((T, B, E), emb)
        X = _f((T + 1, B, 2 * E), emb)
        xproj = _f((T + 1, B, G * Hd), emb)
        gates = _f((T + 1, B, 4 * Hd), emb)
        hdrop = _f((T, B, Hd), emb) if mask_out is not None else None
        warr = ptr_array(w)
        call('cvae_decoder_forward_noprefix', cell, B, T, E, Hd, V, ptr(init), ptr(go), ptr(emb), ptr(mask_in),
             ptr(mask_out), warr, ptr(X), ptr(xproj), ptr(hs), ptr(cs), ptr(gates), ptr(hdrop), ptr(final), stream())
        ctx.cell, ctx.dims = cell, (B, T, E, Hd, V)
        ctx.save_for_backward(go, mask_in, mask_out, emb, X, hs, cs, gates, hdrop, final, *w)
        return final

    @staticmethod
    def backward(ctx, dfinal):
        go, mask_in, mask_out, emb, X, hs, cs, gates, hdrop, final = ctx.saved_tensors[:10]
        w = list(ctx.saved_tensors[10:])
        B, T, E, Hd, V = ctx.dims
        dfinal = _c(dfinal)
        dinit = _f((B, E), emb)
        dw = [torch.empty_like(t) for t in w]
        demb = torch.zeros_like(emb)
        scratch = _f((query('cvae_decoder_backward_scratch', ctx.cell, B, T, E, Hd),), emb)
        warr, dwarr = ptr_array(w), ptr_array(dw)
        call('cvae_decoder_backward_noprefix', ctx.cell, B, T, E, Hd, V, ptr(go), ptr(mask_in), ptr(mask_out), warr,
             ptr(X), ptr(hs), ptr(cs), ptr(gates), ptr(hdrop), ptr(final), ptr(dfinal), ptr(dinit), dwarr, ptr(demb),
             ptr(scratch), stream())
        return (None, None, None, None, None, demb, *dw)


def decoder_noprefix(cell, go, mask_in, mask_out, emb, w):
    """LM decoder: w = [W_ih (G*Hd, E), W_hh, b_ih, b_hh, W_c, b_c]; the input has no conditioning half, which
    the kernels see as init = 0 and W_ih = [W_ih | 0] (autograd slices the gradient back).  mask_in (T,B,E)."""
    B, T = go.shape
    E = emb.shape[1]
    W_ih = torch.cat([w[0], torch.zeros_like(w[0])], 1)
    init = torch.zeros((B, E), dtype=torch.float32, device=emb.device)
    if mask_in is not None:
        mask_in = torch.cat([mask_in, torch.ones_like(mask_in)], 2)
    return DecoderNoPrefixFn.apply(cell, init, go, mask_in, mask_out, emb, W_ih, *w[1:])


def decoder(cell, train, init, go, mask_in, mask_out, emb, w):
    """w: [W_ih, W_hh, b_ih, b_hh, W_c, b_c] -> final (T,R,E)."""
    if not train:
        with torch.no_grad():
            return DecoderFn.apply(cell, False, init, go, None, None, emb, *w)
    return DecoderFn.apply(cell, True, init, go, mask_in, mask_out, emb, *w)


def decoder_sorted(cell, init, go, live_sent, emb, w):
    """IW-eval decoder on sentence-major rows (cvae_decoder_forward_sorted): init (B*K,E) with row
    b*K+k = sample k of sentence b, go (B,T) ordered by non-increasing length, live_sent = host
    int32 (T,) count of sentences still inside their target at step t.  -> final (T,B*K,E) whose
    rows past live_sent[t]*K are zero-filled and never computed."""
    with torch.no_grad():
        init, go, emb = _c(init), _c(go), _c(emb)
        w = [_c(t) for t in w]
        R, E = init.shape
        B, T = go.shape
        K = R // B
        assert R == B * K and live_sent.dtype == torch.int32 and live_sent.device.type == 'cpu' and \
            live_sent.numel() == T and live_sent.is_contiguous()
        V = emb.shape[0]
        Hd = w[1].shape[1]
        G = gates_of(cell)
        hs = _f((T + 2, R, Hd), emb)
        cs = _f((T + 2, R, Hd), emb) if cell == LSTM else None
        final = torch.zeros((T, R, E), dtype=torch.float32, device=emb.device)
        hp = _f((R, G * Hd), emb)
        X, xproj, xadd = _f((T, B, E), emb), _f((T, B, G * Hd), emb), _f((2, R, G * Hd), emb)
        warr = ptr_array(w)
        call('cvae_decoder_forward_sorted', cell, B, K, T, E, Hd, V, ptr(init), ptr(go), ptr(emb),
             live_sent.data_ptr(), warr, ptr(X), ptr(xproj), ptr(xadd), ptr(hp), ptr(hs), ptr(cs), ptr(final),
             stream())
        return final


def decoder_greedy(cell, init, first, T, emb, w, W_v, b_v, mask_id=3):
    """Greedy decoding (CV/Modules/decoder.py:176-201): init (B,E), first (B,) <go> ids, w = [W_ih, W_hh, b_ih,
    b_hh, W_c, b_c], ToVocab (W_v, b_v); entry mask_id (<unk>) never chosen.  -> ids (B,T) int64."""
    with torch.no_grad():
        init, first, emb, W_v, b_v = _c(init), _c(first), _c(emb), _c(W_v), _c(b_v)
        w = [_c(t) for t in w]
        B, E = init.shape
        V, Hd = emb.shape[0], w[1].shape[1]
        ws = _f((query('cvae_decoder_greedy_scratch', cell, B, T, E, Hd, V),), emb)
        out = torch.empty((B, T), dtype=torch.int64, device=emb.device)
        warr = ptr_array(w)
        call('cvae_decoder_greedy', cell, B, T, E, Hd, V, ptr(init), ptr(first), ptr(emb), warr, ptr(W_v), ptr(b_v),
             int(mask_id), ptr(ws), ptr(out), stream())
        return out


def decoder_beam(cell, init, first, T, K, emb, w, W_v, b_v, mask_id=3, return_nll=False):
    """Beam search (CV/Modules/decoder.py:114-168) with K beams; arguments as `decoder_greedy`.
    -> ids (B,T) int64 of the best beam [, its accumulated negative log-likelihood (B,)]."""
    with torch.no_grad():
        init, first, emb, W_v, b_v = _c(init), _c(first), _c(emb), _c(W_v), _c(b_v)
        w = [_c(t) for t in w]
        B, E = init.shape
        V, Hd = emb.shape[0], w[1].shape[1]
        ws = _f((query('cvae_decoder_beam_scratch', cell, B, K, T, E, Hd, V),), emb)
        out = torch.empty((B, T), dtype=torch.int64, device=emb.device)
        nll = _f((B,), emb)
        warr = ptr_array(w)
        call('cvae_decoder_beam', cell, B, K, T, E, Hd, V, ptr(init), ptr(first), ptr(emb), warr, ptr(W_v), ptr(b_v),
             int(mask_id), ptr(ws), ptr(out), ptr(nll), stream())
        return (out, nll) if return_nll else out


class DecoderPairFn(torch.autograd.Function):
    """Two training decoders of identical shape (Decoder / DecoderCouple) advanced in lock-step:
    every timestep of both recurrences is one launch (cvae_decoder_forward_n).
    Inputs: cell, go, emb, then for each decoder (init, mask_in, mask_out, 6 weights)."""

    @staticmethod
    def forward(ctx, cell, go, emb, *rest):
        nd = len(rest) // 9
        go, emb = _c(go), _c(emb)
        per = [[_c(t) for t in rest[9 * d:9 * d + 9]] for d in range(nd)]
        B, T = go.shape
        V, E = emb.shape
        Hd = per[0][4].shape[1]
        G = gates_of(cell)
        bufs = []
        for d in range(nd):
            mo = per[d][2]
            bufs.append(dict(X=_f((T + 1, B, 2 * E), emb), xproj=_f((T + 1, B, G * Hd), emb), hs=_f((T + 2, B, Hd), emb),
                             cs=_f((T + 2, B, Hd), emb) if cell == LSTM else None, gates=_f((T + 1, B, 4 * Hd), emb),
                             hdrop=_f((T, B, Hd), emb) if mo is not None else None, final=_f((T, B, E), emb)))
        w_all = [t for d in range(nd) for t in per[d][3:]]
        arrs = dict(init=ptr_array([per[d][0] for d in range(nd)]), mi=ptr_array([per[d][1] for d in range(nd)]),
                    mo=ptr_array([per[d][2] for d in range(nd)]), w=ptr_array(w_all))
        for k in ('X', 'xproj', 'hs', 'cs', 'gates', 'hdrop', 'final'):
            arrs[k] = ptr_array([b[k] for b in bufs])
        call('cvae_decoder_forward_n', nd, cell, B, T, E, Hd, V, arrs['init'], ptr(go), ptr(emb), arrs['mi'], arrs['mo'],
             arrs['w'], arrs['X'], arrs['xproj'], arrs['hs'], arrs['cs'], arrs['gates'], arrs['hdrop'], arrs['final'],
             stream())
        ctx.cell, ctx.dims, ctx.nd = cell, (B, T, E, Hd, V), nd
        saved = [go, emb]
        for d in range(nd):
            saved += [per[d][1], per[d][2]] + per[d][3:] + [bufs[d][k] for k in ('X', 'hs', 'cs', 'gates', 'hdrop', 'final')]
        ctx.save_for_backward(*saved)
        return tuple(b['final'] for b in bufs)

    @staticmethod
    def backward(ctx, *dfinals):
        sv = ctx.saved_tensors
        go, emb = sv[0], sv[1]
        nd = ctx.nd
        B, T, E, Hd, V = ctx.dims
        per = [sv[2 + 14 * d: 2 + 14 * (d + 1)] for d in range(nd)]   # mi, mo, 6 w, X, hs, cs, gates, hdrop, final
        dfin = [_c(dfinals[d]) if dfinals[d] is not None else torch.zeros_like(per[d][13]) for d in range(nd)]
        dinit = [_f((B, E), emb) for _ in range(nd)]
        dw = [[torch.empty_like(t) for t in per[d][2:8]] for d in range(nd)]
        demb = torch.zeros_like(emb)
        nscr = query('cvae_decoder_backward_scratch', ctx.cell, B, T, E, Hd)
        scr = [_f((nscr,), emb) for _ in range(nd)]
        pa = lambda idx: ptr_array([per[d][idx] for d in range(nd)])
        w_all = ptr_array([t for d in range(nd) for t in per[d][2:8]])
        dw_all = ptr_array([t for d in range(nd) for t in dw[d]])
        arr_mi, arr_mo, arr_X, arr_hs, arr_cs, arr_g, arr_hd, arr_fin = (pa(0), pa(1), pa(8), pa(9), pa(10), pa(11),
                                                                          pa(12), pa(13))
        arr_df, arr_di, arr_scr = ptr_array(dfin), ptr_array(dinit), ptr_array(scr)
        call('cvae_decoder_backward_n', nd, ctx.cell, B, T, E, Hd, V, ptr(go), arr_mi, arr_mo, w_all, arr_X, arr_hs,
             arr_cs, arr_g, arr_hd, arr_fin, arr_df, arr_di, dw_all, ptr(demb), arr_scr, stream())
        out = [None, None, demb]
        for d in range(nd):
            out += [dinit[d], None, None] + dw[d]
        return tuple(out)


def decoder_pair(cell, go, emb, specs):
    """specs: list of (init, mask_in, mask_out, [W_ih, W_hh, b_ih, b_hh, W_c, b_c]) -> list of final (T,B,E)."""
    flat = []
    for init, mi, mo, w in specs:
        flat += [init, mi, mo] + list(w)
    return DecoderPairFn.apply(cell, go, emb, *flat)


# --------------------------------------------------------------------------- projection + CE
def targets_time_major(eos, R):
    eos = _c(eos)
    B, T = eos.shape
    out = torch.empty((T * R,), dtype=torch.int64, device=eos.device)
    call('cvae_targets_time_major', ptr(eos), B, T, R, ptr(out), stream())
    return out


def targets_sentence_major(eos, K):
    eos = _c(eos)
    B, T = eos.shape
    out = torch.empty((T * B * K,), dtype=torch.int64, device=eos.device)
    call('cvae_targets_sentence_major', ptr(eos), B, T, K, ptr(out), stream())
    return out


class ProjCEFn(torch.autograd.Function):
    """ToVocab + SeqLoss(keep_batch=True): per-sentence summed CE (R,), logits never stored."""

    @staticmethod
    def forward(ctx, final, W, b, tgt_tm, pad):
        final, W, b = _c(final), _c(W), _c(b)
        T, R, E = final.shape
        V = W.shape[0]
        N = T * R
        loss_row, lse, sent = _f((N,), final), _f((N,), final), _f((R,), final)
        ws = _f((query('cvae_projce_workspace', N, E, V),), final)
        # training step: the forward pass also saves the logits of the live rows and the backward pass
        # turns them into the gradient in place instead of recomputing the product (settings.ce_save_logits)
        logits = None
        if any(ctx.needs_input_grad) and settings.ce_save_logits:
            n = query('cvae_projce_logits_workspace', N, E, V)
            if 0 < n * 4 <= settings.ce_save_logits_max_bytes:
                logits = _f((n,), final)
        if logits is not None:
            call('cvae_projce_forward_train', ptr(final), ptr(W), ptr(b), ptr(tgt_tm), pad, 1, T, R, E, V,
                 ptr(loss_row), ptr(lse), ptr(sent), ptr(ws), ptr(logits), stream())
        else:
            call('cvae_projce_forward', ptr(final), ptr(W), ptr(b), ptr(tgt_tm), pad, 1, T, R, E, V, ptr(loss_row),
                 ptr(lse), ptr(sent), ptr(ws), stream())
        ctx.pad = pad
        ctx.logits = logits
        ctx.save_for_backward(final, W, b, tgt_tm, lse)
        return sent

    @staticmethod
    def backward(ctx, gsent):
        final, W, b, tgt_tm, lse = ctx.saved_tensors
        T, R, E = final.shape
        V = W.shape[0]
        gsent = _c(gsent)
        dfinal, dW, db = torch.empty_like(final), torch.empty_like(W), torch.empty_like(b)
        ws = _f((query('cvae_projce_backward_workspace', T * R, E, V),), final)
        logits, ctx.logits = ctx.logits, None            # consumed (overwritten with the gradient)
        if logits is not None:
            call('cvae_projce_backward_saved', ptr(final), ptr(W), ptr(b), ptr(tgt_tm), ctx.pad, ptr(lse), ptr(gsent),
                 1, T, R, E, V, ptr(dfinal), ptr(dW), ptr(db), ptr(ws), ptr(logits), stream())
        else:
            call('cvae_projce_backward', ptr(final), ptr(W), ptr(b), ptr(tgt_tm), ctx.pad, ptr(lse), ptr(gsent), 1, T,
                 R, E, V, ptr(dfinal), ptr(dW), ptr(db), ptr(ws), stream())
        return dfinal, dW, db, None, None


def proj_ce(final, W, b, eos, pad, cache=None):
    """final (T,R,E), eos (B,T) with R % B == 0 -> (R,).  `cache`: a dict shared by the losses of one forward pass
    (the two decoders of a training step read the same targets): the time-major targets are built once."""
    R = final.shape[1]
    if cache is None:
        tgt = targets_time_major(eos, R)
    else:
        key = ('tgt_tm', eos.data_ptr(), tuple(eos.shape), tuple(eos.stride()), eos._version, R)
        hit = cache.get(key)
        if hit is None:
            hit = cache[key] = (targets_time_major(eos, R), eos)      # holds `eos`: its storage stays pinned
        tgt = hit[0]
    return ProjCEFn.apply(final, W, b, tgt, pad)


class ProjCEPairFn(torch.autograd.Function):
    """ProjCEFn for the nd decoders of a training step over ONE set of targets (cvae_projce_forward_train_n /
    cvae_projce_backward_saved_n): inputs pad, tgt_tm, then (final, W, b) per decoder -> nd per-sentence sums."""

    @staticmethod
    def forward(ctx, pad, tgt_tm, *rest):
        nd = len(rest) // 3
        per = [[_c(t) for t in rest[3 * d:3 * d + 3]] for d in range(nd)]
        T, R, E = per[0][0].shape
        V = per[0][1].shape[0]
        N = T * R
        ref = per[0][0]
        nws, nlog = query('cvae_projce_workspace', N, E, V), query('cvae_projce_logits_workspace', N, E, V)
        loss_row, lse, sent = [[_f(shape, ref) for _ in range(nd)] for shape in ((N,), (N,), (R,))]
        ws, logits = [_f((nws,), ref) for _ in range(nd)], [_f((nlog,), ref) for _ in range(nd)]
        arrs = [ptr_array([per[d][k] for d in range(nd)]) for k in range(3)]
        a_loss, a_lse, a_sent, a_ws, a_log = (ptr_array(x) for x in (loss_row, lse, sent, ws, logits))
        call('cvae_projce_forward_train_n', nd, arrs[0], arrs[1], arrs[2], ptr(tgt_tm), pad, T, R, E, V, a_loss, a_lse,
             a_sent, a_ws, a_log, stream())
        ctx.pad, ctx.nd, ctx.logits = pad, nd, logits
        ctx.save_for_backward(tgt_tm, *[t for d in range(nd) for t in per[d]], *lse)
        return tuple(sent)

    @staticmethod
    def backward(ctx, *gsent):
        nd = ctx.nd
        sv = ctx.saved_tensors
        tgt_tm, per, lse = sv[0], [sv[1 + 3 * d:4 + 3 * d] for d in range(nd)], sv[1 + 3 * nd:]
        T, R, E = per[0][0].shape
        V = per[0][1].shape[0]
        g = [_c(gsent[d]) if gsent[d] is not None else torch.zeros((R,), dtype=torch.float32, device=tgt_tm.device)
             for d in range(nd)]
        grads = [[torch.empty_like(t) for t in per[d]] for d in range(nd)]
        logits, ctx.logits = ctx.logits, None             # consumed (overwritten with the gradient)
        if logits is None:                                # a second backward over a retained graph: recompute
            ws = _f((query('cvae_projce_backward_workspace', T * R, E, V),), per[0][0])
            for d in range(nd):
                call('cvae_projce_backward', ptr(per[d][0]), ptr(per[d][1]), ptr(per[d][2]), ptr(tgt_tm), ctx.pad,
                     ptr(lse[d]), ptr(g[d]), 1, T, R, E, V, ptr(grads[d][0]), ptr(grads[d][1]), ptr(grads[d][2]), ptr(ws),
                     stream())
            return (None, None) + tuple(t for d in range(nd) for t in grads[d])
        ws = _f((query('cvae_projce_backward_saved_n_workspace', E),), per[0][0])
        arrs = [ptr_array([per[d][k] for d in range(nd)]) for k in range(3)]
        garr = [ptr_array([grads[d][k] for d in range(nd)]) for k in range(3)]
        a_lse, a_g, a_log = ptr_array(list(lse)), ptr_array(g), ptr_array(logits)
        call('cvae_projce_backward_saved_n', nd, arrs[0], arrs[1], arrs[2], ptr(tgt_tm), ctx.pad, a_lse, a_g, T, R, E, V,
             garr[0], garr[1], garr[2], ptr(ws), a_log, stream())
        return (None, None) + tuple(t for d in range(nd) for t in grads[d])


def proj_ce_pair_ok(finals, Ws):
    """The paired path covers a training step whose decoders have one shape and whose logits fit the save budget."""
    if not (settings.ce_pair and settings.ce_save_logits and torch.is_grad_enabled()):
        return False
    T, R, E = finals[0].shape
    V = Ws[0].shape[0]
    if any(f.shape != finals[0].shape for f in finals) or any(w.shape != Ws[0].shape for w in Ws):
        return False
    n = query('cvae_projce_logits_workspace', T * R, E, V)
    return 0 < n * 4 <= settings.ce_save_logits_max_bytes and T * R > 0


def proj_ce_pair(finals, Ws, bs, eos, pad, cache=None):
    """[final (T,R,E)] * nd, eos (B,T) -> nd per-sentence sums (R,), as nd proj_ce calls."""
    R = finals[0].shape[1]
    if cache is None:
        tgt = targets_time_major(eos, R)
    else:
        key = ('tgt_tm', eos.data_ptr(), tuple(eos.shape), tuple(eos.stride()), eos._version, R)
        hit = cache.get(key)
        if hit is None:
            hit = cache[key] = (targets_time_major(eos, R), eos)
        tgt = hit[0]
    flat = [t for d in range(len(finals)) for t in (finals[d], Ws[d], bs[d])]
    return ProjCEPairFn.apply(pad, tgt, *flat)


def proj_ce_sorted(final, W, b, eos, pad):
    """final (T,B*K,E) on sentence-major rows, eos (B,T) -> (B*K,)."""
    B = eos.shape[0]
    tgt = targets_sentence_major(eos, final.shape[1] // B)
    return ProjCEFn.apply(final, W, b, tgt, pad)


def logits_bt(final, W, b):
    """Materialised (B,T,V) logits for callers that index them (not the hot path)."""
    final, W, b = _c(final), _c(W), _c(b)
    T, B, E = final.shape
    V = W.shape[0]
    tmp, out = _f((T * B, V), final), _f((B, T, V), final)
    call('cvae_logits', ptr(final), ptr(W), ptr(b), T, B, E, V, ptr(tmp), ptr(out), stream())
    return out


class LossAssembleFn(torch.autograd.Function):
    """tot = sum_i coef[i] * mean(term_i) and the means themselves (CV/Experiment.py:753-757), one launch.
    -> (tot (1,), means (n,)); only tot carries a gradient."""

    @staticmethod
    def forward(ctx, coef, *terms):
        terms = [_c(t) for t in terms]
        n, B = len(terms), terms[0].shape[0]
        assert all(t.shape == (B,) for t in terms)
        out = _f((n + 1,), terms[0])
        carr = (ctypes.c_float * n)(*[float(c) for c in coef])
        tarr = ptr_array(terms)
        call('cvae_loss_assemble', n, B, tarr, carr, ptr(out), stream())
        ctx.coef, ctx.n, ctx.B, ctx.like = carr, n, B, terms[0]
        tot, means = out[:1], out[1:]
        ctx.mark_non_differentiable(means)
        return tot, means

    @staticmethod
    def backward(ctx, gtot, _gmeans):
        gtot = _c(gtot)
        d = [_f((ctx.B,), ctx.like) for _ in range(ctx.n)]
        darr = ptr_array(d)
        call('cvae_loss_assemble_backward', ctx.n, ctx.B, ptr(gtot), ctx.coef, darr, stream())
        return (None,) + tuple(d)


def loss_assemble(terms, coef):
    """terms: list of (B,) tensors, coef: python floats -> (tot (1,), means (n,))."""
    return LossAssembleFn.apply(tuple(coef), *terms)


# --------------------------------------------------------------------------- latent space
class ReparamFn(torch.autograd.Function):
    """z = mu + sigma * eps  (Normal.rsample, CV/Modules/vae_couple.py:801)."""

    @staticmethod
    def forward(ctx, mu, sigma, eps):
        mu, sigma, eps = _c(mu), _c(sigma), _c(eps)
        B, Z = mu.shape
        z = torch.empty_like(mu)
        call('cvae_reparam_forward', ptr(mu), ptr(sigma), ptr(eps), 1, B, Z, ptr(z), stream())
        ctx.save_for_backward(eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        eps, = ctx.saved_tensors
        dz = _c(dz)
        B, Z = dz.shape
        dmu, dsigma = torch.zeros_like(dz), torch.zeros_like(dz)
        call('cvae_reparam_backward', ptr(dz), ptr(eps), B, Z, ptr(dmu), ptr(dsigma), stream())
        return dmu, dsigma, None


def rsample_n(mu, sigma, eps):
    """eps (n,B,Z) -> z (n,B,Z), no autograd (eval)."""
    mu, sigma, eps = _c(mu), _c(sigma), _c(eps)
    n, B, Z = eps.shape
    z = torch.empty_like(eps)
    call('cvae_reparam_forward', ptr(mu), ptr(sigma), ptr(eps), n, B, Z, ptr(z), stream())
    return z


class KLFn(torch.autograd.Function):
    """kl_divergence(N(mu,sigma) || prior).sum(1) -> (B,)."""

    @staticmethod
    def forward(ctx, mu, sigma, mu_p, sigma_p):
        mu, sigma, mu_p, sigma_p = _c(mu), _c(sigma), _c(mu_p), _c(sigma_p)
        B, Z = mu.shape
        kl = _f((B,), mu)
        call('cvae_kl_forward', ptr(mu), ptr(sigma), ptr(mu_p), ptr(sigma_p), B, Z, ptr(kl), stream())
        ctx.save_for_backward(mu, sigma, mu_p, sigma_p)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        mu, sigma, mu_p, sigma_p = ctx.saved_tensors
        gkl = _c(gkl)
        B, Z = mu.shape
        dmu, dsigma = torch.empty_like(mu), torch.empty_like(mu)
        dmu_p = torch.empty_like(mu) if mu_p is not None else None
        dsigma_p = torch.empty_like(mu) if mu_p is not None else None
        call('cvae_kl_backward', ptr(mu), ptr(sigma), ptr(mu_p), ptr(sigma_p), ptr(gkl), B, Z, ptr(dmu), ptr(dsigma),
             ptr(dmu_p), ptr(dsigma_p), stream())
        return dmu, dsigma, dmu_p, dsigma_p


def kl_div(mu, sigma, mu_p=None, sigma_p=None):
    return KLFn.apply(mu, sigma, mu_p, sigma_p)


class KLFreeBitsFn(torch.autograd.Function):
    """clamp(kl_divergence(N(mu,sigma) || N(0,1)), min=floor).sum(1) -> (B,)  (CV/Experiment.py:1137-1142)."""

    @staticmethod
    def forward(ctx, mu, sigma, floor):
        mu, sigma = _c(mu), _c(sigma)
        B, Z = mu.shape
        kl = _f((B,), mu)
        call('cvae_kl_fb_forward', ptr(mu), ptr(sigma), B, Z, float(floor), ptr(kl), stream())
        ctx.floor = float(floor)
        ctx.save_for_backward(mu, sigma)
        return kl

    @staticmethod
    def backward(ctx, gkl):
        mu, sigma = ctx.saved_tensors
        B, Z = mu.shape
        dmu, dsigma = torch.empty_like(mu), torch.empty_like(mu)
        call('cvae_kl_fb_backward', ptr(mu), ptr(sigma), ptr(_c(gkl)), B, Z, ctx.floor, ptr(dmu), ptr(dsigma), stream())
        return dmu, dsigma, None


def kl_free_bits(mu, sigma, floor):
    return KLFreeBitsFn.apply(mu, sigma, floor)


class CoupleFn(torch.autograd.Function):
    """euclidean(init, init_c.detach()) [+ IMQ]: CV/Experiment.py:742-751,1892-1923 -> (B,)."""

    @staticmethod
    def forward(ctx, init, init_c, Z, euclid):
        init, init_c = _c(init), _c(init_c.detach())
        B, E = init.shape
        d, loss = _f((B,), init), _f((B,), init)
        call('cvae_couple_forward', ptr(init), ptr(init_c), B, E, Z, int(euclid), ptr(d), ptr(loss), stream())
        ctx.Z, ctx.euclid = Z, int(euclid)
        ctx.save_for_backward(init, init_c, d)
        return loss

    @staticmethod
    def backward(ctx, g):
        init, init_c, d = ctx.saved_tensors
        g = _c(g)
        B, E = init.shape
        dinit = torch.empty_like(init)
        call('cvae_couple_backward', ptr(init), ptr(init_c), ptr(d), ptr(g), B, E, ctx.Z, ctx.euclid, ptr(dinit),
             stream())
        return dinit, None, None, None


def couple_loss(init, init_c, latent_dim, euclidean_dist):
    return CoupleFn.apply(init, init_c, latent_dim, euclidean_dist)


# --------------------------------------------------------------------------- IW / MI
def iw_reduce(z, xent, mu, sigma, mu_p=None, sigma_p=None):
    z, xent, mu, sigma, mu_p, sigma_p = _c(z), _c(xent), _c(mu), _c(sigma), _c(mu_p), _c(sigma_p)
    K, B, Z = z.shape
    nll, kl, xm = _f((B,), z), _f((B,), z), _f((B,), z)
    call('cvae_iw_reduce', ptr(z), ptr(xent), ptr(mu), ptr(sigma), ptr(mu_p), ptr(sigma_p), K, B, Z, ptr(nll), ptr(kl),
         ptr(xm), stream())
    return xm, nll, kl


def mi(z, mu, sigma):
    z, mu, sigma = _c(z), _c(mu), _c(sigma)
    K, B, Z = z.shape
    out = _f((B,), z)
    ws = _f((2 * K * B,), z)
    call('cvae_mi', ptr(z), ptr(mu), ptr(sigma), K, B, Z, ptr(ws), ptr(out), stream())
    return out


def mi_static(z, mu, sigma, out=None):
    """ops.mi without a workspace allocation (cvae_mi_static: the library's static scratch array, K*B <= 131072).
    With `out` given the call allocates nothing at all."""
    z, mu, sigma = _c(z), _c(mu), _c(sigma)
    K, B, Z = z.shape
    if out is None:
        out = _f((B,), z)
    call('cvae_mi_static', ptr(z), ptr(mu), ptr(sigma), K, B, Z, ptr(out), stream())
    return out


# --------------------------------------------------------------------------- optimizer
def clip_adam(params, grads, m, v, alias_off, alias_n, alias_mult, max_norm, lr, betas, eps, step_other, step_alias,
              sumsq, norm_out, late=None):
    """late: None or (offset, n, steps_taken) of the parameter group with its own step count; steps_taken < 0 = the
    group has no gradient this iteration (cvae_clip_adam_late)."""
    if late is None:
        call('cvae_clip_adam', ptr(params), ptr(grads), ptr(m), ptr(v), params.numel(), alias_off, alias_n, alias_mult,
             max_norm, lr, betas[0], betas[1], eps, step_other, step_alias, ptr(sumsq), ptr(norm_out), stream())
    else:
        call('cvae_clip_adam_late', ptr(params), ptr(grads), ptr(m), ptr(v), params.numel(), alias_off, alias_n,
             alias_mult, max_norm, lr, betas[0], betas[1], eps, step_other, step_alias, int(late[0]), int(late[1]),
             int(late[2]), ptr(sumsq), ptr(norm_out), stream())
