"""Harness: the semantics of the reference's Experiment.train_epoch / language_modeling /
estimate_mi / evaluate_dialogue for the hot path (loss assembly, lambda_KL schedule,
clip + Adam with the duplicate-embedding behaviour, corpus reductions), plus what the
reference does not have: data-parallel training and sharded evaluation over RCCL.

Reference: CV/Experiment.py:509-558,578-621,639-775,1323-1396,1453-1503;
           CC/Experiment.py:311-428,538-621.
"""
import collections
import math

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, dist_utils, ops
from .modules import MODEL_CLASSES
from .settings import settings


class HParams(object):
    """The values config.Config carries for the hot path (CV/config.py:40-84, CC/config.py:22-66)."""

    def __init__(self, model='vae-couple', cell='gru', V=10002, E=200, H=128, Z=32, dropout=0.2, lr=1e-3,
                 beta1=0.9, beta2=0.999, clip_norm=30.0, kl_annealing=True, annealing_start=2000,
                 annealing_step=40000, lambda_couple=0.1, lambda_seq_couple=1.0, euclidean_dist=False,
                 max_len=200, batch_size=32, pad_id=0, beta=0.8, cyc_M=4, cyc_R=0.5, lambda_fb=6.0, lambda_BoW=0.0):
        """model: vae | vae-couple | cvae | cvae-couple, or a sibling on the same modules:
        dae, beta-vae[-couple] (lambda_KL capped at beta), cyc-anneal-vae[-couple] (cyclic schedule, cyc_M
        cycles with ramp fraction cyc_R), surprising-fix[-couple] (free bits lambda_fb)."""
        self.__dict__.update(locals())
        del self.__dict__['self']

    def is_couple(self):
        return self.model.endswith('couple')

    def is_dialog(self):
        return self.model.startswith('cvae')


def lambda_kl(iter_num, hp):
    """Experiment.update_lambda_kl (CV/Experiment.py:524-558)."""
    if not hp.kl_annealing:
        return 1.0
    if hp.model.startswith('cyc-anneal-vae'):
        if iter_num < hp.annealing_start:
            return 0.0
        if iter_num > hp.annealing_step + hp.annealing_start:
            return 1.0
        tau = ((iter_num - hp.annealing_start) % int(hp.annealing_step / hp.cyc_M)) / (hp.annealing_step / hp.cyc_M)
        return 1.0 if tau > hp.cyc_R else tau / hp.cyc_R
    max_val = hp.beta if hp.model.startswith('beta-vae') else 1.0
    return float(min(max(np.round((iter_num - hp.annealing_start) / hp.annealing_step, decimals=6), 0), max_val))


def build_model(hp, device, seed=None):
    """Same construction order as Experiment.__init__ (CV/Experiment.py:93-146): one shared
    nn.Embedding initialised N(0,1) (utils/vocab.py:22-23), then the model."""
    if seed is not None:
        torch.manual_seed(seed)
    settings.use_lstm = (hp.cell == 'lstm')
    emb = nn.Embedding.from_pretrained(torch.zeros(hp.V, hp.E).normal_(0, 1), freeze=False)
    cls = MODEL_CLASSES[hp.model]
    model = cls(hid_dim=hp.H, latent_dim=hp.Z, enc_layers=1, dec_layers=1, dropout=hp.dropout, enc_bi=True,
                dec_max_len=hp.max_len, beam_size=1, WEAtt_type='none', encoder_emb=emb, decoder_emb=emb,
                pad_id=hp.pad_id)
    return model.to(device)


class FlatAdam(object):
    """All unique parameters live in one fp32 arena (and their gradients in another), so that
    clip_grad_norm_ + Adam is two kernels (cvae_clip_adam).  Reproduces Experiment.step's
    duplicate listing of the shared embedding: the alias count is the number
    of state_dict keys naming it."""

    def __init__(self, model, hp):
        self.hp = hp
        sd = model.state_dict(keep_vars=True)
        uniq, self.n_alias, emb = [], 0, None
        for k, v in sd.items():
            if not v.requires_grad:
                continue
            if k.endswith('Embedding.weight'):
                self.n_alias += 1
                emb = v
            if not any(v is u for u in uniq):
                uniq.append(v)
        # embedding first: the alias segment is [0, V*E)
        uniq.sort(key=lambda t: 0 if t is emb else 1)
        # every parameter starts on a 16-byte boundary (the kernels read weights with 16-byte loads; a
        # vocabulary such as PTB's 10002 would otherwise misalign everything after the first V-sized bias);
        # the padding elements are zero with zero gradient and take no part in anything
        pad4 = lambda k: (k + 3) // 4 * 4
        n = sum(pad4(p.numel()) for p in uniq)
        dev = uniq[0].device
        self.params = torch.zeros(n, dtype=torch.float32, device=dev)
        # 4 spare floats behind the gradients: [n] carries the recurrent kernels' time-out word through the
        # data-parallel all-reduce (export_status / Trainer.train_step); never read by the optimizer
        self.grads = torch.zeros(n + 4, dtype=torch.float32, device=dev)
        self.n = n
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.alias_n = emb.numel() if emb is not None else 0
        off = 0
        self.views, self.gviews = [], []
        for p in uniq:
            k = p.numel()
            self.params[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.params[off:off + k].view_as(p.data)
            self.gviews.append(self.grads[off:off + k].view_as(p.data))
            p.grad = self.gviews[-1]
            self.views.append(p)
            off += pad4(k)
        self.sumsq = torch.zeros(2, dtype=torch.float32, device=dev)
        self.norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.steps = 0
        self.lr = hp.lr
        self.offsets = []
        off = 0
        for p in uniq:
            self.offsets.append(off)
            off += pad4(p.numel())
        # steps each parameter has taken: torch's Adam keeps one count per parameter and skips a parameter whose
        # gradient is None (the BoW head while lambda_KL <= 1e-5, CV/Experiment.py:711-713): see step()
        self.taken = [0] * len(uniq)
        self._none_now = [False] * len(uniq)
        self.buckets = None            # data-parallel gradient buckets (arm_buckets)
        self.hooks_live = False        # hooks act only inside Trainer.forward_backward's backward pass
        self.issued_early = 0          # buckets of the last step whose collective started inside the backward pass
        self._had_grad = None

    # ---- data-parallel training: bucketed all-reduce overlapped with the backward pass.
    # (The reference is single-process: SURVEY 8e.  Constraint kept: the reduction completes before the clip,
    # CV/Experiment.py:509-522.)  The arena is cut into contiguous buckets of about `bucket_bytes`; a bucket is
    # issued from the autograd hook of the LAST of its parameters to receive a gradient, so the collective of the
    # parameters whose gradients come first (ToVocab, the decoders) runs while the rest of the backward pass
    # (encoder, embedding) is still computing.  Which parameters receive a gradient at all is learnt from the
    # first (un-bucketed) step: BoW and PosteriorGaussianCouple.toSigma never do (None in the reference too).
    def arm_buckets(self, bucket_bytes):
        expected = [i for i, had in enumerate(self._had_grad or []) if had]
        if not expected or bucket_bytes <= 0:
            return
        self.buckets, cur = [], None
        for i in expected:
            lo = self.offsets[i]
            hi = lo + self.views[i].numel()
            if cur is None or (hi - cur['lo']) * 4 > bucket_bytes:
                cur = {'lo': lo, 'hi': hi, 'members': [], 'arrived': 0, 'work': None, 'fired': False, 'early': False}
                self.buckets.append(cur)
            cur['hi'] = hi
            cur['members'].append(i)
        # every bucket also carries the (zero-gradient) padding and never-touched segments up to the next bucket
        for k, b in enumerate(self.buckets):
            b['hi'] = self.buckets[k + 1]['lo'] if k + 1 < len(self.buckets) else self.grads.numel()
        self.buckets[0]['lo'] = 0
        owner = {}
        for b in self.buckets:
            for i in b['members']:
                owner[i] = b
        for i, p in enumerate(self.views):
            if i in owner:
                p.register_post_accumulate_grad_hook(self._make_hook(owner[i]))

    def _make_hook(self, bucket):
        def hook(param):
            if not self.hooks_live or self.buckets is None or bucket['fired']:
                return
            bucket['arrived'] += 1
            self._fire_ready()
        return hook

    def _fire_ready(self):
        """Collectives go out in ONE order on every rank, whatever order the hooks arrive in: highest arena offset
        first (the backward pass produces ToVocab's and the decoders' gradients first, and they sit last in the
        arena), a bucket only after every later bucket.  A bucket that is complete but blocked waits for its turn
        (finish_allreduce issues whatever is left in the same order)."""
        for b in reversed(self.buckets):
            if b['fired']:
                continue
            if b['arrived'] < len(b['members']):
                return
            self._fire(b)

    def _fire(self, bucket):
        src, dst = [], []
        for i in bucket['members']:
            p, g = self.views[i], self.gviews[i]
            if p.grad is not None and p.grad.data_ptr() != g.data_ptr():
                src.append(p.grad)
                dst.append(g)
        if src:
            torch._foreach_copy_(dst, src)
        for i in bucket['members']:
            self.views[i].grad = self.gviews[i]
        bucket['work'] = dist_utils.allreduce_sum_async(self.grads[bucket['lo']:bucket['hi']])
        bucket['fired'] = True
        bucket['early'] = True
        self.issued_early += 1

    def finish_allreduce(self, world):
        """After backward: gathers what the hooks did not, reduces every bucket not issued yet, waits for all of
        them and turns the sums into means.  Un-armed (first step, or dp_bucket_bytes = 0): one all-reduce.
        The spare slot behind the gradients carries this rank's time-out word (0 / 1) into the reduction of the
        LAST arena range (the first bucket of the backward pass is the highest one, so the slot has its own tiny
        collective when that bucket is already in flight); afterwards grads[n] > 0 on every rank iff a
        whole-sequence kernel timed out on some rank, and the local word is raised accordingly."""
        self.hooks_live = False
        _lib.call('cvae_persist_status_export', self.grads[self.n:].data_ptr(), _lib.stream())
        if self.buckets is None:
            self._had_grad = [p.grad is not None for p in self.views]
        # a parameter outside the set learnt on the first step that has a gradient now (a loss term that switched
        # on: the BoW head once lambda_KL > 1e-5): if the bucket that covers its arena range is already in flight,
        # gather_grads' copy comes too late for that collective and the range gets one of its own below
        extra = []
        if self.buckets is not None:
            for i, p in enumerate(self.views):
                if p.grad is not None and not self._had_grad[i]:
                    b = next(b for b in self.buckets if b['lo'] <= self.offsets[i] < b['hi'])
                    if b['fired']:
                        extra.append(i)
        self.gather_grads()
        if self.buckets is None:
            dist_utils.allreduce_mean_(self.grads)
            _lib.call('cvae_persist_status_import', self.grads[self.n:].data_ptr(), _lib.stream())
            return
        try:
            for b in reversed(self.buckets):                 # the same order as _fire_ready
                if not b['fired']:
                    b['work'] = dist_utils.allreduce_sum_async(self.grads[b['lo']:b['hi']])
            for b in self.buckets:
                if b['work'] is not None:
                    b['work'].wait()
            tail = [self.gviews[i].view(-1) for i in extra]  # same parameters, same order on every rank
            if self.buckets[-1]['early']:                    # its range went out before the status slot was written
                tail.append(self.grads[self.n:])
            for t in tail:
                w = dist_utils.allreduce_sum_async(t)
                if w is not None:                            # None: already complete (gloo, staged through the host)
                    w.wait()
        finally:
            self.reset_buckets()
        self.grads.mul_(1.0 / world)
        _lib.call('cvae_persist_status_import', self.grads[self.n:].data_ptr(), _lib.stream())

    def reset_buckets(self):
        """Bucket state of the iteration is dropped (also after a backward pass that raised)."""
        self.hooks_live = False
        for b in self.buckets or []:
            b['work'], b['fired'], b['arrived'], b['early'] = None, False, 0, False

    def zero_grad(self):
        """Gradients are produced outside the arena and gathered once after backward (gather_grads):
        with .grad = None autograd hands over each parameter's gradient tensor as it is, instead of
        adding it into a zeroed arena view with one kernel per parameter (48 per step)."""
        for p in self.views:
            p.grad = None

    def gather_grads(self):
        """One fused copy of every produced gradient into the arena; .grad points into the arena again
        afterwards.  Segments of parameters without a gradient (None in the reference too) keep the
        zeros they were created with: nothing ever writes them."""
        src, dst = [], []
        for i, (p, g) in enumerate(zip(self.views, self.gviews)):
            self._none_now[i] = p.grad is None
            if p.grad is None:
                if self.taken[i] > 0:          # had a gradient in an earlier iteration: the arena still holds it
                    g.zero_()
            elif p.grad.data_ptr() != g.data_ptr():
                src.append(p.grad)
                dst.append(g)
        if src:
            torch._foreach_copy_(dst, src)
        for p, g in zip(self.views, self.gviews):
            p.grad = g

    def _late_group(self):
        """torch's Adam counts steps per parameter and, like clip_grad_norm_, passes over a parameter whose .grad
        is None (CV/Experiment.py:509-522).  Parameters that never had a gradient need nothing: their arena
        gradient, m and v are zero and the update is exactly zero.  Parameters that receive their FIRST gradient
        after the model's first step (the BoW head behind `lambda_KL > 1e-5`, :711-713), or lose it again, form
        the "late" group of cvae_clip_adam_late: one contiguous arena range with one step count.
        -> None or (offset, n, steps taken; -1 = no gradient now)."""
        late = [i for i, none in enumerate(self._none_now)
                if (not none and self.taken[i] != self.steps) or (none and self.taken[i] > 0)]
        if not late:
            return None
        lo, hi = late[0], late[-1]
        state = {(self._none_now[i], self.taken[i]) for i in late}
        if late != list(range(lo, hi + 1)) or len(state) != 1 or self.views[lo] is self.views[0] and self.alias_n:
            raise NotImplementedError('parameters with %d different (gradient present, steps taken) histories: '
                                      'cvae_clip_adam_late takes one contiguous late group' % len(state))
        none, taken = next(iter(state))
        end = self.offsets[hi] + self.views[hi].numel()
        return self.offsets[lo], end - self.offsets[lo], (-1 if none else taken)

    def step(self):
        mult = max(self.n_alias, 1)
        late = self._late_group()
        ops.clip_adam(self.params, self.grads, self.m, self.v, 0, self.alias_n, mult, self.hp.clip_norm, self.lr,
                      (self.hp.beta1, self.hp.beta2), 1e-8, self.steps, self.steps * mult, self.sumsq, self.norm,
                      late=late)
        for i, none in enumerate(self._none_now):
            if not none:
                self.taken[i] += 1
        self.steps += 1


def _set_mode(model, training):
    """model.train(training), skipped when the flag already holds (the recursive walk over the sub-modules costs
    ~0.15 ms of host time per step, and the small shapes are bound by the host)."""
    if any(m.training != training for m in model.modules()):
        model.train(training)


class Trainer(object):
    def __init__(self, hp, device, seed=0, model=None):
        self.hp, self.device = hp, device
        self.model = model if model is not None else build_model(hp, device, seed)
        _set_mode(self.model, True)
        self.opt = FlatAdam(self.model, hp)
        self.iter_num = 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self._armed = False
        # data-parallel jobs: see train_step
        self._pending = collections.deque()      # (event, pinned copy of the reduced status slot, batch) of the last steps
        self._cooldown = 0                       # steps left on the per-timestep kernels after a time-out
        self._persist_prev = 1
        self.fallbacks = []                      # iterations whose batches were re-run on the per-timestep kernels
        if self.world > 1:
            _lib.query('cvae_persist_headroom', settings.dp_persist_headroom)

    # ---- one training iteration: CV/Experiment.py:646-775 / CC/Experiment.py:318-428
    def losses(self, batch):
        hp, m = self.hp, self.model
        if hp.is_dialog():
            out = m(post_bare=batch['post_bare'], post_len=batch['post_len'], resp_go=batch['resp_go'],
                    resp_len=batch['resp_len'], resp_bare=batch['resp_bare'])
            eos = batch['resp_eos']
            if hp.is_couple():
                logits, prior, post, logits_c, init, init_c = out
            else:
                logits, prior, post = out
            kl = ops.kl_div(post.loc, post.scale, prior.loc, prior.scale)
        elif hp.model in ('dae', 'lm'):               # CV/Experiment.py:666-692: reconstruction loss only
            out = m(go=batch['go'], sent_len=batch['len'], bare=batch['bare'])
            logits = out if hp.model == 'lm' else out[0]
            seq = m.criterionSeq(logits, batch['eos'], keep_batch=True).mean(0)
            return {'seq': seq, 'tot': seq, 'lambda_kl': 0.0}
        else:
            out = m(go=batch['go'], sent_len=batch['len'], bare=batch['bare'])
            eos = batch['eos']
            if hp.is_couple():
                logits, post, _, bow_logits, logits_c, init, init_c = out
            else:
                logits, post, _, bow_logits = out
            if hp.model.startswith('surprising-fix'):  # free bits, CV/Experiment.py:1141-1142
                kl = ops.kl_free_bits(post.loc, post.scale, hp.lambda_fb / hp.Z)
            else:
                kl = ops.kl_div(post.loc, post.scale)
        # seq + lambda_KL * kl [+ lambda_seq_couple * seq_c + lambda_couple * couple], each a batch mean
        # (Experiment.py:753-757): one launch for the means and the total (ops.loss_assemble)
        lam = lambda_kl(self.iter_num, hp)
        names, coef = ['seq', 'kl'], [1.0, lam]
        terms = [m.criterionSeq(logits, eos, keep_batch=True), kl]
        if hp.is_couple():
            names += ['seq_c', 'couple']
            coef += [hp.lambda_seq_couple, hp.lambda_couple]
            terms += [m.criterionSeq(logits_c, eos, keep_batch=True),
                      ops.couple_loss(init, init_c, hp.Z, hp.euclidean_dist)]
        bow_off = None
        if hp.lambda_BoW > 0 and not hp.is_dialog():
            # bag-of-words loss (CV/Experiment.py:703-706, 737-740): cross-entropy of BoW(z) against every token of
            # the sentence, per sentence divided by its length.  It enters the total only once lambda_KL > 1e-5
            # (:711-713, 759-761); until then it is computed for the log alone and the BoW head has NO gradient
            # (None in the reference: its clip and Adam pass over the head, FlatAdam._late_group)
            L = batch['bare'].shape[1]
            with torch.set_grad_enabled(lam > 1e-5):
                bow = m.criterionSeq(bow_logits.unsqueeze(1).expand(-1, L, -1), batch['bare'], keep_batch=True)
                bow = bow / batch['len'].to(bow.dtype)
            if lam > 1e-5:
                names.append('bow')
                coef.append(hp.lambda_BoW)
                terms.append(bow)
            else:
                bow_off = bow.mean(0)
        tot, means = ops.loss_assemble(terms, coef)
        res = {nm: means[i] for i, nm in enumerate(names)}
        if bow_off is not None:
            res['bow'] = bow_off
        res['tot'] = tot.view(())
        res['lambda_kl'] = lam
        return res

    def forward_backward(self, batch):
        self.iter_num += 1
        _set_mode(self.model, True)
        self.opt.zero_grad()
        # every term below is a batch mean: the model may leave its per-row outputs in its own (length-sorted)
        # processing order (modules.RowOrder) instead of un-permuting each of them
        keep, settings.restore_row_order = settings.restore_row_order, False
        try:
            res = self.losses(batch)
        finally:
            settings.restore_row_order = keep
        self.opt.hooks_live = self.world > 1
        self.opt.issued_early = 0
        try:
            res['tot'].backward()
        except BaseException:
            self.opt.reset_buckets()           # no stale 'fired' / in-flight state after a failed backward pass
            raise
        if self.world > 1:
            # the batch mean over equal per-rank batches = mean of per-rank gradients; must
            # complete before the clip (Experiment.py:515 clips the full-batch gradient).  Buckets whose
            # gradients were complete earlier in the backward pass are already in flight (FlatAdam.arm_buckets)
            self.opt.finish_allreduce(self.world)
            if self.opt.buckets is None and settings.dp_bucket_bytes > 0 and not self._armed:
                self._armed = True
                self.opt.arm_buckets(settings.dp_bucket_bytes)
        else:
            self.opt.gather_grads()
        return res

    def check_kernels(self, sync=False):
        """Raises CvaeHipError if a whole-sequence recurrent kernel ran out of its bounded in-kernel wait (its
        results, and everything computed from them, are invalid).  The word lives in host-mapped memory, so the
        default check costs a host read and no synchronisation: it sees a failure once the failing kernel's store
        has landed, at the latest one step later; cvae_clip_adam checks the same word on the device and does not
        apply a step while it is set, so the weights are never touched by a timed-out step.  sync=True
        synchronises the device first (use it where the host reads results anyway)."""
        bad = _lib.query('cvae_persist_status') if sync else _lib.query('cvae_persist_status_peek')
        if bad:
            if not sync:
                _lib.query('cvae_persist_status')            # synchronise and clear
            raise _lib.CvaeHipError(
                'a whole-sequence recurrent kernel timed out (iteration %d): its blocks were not all resident at once '
                '-- another kernel shared the GPU (a second process or stream, a profiler).  The optimizer step of the '
                'affected iteration was NOT applied.  Run alone on the device or set CVAE_PERSISTENT_RNN=0.' % self.iter_num)

    def train_step(self, batch):
        """forward + backward (+ gradient all-reduce) + clip + Adam.  Returns the loss dict
        (device scalars: nothing syncs here; read them back in one copy when logging).

        One process: a timed-out whole-sequence kernel raises (check_kernels).  Data-parallel (world > 1): a busy
        chip must not stop the job, and every rank has to take the same decision on the same iteration.  The
        time-out word of every rank rides through the gradient all-reduce (FlatAdam.finish_allreduce), so a
        time-out anywhere raises the word everywhere before the clip: cvae_clip_adam then skips that step -- and
        every later one, until the word is cleared -- on all ranks alike, and the replicas stay bit-identical.
        The host looks at the REDUCED value of step k (the same number on every rank) two steps later, when that
        step has certainly completed (no stall of the host's run-ahead): if it is set, the device is drained, the
        word cleared, the recurrences switched to the per-timestep kernels (cvae_persistent_rnn(0), no restart)
        for settings.dp_fallback_steps steps, and the batches of the skipped steps are re-run in order."""
        if self.world > 1:
            self._dp_check(lag=2)
            if self._cooldown > 0:
                self._cooldown -= 1
                if self._cooldown == 0:
                    _lib.query('cvae_persistent_rnn', self._persist_prev)      # back to what it was before the time-out
        before = (self.iter_num, self.opt.steps, list(self.opt.taken))
        res = self.forward_backward(batch)
        self.opt.step()
        if self.world > 1:
            pin = torch.empty(1, dtype=torch.float32, pin_memory=True)
            pin.copy_(self.opt.grads[self.opt.n:self.opt.n + 1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((ev, pin, batch, before))
        else:
            self.check_kernels()
        return res

    def _dp_check(self, lag):
        """Looks at the reduced status of the step `lag` steps back; handles a time-out (see train_step)."""
        while len(self._pending) > lag:
            ev, pin, _, before = self._pending[0]
            ev.synchronize()
            if float(pin[0]) == 0.0:
                self._pending.popleft()
                continue
            # some rank timed out in that iteration: its step and the ones enqueued since were skipped on the device.
            # The host's counters go back to where they stood before it (iteration number -> lambda_KL, Adam's step
            # counts), so the re-run is the run that would have happened
            redo = list(self._pending)
            self._pending.clear()
            _lib.query('cvae_persist_status')                # drain the device, clear the word
            prev = _lib.query('cvae_persistent_rnn', 0)
            if self._cooldown == 0:
                self._persist_prev = prev
            self._cooldown = settings.dp_fallback_steps + len(redo) + 1
            self.iter_num, self.opt.steps, self.opt.taken = before[0], before[1], list(before[2])
            self.fallbacks.append(before[0] + 1)
            for _, _, b, _ in redo:
                self.train_step(b)
            return

    def dp_flush(self):
        """End of a data-parallel run (or before reading the weights): every outstanding step is checked."""
        if self.world > 1:
            self._dp_check(lag=0)

    # ---- evaluation: CV/Experiment.py:1323-1396 ; CC/Experiment.py:538-621
    def iw_eval(self, batches, n_sample):
        """batches: iterable of batch dicts -- this rank's shard, or with settings.shard_samples the
        SAME batches on every rank (the K samples are sharded inside test_lm instead).  Returns
        corpus metrics (nll, xent, kl, ppl), identical on every rank."""
        _set_mode(self.model, False)
        hp = self.hp
        acc = torch.zeros(5, dtype=torch.float64, device=self.device)   # sum nll, xent, kl, len+1, count
        per_sentence = []
        for b in batches:
            if hp.is_dialog():
                xent, nll, kl = self.model.test_lm(post_bare=b['post_bare'], post_len=b['post_len'],
                                                   resp_go=b['resp_go'], resp_len=b['resp_len'],
                                                   resp_bare=b['resp_bare'], resp_eos=b['resp_eos'],
                                                   n_sample=n_sample)
                ln = b['resp_len']
            else:
                xent, nll, kl, _ = self.model.test_lm(bare=b['bare'], sent_len=b['len'], go=b['go'], eos=b['eos'],
                                                      n_sample=n_sample)
                ln = b['len']
            acc += torch.stack([nll.sum().double(), xent.sum().double(), kl.sum().double(),
                                (ln + 1).sum().double(), torch.tensor(float(nll.numel()), device=self.device).double()])
            per_sentence.append((xent, nll, kl))
        if not settings.shard_samples:
            dist_utils.reduce_sums_(acc)
        a = acc.tolist()
        self.check_kernels(sync=True)
        _set_mode(self.model, True)
        return dict(nll=a[0] / a[4], xent=a[1] / a[4], kl=a[2] / a[4], ppl=math.exp(a[0] / a[3]),
                    n=int(a[4])), per_sentence

    def estimate_mi(self, batches, n_sample=100):
        """CV/Experiment.py:1453-1503: whole batches (the estimator depends on batch
        composition) are sharded over ranks; final reduction of (sum mi, count)."""
        _set_mode(self.model, False)
        acc = torch.zeros(2, dtype=torch.float64, device=self.device)
        for b in batches:
            mi, _ = self.model.estimate_mi(go=b['go'], sent_len=b['len'], bare=b['bare'], n_sample=n_sample)
            acc += torch.stack([mi.sum().double(), torch.tensor(float(mi.numel()), device=self.device).double()])
        dist_utils.reduce_sums_(acc)
        _set_mode(self.model, True)
        a = acc.tolist()
        self.check_kernels(sync=True)
        return a[0] / max(a[1], 1.0)


def text_batch(corpus, idx, device, truncate=True, sort_rows=False):
    bare, go, eos, ln = corpus.batch(idx, device, truncate, sort_rows)
    return dict(bare=bare, go=go, eos=eos, len=ln)


def dialog_batch(posts, resps, idx, device):
    """CC/Experiment.py:331-339 layout (both sides truncated to their batch maxima)."""
    pb, _, _, pl = posts.batch(idx, device, True)
    rb, rg, re_, rl = resps.batch(idx, device, True)
    return dict(post_bare=pb, post_len=pl, resp_bare=rb, resp_go=rg, resp_eos=re_, resp_len=rl)
