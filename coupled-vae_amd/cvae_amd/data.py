"""Synthetic corpora and on-device batch layout (a1).

The reference's file readers and pickled vocabularies are out of scope; what is kept
bit-exact is the tensor layout `process_raw_sent` produces (CV/dataloaders/ptb.py:118-129),
built on the GPU by cvae_pack_batch from CSR token ids."""
import numpy as np
import torch

from . import ops

PAD, GO, EOS, UNK = 0, 1, 2, 3     # CV/utils/vocab.py:29-30

SHAPES = {
    # name: V, E, H, Z, lognormal mu, sigma, cap   
    'ptb': dict(V=10002, E=200, H=128, Z=32, mu=3.0, sigma=0.45, cap=82),
    'yelp': dict(V=20000, E=200, H=256, Z=32, mu=4.4, sigma=0.5, cap=200),
    'yahoo': dict(V=20001, E=200, H=256, Z=32, mu=4.2, sigma=0.5, cap=200),
    'switchboard': dict(V=10000, E=200, H=128, Z=32, mu=2.6, sigma=0.6, cap=200),
    'scaleup': dict(V=50000, E=200, H=1024, Z=32, mu=4.4, sigma=0.5, cap=200),
}


def synthetic_sentences(n, V, mu, sigma, cap, seed=0):
    """ids uniform in [4,V), len = clip(floor(lognormal(mu,sigma)), 1, cap)."""
    rs = np.random.RandomState(seed)
    lens = np.clip(np.floor(rs.lognormal(mu, sigma, size=n)), 1, cap).astype(np.int64)
    return [rs.randint(4, V, size=int(l)).astype(np.int64) for l in lens]


class Corpus(object):
    """Token ids in CSR form on the host; batches are laid out on the device."""

    def __init__(self, sentences, max_len=200):
        self.sentences = [np.asarray(s, dtype=np.int64) for s in sentences]
        self.max_len = max_len
        self.lengths = np.asarray([min(len(s), max_len) for s in self.sentences], dtype=np.int64)

    def __len__(self):
        return len(self.sentences)

    def batch(self, idx, device, truncate=True, sort_rows=False):
        """-> bare (B,L), go (B,L+1), eos (B,L+1), len (B,) int64 on `device`.
        truncate=True: training layout (cut to the batch maximum, Experiment.py:659-664).
        sort_rows=True: the batch's rows in order of non-increasing length (same sentences; the
        loss is a batch mean, so the order is free) -- the backward recurrence then exchanges only the
        rows that are still inside their sentence."""
        idx = list(idx)
        if sort_rows:
            idx = sorted(idx, key=lambda i: -int(self.lengths[i]))
        sents = [self.sentences[i] for i in idx]
        flat = np.concatenate(sents) if sents else np.zeros((0,), np.int64)
        off = np.cumsum([0] + [len(s) for s in sents]).astype(np.int64)
        width = int(self.lengths[list(idx)].max()) if truncate else self.max_len
        flat_d = torch.from_numpy(flat).to(device, non_blocking=True)
        off_d = torch.from_numpy(off).to(device, non_blocking=True)
        if flat_d.numel() == 0:
            flat_d = torch.zeros((1,), dtype=torch.int64, device=device)
        return ops.pack_batch(flat_d, off_d, self.max_len, width, PAD, GO, EOS)

    # ---- 8f.2: the whole corpus resident on the device; a batch costs one copy of B indices
    def to_device(self, device):
        flat = np.concatenate(self.sentences) if self.sentences else np.zeros((1,), np.int64)
        off = np.cumsum([0] + [len(s) for s in self.sentences]).astype(np.int64)
        self.dev_flat = torch.from_numpy(flat).to(device)
        self.dev_off = torch.from_numpy(off).to(device)
        # pinned staging for the per-batch index vector: a ring of slots, each guarded by the event recorded
        # after its last host->device copy (the host runs ahead of the stream in a training loop, so a slot
        # must not be overwritten while its copy is still queued)
        self._ring = [torch.empty((4096,), dtype=torch.int64).pin_memory() for _ in range(8)]
        self._ring_ev = [None] * len(self._ring)
        self._ring_next = 0
        return self

    def batch_resident(self, idx, truncate=True):
        """Same tensors as batch(), built from the device-resident CSR arrays (to_device first)."""
        idx = np.asarray(list(idx), dtype=np.int64)
        width = int(self.lengths[idx].max()) if truncate else self.max_len
        if idx.size <= self._ring[0].numel():
            k = self._ring_next
            self._ring_next = (k + 1) % len(self._ring)
            if self._ring_ev[k] is not None:
                self._ring_ev[k].synchronize()          # the copy that last read this slot has completed
            stage = self._ring[k][:idx.size]
            stage.copy_(torch.from_numpy(idx))
            sel = stage.to(self.dev_flat.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev_flat.device))
            self._ring_ev[k] = ev
        else:
            sel = torch.from_numpy(idx).to(self.dev_flat.device)
        return ops.pack_batch_indexed(self.dev_flat, self.dev_off, sel, self.max_len, width, PAD, GO, EOS)

    def n_tokens(self, idx):
        """PPL denominator: len + 1 per sentence (Experiment.py:1394)."""
        return int((self.lengths[list(idx)] + 1).sum())


def epoch_batches(lengths, batch_size, seed=0, bucket=False, pool=50, drop_last=False):
    """Index lists of one epoch.  bucket=False: a plain shuffle (DataLoader(shuffle=True),
    CV/Experiment.py:641).  bucket=True (8f.2): shuffle, cut into pools of pool*batch_size sentences,
    sort each pool by length, cut into batches, shuffle the batch order -- sentences of similar
    length share a batch, so the per-batch width (and the recurrence's step count) follows the
    data instead of the longest outlier.  Changes batch composition, hence opt-in."""
    rs = np.random.RandomState(seed)
    n = len(lengths)
    perm = rs.permutation(n)
    if bucket:
        lengths = np.asarray(lengths)
        step = pool * batch_size
        perm = np.concatenate([p[np.argsort(-lengths[p], kind='stable')]
                               for p in (perm[i:i + step] for i in range(0, n, step))])
    batches = [perm[i:i + batch_size] for i in range(0, n, batch_size)]
    if drop_last and batches and len(batches[-1]) < batch_size:
        batches.pop()
    if bucket:
        order = rs.permutation(len(batches))
        batches = [batches[i] for i in order]
    return [b.tolist() for b in batches]
