"""Soak of the in-band hand-offs (whole-sequence recurrent kernels) under uneven memory load.

The same N training steps are run three times in deterministic mode (cvae_deterministic(1): every result bit-reproducible):
quiet; with a side stream that streams 256 MB device-to-device copies the whole time (HBM traffic beside the
hand-offs, CUs taken away from the recurrent kernels' blocks at launch); and quiet again.  A hand-off that ever consumed
a stale or half-landed piece would change some bit of some loss or weight: all three runs must agree exactly, and no
bounded spin may run out (cvae_persist_status).  MI355X_MICROARCH.md: "test every hand-off under uneven load, checking
every word"."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
cell = sys.argv[2] if len(sys.argv) > 2 else 'lstm'
_lib.query('cvae_deterministic', 1)
shape = data.SHAPES[os.environ.get('SHAPE', 'yelp')]
hp = harness.HParams(model='vae-couple', cell=cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'])
B = 32
corpus = data.Corpus(data.synthetic_sentences(B * 32, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=77))
batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev) for i in range(32)]
side = torch.cuda.Stream()
big_a = torch.empty(64 << 20, device=dev); big_b = torch.empty(64 << 20, device=dev)      # 256 MB each


def run(load):
    tr = harness.Trainer(hp, dev, seed=1)
    torch.manual_seed(4321)
    stop = threading.Event()

    def hammer():
        torch.cuda.set_device(dev)
        with torch.cuda.stream(side):
            while not stop.is_set():
                big_b.copy_(big_a, non_blocking=True)
                big_a.copy_(big_b, non_blocking=True)
                side.synchronize()

    th = threading.Thread(target=hammer) if load else None
    if th:
        th.start()
    t0 = time.perf_counter()
    tots = [tr.train_step(batches[i % 32])['tot'].detach() for i in range(n)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if th:
        stop.set(); th.join()
    st = _lib.query('cvae_persist_status')
    assert st == 0, 'a bounded spin ran out'
    return torch.stack(tots).cpu(), tr.opt.params.detach().cpu().clone(), dt


ref_l, ref_w, dt0 = run(False)
print('quiet:  %.2f ms/step' % (dt0 / n * 1e3), flush=True)
for tag, load in (('loaded', True), ('quiet2', False)):
    l, w, dt = run(load)
    same = torch.equal(l, ref_l) and torch.equal(w, ref_w)
    print('%s: %.2f ms/step  bit-identical to the first run: %s' % (tag, dt / n * 1e3, same), flush=True)
    assert same, 'results differ (%s): max loss diff %.3e' % (tag, float((l - ref_l).abs().max()))
print('hand-off soak ok (%d steps x 3, %s)' % (n, cell))
