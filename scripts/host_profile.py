"""Diagnostic: where the HOST spends a training step (cProfile over n steps; the device runs asynchronously, so at the
small shapes this is the step time).  SHAPE=ptb|yelp|switchboard python scripts/host_profile.py [n]"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'coupled-vae_amd'))
import torch
from cvae_amd import _lib, data, harness
dev = torch.device('cuda:0'); _lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
shape = data.SHAPES[os.environ.get('SHAPE', 'ptb')]
hp = harness.HParams(model='vae-couple', cell='lstm', V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'], batch_size=32)
tr = harness.Trainer(hp, dev, seed=0)
corpus = data.Corpus(data.synthetic_sentences(256, shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000))
batches = [harness.text_batch(corpus, range(i * 32, (i + 1) * 32), dev) for i in range(8)]
for i in range(5):
    tr.train_step(batches[i % 8])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    tr.train_step(batches[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host issue %.3f ms/step, with device drain %.3f ms/step' % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    tr.train_step(batches[i % 8])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumulative').print_stats(60)
