#!/usr/bin/env python3
"""bench.py -- Coupled-VAE hot path on MI355X: train tokens/s + IW-NLL sentences/s.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload at N=1 (BASELINE.json configs[1]): Yelp-shape Coupled-VAE (V=20000, E=200, H=256,
Z=32, IMQ coupling), batch 32, synthetic corpus (lengths lognormal(4.4,0.5) capped at 200),
LSTM cells (the cell BASELINE's metric names; --cell gru for the reference default).
A "step" = one training step (forward, backward, clip, Adam) on one batch of 32 sentences.
After the K timed steps the IW-NLL evaluation (n_sample=500) of --iw-batches batches is
timed separately.  `value` = training tokens/s of the whole job (a token = one non-pad
target = len+1 per sentence, the PPL denominator), `iw_nll.value` = sentences/s.
N > 1: one process per GPU, batch 32 PER GPU (weak scaling), gradients all-reduced over
RCCL before the clip; evaluation batches sharded over ranks, one final all-reduce.

Inputs are resident in HBM before the timed regions.  Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'coupled-vae_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402
import torch.distributed as dist   # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, dense fp32-input matrix (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TFLOPS = 2516.6   # dense bf16 (16x the fp32-input rate: v_mfma_f32_32x32x16_bf16)
PEAK_HBM_GBS = 8000.0
# kernels whose product runs as error-compensated split-bf16: operands split into two bf16 terms and three bf16
# MFMA passes per fp32 product (bf16x3: forward products), or three terms and six passes (bf16x6: the gradient
# products of gemm_wide).  Their algorithmic fp32 FLOP/s are priced against the bf16 peak / passes
SPLIT_BF16_KERNELS = {'ce_core_kernel': 3, 'rnn_large_step_kernel': 3, 'gemm_wide_kernel': 6, 'gemm_nt_kernel': 3,
                      'rnn_fwd_persist_kernel': 3, 'rnn_bwd_persist_kernel': 6}
SPLIT_BF16 = True
KERNELS = {0: ('gemm_tiled_kernel', 'mfma'), 1: ('skinny_kernel', 'mfma'), 2: ('cell_fwd_kernel', 'hbm'),
           3: ('cell_bwd_kernel', 'hbm'), 4: ('ce_core_kernel', 'mfma'), 5: ('gemm_wide_kernel', 'mfma'),
           6: ('rnn_fwd_persist_kernel', 'mfma'), 7: ('rnn_bwd_persist_kernel', 'mfma'),
           8: ('rnn_large_step_kernel', 'mfma'), 9: ('gemm_nt_kernel', 'mfma'), 10: ('clip_adam_kernel', 'hbm'),
           11: ('ce_grad_from_logits_kernel', 'hbm'), 12: ('iw_reduce_kernel', 'hbm')}
HBM_KERNELS = ('clip_adam_kernel', 'ce_grad_from_logits_kernel', 'iw_reduce_kernel')


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='yelp', choices=['ptb', 'yelp', 'yahoo'])
    ap.add_argument('--cell', default='lstm', choices=['gru', 'lstm'])
    ap.add_argument('--iw-samples', type=int, default=500)
    ap.add_argument('--iw-batches', type=int, default=2, help='evaluation batches per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-iw', action='store_true')
    ap.add_argument('--no-ptb', action='store_true', help='skip the PTB-shape LSTM sub-record')
    ap.add_argument('--no-fp32', action='store_true', help='skip the fp32-input-MFMA sub-record')
    ap.add_argument('--iw-shard', default='sentences', choices=['sentences', 'samples'],
                    help='N > 1: shard evaluation batches over ranks (weak scaling, default) or the K samples of '
                         'every sentence (same batches on every rank, all-gather of the cross-entropy rows)')
    return ap.parse_args()


def log(rank, msg):
    """Progress on stderr (stdout carries only the JSON line)."""
    if rank == 0 or os.environ.get('CVAE_BENCH_VERBOSE') == '1':
        sys.stderr.write('[bench rank %d %s] %s\n' % (rank, time.strftime('%H:%M:%S'), msg))
        sys.stderr.flush()


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=dev if dist.get_backend() != 'gloo' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def profile_read(lib):
    out = {}
    for k, (name, bound) in KERNELS.items():
        n, ms, work = ctypes.c_int64(0), ctypes.c_double(0), ctypes.c_double(0)
        rc = lib.cvae_profile_read(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(work))
        assert rc == 0
        out[name] = dict(launches=n.value, ms=ms.value, work=work.value, bound=bound)
    return out


def pmc_traffic(kernel):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r03_pmc_hbm_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs of this bench,
    scripts/collect_profiles_r03.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Counters
    cannot be read from inside this process, so the figure is the recorded one for the same workload, or None."""
    if kernel != 'ce_core_kernel':
        return None
    for name, split in (('r03_pmc_hbm_summary.json', True), ('r02_pmc_hbm_summary.json', None), ('r01_pmc_hbm_summary.json', None)):
        path = os.path.join(ROOT, 'profiles', name)
        try:
            rec = json.load(open(path))['ce_core_forward_eval_launch']
            # the record must be of the variant this run launches (split-bf16 or fp32-input)
            is_split = split if split is not None else ('ce_core_bf3_kernel' in rec.get('kernel', ''))
            if is_split != SPLIT_BF16:
                continue
            return round(rec['fetch_bytes_corrected_avg'] + rec['write_bytes_avg'])
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def roofline_of(prof, region):
    """The kernel with the largest total time in the profiled pass."""
    name, rec = max(((k, v) for k, v in prof.items() if k not in HBM_KERNELS), key=lambda kv: kv[1]['ms'])
    if rec['launches'] == 0 or rec['ms'] == 0:
        return None
    per_s = rec['work'] / (rec['ms'] * 1e-3)
    arith = 'f32'
    if rec['bound'] == 'mfma':
        ach, peak, unit = per_s / 1e12, PEAK_F32_MFMA_TFLOPS, 'TFLOP/s'
        if SPLIT_BF16 and name in SPLIT_BF16_KERNELS:
            passes = SPLIT_BF16_KERNELS[name]
            peak = round(PEAK_BF16_MFMA_TFLOPS / passes, 1)
            arith = ('bf16x3 (hi.hi + hi.lo + lo.hi, fp32 accumulation)' if passes == 3 else
                     'bf16x6 (three-term split, the six leading products, fp32 accumulation)')
    else:
        ach, peak, unit = per_s / 1e9, PEAK_HBM_GBS, 'GB/s'
    out = dict(kernel=name, region=region, bound=rec['bound'], achieved=round(ach, 3), peak=peak, unit=unit,
               frac=round(ach / peak, 4), traffic=pmc_traffic(name), launches=rec['launches'],
               avg_launch_ms=round(rec['ms'] / rec['launches'], 5),
               work_per_launch=rec['work'] / rec['launches'], arithmetic=arith)
    if arith != 'f32':
        out['note'] = ('achieved = algorithmic fp32 FLOP/s (2*M*N*K per launch); peak = dense bf16 MFMA peak / %d passes; '
                       'against the fp32-input MFMA peak of %.1f TFLOP/s the same figure is %.2fx'
                       % (SPLIT_BF16_KERNELS[name], PEAK_F32_MFMA_TFLOPS, ach / PEAK_F32_MFMA_TFLOPS))
    return out


def pmc_hbm_bytes(kernel):
    """HBM-side bytes per launch of a bandwidth kernel from the committed PMC passes (profiles/r03_pmc_hbm_summary.json,
    FETCH_SIZE doubled as the guide prescribes for gfx950), or None."""
    try:
        rec = json.load(open(os.path.join(ROOT, 'profiles', 'r03_pmc_hbm_summary.json')))[kernel]
        return round(rec['fetch_bytes_corrected_avg'] + rec['write_bytes_avg'])
    except (OSError, KeyError, ValueError, TypeError):
        return None


def roofline_hbm(profs):
    """Achieved GB/s of the bandwidth-shaped kernels (algorithmic bytes = every input and output once, SURVEY 8d,
    divided by the HIP-event duration of their launches) against the 8 TB/s HBM peak."""
    out = {}
    for prof in profs:
        for name in HBM_KERNELS:
            rec = (prof or {}).get(name)
            if not rec or rec['launches'] == 0 or rec['ms'] == 0 or name in out:
                continue
            gbs = rec['work'] / (rec['ms'] * 1e-3) / 1e9
            out[name] = dict(bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                             frac=round(gbs / PEAK_HBM_GBS, 4), launches=rec['launches'],
                             avg_launch_ms=round(rec['ms'] / rec['launches'], 5),
                             bytes_per_launch=round(rec['work'] / rec['launches']), traffic=pmc_hbm_bytes(name))
    return out


def host_cores():
    """(physical cores of socket 0, threads this process may use).  SURVEY 8d: the CPU baseline runs on the physical
    cores of one socket; on the GPU box the job's CPU share is smaller (OMP_NUM_THREADS), which caps it."""
    cores = set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = int(line.split(':')[1])
            elif line.startswith('core id'):
                core = int(line.split(':')[1])
            elif not line.strip():
                if phys == 0 and core is not None:
                    cores.add(core)
                phys = core = None
    except OSError:
        pass
    socket = len(cores) or (os.cpu_count() or 1)
    share = socket
    try:
        share = min(share, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    if os.environ.get('OMP_NUM_THREADS', '').isdigit():
        share = min(share, int(os.environ['OMP_NUM_THREADS']))
    return socket, max(share, 1)


def cpu_baseline(args, shape):
    """The oracle ("port": CPU restatement proven equal to the reference, tests/test_oracle_golden.py) timed on
    this host's cores on a bounded sample of the same workload, by SURVEY 8d's protocol: 3 warm-up + 10 timed
    training steps on the same synthetic batches (reference row order), threads = the physical cores of one
    socket (capped by the job's CPU share); IW-NLL timed on one full batch of 32 sentences at a stated K."""
    from oracle import cvae_oracle as O
    socket_cores, n_thr = host_cores()
    torch.set_num_threads(n_thr)
    model, cell = 'vae-couple', args.cell
    P = O.tie_aliases(O.init_params(model, cell, shape['V'], shape['E'], shape['H'], shape['Z'], seed=0))
    B, n_warm, n_steps = 32, 3, 10
    sents = O.synthetic_sentences(B * (n_warm + n_steps), shape['V'], shape['mu'], shape['sigma'], shape['cap'], seed=1000)
    g = torch.Generator().manual_seed(0)

    def drop(*s):
        return (torch.rand(*s, generator=g) > 0.2).float() / 0.8

    Hd, E = 2 * shape['H'], shape['E']
    cfg = dict(latent_dim=shape['Z'], lambda_kl=0.5, lambda_couple=0.1, lambda_seq_couple=1.0, euclidean_dist=False)
    keys = [k for k in P.keys() if not k.startswith('BoW') and 'PosteriorGaussianCouple.toSigma' not in k]
    state = {}
    tokens, dt = 0, 0.0
    for i in range(n_warm + n_steps):
        bare, go, eos, ln = (torch.from_numpy(x) for x in O.make_batch(sents[i * B:(i + 1) * B], 200, truncate=True))
        batch = dict(bare=bare, go=go, eos=eos, len=ln)
        L, T = bare.shape[1], go.shape[1]
        noise = dict(enc_drop=drop(L, B, E), eps=torch.randn(B, shape['Z'], generator=g),
                     dec_in_drop=drop(T, B, 2 * E), dec_out_drop=drop(T, B, Hd),
                     decc_in_drop=drop(T, B, 2 * E), decc_out_drop=drop(T, B, Hd))
        t0 = time.perf_counter()
        O.train_step(P, model, cell, batch, noise, cfg, state, keys)
        if i >= n_warm:
            dt += time.perf_counter() - t0
            tokens += int((ln + 1).sum())
    # IW-NLL: one full batch of 32 sentences, K = the reference's validation setting (10; Experiment.py:1330-1335)
    Ks = 10
    bare, go, eos, ln = (torch.from_numpy(x) for x in O.make_batch(sents[:B], 200, truncate=True))
    sb = dict(bare=bare, go=go, eos=eos, len=ln)
    eps = torch.randn(Ks, B, shape['Z'], generator=g)
    for pp in {id(v): v for v in P.values()}.values():
        pp.requires_grad_(False)
    t0 = time.perf_counter()
    with torch.no_grad():
        O.iw_nll(P, model, cell, sb, eps)
    dt_iw = time.perf_counter() - t0
    return dict(value=round(tokens / dt, 1), unit='tokens/s', cores=n_thr, socket_physical_cores=socket_cores,
                kind='port',
                sample='%d timed train steps after %d warm-up (batch 32, same shape and length distribution, reference '
                       'row order) of the CPU oracle on %d threads; IW-NLL: one batch of %d sentences at K=%d, measured '
                       '(iw_nll_sentences_per_s_K%d); the K=%d figure is that value x %d/%d (the cost is linear in K)'
                       % (n_steps, n_warm, n_thr, B, Ks, Ks, args.iw_samples, Ks, args.iw_samples),
                ms_per_step=round(dt / n_steps * 1e3, 1),
                **{'iw_nll_sentences_per_s_K%d' % Ks: round(B / dt_iw, 4),
                   'iw_nll_sentences_per_s': round(B / dt_iw * Ks / args.iw_samples, 4)})


def rnn_live_flops(hp, batches):
    """Algorithmic FLOPs of the recurrent products over LIVE (row, step) pairs only: the encoder's two directions
    run len steps per sentence, each decoder the prefix step + len+1 target steps.  (The library's own figure for
    the whole-sequence kernels counts the padded T*R pairs of the launch.)  Same for forward and backward."""
    G = 3 if hp.cell == 'gru' else 4
    H, Hd = hp.H, 2 * hp.H
    nd = 2 if hp.is_couple() else 1
    tot = 0.0
    for b in batches:
        ln = b['len'].double()
        tot += 2.0 * G * H * H * 2 * float(ln.sum()) + 2.0 * G * Hd * Hd * nd * float((ln + 2).sum())
    return tot


def run_training(tr, batches, tokens, args, world, dev, rank):
    """W warm-up + K timed steps; returns (seconds for the K steps, tokens/s of the whole job)."""
    from cvae_amd import _lib
    for i in range(args.warmup):
        tr.train_step(batches[i])
    barrier_sync(world)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        tr.train_step(batches[i])
    tr.dp_flush()                      # data-parallel: a time-out in the last steps is handled before the clock stops
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, dev)
    if _lib.query('cvae_persist_status') != 0:
        raise _lib.CvaeHipError('a whole-sequence recurrent kernel timed out inside the timed region: no figure')
    tok = torch.tensor([float(sum(tokens[args.warmup:args.warmup + args.steps]))], dtype=torch.float64, device=dev)
    if world > 1:
        if dist.get_backend() == 'gloo':
            tok = tok.cpu()
        dist.all_reduce(tok)
    return dt, float(tok.item()) / dt


def run_iw(tr, data, harness, shape, B, n_batches, K, by_samples, world, dev, rank, seed=5000):
    ev = data.Corpus(data.synthetic_sentences(B * (n_batches + 1), shape['V'], shape['mu'], shape['sigma'],
                                              shape['cap'], seed=seed + (0 if by_samples else rank)))
    ebatches = [harness.text_batch(ev, range(i * B, (i + 1) * B), dev, truncate=False) for i in range(n_batches + 1)]
    tr.iw_eval(ebatches[:1], K)                       # warm-up
    barrier_sync(world)
    t0 = time.perf_counter()
    metrics, _ = tr.iw_eval(ebatches[1:], K)
    barrier_sync(world)
    dt_iw = max_over_ranks(time.perf_counter() - t0, world, dev)
    return ebatches, dict(value=round(metrics['n'] / dt_iw, 3), unit='sentences/s', n_sample=K, sentences=metrics['n'],
                          seconds=round(dt_iw, 4))


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback)'
    # CVAE_BENCH_REHEARSE=1: every rank on GPU 0 with gloo, to rehearse the N > 1 control flow on a
    # one-GPU box (run it with CVAE_PERSISTENT_RNN=0: the whole-sequence kernels of two processes
    # cannot be co-resident on one GPU).  Not a measurement.
    rehearse = os.environ.get('CVAE_BENCH_REHEARSE') == '1'
    if os.environ.get('CVAE_BENCH_WATCHDOG'):          # dump every thread's stack and exit if stuck
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['CVAE_BENCH_WATCHDOG']), exit=True)
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)
    from cvae_amd import _lib, data, harness, settings
    _lib.init(local_rank)
    gathered_devices = ['cuda:%d' % local_rank]
    if world > 1:                                   # which device every rank drives (the `collective` record)
        gathered_devices = [None] * world
        dist.all_gather_object(gathered_devices, '%s/cuda:%d' % (os.uname().nodename, local_rank))
    global SPLIT_BF16
    SPLIT_BF16 = bool(_lib.query('cvae_split_bf16', 1))          # returns the previous (= configured) setting ...
    _lib.query('cvae_split_bf16', int(SPLIT_BF16))               # ... which stays in force

    shape = data.SHAPES[args.workload]
    hp = harness.HParams(model='vae-couple', cell=args.cell, V=shape['V'], E=shape['E'], H=shape['H'], Z=shape['Z'],
                         annealing_start=1000 if args.workload != 'ptb' else 2000,
                         euclidean_dist=(args.workload == 'yahoo'))
    log(rank, 'world %d, device %s: building the model' % (world, dev))
    tr = harness.Trainer(hp, dev, seed=0)          # same seed: identical initial weights on every rank
    tr.iter_num = 20000                            # mid-annealing: lambda_KL > 0, every loss term live
    B = hp.batch_size
    n_batches = max(args.steps + args.warmup, 1)
    sents = data.synthetic_sentences(B * n_batches, shape['V'], shape['mu'], shape['sigma'], shape['cap'],
                                     seed=1000 + rank)
    corpus = data.Corpus(sents)
    # rows in the corpus' (random) order, as the reference's shuffling DataLoader hands them over; the product
    # orders a batch's rows by length itself (modules.RowOrder)
    batches = [harness.text_batch(corpus, range(i * B, (i + 1) * B), dev) for i in range(n_batches)]
    tokens = [corpus.n_tokens(range(i * B, (i + 1) * B)) for i in range(n_batches)]
    torch.manual_seed(1234 + rank)                 # dropout / eps streams differ per rank

    log(rank, 'training: %d warm-up + %d timed steps' % (args.warmup, args.steps))
    dt, train_tok_s = run_training(tr, batches, tokens, args, world, dev, rank)
    log(rank, 'training done: %.3f ms/step' % (dt / max(args.steps, 1) * 1e3))

    # ---- informational: the same corpus with length-bucketed batches from the device-resident corpus
    # (batch composition differs from the reference's plain shuffle, so this is NOT `value`)
    corpus.to_device(dev)
    order = data.epoch_batches(corpus.lengths, B, seed=7 + rank, bucket=True, drop_last=True)
    bb = [dict(zip(('bare', 'go', 'eos', 'len'), corpus.batch_resident(ix))) for ix in order[:n_batches]]
    btok = [corpus.n_tokens(ix) for ix in order[:n_batches]]
    for i in range(min(args.warmup, len(bb))):
        tr.train_step(bb[i])
    barrier_sync(world)
    t0 = time.perf_counter()
    for i in range(args.warmup, len(bb)):
        tr.train_step(bb[i])
    barrier_sync(world)
    dt_b = max_over_ranks(time.perf_counter() - t0, world, dev)
    tok_b = torch.tensor([float(sum(btok[args.warmup:]))], dtype=torch.float64, device=dev)
    if world > 1:
        if dist.get_backend() == 'gloo':
            tok_b = tok_b.cpu()
        dist.all_reduce(tok_b)
    bucketed = dict(value=round(float(tok_b.item()) / max(dt_b, 1e-9), 1), unit='tokens/s',
                    ms_per_step=round(dt_b / max(len(bb) - args.warmup, 1) * 1e3, 3),
                    note='length-bucketed batches (pool 50 x batch), same sentences; informational')
    log(rank, 'bucketed training: %.3f ms/step' % bucketed['ms_per_step'])

    # ---- IW-NLL evaluation (test.py path): sentences sharded over ranks
    iw = None
    if not args.no_iw:
        by_samples = args.iw_shard == 'samples' and world > 1
        settings.shard_samples = by_samples
        log(rank, 'IW-NLL: K=%d, %d batches (+1 warm-up), shard=%s' % (args.iw_samples, args.iw_batches, args.iw_shard))
        ebatches, iw = run_iw(tr, data, harness, shape, B, args.iw_batches, args.iw_samples, by_samples, world, dev, rank)
        iw['shard'] = args.iw_shard if world > 1 else 'none'

    # ---- the shape BASELINE.json's metric names: PTB-shape LSTM-VAE with coupling (configs[0] on the GPU), batch 32,
    # train step + IW-NLL at the reference's test setting K = 100 (Experiment.py:1330-1335)
    ptb = None
    if args.workload != 'ptb' and not args.no_ptb:
        ps = data.SHAPES['ptb']
        php = harness.HParams(model='vae-couple', cell='lstm', V=ps['V'], E=ps['E'], H=ps['H'], Z=ps['Z'], annealing_start=2000)
        ptr = harness.Trainer(php, dev, seed=0)
        ptr.iter_num = 20000
        pc = data.Corpus(data.synthetic_sentences(B * n_batches, ps['V'], ps['mu'], ps['sigma'], ps['cap'], seed=2000 + rank))
        pb = [harness.text_batch(pc, range(i * B, (i + 1) * B), dev) for i in range(n_batches)]
        ptok = [pc.n_tokens(range(i * B, (i + 1) * B)) for i in range(n_batches)]
        pdt, ptoks = run_training(ptr, pb, ptok, args, world, dev, rank)
        ptb = dict(workload='PTB-shape Coupled-VAE (vae-couple, lstm): V=%d E=%d H=%d Z=%d, batch %d per GPU, lengths '
                            'lognormal(%.1f,%.2f) cap %d' % (ps['V'], ps['E'], ps['H'], ps['Z'], B, ps['mu'], ps['sigma'], ps['cap']),
                   value=round(ptoks, 1), unit='tokens/s', ms_per_step=round(pdt / max(args.steps, 1) * 1e3, 3))
        if not args.no_iw:
            settings.shard_samples = False
            _, ptb['iw_nll'] = run_iw(ptr, data, harness, ps, B, 8, 100, False, world, dev, rank, seed=6000)
            settings.shard_samples = args.iw_shard == 'samples' and world > 1
        log(rank, 'PTB shape: %.3f ms/step' % ptb['ms_per_step'])
        del ptr, pb

    # ---- roofline legs: the same work once more with the library's per-kernel HIP events on (events
    # are recorded on the launch stream inside the library: cvae_profile_enable / cvae_profile_read).
    # `roofline` = the kernel with the largest total time over the whole job, which is ce_core_kernel in
    # the IW-NLL evaluation; `roofline_train` = the dominant kernel of the training step alone.
    # Every rank runs the legs (their collectives must match); rank 0 records and reads the events.
    roof = roof_train = None
    roof_all = {}

    def table(prof):
        return {k: dict(launches=v['launches'], ms=round(v['ms'], 3),
                        rate=round(v['work'] / max(v['ms'], 1e-9) / (1e9 if v['bound'] == 'mfma' else 1e6), 2),
                        unit='TFLOP/s' if v['bound'] == 'mfma' else 'GB/s') for k, v in prof.items()}

    def leg(fn):
        if rank == 0:
            _lib.call('cvae_profile_enable', 1)
        fn()
        torch.cuda.synchronize()
        prof = profile_read(_lib.lib) if rank == 0 else None
        if rank == 0:
            _lib.call('cvae_profile_enable', 0)
        return prof

    log(rank, 'roofline legs')
    n_leg = min(3, n_batches)
    prof_t = leg(lambda: [tr.train_step(batches[i]) for i in range(n_leg)])
    if rank == 0:
        live = rnn_live_flops(hp, batches[:n_leg])
        for k in ('rnn_fwd_persist_kernel', 'rnn_bwd_persist_kernel'):
            if prof_t[k]['launches']:
                prof_t[k]['work'] = live               # live (row, step) pairs, not the launches' padded T x R
    prof_e = None if args.no_iw else leg(lambda: tr.iw_eval(ebatches[1:], args.iw_samples))
    if rank == 0:
        roof_train = roofline_of(prof_t, 'train step (%d steps)' % n_leg)
        roof_all['train'] = table(prof_t)
        roof = roof_train
        if prof_e is not None:
            roof_eval = roofline_of(prof_e, 'IW-NLL evaluation (%d batches)' % args.iw_batches)
            roof_all['iw_nll'] = table(prof_e)
            # the kernel with the largest total time over the whole job (both legs scaled to what the job ran)
            k_train = max(v['ms'] for k, v in prof_t.items() if k not in HBM_KERNELS) / n_leg * (args.steps + args.warmup)
            k_eval = max(v['ms'] for k, v in prof_e.items() if k not in HBM_KERNELS) * (args.iw_batches + 1.0) / args.iw_batches
            roof = roof_eval if (roof_eval and k_eval >= k_train) else roof_train
    roof_hbm = roofline_hbm([prof_t, prof_e]) if rank == 0 else None
    barrier_sync(world)

    # ---- the same step and evaluation on the fp32-input MFMA kernels (cvae_split_bf16(0)): what the split-bf16
    # products buy, beside `value` in the same line
    fp32 = None
    if SPLIT_BF16 and not args.no_fp32:
        _lib.query('cvae_split_bf16', 0)
        try:
            fdt, ftoks = run_training(tr, batches, tokens, args, world, dev, rank)
            fp32 = dict(arithmetic='fp32-input MFMA everywhere (cvae_split_bf16(0))', value=round(ftoks, 1), unit='tokens/s',
                        ms_per_step=round(fdt / max(args.steps, 1) * 1e3, 3))
            if not args.no_iw:
                _, fiw = run_iw(tr, data, harness, shape, B, args.iw_batches, args.iw_samples,
                                args.iw_shard == 'samples' and world > 1, world, dev, rank)
                fp32['iw_nll'] = fiw
        finally:
            _lib.query('cvae_split_bf16', 1)
        log(rank, 'fp32-input kernels: %.3f ms/step' % fp32['ms_per_step'])

    if rank == 0:
        out = {
            'metric': 'train tokens/s (Coupled-VAE train step) + IW-NLL eval sentences/s (field iw_nll)',
            'value': round(train_tok_s, 1), 'unit': 'tokens/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / max(args.steps, 1) * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (split-bf16x3/x6 products)' if SPLIT_BF16 else 'f32',
            'dtype_note': ('fp32 storage, accumulation and results; the hidden x vocab product and the many-row gate product of '
                           'IW evaluation run as error-compensated split-bf16 (3 bf16 MFMA passes per fp32 product; error vs fp64 '
                           'equal to the fp32-input MFMA kernels: tests/test_gpu_kernels.py::test_projce_bf16x3_error), the '
                           'sequence-wide gradient products and the backward recurrence with a three-term split (6 passes; '
                           'scripts/grad_error_modes.py), x W^T projections and the forward recurrence with the two-term split'
                           if SPLIT_BF16 else 'fp32-input MFMA everywhere (CVAE_SPLIT_BF16=0)'), 'data': 'synthetic',
            'config': {'workload': '%s-shape Coupled-VAE (vae-couple, %s): V=%d E=%d H=%d Z=%d, batch %d per GPU, '
                                   'lengths lognormal(%.1f,%.2f) cap %d, rows in corpus (shuffled) order; train step + IW-NLL K=%d'
                                   % (args.workload, args.cell, shape['V'], shape['E'], shape['H'], shape['Z'], B,
                                      shape['mu'], shape['sigma'], shape['cap'], args.iw_samples),
                       'headline': ('`value` is BASELINE.json configs[1] (the 1-GPU configuration: Yelp shape, train step; '
                                    'IW-NLL K=500 in `iw_nll`); `ptb_lstm` is the shape the metric string names '
                                    '(configs[0], the reference\'s CPU-runnable case) measured the same way'
                                    if args.workload == 'yelp' else 'workload selected with --workload'),
                       'parallelism': 'dp%d' % world},
            'iw_nll': iw, 'ptb_lstm': ptb, 'train_bucketed': bucketed, 'fp32_mfma': fp32, 'roofline': roof,
            'roofline_train': roof_train, 'roofline_hbm': roof_hbm, 'kernels': roof_all,
        }
        if world > 1:
            out['collective'] = dict(backend=dist.get_backend(), ranks=dist.get_world_size(),
                                     devices=sorted(set(gathered_devices)),
                                     gradient_all_reduce='flat arena, buckets of %d MB issued from autograd hooks'
                                                         % (settings.dp_bucket_bytes >> 20),
                                     buckets_issued_early_last_step=tr.opt.issued_early,
                                     persist_headroom_cus=settings.dp_persist_headroom,
                                     iterations_rerun_on_per_step_kernels=list(tr.fallbacks))
        if world == 1 and not args.no_cpu_baseline:
            log(rank, 'CPU baseline (oracle)')
            out['cpu_baseline'] = cpu_baseline(args, shape)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
