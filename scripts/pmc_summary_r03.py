"""Summaries of the round-3 rocprofv3 passes over bench.py (separate --pmc runs; see scripts/collect_profiles_r03.sh).

usage: python scripts/pmc_summary_r03.py hbm  <fetch.csv> <write.csv> <out.json>
       python scripts/pmc_summary_r03.py mfma <sq.csv> <out.json>

hbm:  FETCH_SIZE / WRITE_SIZE are reported in KB (1024 B) per dispatch; on gfx950 FETCH_SIZE tallies 128-byte requests
      at 64 bytes and is doubled here, as MI355X_MICROARCH.md prescribes (WRITE_SIZE is exact for 16-byte streaming
      stores and float atomics).  One record per kernel bench.py prices: the CE core's evaluation launches
      (`ce_core_forward_eval_launch` = roofline.traffic) and the three bandwidth kernels of `roofline_hbm`.
mfma: SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE per dispatch; the matrix-pipe
      utilisation of a kernel is MFMA_BUSY / (1024 SIMDs x launch cycles), launch cycles = GRBM_GUI_ACTIVE / 8
      (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs; checked: 443e6 / 8 = 55e6 cycles = the 29 ms evaluation launch
      of the CE core at 1.9 GHz): the share of the chip's SIMD-cycles in which a matrix instruction was executing."""
import collections
import csv
import json
import sys

KEYS = collections.OrderedDict([
    # template arguments <NB, GRAD, SAVE[, FAST]>: the evaluation launch is GRAD = SAVE = false, the training forward SAVE = true
    ('ce_core_forward_eval_launch', lambda n: 'ce_core_bf3_kernel<' in n and n.split('ce_core_bf3_kernel<')[1].split('>')[0].replace(' ', '').split(',')[1:3] == ['false', 'false']),
    ('ce_core_train_forward', lambda n: 'ce_core_bf3_kernel<' in n and n.split('ce_core_bf3_kernel<')[1].split('>')[0].replace(' ', '').split(',')[1:3] == ['false', 'true']),
    ('clip_adam_kernel', lambda n: 'clip_adam_kernel' in n),
    ('ce_grad_from_logits_kernel', lambda n: 'ce_grad_from_logits_kernel' in n),
    ('iw_reduce_kernel', lambda n: 'iw_reduce_kernel' in n),
    ('rnn_large_step_bf3_kernel', lambda n: 'rnn_large_step_bf3_kernel' in n),
    ('rnn_large_wreg_kernel', lambda n: 'rnn_large_wreg_kernel' in n),
    ('gemm_wide_bf3_kernel', lambda n: 'gemm_wide_bf3_kernel' in n),
    ('gemm_nt_bf3_kernel', lambda n: 'gemm_nt_bf3' in n),
    ('rnn_fwd_persist_tag_kernel', lambda n: 'rnn_fwd_persist_tag_kernel' in n),
    ('rnn_bwd_persist_tag_kernel', lambda n: 'rnn_bwd_persist_tag_kernel' in n),
])


def load(path):
    """kernel name -> counter -> list of per-dispatch values (dispatch order)"""
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        agg.setdefault(r['Kernel_Name'], collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    return agg


def pick(agg, pred, counter):
    vals = []
    for k, c in agg.items():
        if pred(k) and counter in c:
            vals += c[counter]
    return vals


def big(vals):
    """the launches above half the largest one (drops warm-up-sized and tail launches of the same kernel)"""
    if not vals:
        return []
    m = max(vals)
    return [i for i, v in enumerate(vals) if v > 0.5 * m]


def hbm(fetch_csv, write_csv, out_path):
    fa, wa = load(fetch_csv), load(write_csv)
    out = collections.OrderedDict()
    for key, pred in KEYS.items():
        f, w = pick(fa, pred, 'FETCH_SIZE'), pick(wa, pred, 'WRITE_SIZE')
        if not f:
            continue
        sel = big(f)
        fk = sum(f[i] for i in sel) / len(sel)
        wk = sum(w[i] for i in sel) / len(sel) if len(w) == len(f) else (sum(w) / len(w) if w else float('nan'))
        out[key] = dict(launches=len(sel), fetch_kb_raw_avg=round(fk, 2), fetch_bytes_corrected_avg=round(fk * 2048),
                        write_bytes_avg=round(wk * 1024))
    out['note'] = ('FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); both count L2->fabric '
                   'requests incl. Infinity-Cache hits; separate --pmc passes of: bench.py --no-cpu-baseline --no-ptb --no-fp32 '
                   '--steps 3 --warmup 1 --iw-batches 2; per kernel the mean over its launches above half its largest one')
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != 'note'}))


def mfma(sq_csv, out_path, cus=256):
    agg = load(sq_csv)
    out = collections.OrderedDict()
    for key, pred in KEYS.items():
        busy, gui = pick(agg, pred, 'SQ_VALU_MFMA_BUSY_CYCLES'), pick(agg, pred, 'GRBM_GUI_ACTIVE')
        sqb, wav = pick(agg, pred, 'SQ_BUSY_CYCLES'), pick(agg, pred, 'SQ_WAVE_CYCLES')
        if not busy or not gui:
            continue
        sel = big(gui)
        mean = lambda v: sum(v[i] for i in sel) / len(sel) if len(v) == len(gui) else float('nan')
        b, g = mean(busy), mean(gui)
        out[key] = dict(launches=len(sel), SQ_VALU_MFMA_BUSY_CYCLES=round(b), GRBM_GUI_ACTIVE=round(g),
                        SQ_BUSY_CYCLES=round(mean(sqb)) if sqb else None, SQ_WAVE_CYCLES=round(mean(wav)) if wav else None,
                        mfma_pipe_busy_frac=round(b / (4.0 * cus * g / 8.0), 4))
    out['note'] = ('one --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE) of: bench.py '
                   '--no-cpu-baseline --no-ptb --no-fp32 --steps 3 --warmup 1 --iw-batches 2; mfma_pipe_busy_frac = '
                   'MFMA_BUSY / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs): SIMD-cycles with a matrix instruction executing over '
                   'the SIMD-cycles of the launch (a whole-sequence recurrent launch occupies 128-256 CUs and waits on '
                   'hand-offs most of the time: its fraction is small by construction)')
    json.dump(out, open(out_path, 'w'), indent=1)
    print(json.dumps({k: v.get('mfma_pipe_busy_frac') for k, v in out.items() if k != 'note'}))


if __name__ == '__main__':
    if sys.argv[1] == 'hbm':
        hbm(*sys.argv[2:5])
    else:
        mfma(*sys.argv[2:4])
