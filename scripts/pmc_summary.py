"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into profiles/<out>.json.

usage: python scripts/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB (1024 bytes) per dispatch; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes and is doubled here, as
MI355X_MICROARCH.md prescribes.  The record `ce_core_forward_eval_launch` averages the large
IW-evaluation launches of the CE core kernel (GRAD=false, SAVE=false; those above half the largest one): it is the
`roofline.traffic` figure bench.py reports for its dominant kernel."""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        agg.setdefault(r['Kernel_Name'][:100], []).append(float(r['Counter_Value']))
    out = [dict(kernel=k, launches=len(v), total_kb=round(sum(v), 1), avg_kb=round(sum(v) / len(v), 1),
                max_kb=round(max(v), 1)) for k, v in agg.items()]
    out.sort(key=lambda d: -d['total_kb'])
    return out, agg


def main():
    fetch_csv, write_csv, out_path = sys.argv[1:4]
    fetch, fa = per_kernel(fetch_csv, 'FETCH_SIZE')
    write, wa = per_kernel(write_csv, 'WRITE_SIZE')
    rec = None

    def is_eval_ce(name):
        # ce_core_bf3_kernel<NB, GRAD, SAVE> (split-bf16, the default) or ce_core_kernel<NQ, GRAD, EXACT, SAVE>
        # (CVAE_SPLIT_BF16=0): the evaluation launches are GRAD = false AND SAVE = false (the training forward
        # is <.., SAVE = true>: it also stores the live rows' logits)
        for stem, n in (('ce_core_bf3_kernel<', 3), ('ce_core_kernel<', 4)):
            if stem in name:
                targs = [t.strip() for t in name.split(stem)[1].split('>')[0].split(',')]
                return len(targs) == n and targs[1] == 'false' and targs[-1] == 'false'
        return False

    key = [k for k in fa if is_eval_ce(k)]
    key.sort(key=lambda k: -max(fa[k]))
    if key:
        f, w = fa[key[0]], wa.get(key[0], [])
        big = [i for i, v in enumerate(f) if v > 0.5 * max(f)]
        fk = sum(f[i] for i in big) / len(big)
        wk = sum(w[i] for i in big) / len(big) if len(w) == len(f) else float('nan')
        rec = dict(kernel=key[0], launches=len(big), fetch_kb_raw_avg=round(fk, 2), fetch_bytes_corrected_avg=round(fk * 1024 * 2),
                   write_bytes_avg=round(wk * 1024),
                   note='FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); counts '
                        'L2->fabric requests incl. Infinity-Cache hits; separate --pmc passes (FETCH_SIZE, WRITE_SIZE) '
                        'of: bench.py --no-cpu-baseline --steps 3 --warmup 1 --iw-batches 2')
    json.dump({'FETCH_SIZE': fetch[:16], 'WRITE_SIZE': write[:16], 'ce_core_forward_eval_launch': rec},
              open(out_path, 'w'), indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
